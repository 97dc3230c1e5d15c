************************************************************************
*
*  The data in this file represents a dummy problem used to set up
*  storage and pointers for later manipulation.  The problem
*  consists of one row, one column, and one element.
*
************************************************************************
NAME          DUMMY
ROWS
 N  DOBJ
 G  DROW1
COLUMNS
    DCOL1     DOBJ             1.0
    DCOL1     DROW1            1.0
ENDATA
