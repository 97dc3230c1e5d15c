"""MPS / SIF reader: free and fixed column formats, exact decimal
parsing, conversion to a ``GeneralForm``.  One-shot, serial CPU code (exact ``Fraction``): it feeds
the pivot engine, it is not accelerated.

Restates (file:line under /root/reference/src/io/):
  mps/parse/mod.rs:37-90      section order NAME [OBJSENSE] ROWS COLUMNS [RHS] [RANGES] [BOUNDS] ENDATA
  mps/parse/mod.rs:101-107    comment and empty lines are dropped
  mps/parse/mod.rs:249-313    ROWS: one N row; constraint rows SORTED BY NAME (row index = name rank)
  mps/parse/mod.rs:353-480    COLUMNS: file order, at most two (row, value) pairs per line, MARKER lines
  mps/parse/mod.rs:536-640    RHS / RANGES groups, at most two pairs per line
  mps/parse/mod.rs:692-744    BOUNDS: FR MI PL BV LO UP FX LI UI (SC unimplemented)
  mps/parse/free.rs:9-93      free format = whitespace split
  mps/parse/fixed.rs:13-128   fixed format = character fields 1..3, 4..12, 14..22, 24..36, 39..47, 49..61
  mps/number/parse.rs:53-121  plain decimals only ([-]ddd[.ddd], no exponent) -> exact int / 10^k
  mps/convert.rs:28-500       MPS -> GeneralForm (bounds semantics, ranges, merged right-hand sides)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Optional, Tuple

SECTIONS = ("ROWS", "COLUMNS", "RHS", "BOUNDS", "RANGES", "ENDATA")
FIELDS = [(0, 1), (1, 3), (4, 12), (14, 22), (24, 36), (39, 47), (49, 61)]   # fixed.rs:120-128
COMMENT = "*"


class MPSError(ValueError):
    pass


def parse_number(text: str) -> Fraction:
    """mps/number/parse.rs:84-121: sign, integer part, mantissa; no exponents."""
    if not text:
        raise MPSError("empty number")
    sign = 1
    if text[0] == "-":
        sign, text = -1, text[1:]

    def part(t: str) -> int:
        if t == "":
            return 0
        if not (t.isdigit() or (t[0] == "+" and t[1:].isdigit())):
            raise MPSError(f"cannot parse number part {t!r}")
        return int(t)
    if "." in text:
        idx = text.index(".")
        k = len(text) - idx - 1
        value = part(text[:idx]) * 10 ** k + part(text[idx + 1:])
        return Fraction(sign * value, 10 ** k)
    return Fraction(sign * part(text), 1)


@dataclass
class MPS:
    """io/mps/mod.rs `MPS` (rows sorted by name; columns in file order)."""
    name: str
    maximize: bool
    cost_row_name: str
    cost_values: List[Tuple[int, Fraction]]                   # (column, value)
    rows: List[Tuple[str, str]]                               # (name, 'L' | 'E' | 'G')
    columns: List[Tuple[str, bool, List[Tuple[int, Fraction]]]]   # (name, integer?, sorted (row, value))
    rhss: List[Tuple[str, List[Tuple[int, Fraction]]]]
    ranges: List[Tuple[str, List[Tuple[int, Fraction]]]]
    bounds: List[Tuple[str, List[Tuple[int, Tuple[str, Optional[Fraction]]]]]]


# ---- column retrievers ------------------------------------------------------------------------
class _Free:
    @staticmethod
    def one_and_two(line):
        p = line.split()
        if len(p) < 2:
            raise MPSError("Could not read second field")
        return p[0], p[1]

    @staticmethod
    def column_line(line):
        p = line.split()
        if len(p) < 3:
            raise MPSError("Could not read fourth field")
        if p[1] == "'MARKER'":
            return ("marker", p[2])
        return ("data", p[0], p[1], p[2], p[3:])

    @staticmethod
    def two_through_four(line):
        p = line.split()
        if len(p) < 3:
            raise MPSError("Could not read fourth field")
        return p[0], p[1], p[2], p[3:]

    @staticmethod
    def five_and_six(rest):
        return (rest[0], rest[1]) if len(rest) >= 2 else None

    @staticmethod
    def one_through_three(line):
        p = line.split()
        if len(p) < 3:
            raise MPSError("Could not read third field")
        return p[0], p[1], p[2], p[3:]

    @staticmethod
    def four(rest):
        if not rest:
            raise MPSError("Could not read value for bound.")
        return rest[0]


class _Fixed:
    @staticmethod
    def one_and_two(line):
        if len(line) <= FIELDS[2][0]:
            raise MPSError("Line is too short.")
        name = line[FIELDS[2][0]:min(FIELDS[2][1], len(line))].rstrip()
        if not name:
            raise MPSError("Empty row name.")
        return line[FIELDS[1][0]:FIELDS[1][1]], name

    @staticmethod
    def column_line(line):
        if len(line) < FIELDS[4][1]:
            raise MPSError("Line is too short.")
        if line[FIELDS[3][0]:FIELDS[3][1]] == "'MARKER'":
            if len(line) < FIELDS[5][1]:
                raise MPSError("Line is too short to be a marker line.")
            return ("marker", line[FIELDS[5][0]:FIELDS[5][1]])
        return ("data", line[FIELDS[2][0]:FIELDS[2][1]].rstrip(), line[FIELDS[3][0]:FIELDS[3][1]].rstrip(),
                line[FIELDS[4][0]:FIELDS[4][1]].lstrip(), line[FIELDS[4][1]:])

    @staticmethod
    def two_through_four(line):
        if len(line) < FIELDS[4][1]:
            raise MPSError("Line is too short.")
        return (line[FIELDS[2][0]:FIELDS[2][1]].rstrip(), line[FIELDS[3][0]:FIELDS[3][1]].rstrip(),
                line[FIELDS[4][0]:FIELDS[4][1]].lstrip(), line[FIELDS[4][1]:])

    @staticmethod
    def five_and_six(rest):
        base = FIELDS[4][1]
        if len(rest) >= FIELDS[6][1] - base:
            five = rest[FIELDS[5][0] - base:FIELDS[5][1] - base].rstrip()
            six = rest[FIELDS[6][0] - base:FIELDS[6][1] - base].lstrip()
            if five and six:
                return five, six
        return None

    @staticmethod
    def one_through_three(line):
        if len(line) < FIELDS[3][0]:
            raise MPSError("Line is too short.")
        return (line[FIELDS[1][0]:FIELDS[1][1]], line[FIELDS[2][0]:FIELDS[2][1]].rstrip(),
                line[FIELDS[3][0]:FIELDS[3][1]].rstrip(), line[FIELDS[3][1]:])

    @staticmethod
    def four(rest):
        end = FIELDS[4][1] - FIELDS[3][1]
        if len(rest) < end:
            raise MPSError("Line doesn't contain another value, it's too short.")
        return rest[FIELDS[4][0] - FIELDS[3][1]:end].lstrip()


# ---- parser -------------------------------------------------------------------------------------
def _parse(text: str, cr) -> MPS:
    lines = [(n + 1, ln) for n, ln in enumerate(text.splitlines())
             if not ln.lstrip().startswith(COMMENT) and ln != ""]
    pos = 0

    def same_section(ln):
        return ln.startswith(" ")

    def next_section(ln, allowed):
        if ln not in SECTIONS:
            raise MPSError(f"Unknown section header {ln!r}.")
        if ln != "ENDATA" and ln not in allowed:
            raise MPSError(f"Unexpected section {ln!r}.")
        return ln

    # NAME (parse/mod.rs:171-195): the line must start with NAME, name = first token after it
    if not lines or not lines[0][1].startswith("NAME"):
        raise MPSError("No NAME line.")
    after = lines[0][1][4:].split()
    if not after:
        raise MPSError("No name found.")
    name = after[0]
    pos = 1
    # optional OBJSENSE (parse/mod.rs:211-247); trailing blanks are ignored, the ROWS header follows
    maximize = False
    if pos >= len(lines):
        raise MPSError("No line to read, is the program more than a name?")
    head = lines[pos][1].rstrip()
    if head == "ROWS":
        pos += 1
    elif head == "OBJSENSE":
        if pos + 2 >= len(lines) + 1 or pos + 1 >= len(lines):
            raise MPSError("Program can't end in the OBJSENSE section.")
        direction = lines[pos + 1][1].rstrip()
        if pos + 2 >= len(lines) or not lines[pos + 2][1].startswith("ROWS"):
            raise MPSError("Expected the ROWS section next.")
        if direction in ("  MAXIMIZE", "  MAX"):
            maximize = True
        elif direction in ("  MINIMIZE", "  MIN"):
            maximize = False
        else:
            raise MPSError(f"Can't read objective {direction}")
        pos += 3
    else:
        raise MPSError(f"Line contents {head!r} were unexpected")

    # ROWS
    rows: List[Tuple[str, str]] = []
    cost_row = None
    while True:
        if pos >= len(lines):
            raise MPSError("Section ended sooner than expected.")
        ln = lines[pos][1]
        pos += 1
        if not same_section(ln):
            next_section(ln, ("COLUMNS",))
            break
        rtype, rname = cr.one_and_two(ln)
        kind = rtype[0:1]                                          # RowType::from_str: first character
        if kind == "N":
            if cost_row is not None:
                raise MPSError("Second cost row detected. This is not supported.")
            cost_row = rname
        elif kind in ("L", "E", "G"):
            rows.append((rname, kind))
        else:
            raise MPSError(f"Row type {rtype!r} unknown.")
    if cost_row is None:
        raise MPSError("No cost name read.")
    rows.sort(key=lambda r: r[0])                                  # parse/mod.rs:296
    names = [r[0] for r in rows]
    if cost_row in names:
        raise MPSError("Cost row name found in other rows.")
    if len(set(names)) != len(names):
        raise MPSError("Duplicate row name found.")
    row_index = {n: i for i, n in enumerate(names)}

    # COLUMNS
    columns: List[Tuple[str, bool, List[Tuple[int, Fraction]]]] = []
    cost_values: List[Tuple[int, Fraction]] = []
    cur_name, cur_vals = None, []
    integer = False

    def save_column(new_name):
        nonlocal cur_name, cur_vals
        if cur_name is not None:
            vals = sorted(cur_vals, key=lambda t: t[0])
            if any(vals[k][0] == vals[k + 1][0] for k in range(len(vals) - 1)):
                raise MPSError(f"Duplicate row for column {cur_name!r}")
            columns.append((cur_name, integer, vals))
            cur_vals = []
        cur_name = new_name

    section = None
    while True:
        if pos >= len(lines):
            raise MPSError("Section ended sooner than expected.")
        ln = lines[pos][1]
        pos += 1
        if not same_section(ln):
            section = next_section(ln, ("RHS", "RANGES", "BOUNDS"))
            save_column(None)
            break
        content = cr.column_line(ln)
        if content[0] == "marker":
            save_column(None)
            marker = content[1].strip()
            if marker == "'INTORG'":
                integer = True
            elif marker == "'INTEND'":
                integer = False
            else:
                raise MPSError(f"Marker type {marker!r} unknown.")
            continue
        _, cname, rname, vtext, rest = content
        if cur_name is not None:
            if cur_name != cname:
                save_column(cname)
        else:
            cur_name = cname

        def save_pair(rn, vt):
            value = parse_number(vt)
            if rn in row_index:
                cur_vals.append((row_index[rn], value))
            elif rn == cost_row:
                cost_values.append((len(columns), value))
            else:
                raise MPSError(f"Row {rn!r} not known.")
        save_pair(rname, vtext)
        extra = cr.five_and_six(rest)
        if extra is not None:
            save_pair(*extra)
    col_index = {c[0]: j for j, c in enumerate(columns)}

    def value_section(allowed):
        nonlocal pos
        groups: List[Tuple[str, List[Tuple[int, Fraction]]]] = []
        gname, gvals = None, []

        def save_group(new):
            nonlocal gname, gvals
            if gname is not None:
                vals = sorted(gvals, key=lambda t: t[0])
                for k in range(len(vals) - 1):
                    if vals[k][0] == vals[k + 1][0]:
                        raise MPSError(f"Duplicate row id {vals[k][0]} for group {gname!r}")
                groups.append((gname, vals))
                gvals = []
            gname = new
        while True:
            if pos >= len(lines):
                raise MPSError("Section ended sooner than expected.")
            ln = lines[pos][1]
            pos += 1
            if not same_section(ln):
                nxt = next_section(ln, allowed)
                save_group(None)
                return groups, nxt
            g, rn, vt, rest = cr.two_through_four(ln)
            if gname is not None:
                if gname != g:
                    save_group(g)
            else:
                gname = g
            for pair in ((rn, vt), cr.five_and_six(rest)):
                if pair is None:
                    continue
                if pair[0] not in row_index:
                    raise MPSError(f"Row {pair[0]!r} not known.")
                gvals.append((row_index[pair[0]], parse_number(pair[1])))

    rhss, ranges, bounds = [], [], []
    if section == "RHS":
        rhss, section = value_section(("RANGES", "BOUNDS"))
    if section == "RANGES":
        ranges, section = value_section(("BOUNDS",))
    seen = set()
    for _g, vals in ranges:
        for (i, _v) in vals:
            if i in seen:
                raise MPSError("Each row can have at most one range value")
            seen.add(i)
    if section == "BOUNDS":
        bname, bvals = None, []

        def save_bound(new):
            nonlocal bname, bvals
            if bname is not None:
                bounds.append((bname, sorted(bvals, key=lambda t: t[0])))   # stable; duplicates allowed
                bvals = []
            bname = new
        while True:
            if pos >= len(lines):
                raise MPSError("Section ended sooner than expected.")
            ln = lines[pos][1]
            pos += 1
            if not same_section(ln):
                next_section(ln, ())
                save_bound(None)
                break
            btype, bn, cname, rest = cr.one_through_three(ln)
            btype = btype.strip()
            if cname not in col_index:
                raise MPSError(f"Column name {cname!r} unknown")
            if bname is not None:
                if bname != bn:
                    save_bound(bn)
            else:
                bname = bn
            if btype in ("FR", "MI", "PL", "BV"):
                bvals.append((col_index[cname], (btype, None)))
            elif btype in ("LO", "UP", "FX", "LI", "UI"):
                bvals.append((col_index[cname], (btype, parse_number(cr.four(rest)))))
            elif btype == "SC":
                raise NotImplementedError("SC bounds (parse/mod.rs:735)")
            else:
                raise MPSError(f"Bound type {btype!r} unknown.")
    if pos < len(lines):
        raise MPSError("File parsed successfully, but it has nonempty lines at the end")
    return MPS(name, maximize, cost_row, cost_values, rows, columns, rhss, ranges, bounds)


def parse(text: str) -> MPS:
    """Free format (`io::import` uses it for both .mps and .SIF, io/mod.rs:49)."""
    return _parse(text, _Free)


def parse_fixed(text: str) -> MPS:
    """Fixed format (used by the Netlib tests, tests/netlib/mod.rs:54)."""
    return _parse(text, _Fixed)


def import_file(path: str, fixed: bool = False) -> MPS:
    with open(path) as f:
        text = f.read()
    return parse_fixed(text) if fixed else parse(text)
