cd $GRAFT_REPO_ROOT
for fuse in 0 256; do for v in 1 2; do timeout -k 5 60 scripts/microbench/ell_pass scripts/microbench/basis_25fv47_100.txt $v $fuse; done; done
for v in 1 2; do timeout -k 5 60 scripts/microbench/ell_pass 790 6 4 $v 0; done
