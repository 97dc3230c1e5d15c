set -e
cd $GRAFT_REPO_ROOT
export RELP_DEBUG=1
run() { echo "== $@"; timeout -k 10 300 python scripts/lu_large.py "$@" 2>&1 | grep -v "^\[relp\] refactor\|^\[relp\] schedule" | tail -n 8; }
run netlib/25FV47.SIF 1 lu 100000
run miplib/acc-tight4.mps 0 lu 60000
run miplib/acc-tight4.mps 0 tableau 60000
run netlib/80BAU3B.SIF 1 lu 100000 -1 0 1
run netlib/DFL001.SIF 1 lu 40000 -1 1 1
run netlib/DFL001.SIF 1 tableau 40000 -1 1 1
run netlib/GREENBEB.SIF 1 lu 100000 -1 1 1
unset RELP_DEBUG
timeout -k 10 900 python -m pytest tests/test_gpu_big_pins.py -x -q 2>&1 | tail -n 15
