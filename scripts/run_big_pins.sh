# GPU probe of the big Netlib pins with the f64 safeguards on (one step after the other; a failed step ends the run)
set -e
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 240 python scripts/big_pins.py "$@" > gpurun_out/bp3_$1_$2.log 2>&1 || { tail -n 5 gpurun_out/bp3_$1_$2.log; echo "FAILED $@"; exit 1; }; tail -n 3 gpurun_out/bp3_$1_$2.log; }
run GREENBEA lu -1 100 1 1
run GREENBEA tableau -1 100 1 1 200
run GREENBEA revised -1 100 1 1 200
run GREENBEA revised -1 100 1 1 1000
run GREENBEA tableau -1 100 1 1 1000
run 80BAU3B tableau 32 100 0 1 200
run 80BAU3B tableau 32 100 1 1 200
