set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15
