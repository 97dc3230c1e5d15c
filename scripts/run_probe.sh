set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sh in 0 1; do
  RELP_LU_LOOKAHEAD_SHORT=$sh timeout -k 10 200 python scripts/lu_large.py netlib/25FV47.SIF 0 lu 30000 11 > gpurun_out/fv11.log 2>&1 || true
  echo "25FV47 block 11, short look-ahead $sh"; grep "optimal\|lookahead" gpurun_out/fv11.log | tail -n 2 | cut -c1-300
done
for b in 8 11 16; do
  timeout -k 10 200 python scripts/lu_large.py netlib/25FV47.SIF 0 lu 30000 $b > gpurun_out/fv11.log 2>&1 || true
  echo "block $b"; grep "optimal" gpurun_out/fv11.log | tail -n 1 | cut -c1-200
done
