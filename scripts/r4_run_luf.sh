cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_lu_device.py -x -q -m gpu > gpurun_out/r4/lu_device.log 2>&1; tail -n 3 gpurun_out/r4/lu_device.log
for args in "11 1"; do RELP_DEBUG=1 timeout -k 10 120 python scripts/r4_luf_profile.py $args 2>&1 | grep -E "device factorisation,|schedule |block" | tail -n 6 | cut -c1-460; done
