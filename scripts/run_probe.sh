# scratch script of the GPU box runs (gpurun -- 'bash scripts/run_probe.sh')
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_lu_device.py -m gpu -q > gpurun_out/r4/lu_device.log 2>&1; echo "pytest exit $?"; tail -n 8 gpurun_out/r4/lu_device.log
for args in "-1 1" "11 1"; do RELP_DEBUG=1 timeout -k 10 120 python scripts/r4_luf_profile.py $args 2>&1 | grep -E "device factorisation,|schedule |block" | tail -n 7; done
