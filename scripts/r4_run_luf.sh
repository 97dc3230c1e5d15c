cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for args in "11 1"; do RELP_DEBUG=1 timeout -k 10 120 python scripts/r4_luf_profile.py $args 2>&1 | grep -E "device factorisation,|schedule |block" | tail -n 6 | cut -c1-460; done
