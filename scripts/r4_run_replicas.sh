cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
# R replicas of the LU engine on one GPU: refactorisation on the host (look-ahead) and on the device, by hardware queues
for q in 16 32; do
  for dev in False True; do
    echo "GPU_MAX_HW_QUEUES=$q device_factorisation=$dev"
    GPU_MAX_HW_QUEUES=$q timeout -k 10 280 python -c "
import sys; sys.path.insert(0, '.')
import bench
out = bench.sparse_replicas(counts=(1, 8, 16, 32), device_factorisation=$dev)
print({k: (round(v['value']), round(v['seconds'], 2), v['every_replica_walks_the_solo_pivots']) for k, v in out['replicas'].items()})
" 2>&1 | tail -n 1
  done
done
