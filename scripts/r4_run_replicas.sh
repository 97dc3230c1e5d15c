cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for q in 4 8 16 32; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python -c "
import sys; sys.path.insert(0, '.')
import bench
out = bench.sparse_replicas(counts=(8, 32))
print({k: (round(v['value']), round(v['seconds'], 2), v['every_replica_walks_the_solo_pivots']) for k, v in out['replicas'].items()})
" 2>&1 | tail -n 1
done
