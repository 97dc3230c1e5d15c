"""R independent LU engines on one GPU with the refactorisations on the host (default) and on the device (f4): aggregate it/s."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
for dev in (False, True):
    out = bench.sparse_replicas(counts=(1, 8, 32, 64, 128), device_factorisation=dev)
    print("device factorisation" if dev else "host factorisation", {k: (round(v["value"]), round(v["seconds"], 2), v["all_optimal"], v["every_replica_walks_the_solo_pivots"]) for k, v in out["replicas"].items()}, flush=True)
