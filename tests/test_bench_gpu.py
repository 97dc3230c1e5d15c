"""The driver's exact benchmark command as a child process on the GPU box: one complete JSON line (round 1's run died
with KeyError 'GBps' and left BENCH_r01.json empty)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_command_emits_one_complete_json_line():
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["metric"] == "simplex iterations/sec" and out["unit"] == "iterations/s" and out["n_gpus"] == 1
    assert out["value"] > 1000 and out["higher_is_better"] is True and out["dtype"] == "f64" and out["vs_baseline"] is None
    block = out["config"]["update_block"]
    assert block == 64 and out["steps"] % block == 0 and out["steps"] >= 4 * block and out["timing"]["steps_requested"] == 20
    assert out["timing"]["windows"] == 5 and len(out["timing"]["window_ms"]) == 5
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) <= 1e-6 * out["value"]
    roof = out["roofline"]
    assert roof["kernel"] == "k_tab_flush_lds" and roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0.2 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["launches_timed"] == 5 * out["steps"] // block              # every flush of the timed windows is bracketed
    assert 0.0 < roof["pivot"]["frac"] < 1.0 and roof["pivot"]["algorithmic_bytes"] > 0
    assert 0.3 < roof["revised_engine"]["ftran"]["frac"] < 1.0 and 0.3 < roof["revised_engine"]["price"]["frac"] < 1.0
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "pivots" in cpu["sample"]
    assert out["revised_engine"]["value"] > 100 and out["c2"]["value"] > 1000
    assert out["c4"]["value"] > 1000 and out["c4"]["roofline"]["kernel"] == "k_tab_flush_lds"
    sp = out["sparse_engine"]
    assert sp["outcome"] == "optimal" and abs(sp["objective"] - 5.5018459e+03) < 1e-4 and sp["tolerances"] == "relp_default_config"
    assert sp["kernel_launches_per_pivot"] <= 2.0 / 11 and sp["pivot_kernel_phase_share"]["u_solve"] > 0      # persistent pivot kernel
    assert sp["reference_cadence_update_block_11"]["outcome"] == "optimal"
    for other in ("explicit_inverse_engine", "tableau_engine"):
        assert sp[other]["outcome"] == "optimal" and abs(sp[other]["objective"] - 5.5018459e+03) < 1e-4
