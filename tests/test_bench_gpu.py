"""The driver's benchmark command as a child process on the GPU box: one complete JSON line (round 1's run died with KeyError
'GBps' and left BENCH_r01.json empty).  [r4] in its `--quick` form; the full line is checked on the recorded run of the round."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_headline(out, quick):
    assert out["metric"] == "simplex iterations/sec" and out["unit"] == "iterations/s" and out["n_gpus"] == 1
    assert out["value"] > 1000 and out["higher_is_better"] is True and out["dtype"] == "f64" and out["vs_baseline"] is None
    block = out["config"]["update_block"]
    assert block == 64 and out["steps"] % block == 0 and out["steps"] >= 4 * block
    assert out["steps_requested"] == 20 and out["timing"]["steps_requested"] == 20 and "AT LEAST" in out["config"]["steps_note"]
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
    sp = out["sparse_engine"]
    assert sp["outcome"] == "optimal" and abs(sp["objective"] - 5.5018459e+03) < 1e-4 and sp["tolerances"] == "relp_default_config"
    assert sp["kernel_launches_per_pivot"] <= 2.0 / 11 and sp["pivot_kernel_phase_share"]["u_solve"] > 0      # persistent pivot kernel
    assert sp["reference_cadence_update_block_11"]["outcome"] == "optimal"
    pp = sp["reference_cadence_update_block_11_pipelined"]              # (RELP_LU_PIPELINE_SHORT: the host factorises behind the kernel's back)
    assert pp["outcome"] == "optimal" and abs(pp["objective"] - 5.5018459e+03) < 1e-4 and pp["lookahead_installs"] >= pp["refactorisations"] - 4
    for other in ("explicit_inverse_engine", "tableau_engine"):
        assert sp[other]["outcome"] == "optimal" and abs(sp[other]["objective"] - 5.5018459e+03) < 1e-4
    # [r4] the sparse path's CPU baseline is the reference's own back-end for it (LUDecomposition + eta file, oracle/relp_f64_lu.h),
    # the sparse-rows back-end beside it; the LU engine at the same cadence walks the LU oracle's pivots
    assert sp["cpu_baseline"]["kind"] == "port" and "LUDecomposition" in sp["cpu_baseline"]["back_end"] and sp["cpu_baseline"]["value"] > 0
    assert sp["cpu_baseline"]["lu_engine_at_the_same_cadence_walks_the_same_pivots_for"] >= 1000
    assert "BasisInverseRows" in sp["cpu_baseline_rows_back_end"]["back_end"] and sp["cpu_baseline_rows_back_end"]["value"] > 0
    c1 = out["c1"]
    assert c1["exact_cpu"]["objective_is_the_reference_pin"] is True and c1["exact_cpu"]["pivots"] > 100
    for label in ("lu", "revised", "tableau"):
        assert c1[label]["outcome"] == "optimal" and c1[label]["trace_identical_to_exact"] is True
        assert abs(c1[label]["objective"] - 24975305659811992079614961229 / 120651674036153428931840) < 1e-6
    stride = out["kernel_event_stride"]
    assert stride % block != 0                                            # (no aliasing with the block)
    if quick:
        assert "c4" not in out and "c5" not in out and "scale" not in sp and "large" not in sp and "replicas" not in sp


def check_full_sections(out):
    """The sections `--quick` leaves out (the driver's full command measures them; the recorded line of the round is checked
    against this in the CPU tier, tests/test_bench_json.py)."""
    assert out["c4"]["update_block"] == 96 and out["c4"]["value"] > 1000 and out["c4"]["roofline"]["kernel"] == "k_tab_flush_lds"
    sp = out["sparse_engine"]
    c5 = out["c5"]
    for label in ("lu", "revised", "tableau"):
        assert c5["50v-10"][label]["outcome"] == "optimal" and abs(c5["50v-10"][label]["objective"] - 2879.065687) < 1e-3
        assert c5["50v-10"][label]["degenerate_pivots"] > 0
        assert c5["acc-tight4"][label]["pivots"] >= 30000 and c5["acc-tight4"][label]["degenerate_pivots"] > 0
    large = sp["large"]
    for name, tol in (("GREENBEA", 1.0), ("GREENBEB", 10.0), ("80BAU3B", 1e-3)):
        for label in ("lu", "revised", "tableau"):
            assert large[name][label]["outcome"] == "optimal" and abs(large[name][label]["pin_error"]) < tol, (name, label)
    assert large["80BAU3B"]["lu"]["pivot_kernel_clocks_per_pivot"] and large["DFL001"]["lu"]["pivot_kernel_clocks_per_pivot"]
    # the LU engine beyond one CU's LDS: third kernel layout against the loop it used to fall back to and the dense engines
    scale = sp["scale"]
    assert scale["lu"]["kernel_layout"]["persistent_kernel"] and scale["lu"]["kernel_layout"]["layout"] == 2
    assert not scale["lu_product_form_fallback"]["kernel_layout"]["persistent_kernel"]
    for leg in ("lu_product_form_fallback", "tableau", "revised"):
        assert scale[leg]["first_250_pivots_equal_the_lu_engines"] is True, leg
    assert scale["lu"]["rows"] == 63988 and scale["lu"]["pivots"] == 20000 and scale["lu"]["value"] > 1000
    assert scale["lu_over_fallback"] > 5 and scale["lu"]["pivot_kernel_clocks_per_pivot"] > 0
    assert scale["cpu_baseline"]["lu_engine_takes_the_same_5000_pivots"] is True and scale["cpu_baseline"]["value"] > 0
    assert abs(scale["cpu_baseline"]["objective_after_sample"] - scale["cpu_baseline"]["lu_engine_objective_after_sample"]) < 1e-6
    assert scale["cpu_baseline_lu_back_end"]["value"] is not None and "LUDecomposition" in scale["cpu_baseline_lu_back_end"]["back_end"]
    # SURVEY 8e, "LU engine: replicas only": R independent engines on one GPU, each walking the solo pivot sequence
    rep = sp["replicas"]["replicas"]
    for r in ("1", "8", "32", "64"):
        assert rep[r]["all_optimal"] is True and rep[r]["every_replica_walks_the_solo_pivots"] is True and rep[r]["value"] > 0
    assert rep["8"]["value"] > 2.0 * rep["1"]["value"]


def test_driver_command_quick_form_emits_one_complete_json_line():
    """`--quick` (the full command took 130 of the tier's 372 s and the driver runs it anyway): the same
    launcher, the same assembly, every key of the headline, `roofline` and `cpu_baseline`, the 25FV47 section and c1 / c2 measured
    as always; the long sections are left to the driver's own run, whose recorded line tests/test_bench_json.py checks."""
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--quick"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    check_headline(json.loads(lines[0]), quick=True)


def test_sparse_path_as_replicas_is_what_gpus_n_engine_lu_runs():
    """`bench.py --gpus N --engine lu` (replicas only, no data-path collective): here N = 1 with four replicas."""
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--engine", "lu", "--replicas", "4"]
    out = _one_line(subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600))
    assert out["config"]["engine"] == "lu" and out["config"]["replicas_per_gpu"] == 4 and out["scaling"] == "weak" and out["n_gpus"] == 1
    assert out["replicas"]["all_optimal"] is True and out["replicas"]["every_replica_walks_the_solo_pivots"] is True
    assert out["value"] > 10000 and abs(out["replicas"]["objective"] - 5.5018459e+03) < 1e-4


def _one_line(p):
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_multi_rank_command_rehearsed_with_two_ranks_on_one_gpu():
    """The driver's N = 2 path, `python bench.py --gpus 2 ...`, as a fresh child process (it starts torch.distributed.run
    itself): RELP_BENCH_REHEARSE=1 puts both ranks on this box's one GPU and lets gloo carry the exchange (RCCL refuses two
    ranks on one device); launcher, rank wiring, sharded engines, barrier + max-over-ranks timing and the JSON assembly are the
    ones an 8-GPU node runs.  No scaling curve has been measured anywhere yet: this checks the path, not a speed."""
    env = dict(os.environ, RELP_BENCH_REHEARSE="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"]
    out = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 10       # (gloo on the host carries every exchange)
    assert "sharded x2" in out["config"]["parallelism"] and "shard_loop" in out["config"]
    assert out["roofline"]["kernel"] == "k_tab_flush_lds" and out["roofline"]["frac"] > 0
    assert out["c4"]["value"] > 10 and "sharded x2" in out["c4"]["workload"]
    assert "cpu_baseline" not in out                                       # rank 0 at N = 1 only


def test_sparse_replicas_over_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2 --engine lu`: what an N-GPU run of the sparse path consists of (replicas only) -- one
    process per GPU, every rank its own LU engines, no data-path collective, barrier + max-over-ranks timing, value = all pivots /
    that time.  Rehearsed with both ranks on this box's one GPU (gloo carries the two scalar reductions)."""
    env = dict(os.environ, RELP_BENCH_REHEARSE="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--engine", "lu", "--replicas", "2"]
    out = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "replicas x2"
    assert out["replicas"]["every_replica_walks_the_solo_pivots"] is True and out["value"] > 10000


def test_sharded_loop_over_rccl_with_one_rank():
    """`--force-sharded` at N = 1: the native multi-GPU loop (relp_shard_run) with RCCL itself -- ncclCommInitRank from the id
    torch.distributed broadcasts, ncclAllGather between the kernels on the engine's stream -- on a communicator of one rank."""
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--force-sharded", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    out = _one_line(subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500))
    assert out["n_gpus"] == 1 and out["value"] > 1000
    assert "native" in out["config"]["shard_loop"] and out["c4"]["value"] > 1000
    assert out["roofline"]["kernel"] == "k_tab_flush_lds"
