"""bench.py's JSON assembly, without a GPU: window plan, algorithmic bytes and the one output line for profile
dictionaries as the engines produce them at K = 1, 20, 64, 200 (round 1's driver run died with KeyError 'GBps'
because no flush fell into a 20-pivot window; windows are whole update blocks now and no key is assumed)."""
import argparse
import json
import os
import subprocess
import sys

import math

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def ns(steps, warmup=5, workload="dense10k", events=True):
    return argparse.Namespace(steps=steps, warmup=warmup, workload=workload, no_kernel_events=not events, event_stride=16)


@pytest.mark.parametrize("k,block,expect", [(1, 64, 256), (20, 64, 256), (64, 64, 256), (200, 64, 256), (257, 64, 320),
                                            (1000, 64, 1024), (20, 0, 20), (1, 0, 1), (20, 128, 512), (0, 64, 256)])
def test_window_steps_are_whole_blocks(k, block, expect):
    assert bench.window_steps(k, block) == expect
    if block:
        assert bench.window_steps(k, block) % block == 0 and bench.window_steps(k, block) >= 4 * block


def tableau_prof(steps, block, stride=16, with_flush=True):
    sampled = max(steps // stride, 1)
    prof = {"price": (sampled, sampled * 0.0116), "ftran": (sampled, sampled * 0.0099), "ratio": (sampled, sampled * 0.0075),
            "select_column": (0, 0.0), "update_inverse": (0, 0.0)}
    if with_flush:
        prof["flush"] = (steps // block, (steps // block) * 0.655)
    return prof


def revised_prof(steps, block, stride=16):
    sampled = max(steps // stride, 1)
    prof = {"price": (sampled, sampled * 0.140), "ftran": (sampled, sampled * 0.135), "ratio": (sampled, sampled * 0.008),
            "apply_w": (sampled, sampled * 0.007), "update_w": (sampled, sampled * 0.009), "select_column": (sampled, sampled * 0.008)}
    if block:
        prof["flush"] = (steps // block, (steps // block) * 0.53)
    else:
        prof["update_inverse"] = (sampled, sampled * 0.276)
    return prof


@pytest.mark.parametrize("k", [1, 20, 64, 200])
@pytest.mark.parametrize("events", [True, False])
def test_line_for_every_step_count(k, events):
    m, n, _ = bench.WORKLOADS["dense10k"]
    steps = bench.window_steps(k, 64)
    windows = [steps * 33e-6 * f for f in (1.0, 1.02, 0.99, 1.01, 1.03)]
    res = {"steps": steps, "window_s": windows, "prof": tableau_prof(steps * 5, 64) if events else {}, "block": 64, "objective": -1.0}
    primary = bench.section(res, "dense10k", m, n, 1, "tableau")
    res2 = {"steps": steps, "window_s": [w * 10 for w in windows], "prof": revised_prof(steps * 5, 64) if events else {},
            "block": 64, "objective": -1.0}
    secondary = bench.section(res2, "dense10k", m, n, 1, "revised")
    cpu = {"value": 6.1, "unit": "iterations/s", "cores": 1, "kind": "port", "sample": "x"}
    out = bench.assemble(ns(k, events=events), 1, "tableau", primary, secondary, None, None, None, cpu)
    line = json.loads(json.dumps(out))
    assert line["steps"] == steps and line["steps"] % 64 == 0 and line["timing"]["steps_requested"] == k
    assert abs(line["value"] - steps / sorted(windows)[2]) < 1e-6 and line["higher_is_better"] is True
    assert abs(line["ms_per_step"] - sorted(windows)[2] * 1e3 / steps) < 1e-12
    assert line["cpu_baseline"]["kind"] == "port" and line["vs_baseline"] is None and line["dtype"] == "f64"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    # the whole pivot is always priced, whatever the events say
    assert roof["pivot"]["frac"] == pytest.approx(roof["pivot"]["achieved"] / 8000.0, rel=1e-3)
    assert 50e6 < roof["pivot"]["algorithmic_bytes"] < 80e6           # ~17 MB + 3.2 GB / 64
    if events:
        assert roof["kernel"] == "k_tab_flush_lds" and roof["frac"] == pytest.approx(roof["achieved"] / 8000.0, rel=1e-3)
        assert roof["achieved"] == pytest.approx(16.0 * m * (n + m) / 655e-6 / 1e9, rel=1e-3)
        assert roof["revised_engine"]["ftran"]["frac"] == pytest.approx(8.0 * m * m / 135e-6 / 1e9 / 8000.0, rel=2e-3)
        assert roof["revised_engine"]["price"]["achieved"] > 0
    else:
        assert roof["kernel"] is None and roof["frac"] is None and "note" in roof
        assert line["kernel_event_stride"] is None


def test_window_without_a_flush_in_the_profile_does_not_crash():
    """The exact shape of round 1's crash: a tableau profile without a 'flush' entry."""
    m, n, _ = bench.WORKLOADS["dense10k"]
    res = {"steps": 20, "window_s": [20 * 22e-6] * 5, "prof": tableau_prof(20, 64, with_flush=False), "block": 64, "objective": 0.0}
    primary = bench.section(res, "dense10k", m, n, 1, "tableau")
    assert primary["roofline"] is None
    out = bench.assemble(ns(20), 1, "tableau", primary)
    assert out["roofline"]["frac"] is None and out["roofline"]["pivot"]["achieved"] > 0


def test_revised_engine_as_primary_and_sharded_sections():
    m, n, _ = bench.WORKLOADS["dense10k"]
    for block in (0, 64):
        steps = bench.window_steps(20, block)
        res = {"steps": steps, "window_s": [steps * 330e-6] * 5, "prof": revised_prof(steps * 5, block), "block": block, "objective": 0.0}
        primary = bench.section(res, "dense10k", m, n, 1, "revised")
        out = bench.assemble(ns(20), 1, "revised", primary)
        assert out["roofline"]["kernel"] in ("k_price_all", "k_ftran", "k_update_inverse_vectors", "k_flush_apply")
        assert 0.0 < out["roofline"]["frac"] < 1.0
    # 8 ranks: the local shard's bytes shrink, the line says so
    alg1 = bench.algorithmic_bytes("tableau", m, n, 1, 64)
    alg8 = bench.algorithmic_bytes("tableau", m, n, 8, 64)
    assert alg8["launch"]["flush"] == pytest.approx(alg1["launch"]["flush"] / 8)
    res = {"steps": 256, "window_s": [256 * 30e-6] * 5, "prof": tableau_prof(1280, 64), "block": 64, "objective": 0.0}
    primary = bench.section(res, "dense10k", m, n, 8, "tableau")
    c4 = dict(bench.section(res, "c4", 10000, 50000, 8, "tableau"), workload="c4")
    out = bench.assemble(ns(200), 8, "tableau", primary, c4=c4, loop_kind="native")
    assert out["n_gpus"] == 8 and "cpu_baseline" not in out and out["c4"]["roofline"]["kernel"] == "k_tab_flush_lds"
    assert "x8" in out["config"]["parallelism"] and out["config"]["shard_loop"] == "native"


def test_gpus_flag_without_a_launcher_starts_the_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus 2` outside torch.distributed.run must launch the ranks itself, not exit."""
    seen = {}

    def fake_run(cmd, stdout=None):
        seen["cmd"] = cmd
        return subprocess.CompletedProcess(cmd, 0, stdout=b'{"ok": true}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "20"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "20"]


@pytest.mark.parametrize("block", [0, 1, 11, 32, 48, 64, 96, 128])
@pytest.mark.parametrize("stride", [1, 16, 64, 96])
def test_event_stride_never_aliases_with_the_update_block(stride, block):
    """stride 64 on block 64 bracketed the first pivot of every block only.  The sampled positions of a block
    must cover all of 0 .. K - 1."""
    got = bench.coprime_stride(stride, block)
    assert got >= stride
    if block > 1:
        assert got % block != 0 and math.gcd(got, block) == 1
        assert {(k * got) % block for k in range(block)} == set(range(block))


def test_recorded_bench_line_of_the_round_is_complete():
    """The driver's exact command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) run on an MI355X and recorded as
    profiles/r04_bench.json: every section the GPU tier's `--quick` form leaves out (c4, c5, sparse.large, sparse.scale,
    sparse.replicas) is checked here against the same assertions (tests/test_bench_gpu.py)."""
    path = os.path.join(ROOT, "profiles", "r04_bench.json")
    if not os.path.exists(path):
        pytest.skip("no recorded line yet")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_bench_gpu
    with open(path) as f:
        out = json.load(f)
    test_bench_gpu.check_headline(out, quick=False)
    test_bench_gpu.check_full_sections(out)
