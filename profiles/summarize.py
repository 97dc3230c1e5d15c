#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (written by profiles/collect.sh) into the committed
evidence: profiles/<tag>_kernel_stats.csv (rocprofv3 --stats output, verbatim),
profiles/<tag>_summary.md and profiles/traffic.json (HBM bytes per launch from the PMC passes,
FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950).

Launches that ran as no-ops (the loop kernels return at once when the PivotRecord says the loop
ended) are excluded from the per-kernel averages: a launch counts when its duration is at least
half of that kernel's 90th-percentile launch (not the longest: one slow outlier would hide the rest).
"""
import collections
import csv
import glob as _glob
import json
import os
import shutil
import statistics
import sys


def newest(pattern):
    """Matches of `pattern`, newest first: gpurun merges every collection into the same directory, older runs stay."""
    return sorted(_glob.glob(pattern), key=os.path.getmtime, reverse=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, value_col=None):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("relp::", "").replace("void ", "").split("<")[0]
        dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        val = float(row[value_col]) if value_col else 0.0
        agg[name].append((dur, val))
    out = {}
    for name, rows in agg.items():
        durs = sorted(d for d, _ in rows)
        mx = durs[min(len(durs) - 1, int(0.9 * len(durs)))]
        eff = [(d, v) for d, v in rows if d >= 0.5 * mx]
        out[name] = {"launches": len(rows), "effective": len(eff),
                     "avg_us": statistics.mean(d for d, _ in eff) / 1e3,
                     "value": statistics.mean(v for _, v in eff),
                     # every launch, no filter: what rocprofv3's own *_kernel_stats.csv reports
                     "all_avg_us": statistics.mean(durs) / 1e3, "total_ms": sum(durs) / 1e6}
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    workload = sys.argv[2] if len(sys.argv) > 2 else "dense10k"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    trace = per_kernel(newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0])
    fetch = per_kernel(newest(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv"))[0], "Counter_Value")
    write = per_kernel(newest(os.path.join(src, "pmc_write", "*", "*_counter_collection.csv"))[0], "Counter_Value")
    # calibration of the counters for 8-byte-per-lane tile accesses: k_tile<true> in place moves 8 m n bytes each way
    cal_f = cal_w = None
    cal_src = os.path.join(ROOT, "gpurun_out", f"prof_{sys.argv[3]}") if len(sys.argv) > 3 else src     # calibration of another collection
    try:
        cf = per_kernel(newest(os.path.join(cal_src, "cal_fetch", "*", "*_counter_collection.csv"))[0], "Counter_Value")
        cw = per_kernel(newest(os.path.join(cal_src, "cal_write", "*", "*_counter_collection.csv"))[0], "Counter_Value")
        known = 8.0 * 10000 * 20000
        cal_f = known / (cf["k_tile"]["value"] * 1024.0)
        cal_w = known / (cw["k_tile"]["value"] * 1024.0)
    except Exception as e:          # noqa: BLE001
        print("no calibration run:", e)
    lines = [f"# rocprofv3 summary `{tag}` ({workload})", "",
             "Command: `bash profiles/collect.sh` (rocprofv3 --kernel-trace --stats; separate --pmc FETCH_SIZE and",
             "--pmc WRITE_SIZE passes).  Durations from the kernel trace, effective launches only (see",
             "profiles/summarize.py).  HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per launch.", "",
             "| kernel | launches (effective) | avg us | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM MB / launch |",
             "|---|---|---|---|---|---|"]
    traffic = {}
    for name in sorted(trace, key=lambda k: -trace[k]["avg_us"] * trace[k]["effective"]):
        if not name.startswith("k_"):
            continue
        t = trace[name]
        f = fetch.get(name, {}).get("value", 0.0)
        w = write.get(name, {}).get("value", 0.0)
        hbm = (2.0 * f + w) * 1024.0
        note = ""
        if name in ("k_tab_flush", "k_tab_flush_lds", "k_flush_apply"):
            # 8-byte-per-lane loads: the x2 FETCH_SIZE correction is calibrated for 16-byte-per-lane streams only
            # (MI355X_MICROARCH.md section HBM: other widths uncalibrated); the raw figure already matches the bytes
            # of the matrix, so raw FETCH_SIZE + WRITE_SIZE is reported for these kernels (k_tab_flush_lds reads its
            # T0 tile 8 bytes per lane and under-reports further: treat its figure as a lower bound)
            if cal_f and name in ("k_tab_flush", "k_tab_flush_lds"):
                hbm = (f * cal_f + w * cal_w) * 1024.0
                note = f" (8-byte tile accesses: FETCH_SIZE x {cal_f:.3f}, WRITE_SIZE x {cal_w:.3f}, calibrated on scripts/microbench/tile_rmw.hip)"
            else:
                hbm = (f + w) * 1024.0
                note = " (raw FETCH_SIZE, 8-byte loads)"
        traffic[name[2:]] = hbm
        lines.append(f"| {name} | {t['launches']} ({t['effective']}) | {t['avg_us']:.1f} | {f:.1f} | {w:.1f} | {hbm / 1e6:.1f}{note} |")
    # the sparse path: launches per pivot of the LU engine (one whole 25FV47 solve)
    try:
        lstats = newest(os.path.join(src, "lu_trace", "*", "*_kernel_stats.csv"))[0]
        shutil.copy(lstats, os.path.join(ROOT, "profiles", f"{tag}_lu_25fv47_kernel_stats.csv"))
        lt = per_kernel(newest(os.path.join(src, "lu_trace", "*", "*_kernel_trace.csv"))[0])
        log = open(os.path.join(src, "lu_trace.log")).read()
        pivots = int([ln for ln in log.splitlines() if ln.startswith("optimal")][0].split()[1])
        total = sum(v["launches"] for k, v in lt.items())
        lines += ["", f"## LU engine, Netlib 25FV47, whole solve ({pivots} pivots)", "",
                  f"{total} kernel launches in all = {total / pivots:.3f} per pivot.  Every launch counts here (the pivot",
                  "kernel's launches legitimately last 0.3 - 3 ms: they end when the update file is full or the phase is over), so",
                  f"the rows equal those of `{tag}_lu_25fv47_kernel_stats.csv`.", "",
                  "| kernel | launches | avg us | total ms |", "|---|---|---|---|"]
        for name in sorted(lt, key=lambda k: -lt[k]["total_ms"])[:8]:
            v = lt[name]
            lines.append(f"| {name} | {v['launches']} | {v['all_avg_us']:.1f} | {v['total_ms']:.1f} |")
    except Exception as e:          # noqa: BLE001
        print("no LU trace:", e)
    open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allt[workload] = traffic
    allt["_source"] = f"profiles/{tag}_summary.md"
    json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
