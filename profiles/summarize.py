#!/usr/bin/env python3
"""Turns one gpurun_out/prof_<tag>/ directory (written by profiles/run_profiles.sh on the GPU box) into
the tracked summary under profiles/<tag>/ and refreshes profiles/pmc_traffic.json, which bench.py reads
for roofline.traffic (HBM bytes per launch of each kernel from the rocprofv3 PMC passes).

usage: python profiles/summarize.py <tag> [workload-key]
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "500x2000xH4096xE1xfixedxmfmaxfast"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    for log in ("bench_trace.log",):
        for line in open(os.path.join(src, log)):
            if line.startswith("{"):
                with open(os.path.join(dst, "bench_under_rocprof.json"), "w") as f:
                    f.write(line)
    kern = {}
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        path = os.path.join(src, name, "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "psdev::" not in k or r["Counter_Name"] != ctr:
                continue
            short = k.split("psdev::")[1].split("<")[0].split("(")[0]
            short = {"ps_ransac_score_fast": "ps_ransac_score"}.get(short, short)   # bench.py's name of the timing slot
            agg[short].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            kern.setdefault(k, {})[ctr + "_KB_per_launch"] = sum(v) / len(v)
            kern[k]["launches"] = len(v)
    summary = {
        "command": "profiles/run_profiles.sh " + tag + "  (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE: "
                   "three separate passes of the same bench.py command)",
        "note": "FETCH_SIZE / WRITE_SIZE are KB per dispatch. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
                "counts 128-B requests at 64 B, i.e. reports half of the bytes read -> doubled before use. Calibrated on "
                "this access pattern in round 1 (scalar s_load_dwordx4/x16 + 16-byte vector loads: the VALU matcher read every "
                "descriptor row of the 500-frame sequence, 32.0 MB unique, and reported FETCH_SIZE = 16.4 MB = 32.7 MB after "
                "doubling). "
                "WRITE_SIZE is used as is. traffic = 2 x FETCH_SIZE + WRITE_SIZE.",
        "kernels": kern}
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # SQ / MFMA counters (averages per dispatch)
    sq = os.path.join(src, "sq", "bench_counter_collection.csv")
    if os.path.exists(sq):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(sq)):
            k = r["Kernel_Name"]
            if "psdev::" in k:
                agg[k.split("psdev::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(dst, "sq_counters.json"), "w") as f:
            json.dump({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}, f, indent=1)
    # per-launch durations from the kernel trace, in launch order (the --stats average mixes warm-up launches, the timed
    # region and bench.py's single-chain leg; with the default profile command --streams 1 --steps 5 --warmup 2 the
    # launches are: 2 warm-up, 5 timed, 1 leg warm-up, 5 leg (these are what `kernel_ms` averages), 1 statistics pass)
    tr = os.path.join(src, "trace", "bench_kernel_trace.csv")
    if os.path.exists(tr):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(tr)):
            k = r["Kernel_Name"]
            if "psdev::" in k:
                per[k.split("psdev::")[1].split("(")[0]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
        out = {}
        # launch order of bench.py --streams 1: W warm-up, K timed, 1 leg warm-up, L = max(3, min(8, K)) leg launches
        # (what `kernel_ms` averages), 1 statistics pass
        K = W = None
        try:
            line = json.loads(open(os.path.join(dst, "bench_under_rocprof.json")).read())
            K, W = int(line["steps"]), int(line["warmup"])
        except Exception:
            pass
        for k, v in per.items():
            v.sort()
            ms = [round(x[1], 4) for x in v]
            leg = None
            if K is not None:
                L = max(3, min(8, K))
                if len(ms) == W + K + 1 + L + 1:
                    leg = sum(ms[-(L + 1):-1]) / L
            out[k] = {"launch_ms": ms, "all_launches_mean_ms": sum(ms) / len(ms), "single_chain_leg_mean_ms": leg}
        with open(os.path.join(dst, "kernel_launch_ms.json"), "w") as f:
            json.dump(out, f, indent=1)
    for extra in ("bench_default.json",):
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    allt = json.load(open(tf)) if os.path.exists(tf) else {}
    allt[key] = {k: int((2.0 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024)
                 for k, v in kern.items()}
    allt[key]["_source"] = f"profiles/{tag}/pmc_summary.json"
    with open(tf, "w") as f:
        json.dump(allt, f, indent=1)
    print(json.dumps(summary["kernels"], indent=1))


if __name__ == "__main__":
    main()
