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
    def short_name(k):
        short = k.split("psdev::")[1].split("<")[0].split("(")[0]
        # bench.py's name of the timing slot: all builds / stages of the scoring kernel are one step of the path
        return {"ps_ransac_score_fast": "ps_ransac_score", "ps_ransac_score_euclid": "ps_ransac_score",
                "ps_stage_reorder": "ps_ransac_score",   # (the scoring step's own launch between stage 0 and stage 1)
                "ps_ransac_score_mfma": "ps_ransac_score"}.get(short, short)

    # A kernel may be launched several times per step (the staged scoring: prefix + three stages): every figure below
    # is PER STEP = sum over the step's launches; steps = dispatches of the once-per-step kernel ps_crosscheck_prep.
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
            agg[short_name(k)].append(float(r["Counter_Value"]))
        steps = max(len(agg.get("ps_crosscheck_prep", [])), 1)
        for k, v in agg.items():
            kern.setdefault(k, {})[ctr + "_KB_per_launch"] = sum(v) / steps      # (per step; key name kept for bench.py)
            kern[k]["launches"] = len(v)
            kern[k]["launches_per_step"] = len(v) / steps
    summary = {
        "command": "profiles/run_profiles.sh " + tag + "  (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE: "
                   "three separate passes of the same bench.py command)",
        "note": "FETCH_SIZE / WRITE_SIZE are KB per STEP of the path (sum over the step's launches of the kernel: the staged "
                "scoring launches its kernel four times per step). gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
                "counts 128-B requests at 64 B, i.e. reports half of the bytes read -> doubled before use. Calibrated on "
                "this access pattern in round 1 (scalar s_load_dwordx4/x16 + 16-byte vector loads: the VALU matcher read every "
                "descriptor row of the 500-frame sequence, 32.0 MB unique, and reported FETCH_SIZE = 16.4 MB = 32.7 MB after "
                "doubling). "
                "WRITE_SIZE is used as is. traffic = 2 x FETCH_SIZE + WRITE_SIZE.",
        "kernels": kern}
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # SQ / MFMA counters (sums per step, by full kernel name)
    sq = os.path.join(src, "sq", "bench_counter_collection.csv")
    if os.path.exists(sq):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        bygrid = collections.defaultdict(lambda: collections.defaultdict(list))
        nstep = 0
        for r in csv.DictReader(open(sq)):
            k = r["Kernel_Name"]
            if "psdev::" in k:
                name = k.split("psdev::")[1].split("(")[0]
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
                bygrid[f"{name} grid {grid}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if k.startswith("ps_crosscheck_prep"):
                nstep = max(nstep, max(len(v) for v in cs.values()))
        nstep = max(nstep, 1)
        with open(os.path.join(dst, "sq_counters.json"), "w") as f:
            json.dump({k: dict({c: sum(v) / nstep for c, v in cs.items()}, launches_per_step=max(len(v) for v in cs.values()) / nstep)
                       for k, cs in agg.items()}, f, indent=1)
        # the same per launch shape (a kernel launched twice per step with different grids does different things: stage 0 =
        # models only (one work-group per pair: the pure sample -> Umeyama -> SVD prologue), then the sweep)
        with open(os.path.join(dst, "sq_counters_by_grid.json"), "w") as f:
            json.dump({k: dict({c: sum(v) / nstep for c, v in cs.items()}, launches_per_step=max(len(v) for v in cs.values()) / nstep)
                       for k, cs in bygrid.items()}, f, indent=1)
    # per-step durations from the kernel trace, in launch order (the --stats average mixes warm-up launches, the timed
    # region and bench.py's single-chain leg).  Launch order of bench.py --streams 1 --repeats 1 --no-other-modes:
    # W warm-up steps, K timed, 1 leg warm-up, L = max(3, min(8, K)) leg steps (what `kernel_ms` averages), 1 statistics pass
    tr = os.path.join(src, "trace", "bench_kernel_trace.csv")
    if os.path.exists(tr):
        launches = collections.defaultdict(list)
        for r in csv.DictReader(open(tr)):
            k = r["Kernel_Name"]
            if "psdev::" in k:
                launches[short_name(k)].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
        step_starts = sorted(t for t, _ in launches.get("ps_crosscheck_prep", []))
        out = {}
        K = W = None
        seq = None
        try:
            line = json.loads(open(os.path.join(dst, "bench_under_rocprof.json")).read())
            K, W = int(line["steps"]), int(line["warmup"])
            seq = line.get("launch_sequence")
        except Exception:
            pass
        import bisect
        for k, v in launches.items():
            v.sort()
            per_step = [0.0] * len(step_starts)
            # a launch belongs to the step whose cross-check kernel started last before it (kernel 1 runs before
            # kernel 2 of its own step: it is attributed through the NEXT cross-check start)
            for t, ms in v:
                i = bisect.bisect_right(step_starts, t) - 1
                if k in ("ps_expand_query_fp4", "ps_hamming_mfma", "ps_hamming_nn"):
                    i += 1
                if 0 <= i < len(per_step):
                    per_step[i] += ms
            ms_list = [round(x, 4) for x in per_step]
            leg = None
            if seq is not None:   # round 4: bench.py reports its own call sequence
                L, tail = int(seq["leg"]), int(seq["stats"])
                total = sum(int(seq[x]) for x in ("warm", "timed", "instrumented", "leg_warm", "leg", "stats"))
                if L and len(ms_list) == total:
                    leg = sum(ms_list[len(ms_list) - tail - L:len(ms_list) - tail]) / L
            elif K is not None:
                L = max(3, min(8, K))
                if len(ms_list) == W + K + 1 + L + 1:
                    leg = sum(ms_list[-(L + 1):-1]) / L
            out[k] = {"step_ms": ms_list, "launches_per_step": len(v) / max(len(step_starts), 1),
                      "all_steps_mean_ms": sum(ms_list) / max(len(ms_list), 1), "single_chain_leg_mean_ms": leg}
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
