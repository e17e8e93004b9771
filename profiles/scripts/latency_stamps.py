"""Single-pair latency study (BASELINE configs[1]): HIP-event time of every kernel of one 2000-keypoint pair, and the phase
breakdown of kernels 2 and 4 from in-kernel shader-clock stamps (option "stamps"; ps_debug_stamps).
Run on the GPU box: python3 profiles/scripts/latency_stamps.py > gpurun_out/latency_stamps.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

K2 = ["best[q] built (init + atomicMin pass)", "matches compacted + depth filter + records", "bounds / operands / counters"]
K4 = ["(1) selection replay", "(2) winner model + inlier pass + compaction", "(3a) refit: wave Umeyama + Jacobi SVD",
      "(3b) Euclidean re-selection + mask", "(4) pointInlierRatio + pose + stats"]
out = {}
seq = synth.make_sequence(2, 2000, config=3, index=0)
for name, ev, est, H in (("E1/fixed/4096", 1, EST_FIXED, 4096), ("E0/ransac/487", 0, EST_RANSAC, 487), ("E0/fixed/4096", 0, EST_FIXED, 4096)):
    ctx = api.Context(0)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    prm = default_ransac_params(ev)
    cfg, _ = make_config(est, H, seed=3)
    for _ in range(20):
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    per_pair_us = (time.perf_counter() - t0) / n * 1e6
    ctx.enable_timing(True)
    for _ in range(50):
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    kern = {k: v[0] / max(v[1], 1) * 1e3 for k, v in ctx.kernel_time_totals().items()}      # us per launch
    ctx.enable_timing(False)
    ctx.set_option("stamps", 1)
    acc = np.zeros(16)
    reps = 30
    for _ in range(reps):
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
        torch.cuda.synchronize()
        acc += np.array(ctx.stamps(), dtype=np.float64)
    st = ctx.stamps()
    ctx.set_option("stamps", 0)
    s = np.array(st, dtype=np.float64)
    k2_ticks = np.diff(s[0:4])
    k4_ticks = np.diff(s[4:10])
    # ticks -> microseconds through each kernel's own HIP-event duration
    k2_us = k2_ticks / max(k2_ticks.sum(), 1) * kern.get("ps_crosscheck_prep", 0.0)
    k4_us = k4_ticks / max(k4_ticks.sum(), 1) * kern.get("ps_select_refit", 0.0)
    out[name] = {"wall_us_per_pair_device_resident": per_pair_us, "kernel_us": kern, "kernel_us_sum": sum(kern.values()),
                 "ps_crosscheck_prep_phases_us": dict(zip(K2, [round(float(x), 2) for x in k2_us])),
                 "ps_select_refit_phases_us": dict(zip(K4, [round(float(x), 2) for x in k4_us])),
                 "ticks": {"kernel2": [int(x) for x in k2_ticks], "kernel4": [int(x) for x in k4_ticks]},
                 "note": "phase shares are from s_memtime stamps of work-group 0 (in launch order one pair = one work-group), "
                         "scaled to the kernel's HIP-event duration (which includes launch and drain)"}
    ctx.close()
print(json.dumps(out, indent=1))
