#!/usr/bin/env python3
"""Where the staged scoring starts to pay: time of one batched call with option prune = 2 (staged whenever possible) over the
same call with prune = 0 (complete scoring), across batch size x frame size x hypotheses x metric x schedule.  The cost model
of prepare_score (ps_capi.hip) is fitted on the `units` column: pairs x (ceil(H / 256) - 1) x frame capacity.
    python profiles/scripts/staged_crossover.py [--quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from putslam_amd import api, synth  # noqa: E402
from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config  # noqa: E402
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs  # noqa: E402

quick = "--quick" in sys.argv
ctxs = {pr: api.Context(0) for pr in (2, 0, 1)}
for pr, c in ctxs.items():
    c.set_option("prune", pr)


def measure(seq, ev, est, H, calls):
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    prm = default_ransac_params(ev)
    cfg, _ = make_config(est, H, seed=3)
    out = {}
    pbs = {pr: PairBatchDevice(seq["pairs"], fs.max_kpts) for pr in ctxs}
    for pr, c in ctxs.items():
        for _ in range(4):
            run_pairs(c, prm, cfg, TUM_FR1_K, fs, pbs[pr])
    torch.cuda.synchronize()
    best = {pr: 1e9 for pr in ctxs}
    for rep in range(3):                      # the forms alternate: same clock state, same box
        for pr, c in ctxs.items():
            t0 = time.perf_counter()
            for _ in range(calls):
                run_pairs(c, prm, cfg, TUM_FR1_K, fs, pbs[pr])
            torch.cuda.synchronize()
            best[pr] = min(best[pr], (time.perf_counter() - t0) / calls * 1e3)
    staged_default = ctxs[1].get_option("last_staged_pairs") > 0
    return best, staged_default


SCHED = [("E1/fixed", 1, EST_FIXED), ("E0/fixed", 0, EST_FIXED), ("E0/ransac", 0, EST_RANSAC), ("E1/usac", 1, EST_USAC)]
print("# ms per call: staged (prune=2) / complete (prune=0) = ratio | default (prune=1) took the staged form? | units = P x (ceil(H/256)-1) x cap")
worst = 1.0
for name, ev, est in SCHED:
    for kpts in ((500, 2000) if quick else (500, 1000, 2000, 4000)):
        for H in ((4096,) if quick else ((1024, 4096, 16384) if est == EST_FIXED else (487, 4000))):
            Heff = min(H, 487) if est == EST_RANSAC else H      # (the RANSAC schedule never consumes more: RANSAC.cpp:30,450-453)
            hb1 = (Heff + 255) // 256 - 1
            if hb1 < 1:
                continue
            for P in (4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 499):
                units = P * hb1 * kpts
                if units < 4.0e4 or units > 1.5e7:
                    continue
                seq = synth.make_sequence(P + 1, kpts, config=3, index=7)
                best, staged_default = measure(seq, ev, est, H, 10)
                ratio = best[2] / best[0]
                chosen = best[1] / min(best[2], best[0])
                worst = max(worst, chosen)
                print(f"{name:9s} kpts {kpts:5d} H {H:6d} P {P:4d} units {units:9.3g}: {best[2]:7.4f} / {best[0]:7.4f} = {ratio:5.2f}   "
                      f"default {'staged  ' if staged_default else 'complete'} {best[1]:7.4f} = {chosen:4.2f} x the better form", flush=True)
print(f"# worst default / better form: {worst:.3f}")
