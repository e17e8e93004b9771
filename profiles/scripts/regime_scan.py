#!/usr/bin/env python3
"""Looks for performance cliffs: the batched call over a grid of batch sizes, frame sizes, schedules and metrics -- time per call
and per pair, so that a regime that falls off the curve of its neighbours stands out (the USAC cap of 850 000 did, round 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

def measure(ctx, seq, ev, est, H, calls):
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"]); pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    prm = default_ransac_params(ev); cfg, _ = make_config(est, H, seed=3)
    for _ in range(5): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e3

ctx = api.Context(0)
SCHED = (("E0/ransac/487", 0, EST_RANSAC, 487), ("E1/fixed/4096", 1, EST_FIXED, 4096), ("E0/usac/850000", 0, EST_USAC, 850000),
         ("E2/fixed/1000", 2, EST_FIXED, 1000), ("E4/ransac/1157", 4, EST_RANSAC, 1157))
print("batch size (2000 keypoints): ms per call | us per pair")
for P in (1, 2, 4, 8, 16, 17, 32, 64, 128, 256, 499):
    seq = synth.make_sequence(P + 1, 2000, config=3, index=0)
    row = []
    for name, ev, est, H in SCHED:
        ms = measure(ctx, seq, ev, est, H, 20 if P > 64 else 60)
        row.append(f"{name} {ms:.3f} | {ms / P * 1e3:.1f}")
    print(f"P={P:4d}  " + "   ".join(row), flush=True)
print("frame size (8 pairs): ms per call")
for n in (64, 500, 1000, 2000, 4000, 8000, 16384):
    seq = synth.make_sequence(9, n, config=3, index=1)
    row = []
    for name, ev, est, H in SCHED:
        ms = measure(ctx, seq, ev, est, H, 20)
        row.append(f"{name} {ms:.3f}")
    print(f"kpts={n:5d}  " + "   ".join(row), flush=True)
