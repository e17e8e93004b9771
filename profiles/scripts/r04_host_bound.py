#!/usr/bin/env python3
"""Is the timed region of bench.py bound by the host's submission rate?  The same step as bench.py (3 chains, 499 pairs x 2000
keypoints, H = 4096, errorVersion 1), timed (a) to the return of the LAST enqueue of N steps and (b) to the closing synchronize."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_split

seq = synth.make_sequence(500, 2000, config=3, index=0)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
P = len(seq["pairs"])
for S in (1, 2, 3, 4):
    ctxs = [api.Context(0) for _ in range(S)]
    chains = [torch.cuda.Stream() for _ in range(S)]
    bounds = [P * i // S for i in range(S + 1)]
    for ev, est, H in ((1, EST_FIXED, 4096), (0, EST_FIXED, 4096), (0, EST_RANSAC, 487)):
        prm = default_ransac_params(ev)
        def step():
            run_pairs_split(ctxs, chains, prm, est, H, 0xB0B0, TUM_FR1_K, fs, pb, bounds=bounds, join=False)
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        N = 100
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"chains {S} E{ev} est {est} H {H}: enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, total {1e3 * (t2 - t0) / N:.3f} ms/step "
              f"-> {P * N / (t2 - t0):.0f} pairs/s", flush=True)
    for c in ctxs:
        c.close()
