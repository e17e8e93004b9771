#!/usr/bin/env python3
"""A/B of library builds (or of option sets of one build) in ONE process on one box: kernel times differ by +-3 % from
process to process on this pool (clock state, placement), so the variants alternate inside a single run --
A B A B ... -- on the same device-resident data, each step as ONE launch chain with HIP events around every kernel.

    python3 profiles/scripts/ab_libs.py [--ev 1] [--est fixed] [--hyp 4096] [--pairs 499] [--rounds 8] \
        name=path/to/lib.so[:opt=val,opt=val] ...
"""
import argparse, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_split

ap = argparse.ArgumentParser()
ap.add_argument("--ev", type=int, default=1)
ap.add_argument("--est", default="fixed")
ap.add_argument("--hyp", type=int, default=4096)
ap.add_argument("--pairs", type=int, default=499)
ap.add_argument("--kpts", type=int, default=2000)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--inliers", type=float, default=0.7)
ap.add_argument("--noise", type=float, default=0.004)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
est = {"fixed": EST_FIXED, "ransac": EST_RANSAC, "usac": EST_USAC}[a.est]
seq = synth.make_sequence(a.pairs + 1, a.kpts, config=3, index=0, inlier_frac=a.inliers, noise=a.noise)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
P = len(seq["pairs"])
prm = default_ransac_params(a.ev)
chain = torch.cuda.Stream()
vs = []
for v in a.variants:
    name, rest = v.split("=", 1)
    path, _, opts = rest.partition(":")
    c = api.Context(0, lib=os.path.join(ROOT, path) if path else None)
    for o in filter(None, opts.split(",")):
        k, val = o.split("=")
        c.set_option(k, int(val))
    vs.append((name, c))
ref = None
times = {n: [] for n, _ in vs}
wall = {n: [] for n, _ in vs}
import time
for r in range(a.rounds + 1):
    for name, c in vs:
        c.enable_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run_pairs_split([c], [chain], prm, est, a.hyp, 0xB0B0, TUM_FR1_K, fs, pb, bounds=[0, P], join=False)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / a.steps * 1e3
        k = {kk: v[0] / max(v[1], 1) for kk, v in c.kernel_time_totals().items()}
        c.enable_timing(False)
        out = pb.download()
        sig = (out["pose"].tobytes(), out["inlierMask"].tobytes())
        if ref is None:
            ref = sig
        assert sig == ref, f"variant {name} produced different results"
        if r > 0:   # round 0 = warm-up
            times[name].append(k)
            wall[name].append(w)
for name, _ in vs:
    keys = sorted(times[name][0])
    med = {k: float(np.median([t[k] for t in times[name]])) for k in keys}
    print(f"{name:>14s}: step {np.median(wall[name]):.4f} ms (min {min(wall[name]):.4f})  kernel sum {sum(med.values()):.4f}  " +
          "  ".join(f"{k.replace('ps_', '')} {v:.4f}" for k, v in med.items()), flush=True)
