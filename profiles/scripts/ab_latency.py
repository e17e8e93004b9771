#!/usr/bin/env python3
"""Single-pair latency A/B of library builds in ONE process on one box (box-to-box the figure moves by +-2 us):
the variants alternate A B C A B C ..., each turn = N device-resident calls of one 2000-keypoint pair, timing off.

    python3 profiles/scripts/ab_latency.py [--rounds 10] [--calls 300] name=path/to/lib.so[:opt=val,opt=val] ...
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
seq = synth.make_sequence(2, 2000, config=3, index=0)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
vs = []
for v in a.variants:
    name, rest = v.split("=", 1)
    path, _, opts = rest.partition(":")
    c = api.Context(0, lib=os.path.join(ROOT, path) if path else None)
    for o in filter(None, opts.split(",")):
        k, val = o.split("=")
        c.set_option(k, int(val))
    vs.append((name, c, PairBatchDevice(seq["pairs"], fs.max_kpts)))
for label, ev, est, H in (("E0/ransac/487", 0, EST_RANSAC, 487), ("E1/fixed/4096", 1, EST_FIXED, 4096)):
    prm = default_ransac_params(ev)
    cfg, _ = make_config(est, H, seed=3)
    t = {n: [] for n, _, _ in vs}
    enq = {n: [] for n, _, _ in vs}  # host time to queue one call (the loop is bound by the device when this is well below the total)
    for r in range(a.rounds + 1):
        for name, c, pb in vs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.calls):
                run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            if r:
                t[name].append((time.perf_counter() - t0) / a.calls * 1e6)
                enq[name].append((t1 - t0) / a.calls * 1e6)
    outs = [pb.download() for _, _, pb in vs]
    same = all(o["pose"].tobytes() == outs[0]["pose"].tobytes() and o["stats"].tobytes() == outs[0]["stats"].tobytes() for o in outs)
    print(label, " ".join(f"{n}: median {np.median(x):.1f} us (min {min(x):.1f}, host enqueue {np.median(enq[n]):.1f})" for n, x in t.items()),
          "| same results:", same, flush=True)
