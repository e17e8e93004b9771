#!/usr/bin/env python3
"""Staged (default) against complete scoring (option prune = 0) on data the defaults were NOT tuned on: inlier ratio and
noise sweeps, errorVersion 0 and 1, 200 pairs x 2000 keypoints, H = 4096 fixed.  Prints the scoring step's time for both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

for frac, noise in ((0.9, 0.002), (0.7, 0.004), (0.5, 0.006), (0.3, 0.01), (0.15, 0.02), (0.05, 0.02)):
    seq = synth.make_sequence(201, 2000, config=9, index=int(frac * 100), inlier_frac=frac, noise=noise)
    for ev in (1, 0):
        prm = default_ransac_params(ev)
        cfg, _ = make_config(EST_FIXED, 4096, seed=3)
        row = []
        for prune in (1, 0):
            c = api.Context(0)
            c.set_option("prune", prune)
            fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
            pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
            for _ in range(3):
                run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
            c.synchronize()
            c.enable_timing(True)
            for _ in range(6):
                run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
            c.synchronize()
            t = {k: v[0] / max(v[1], 1) for k, v in c.kernel_time_totals().items()}
            g = pb.download()
            row.append(t.get("ps_ransac_score", float("nan")))
            c.close()
        st = g["stats"]
        print(f"inliers {frac:4.2f} noise {noise:5.3f} E{ev}: staged {row[0]:.3f} ms  complete {row[1]:.3f} ms  ratio {row[0] / row[1]:.2f}   "
              f"(M {st['numMatchesValid'].mean():.0f}, best {st['bestInlierCount'].mean():.0f}, accepted {int(st['accepted'].sum())}/200)", flush=True)
