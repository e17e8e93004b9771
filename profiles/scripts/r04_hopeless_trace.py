#!/usr/bin/env python3
"""One row of data_sweep.py (15 % inliers, errorVersion 0, 200 pairs) for a rocprofv3 kernel trace: staged (argv[1] = 1) or
complete (0) scoring, 10 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
prune = int(sys.argv[1]); frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15; ev = int(sys.argv[3]) if len(sys.argv) > 3 else 0
seq = synth.make_sequence(201, 2000, config=9, index=int(frac * 100), inlier_frac=frac, noise=0.02 if frac < 0.2 else 0.004)
prm = default_ransac_params(ev)
cfg, _ = make_config(EST_FIXED, 4096, seed=3)
c = api.Context(0)
c.set_option("prune", prune)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
for _ in range(10):
    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
c.synchronize()
