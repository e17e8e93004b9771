#!/usr/bin/env python3
"""Success rate of stage 1's one-direction pre-test on the bench workload (score_stats counters 2 / 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seq = synth.make_sequence(200, 2000, config=2, index=0)
prm = default_ransac_params(ev)
cfg, _ = make_config(EST_FIXED, 4096, seed=1)
c = api.Context(0)
c.set_option("score_stats", 1)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
c.synchronize()
st = c.score_stats_ex()
print("parked", st[0], "evals", st[1], "pretest ok", st[2], "pretest failed", st[3], "rate", st[2] / max(st[2] + st[3], 1))
