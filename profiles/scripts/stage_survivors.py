#!/usr/bin/env python3
"""How many hypotheses survive stages 1 and 2 of the staged scoring, and where the stages end, on the bench workload.
usage: python profiles/scripts/stage_survivors.py [errorVersion] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 500
seq = synth.make_sequence(frames, 2000, config=2, index=0)
prm = default_ransac_params(ev)
cfg, _ = make_config(EST_FIXED, 4096, seed=1)
c = api.Context(0)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
c.synchronize()
P = len(seq["pairs"])
s = c.stage_survivors(P)
g = pb.download()
M = g["stats"]["numMatchesValid"].astype(int)
best = g["stats"]["bestInlierCount"].astype(int)
for name, v in (("stage 1 survivors", s[0]), ("stage 2 survivors", s[1]), ("valid matches", M), ("best count", best), ("miss", M - best)):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100]).astype(int)
    print(f"{name:20s} mean {v.mean():8.1f}   min/p10/p50/p90/p99/max {q.tolist()}")
print("pairs with > 256 stage-2 survivors:", int((s[1] > 256).sum()), " > 1024 stage-1 survivors:", int((s[0] > 1024).sum()))
