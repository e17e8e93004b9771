"""Pruned scoring statistics (GPU box): share of the sweep the pruned launch still computes, per regime."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import *
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
seq2 = synth.make_sequence(41, 2000, config=3, index=0)
for ev, estn, Hh in ((1, EST_FIXED, 4096), (0, EST_FIXED, 4096), (1, EST_RANSAC, 487), (2, EST_FIXED, 4096)):
    c = api.Context(0); c.set_option("score_stats", 1)
    fs = FrameSetDevice(seq2["desc"], seq2["pts"], seq2["nkpts"]); pb = PairBatchDevice(seq2["pairs"], fs.max_kpts)
    cfg2, _ = make_config(estn, Hh, seed=5)
    run_pairs(c, default_ransac_params(ev), cfg2, TUM_FR1_K, fs, pb); g = pb.download()
    st = c.score_stats_ex()
    print("E%d est %d H %d stats_ex" % (ev, estn, Hh), st, "computed/full = %.3f" % (st[2] / max(st[3], 1)),
          "mean best", g["stats"]["bestInlierCount"].mean(), "M", g["stats"]["numMatchesValid"].mean())
