"""Staged scoring statistics (GPU box): share of the complete (hypothesis x match) sweep that is still evaluated."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import *
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
seq2 = synth.make_sequence(41, 2000, config=3, index=0)
out = {}
for ev, estn, Hh, name in ((1, EST_FIXED, 4096, "E1/fixed/4096"), (0, EST_FIXED, 4096, "E0/fixed/4096"), (1, EST_RANSAC, 487, "E1/ransac/487"),
                           (0, EST_RANSAC, 487, "E0/ransac/487"), (2, EST_FIXED, 4096, "E2/fixed/4096"), (4, EST_FIXED, 4096, "E4/fixed/4096")):
    ev_made = []
    for prune in (1, 0):
        c = api.Context(0); c.set_option("score_stats", 1); c.set_option("prune", prune)
        fs = FrameSetDevice(seq2["desc"], seq2["pts"], seq2["nkpts"]); pb = PairBatchDevice(seq2["pairs"], fs.max_kpts)
        cfg2, _ = make_config(estn, Hh, seed=5)
        run_pairs(c, default_ransac_params(ev), cfg2, TUM_FR1_K, fs, pb); g = pb.download()
        ev_made.append(c.score_stats_ex()[1])
    out[name] = {"evaluations_staged": ev_made[0], "evaluations_complete": ev_made[1], "frac": ev_made[0] / max(ev_made[1], 1),
                 "mean_best_count": float(g["stats"]["bestInlierCount"].mean()), "mean_valid_matches": float(g["stats"]["numMatchesValid"].mean())}
print(json.dumps(out))
