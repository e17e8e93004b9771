"""Share of (hypothesis, match) evaluations the decision-exact scoring kernels hand to the value-exact code, per mode
(run on the GPU box: python3 profiles/scripts/parked_stats.py)."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config

ctx = api.Context(0)
ctx.set_option("score_stats", 1)
out = {}
for mode in (0, 1, 2, 4):
    tot_p = tot_e = 0
    for idx in range(6):
        a, b = synth.make_pair(2000, config=3, index=100 + idx)
        m = ctx.match_hamming256(a["desc"], b["desc"])
        prm = default_ransac_params(mode)
        cfg, _ = make_config(EST_FIXED, 4096, seed=idx)
        ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
        p, e = ctx.score_stats()
        tot_p += p
        tot_e += e
    out["errorVersion%d" % mode] = {"recounted": tot_p, "evaluations": tot_e, "frac": (tot_p / tot_e) if tot_e else None}
print(json.dumps(out))
