import numpy as np, sys
sys.path.insert(0, '.')
from putslam_amd import api, synth
from putslam_amd._abi import *
ctx = api.Context(0)
seq = synth.make_sequence(33, 2000, config=3, index=1)
prm = default_ransac_params(REPROJECTION_ERROR)
for p in (11, 2):
    cfg, _ = make_config(EST_FIXED, 4096, seed=42 + p)
    m = ctx.match_hamming256(seq["desc"][p], seq["desc"][p + 1])
    ctx.set_option("score", 1)
    ref = ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1], m)
    ctx.set_option("score", 2)
    for ms in (0, 1, 2, 4, 8):
        ctx.set_option("msplit", ms)
        for rep in range(3):
            g = ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1], m)
            d = np.nonzero(g != ref)[0]
            print("pair", p, "msplit", ms, "rep", rep, "M", len(m), "differing hyps", len(d), d[:8], (g[d] - ref[d])[:8])
    ctx.set_option("msplit", 0)
