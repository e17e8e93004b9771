#!/usr/bin/env python3
"""Counterpart of demos/demoUSAC.cpp:24-98,212-294: the same correspondences through RANSAC (TEST_Ransac) and
through USAC (TEST_USAC) with the demo's thresholds 0.02 / 2.0 / 0.0002 / 0.1 / 3 (:72-81).  The recorded
fixture of the reference (resources/USAC/*.features|.matches|.ransac) is not in its repository, so the input is
a synthetic pair in the same shape (features `id u v x y z`, matches `prevId curId`)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    from putslam_amd import api, synth
    from putslam_amd._abi import EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config
    a, b = synth.make_pair(500, config=2, index=164)
    ctx = api.Context(0)
    matches = ctx.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(0)
    prm.inlierThresholdEuclidean, prm.inlierThresholdReprojection = 0.02, 2.0
    prm.inlierThresholdMahalanobis, prm.minimalInlierRatioThreshold = 0.0002, 0.1
    out = {}
    for name, est, H in (("RANSAC", EST_RANSAC, 3911), ("USAC", EST_USAC, 4096)):
        cfg, _ = make_config(est, H, seed=7)
        r = ctx.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], matches)
        out[name] = r
        print(f"{name}: {int(r['stats']['numInliers'])} inliers of {int(r['stats']['numMatchesValid'])} valid matches, "
              f"{int(r['stats']['iterationsRun'])} iterations, best sample {int(r['stats']['bestHypothesis'])}")
        print(np.array2string(r["pose"], precision=5, suppress_small=True))
    return out, b["T_prev_from_cur"]


if __name__ == "__main__":
    main()
