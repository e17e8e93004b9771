#!/usr/bin/env python3
"""Generates the committed golden vectors (tests/golden/*.npz) with the CPU oracle.

The reference ships no test vectors for this path (its only recorded fixture,
resources/USAC/*.features|.matches|.ransac, is absent from the repository) and cannot be built here
(OpenCV + Eigen missing), so the vectors pin the ORACLE: seeded synthetic inputs (putslam_amd/synth.py)
+ the oracle's outputs.  tests/test_golden.py checks the oracle (CPU) and the HIP path (GPU) against
them, so neither can drift silently.

    python tests/golden/make_golden.py          # rewrites the .npz files
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as po  # noqa: E402
from putslam_amd import synth  # noqa: E402
from putslam_amd._abi import (ADAPTIVE_ERROR, EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_AND_REPROJECTION_ERROR,  # noqa: E402
                              EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config)

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(64, 11), (500, 12), (2000, 13)]
MODES = [EUCLIDEAN_ERROR, REPROJECTION_ERROR, EUCLIDEAN_AND_REPROJECTION_ERROR, ADAPTIVE_ERROR]
ESTS = [(EST_RANSAC, 487), (EST_USAC, 600), (EST_FIXED, 512)]
SEED = 20261003


def build_case(n, index):
    a, b = synth.make_pair(n, config=2, index=index)
    out = dict(desc_a=a["desc"], desc_b=b["desc"], pts_a=a["pts"], pts_b=b["pts"], gt=b["T_prev_from_cur"])
    m = po.match_hamming256(a["desc"], b["desc"])
    out["matches"] = m
    for mode in MODES:
        prm = default_ransac_params(mode)
        cfg, _ = make_config(EST_FIXED, 512, seed=SEED)
        counts, M = po.hypothesis_counts(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
        out[f"counts_m{mode}"] = counts
        out[f"mvalid_m{mode}"] = np.int32(M)
        for est, H in ESTS:
            cfg, _ = make_config(est, H, seed=SEED)
            r = po.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
            k = f"m{mode}_e{est}"
            out[f"pose_{k}"] = r["pose"]
            out[f"mask_{k}"] = r["mask"]
            out[f"stats_{k}"] = np.array([r["stats"]])
    return out


def main():
    for n, index in CASES:
        path = os.path.join(HERE, f"pair_n{n}.npz")
        np.savez_compressed(path, **build_case(n, index))
        print("wrote", path, os.path.getsize(path), "bytes")
    # demoKabsch-shaped case (BASELINE config 1): 500 points, t = (0.1, 0.2, -0.3), R = I, sigma = (.01,.02,.03)
    rng = np.random.Generator(np.random.PCG64(SEED))
    A = rng.uniform(-1.5, 1.5, (500, 3))
    B = A + np.array([0.1, 0.2, -0.3]) + rng.normal(0, 1, (500, 3)) * [0.01, 0.02, 0.03]
    np.savez_compressed(os.path.join(HERE, "kabsch_demo.npz"), A=A, B=B, T=po.kabsch_f64(A, B))


if __name__ == "__main__":
    main()
