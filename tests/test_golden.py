"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle):
CPU leg keeps the oracle from drifting, GPU leg checks the HIP path against the same files."""
import os

import numpy as np
import pytest

from putslam_amd._abi import (ADAPTIVE_ERROR, EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_AND_REPROJECTION_ERROR,
                              EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config)

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = [EUCLIDEAN_ERROR, REPROJECTION_ERROR, EUCLIDEAN_AND_REPROJECTION_ERROR, ADAPTIVE_ERROR]
ESTS = [(EST_RANSAC, 487), (EST_USAC, 600), (EST_FIXED, 512)]
SEED = 20261003


def _load(n):
    return np.load(os.path.join(HERE, f"pair_n{n}.npz"))


def _check(impl, g, counts_fn):
    m = impl.match_hamming256(g["desc_a"], g["desc_b"])
    assert m.tobytes() == g["matches"].tobytes()
    for mode in MODES:
        prm = default_ransac_params(mode)
        cfg, _ = make_config(EST_FIXED, 512, seed=SEED)
        counts = counts_fn(prm, cfg, g["pts_a"], g["pts_b"], m)
        assert np.array_equal(counts, g[f"counts_m{mode}"])
        for est, H in ESTS:
            cfg, _ = make_config(est, H, seed=SEED)
            r = impl.ransac_rigid3d(prm, cfg, TUM_FR1_K, g["pts_a"], g["pts_b"], m)
            k = f"m{mode}_e{est}"
            assert np.array_equal(r["mask"], g[f"mask_{k}"]), k
            assert r["pose"].tobytes() == g[f"pose_{k}"].tobytes(), k
            gs = g[f"stats_{k}"][0]
            for f in gs.dtype.names:
                assert r["stats"][f] == gs[f] or (np.isnan(r["stats"][f]) and np.isnan(gs[f])), (k, f)


@pytest.mark.parametrize("n", [64, 500, 2000])
def test_oracle_reproduces_golden(oracle, n):
    g = _load(n)
    _check(oracle, g, lambda prm, cfg, pa, pb, m: oracle.hypothesis_counts(prm, cfg, TUM_FR1_K, pa, pb, m)[0])


def test_oracle_kabsch_golden(oracle):
    g = np.load(os.path.join(HERE, "kabsch_demo.npz"))
    T = oracle.kabsch_f64(g["A"], g["B"])
    assert np.array_equal(T, g["T"])
    # BASELINE config 1 (demoKabsch.cpp:23,25,1013): t = (0.1, 0.2, -0.3), R = I, sigma = (0.01, 0.02, 0.03), N = 500
    assert np.abs(T[:3, 3] - [0.1, 0.2, -0.3]).max() < 3 * 0.03 / np.sqrt(500) * 3
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 500, 2000])
def test_hip_reproduces_golden(ctx, n):
    g = _load(n)
    _check(ctx, g, lambda prm, cfg, pa, pb, m: ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, pa, pb, m))


@pytest.mark.gpu
def test_hip_kabsch_golden(ctx):
    g = np.load(os.path.join(HERE, "kabsch_demo.npz"))
    assert np.abs(ctx.kabsch_f64(g["A"], g["B"]) - g["T"]).max() < 1e-12
