"""Kernel 3 twins for the reprojection metric: the decision-exact kernel (ps_ransac_score_fast, default) and the value-exact
kernel (ps_ransac_score<1>) must give the oracle's inlier count for EVERY hypothesis
(reference src/TransformEst/RANSAC.cpp:325-375), over thresholds, camera scales, noise levels and degenerate data."""
import numpy as np
import pytest

from putslam_amd import api, synth
from putslam_amd._abi import (DMATCH_DTYPE, EST_FIXED, EST_RANSAC, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params,
                              make_config)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fctx():
    c = api.Context(0)
    c.set_option("score", 1)
    c.set_option("score_stats", 1)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ectx():
    c = api.Context(0)
    c.set_option("score", 0)
    yield c
    c.close()


def _counts(fctx, ectx, oracle, prm, cfg, K, a, b, m):
    g = fctx.debug_ransac_counts(prm, cfg, K, a["pts"], b["pts"], m)
    parked, evals = fctx.score_stats()
    e = ectx.debug_ransac_counts(prm, cfg, K, a["pts"], b["pts"], m)
    c, M = oracle.hypothesis_counts(prm, cfg, K, a["pts"], b["pts"], m)
    assert np.array_equal(e, c), "value-exact kernel differs from the oracle"
    assert np.array_equal(g, c), "decision-exact kernel differs from the oracle"
    assert parked <= evals      # (the share itself is checked at the shipped threshold below)
    return parked, evals, M


@pytest.mark.parametrize("thr", [0.05, 0.5, 2.0, 2.0000001, 7.3, 40.0, 900.0])
def test_score_variants_thresholds(fctx, ectx, oracle, thr):
    a, b = synth.make_pair(900, config=2, index=4000 + int(thr * 10))
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(REPROJECTION_ERROR)
    prm.inlierThresholdReprojection = thr
    cfg, _ = make_config(EST_FIXED, 3000, seed=17)
    parked, evals, M = _counts(fctx, ectx, oracle, prm, cfg, TUM_FR1_K, a, b, m)
    assert M > 100 and evals >= 3000 * M
    if thr == 2.0:
        assert 0 < parked < 0.02 * evals, (parked, evals)    # the band is narrow: a fraction of a per cent is re-done


@pytest.mark.parametrize("scale", [1e-3, 0.03, 1.0, 37.0, 1e3])
def test_score_variants_camera_scale(fctx, ectx, oracle, scale):
    a, b = synth.make_pair(600, config=2, index=4100)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 2048, seed=23)
    K = (TUM_FR1_K * np.float32(scale)).astype(np.float32)
    _counts(fctx, ectx, oracle, prm, cfg, K, a, b, m)


@pytest.mark.parametrize("noise,frac", [(1e-4, 0.9), (0.004, 0.7), (0.02, 0.25), (0.05, 0.1)])
def test_score_variants_noise(fctx, ectx, oracle, noise, frac):
    a, b = synth.make_pair(1000, config=2, index=4200, inlier_frac=frac, noise=noise)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 4096, seed=31)
    _counts(fctx, ectx, oracle, prm, cfg, TUM_FR1_K, a, b, m)


def test_score_variants_degenerate_geometry(fctx, ectx, oracle):
    """Points at the depth-filter limits, on the optical axis, coincident, huge lateral offsets, and a frame whose
    points project next to the principal point: projected depths near zero and invalid 3-point models must be decided
    by the value-exact code (or never matter), never by the band."""
    rng = np.random.default_rng(99)
    n = 400
    prev = np.zeros((n, 3), np.float32)
    prev[:, 2] = rng.choice(np.float32([0.1, 0.1000001, 0.5, 3.0, 6.0, 5.999999]), n)
    prev[:, 0] = (rng.standard_normal(n) * rng.choice([1e-6, 0.01, 1.0, 50.0], n)).astype(np.float32)
    prev[:, 1] = (rng.standard_normal(n) * rng.choice([1e-6, 0.01, 1.0, 50.0], n)).astype(np.float32)
    prev[::17] = prev[1]                                    # coincident points -> invalid (NaN) samples
    ang = np.deg2rad(170.0)                                 # a near-half-turn about y: depths change sign for many points
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    cur = ((prev.astype(np.float64) - [0.2, -0.1, 3.0]) @ R).astype(np.float32)
    cur[:, 2] = np.clip(np.abs(cur[:, 2]), 0.1, 6.0)
    m = np.zeros(n, DMATCH_DTYPE)
    m["queryIdx"] = np.arange(n)
    m["trainIdx"] = rng.permutation(n)
    m["distance"] = 10.0
    a, b = dict(pts=prev), dict(pts=cur)
    for thr in (2.0, 25.0):
        prm = default_ransac_params(REPROJECTION_ERROR)
        prm.inlierThresholdReprojection = thr
        cfg, _ = make_config(EST_FIXED, 4096, seed=3)
        _counts(fctx, ectx, oracle, prm, cfg, TUM_FR1_K, a, b, m)


def test_score_variants_full_results(fctx, ectx, oracle):
    """Whole RANSAC call (selection, refit, final mask, pose, stats) through both kernels vs the oracle."""
    for idx in range(4):
        a, b = synth.make_pair(1200, config=2, index=4300 + idx, inlier_frac=0.5, noise=0.006)
        m = oracle.match_hamming256(a["desc"], b["desc"])
        prm = default_ransac_params(REPROJECTION_ERROR)
        cfg, _ = make_config(EST_RANSAC, 1157, seed=idx)
        c = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
        for ctx in (fctx, ectx):
            g = ctx.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
            assert np.array_equal(g["mask"], c["mask"]) and g["pose"].tobytes() == c["pose"].tobytes()
            assert g["stats"]["bestHypothesis"] == c["stats"]["bestHypothesis"]
            assert g["stats"]["bestInlierCount"] == c["stats"]["bestInlierCount"]


@pytest.mark.parametrize("thr", [2.0, 0.05])
def test_score_variants_batch_on_a_full_chip(oracle, thr):
    """Both kernels over a batch large enough that several work-groups share every CU: identical results, run to run and
    kernel to kernel (the configuration that exposed a source-operand hazard of round 2's matrix-core scoring experiment,
    profiles/variants/ps_score_mfma.h.txt -- every single-pair test, at most one work-group per CU, had passed).  At 0.05 px
    so many evaluations fall inside the band that the kernel drains its queue from inside the loop."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(33, 2000, config=3, index=1)
    prm = default_ransac_params(REPROJECTION_ERROR)
    prm.inlierThresholdReprojection = thr
    cfg, _ = make_config(EST_FIXED, 4096, seed=42)
    outs = {}
    for score in (1, 1, 0):
        c = api.Context(0)
        c.set_option("score", score)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        c.close()
        if "ref" not in outs:
            outs["ref"] = g
            continue
        r = outs["ref"]
        assert g["pose"].tobytes() == r["pose"].tobytes(), "score=%d" % score
        assert np.array_equal(g["inlierMask"], r["inlierMask"]), "score=%d" % score
        for k in ("bestHypothesis", "bestInlierCount", "numInliers", "numMatchesValid"):
            assert np.array_equal(g["stats"][k], r["stats"][k]), (score, k)
    # and one pair of the batch against the oracle, every hypothesis, at the automatic (large) match-range split
    p = 11
    cfgp, _ = make_config(EST_FIXED, 4096, seed=42 + p)
    c = api.Context(0)
    m = c.match_hamming256(seq["desc"][p], seq["desc"][p + 1])
    want, _ = oracle.hypothesis_counts(prm, cfgp, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1], m)
    for _ in range(3):
        got = c.debug_ransac_counts(prm, cfgp, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1], m)
        assert np.array_equal(got, want)
    c.close()
