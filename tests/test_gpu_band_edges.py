"""Directed band-edge tests of the decision-exact scoring kernels (ps_score_fast.h, ps_score_euclid.h).

The kernels decide most (hypothesis, match) evaluations from a cheap value with a proven error band and hand the
evaluations INSIDE the band to the value-exact code.  Random data rarely lands on the band, so these tests put evaluations
exactly on the decision boundary: for chosen (hypothesis, match) evaluations the oracle returns the error VALUE as the
reference computes it (RANSAC.cpp:266-272,346-366); the threshold is then set to the two neighbouring doubles between
which the reference's strict '<' flips for that evaluation, and one step further down.  Required each time:
  (a) every hypothesis's count equals the oracle's (so the boundary evaluations were decided like the reference), and
  (b) ps_debug_score_stats reports at least as many value-exact re-evaluations as there are boundary evaluations
      (they were parked, not decided by the cheap value).
Each chosen match is present DUP times in the match list (identical points -> identical error under the same
hypothesis), so one launch carries DUP boundary evaluations.

Also here: the limits where the kernels' bounds stop holding (boundsOk, ps_score_fast.h / ps_score_euclid.h) --
coordinates / camera constants are swept across them; the wavefronts beyond run the value-exact loop, counts stay equal."""
import numpy as np
import pytest

from putslam_amd import api, synth
from putslam_amd._abi import (ADAPTIVE_ERROR, DMATCH_DTYPE, EST_FIXED, EUCLIDEAN_AND_REPROJECTION_ERROR, EUCLIDEAN_ERROR,
                              REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config)

pytestmark = pytest.mark.gpu

DUP = 25            # copies of the boundary match in the list
H = 64              # hypotheses per launch (one wavefront)
PICKS = 136         # distinct (hypothesis, match) evaluations per variant: 136 x 25 x 3 thresholds > 10 000 evaluations


def _ctx(score):
    c = api.Context(0)
    c.set_option("score", score)
    c.set_option("score_stats", 1)
    return c


def _scene(rng, n=150):
    prev = (rng.uniform(-1.5, 1.5, (n, 3)) + [0, 0, 3.2]).astype(np.float32)
    R, t = synth.random_motion(rng, 4.0, 0.08)
    cur = ((prev.astype(np.float64) - t) @ R + rng.normal(0, 0.004, (n, 3))).astype(np.float32)
    cur[:, 2] = np.clip(cur[:, 2], 0.2, 5.8)
    bad = rng.random(n) < 0.3
    cur[bad] = (rng.uniform(-1.5, 1.5, (int(bad.sum()), 3)) + [0, 0, 3.2]).astype(np.float32)
    return prev, cur


def _boundary(oracle, mode, which, T, K, pp, cp, e):
    """Neighbouring doubles (t0, t1): the reference's test of this evaluation fails at t0 and passes at t1."""
    def passes(thr):
        thrE, thrR = (thr, 1e5) if which == "E" else (1e5, thr)
        return bool(oracle.is_inlier(mode, T, K, pp, cp, thrE, thrR))
    t = float(e)
    if mode == ADAPTIVE_ERROR:
        t = float(e) / float(pp[2])
    for _ in range(64):
        if passes(t):
            break
        t = np.nextafter(t, np.inf)
    assert passes(t)
    for _ in range(64):
        lower = np.nextafter(t, -np.inf)
        if not passes(lower):
            return lower, t
        t = lower
    raise AssertionError("no boundary found")


VARIANTS = [  # (errorVersion, which threshold binds, "score" option)
    (REPROJECTION_ERROR, "R", 1),
    (EUCLIDEAN_ERROR, "E", 1), (ADAPTIVE_ERROR, "E", 1),
    (EUCLIDEAN_AND_REPROJECTION_ERROR, "E", 1), (EUCLIDEAN_AND_REPROJECTION_ERROR, "R", 1),
]


@pytest.mark.parametrize("mode,which,score", VARIANTS)
def test_evaluations_on_the_decision_boundary(oracle, mode, which, score):
    ctx = _ctx(score)
    rng = np.random.default_rng(1000 * mode + (7 if which == "R" else 0) + score)
    done = 0
    boundary_evals = 0
    scene_id = 0
    while done < PICKS:
        prev, cur = _scene(rng)
        n = prev.shape[0]
        scene_id += 1
        for _ in range(8):                       # several picks per scene
            if done >= PICKS:
                break
            i = int(rng.integers(0, n))          # the match that will sit on the boundary
            order = np.concatenate([np.arange(n), np.full(DUP - 1, i)])
            rng.shuffle(order)
            m = np.zeros(order.size, DMATCH_DTYPE)
            m["queryIdx"] = order
            m["trainIdx"] = order
            cfg, _ = make_config(EST_FIXED, H, seed=int(rng.integers(1, 1 << 30)))
            h = int(rng.integers(0, H))
            T, valid, _ = oracle.hypothesis_model(cfg, prev, cur, m, h)
            if T is None:
                continue
            err = oracle.eval_errors(T, TUM_FR1_K, prev[i], cur[i])
            e = err[0] if which == "E" else max(err[1], err[2])
            if not np.isfinite(e) or e <= 1e-9 or e > 1e5:
                continue
            t0, t1 = _boundary(oracle, mode, which, T, TUM_FR1_K, prev[i], cur[i], e)
            counts_at = []
            for thr in (t0, t1, np.nextafter(t0, -np.inf)):
                prm = default_ransac_params(mode)
                prm.minimalNumberOfMatches = 3
                if which == "E":
                    prm.inlierThresholdEuclidean = thr
                    prm.inlierThresholdReprojection = 1e5
                else:
                    prm.inlierThresholdReprojection = thr
                    prm.inlierThresholdEuclidean = 1e5
                got = ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, prev, cur, m)
                parked, evals = ctx.score_stats()
                want, M = oracle.hypothesis_counts(prm, cfg, TUM_FR1_K, prev, cur, m)
                assert np.array_equal(got, want), (mode, which, score, scene_id, h, i, thr)
                assert evals >= H * M, "the decision-exact path did not run"
                assert parked >= DUP, ("boundary evaluations were not handed to the value-exact code", parked, thr)
                counts_at.append(int(want[h]))
                boundary_evals += DUP
            # the reference's decision of exactly these DUP evaluations flips between the two neighbouring thresholds
            assert counts_at[1] - counts_at[0] >= DUP and counts_at[2] <= counts_at[0]
            done += 1
    assert boundary_evals >= 10000
    ctx.close()


@pytest.mark.parametrize("mode,score", [(REPROJECTION_ERROR, 1), (EUCLIDEAN_AND_REPROJECTION_ERROR, 1)])
def test_bounds_limits_reprojection_kernels(oracle, mode, score):
    """boundsOk of the reprojection kernels (ps_score_fast.h): S * fmaxK <= 2^40 and Umax <= 1e7 per wavefront / pair,
    camera constants <= 1e6.  Lateral coordinates and the focal length are swept across the limits: below them the
    decision-exact loop runs, beyond them the value-exact loop of the same kernel; counts equal the oracle's throughout."""
    ctx = _ctx(score)
    rng = np.random.default_rng(5)
    n = 300
    ran_fast = set()
    for lateral in (1.0, 1e3, 1.5e4, 2.5e4, 1e5, 1.2e6, 3e6, 1e9):
        for fscale in (1.0, 1e3, 1.9e3, 2.1e3):           # fx = 517.3 * fscale: 2.1e3 puts it beyond 1e6
            prev = (rng.uniform(-1, 1, (n, 3)) * [lateral, lateral, 1] + [0, 0, 3]).astype(np.float32)
            R, t = synth.random_motion(rng, 2.0, 0.03)
            cur = ((prev.astype(np.float64) - t) @ R).astype(np.float32)
            cur[:, 2] = np.clip(cur[:, 2], 0.1, 6.0)
            m = np.zeros(n, DMATCH_DTYPE)
            m["queryIdx"] = np.arange(n)
            m["trainIdx"] = np.arange(n)
            K = TUM_FR1_K.copy()
            K[0] *= fscale
            K[4] *= fscale
            prm = default_ransac_params(mode)
            prm.inlierThresholdEuclidean = 0.04 * max(lateral, 1.0) ** 0.5
            prm.inlierThresholdReprojection = 2.0 * fscale
            cfg, _ = make_config(EST_FIXED, 512, seed=int(lateral) % 1000 + int(fscale))
            got = ctx.debug_ransac_counts(prm, cfg, K, prev, cur, m)
            parked, evals = ctx.score_stats()
            want, M = oracle.hypothesis_counts(prm, cfg, K, prev, cur, m)
            assert np.array_equal(got, want), (lateral, fscale)
            ran_fast.add(evals > 0)
    assert ran_fast == {True, False}          # the sweep crossed the limits
    ctx.close()


@pytest.mark.parametrize("mode", [EUCLIDEAN_ERROR, ADAPTIVE_ERROR])
def test_bounds_limits_euclidean_kernel(oracle, mode):
    """boundsOk of ps_ransac_score_euclid: S and the pair's largest coordinate <= 1e15, S >= 1e-20."""
    ctx = _ctx(1)
    rng = np.random.default_rng(6)
    n = 257
    ran_fast = set()
    for lateral in (1.0, 1e6, 1e12, 3e14, 9e14, 1.1e15, 1e16, 1e18):
        prev = (rng.uniform(-1, 1, (n, 3)) * [lateral, lateral, 1] + [0, 0, 3]).astype(np.float32)
        R, t = synth.random_motion(rng, 2.0, 0.03)
        cur = ((prev.astype(np.float64) - t) @ R).astype(np.float32)
        cur[:, 2] = np.clip(cur[:, 2], 0.1, 6.0)
        m = np.zeros(n, DMATCH_DTYPE)
        m["queryIdx"] = np.arange(n)
        m["trainIdx"] = np.arange(n)
        prm = default_ransac_params(mode)
        prm.inlierThresholdEuclidean = min(0.04 * max(lateral, 1.0), 9e9)
        cfg, _ = make_config(EST_FIXED, 512, seed=11)
        got = ctx.debug_ransac_counts(prm, cfg, TUM_FR1_K, prev, cur, m)
        parked, evals = ctx.score_stats()
        want, M = oracle.hypothesis_counts(prm, cfg, TUM_FR1_K, prev, cur, m)
        assert np.array_equal(got, want), lateral
        ran_fast.add(evals > 0)
    assert ran_fast == {True, False}
    ctx.close()
