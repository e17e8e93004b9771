"""Pruned scoring (ps_score_euclid.h / ps_score_fast.h, launch B): hypotheses that provably cannot become records of the
sequential selection (RANSAC.cpp:438-455: strict '>' first-best; adaptive trip limit :450-453, USAC.h:944-971) are
abandoned early and hypotheses beyond the trip limit are never scored.  Every OUTPUT of the path -- selected hypothesis,
its count, iterations run, inlier mask, pose bytes, ratios -- must equal both the unpruned run's and the oracle's, for all
three schedules, good and bad data, and batches large enough for the pruned launch to be used -- with the stages sweeping
the matches in the reordered record (ps_stage_reorder: what the prefix's best hypotheses reject first) and in the original
order."""
import numpy as np
import pytest

from putslam_amd import api, synth
from putslam_amd._abi import (ADAPTIVE_ERROR, EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_AND_REPROJECTION_ERROR,
                              EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config)

pytestmark = pytest.mark.gpu

STAT_FIELDS = ("numMatchesIn", "numMatchesValid", "bestHypothesis", "bestInlierCount", "iterationsRun", "numInliers",
               "accepted", "bestInlierRatio", "pointInlierRatio")


def _run(seq, prm, cfg, prune, reorder=1):
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    c = api.Context(0)
    c.set_option("prune", prune)
    c.set_option("reorder", reorder)   # 1 = also under the adaptive schedules (the default reorders the fixed one only)
    assert c.get_option("reorder") == reorder
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    c.close()
    return g


def _same(a, b, P):
    assert a["pose"].tobytes() == b["pose"].tobytes()
    assert np.array_equal(a["numMatches"], b["numMatches"])
    for p in range(P):
        n = int(a["numMatches"][p])
        assert np.array_equal(a["inlierMask"][p, :n], b["inlierMask"][p, :n]), p
        for f in STAT_FIELDS:
            x, y = a["stats"][p][f], b["stats"][p][f]
            assert x == y or (np.isnan(x) and np.isnan(y)), (p, f, x, y)


CASES = [  # (errorVersion, estimator, H, frames, kpts, inlier_frac, noise)
    (EUCLIDEAN_ERROR, EST_FIXED, 4096, 24, 700, 0.70, 0.004),       # the shipped metric, good data
    (EUCLIDEAN_ERROR, EST_FIXED, 2000, 40, 500, 0.15, 0.02),        # bad data: almost nothing can be abandoned
    (EUCLIDEAN_ERROR, EST_RANSAC, 487, 300, 300, 0.70, 0.004),      # the reference's own schedule: limit << 256
    (EUCLIDEAN_ERROR, EST_RANSAC, 1157, 90, 400, 0.12, 0.03),       # LC schedule on hard data: limits beyond the prefix
    (EUCLIDEAN_ERROR, EST_USAC, 3000, 40, 500, 0.30, 0.01),
    (ADAPTIVE_ERROR, EST_FIXED, 3001, 30, 601, 0.50, 0.006),        # odd sizes
    (ADAPTIVE_ERROR, EST_RANSAC, 1157, 90, 400, 0.10, 0.02),
    (REPROJECTION_ERROR, EST_FIXED, 4096, 60, 700, 0.70, 0.004),   # (reprojection: staged from P (ceil(H/256) - 1) >= 768 on)
    (REPROJECTION_ERROR, EST_RANSAC, 1157, 200, 400, 0.12, 0.03),
    (REPROJECTION_ERROR, EST_USAC, 3000, 80, 500, 0.30, 0.01),
    (EUCLIDEAN_AND_REPROJECTION_ERROR, EST_FIXED, 2048, 120, 500, 0.60, 0.005),
    # few matches: the stage cuts (multiples of 64) come close to each other and to M -- no stage may be left empty while
    # a later one still expects its survivors (round 3: stage 2 was, with M - c1 < 16)
    (REPROJECTION_ERROR, EST_FIXED, 1161, 200, 270, 0.94, 0.0014),
    (EUCLIDEAN_ERROR, EST_FIXED, 1500, 120, 150, 0.80, 0.002),
    (ADAPTIVE_ERROR, EST_FIXED, 1161, 140, 90, 0.90, 0.001),
    # many hypotheses, few pairs (the stress configuration's shape): the survivor lists are long, the list stages loop over
    # them in several passes and with several work-groups per pair
    (REPROJECTION_ERROR, EST_FIXED, 20000, 12, 900, 0.45, 0.008),
    (EUCLIDEAN_ERROR, EST_FIXED, 30000, 5, 800, 0.35, 0.01),
]


@pytest.mark.parametrize("mode,est,H,frames,kpts,frac,noise", CASES)
def test_pruned_equals_unpruned_equals_oracle(oracle, mode, est, H, frames, kpts, frac, noise):
    seq = synth.make_sequence(frames, kpts, config=3, index=mode * 100 + est * 10 + (H % 7), inlier_frac=frac, noise=noise)
    P = len(seq["pairs"])
    prm = default_ransac_params(mode, lc=(H == 1157))
    cfg, _ = make_config(est, H, seed=77)
    pruned = _run(seq, prm, cfg, 1)
    full = _run(seq, prm, cfg, 0)
    _same(pruned, full, P)
    _same(_run(seq, prm, cfg, 1, reorder=0), full, P)   # the stages in the original match order
    # the oracle on a sample of the pairs (the whole batch for the small cases)
    idx = np.arange(P) if P <= 40 else np.unique(np.linspace(0, P - 1, 24).astype(int))
    # pair p of the batch uses seed + p: the oracle is run per sampled pair with that pair's seed
    for j, p in enumerate(idx):
        cfgp, _ = make_config(est, H, seed=77 + int(p))
        cp = oracle.vo_pairs(prm, cfgp, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][p:p + 1], threads=1)
        n = int(cp["numMatches"][0])
        assert np.array_equal(pruned["inlierMask"][p, :n], cp["inlierMask"][0, :n]), p
        assert pruned["pose"][p].tobytes() == cp["pose"][0].tobytes(), p
        for f in STAT_FIELDS:
            x, y = pruned["stats"][p][f], cp["stats"][0][f]
            assert x == y or (np.isnan(x) and np.isnan(y)), (p, f, x, y)


def test_prune_ties_and_late_records(oracle):
    """Adversarial for the abandon rule: many hypotheses with EQUAL counts (duplicated points make whole groups of
    samples identical) and the best hypothesis placed late -- the first of equals must still win, a later strictly better
    one must still be found."""
    from putslam_amd._abi import DMATCH_DTYPE  # noqa: F401
    rng = np.random.default_rng(8)
    frames, kpts = 70, 400
    seq = synth.make_sequence(frames, kpts, config=3, index=999, inlier_frac=0.6, noise=0.0)   # noise-free: exact ties
    # quantise the points so that many 3-point samples give bit-identical models and counts
    seq["pts"] = (np.round(seq["pts"] * 64) / 64).astype(np.float32)
    P = len(seq["pairs"])
    for mode in (EUCLIDEAN_ERROR, REPROJECTION_ERROR):
        for est, H in ((EST_FIXED, 4096), (EST_RANSAC, 1157)):
            prm = default_ransac_params(mode, lc=(H == 1157))
            prm.minimalInlierRatioThreshold = 0.02
            cfg, _ = make_config(est, H, seed=int(rng.integers(1, 1 << 30)))
            pruned = _run(seq, prm, cfg, 1)
            full = _run(seq, prm, cfg, 0)
            _same(pruned, full, P)
            _same(_run(seq, prm, cfg, 1, reorder=0), full, P)


def test_reordered_record_is_a_permutation_with_the_rejected_matches_in_front(oracle):
    """ps_stage_reorder only changes the ORDER the stages sweep the matches in: per pair the order is a permutation of the
    depth-valid matches, and the matches of its pre-test front are rejected by the pair's selected hypothesis (one of the
    voters unless a later hypothesis beat the prefix -- then nearly all; the property asserted is the permutation, the
    front is reported)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(60, 700, config=3, index=4242, inlier_frac=0.7, noise=0.004)
    P = len(seq["pairs"])
    for mode in (REPROJECTION_ERROR, EUCLIDEAN_ERROR):
        prm = default_ransac_params(mode)
        cfg, _ = make_config(EST_FIXED, 4096, seed=5)
        c = api.Context(0)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        perm, front = c.stage_order(P, fs.max_kpts)
        c.close()
        for p in range(P):
            M = int(g["stats"][p]["numMatchesValid"])
            assert sorted(perm[p, :M].tolist()) == list(range(M)), (mode, p)
            assert 0 <= front[p] <= M
        if mode == REPROJECTION_ERROR:
            assert front.sum() > 0   # (how much of the miss budget is FAR off depends on the data: 40 % here, 75 % in the bench)
        else:
            assert (front == 0).all()   # (the pre-test front exists for the reprojection metrics only)
