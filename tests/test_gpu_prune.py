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

# launch-shape / tuning knobs: every value gives the same results; addressed as "debug.<name>" (include/putslam_hip.h)
SHAPE_KNOBS = ("qsplit", "msplit", "gensplit", "singlerest", "pretest", "list_g2", "prefix", "reorder_top", "reorder_margin")
RETIRED_KNOBS = ("list_r3", "list_g3", "reorder_c2div", "reorder_gran")   # constants of the library since round 6
STAT_FIELDS = ("numMatchesIn", "numMatchesValid", "bestHypothesis", "bestInlierCount", "iterationsRun", "numInliers",
               "accepted", "bestInlierRatio", "pointInlierRatio")


def _run(seq, prm, cfg, prune, reorder=1, **options):
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    c = api.Context(0)
    c.set_option("prune", 2 if prune else 0)   # 2 = the staged form whatever the batch size (the default, 1, asks the cost model)
    c.set_option("reorder", reorder)   # 1 = also under the adaptive schedules (the default reorders the fixed one only)
    assert c.get_option("reorder") == reorder
    for name, value in options.items():   # the staged scoring's twins and launch-shape knobs ("debug.<name>", ps_context_set_option)
        name = "debug." + name if name in SHAPE_KNOBS else name
        c.set_option(name, value)
        assert c.get_option(name) == value
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
            # 256 model slots per pair: the late winner has none and is rebuilt by kernel 4 (ModelArgs::modelH)
            _same(_run(seq, prm, cfg, 1, model_room_mib=1), full, P)


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
        c.set_option("prune", 2)
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


# ---- every twin / tuning knob of the staged scoring, per context (round 3 could only set them through the environment) ----
# The selection rule all of them must preserve: strict '>' first-best with the adaptive trip limit, RANSAC.cpp:438-455.
TWINS = [
    dict(pretest=0), dict(gensplit=0), dict(singlerest=0), dict(bail=0),
    dict(prefix=64), dict(prefix=128),
    dict(list_g2=1), dict(list_g2=7), dict(list_g2=64, pretest=0),
    dict(reorder_top=1), dict(reorder_top=16, reorder_margin=1), dict(reorder_margin=300),
    dict(msplit=3), dict(msplit=32, gensplit=0),
    # parked models in proportion to the work (round 5): with room for 1 MiB of them only the leading few hundred hypotheses of
    # every pair have a slot, the others are swept in one piece by stage 1 and rebuilt by kernel 4 when one of them wins
    dict(model_room_mib=1), dict(model_room_mib=1, gensplit=0, pretest=0),
]
TWIN_CASES = [  # (errorVersion, estimator, H, frames, kpts, inlier_frac, noise)
    (EUCLIDEAN_ERROR, EST_FIXED, 4096, 24, 700, 0.70, 0.004),
    (REPROJECTION_ERROR, EST_FIXED, 4096, 60, 700, 0.70, 0.004),
    (EUCLIDEAN_AND_REPROJECTION_ERROR, EST_FIXED, 2048, 120, 500, 0.45, 0.008),
    (ADAPTIVE_ERROR, EST_RANSAC, 1157, 90, 400, 0.10, 0.02),
    (REPROJECTION_ERROR, EST_USAC, 3000, 80, 500, 0.30, 0.01),
    (REPROJECTION_ERROR, EST_FIXED, 1161, 200, 270, 0.94, 0.0014),   # cuts close to each other and to M
    (EUCLIDEAN_ERROR, EST_FIXED, 2000, 40, 500, 0.12, 0.02),         # hopeless data: the one-stage exit of the reorder launch
]
_FULL = {}


@pytest.mark.parametrize("case", range(len(TWIN_CASES)))
@pytest.mark.parametrize("twin", TWINS, ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()))
def test_staged_twins_equal_complete(twin, case):
    mode, est, H, frames, kpts, frac, noise = TWIN_CASES[case]
    seq = synth.make_sequence(frames, kpts, config=3, index=7000 + case, inlier_frac=frac, noise=noise)
    P = len(seq["pairs"])
    prm = default_ransac_params(mode, lc=(H == 1157))
    cfg, _ = make_config(est, H, seed=1234)
    if case not in _FULL:   # the complete sweep (prune = 0) once per case; test_pruned_equals_unpruned_equals_oracle ties it to the oracle
        _FULL[case] = _run(seq, prm, cfg, 0)
    _same(_run(seq, prm, cfg, 1, reorder=1, **twin), _FULL[case], P)
    if "msplit" not in twin:
        _same(_run(seq, prm, cfg, 1, reorder=2, **twin), _FULL[case], P)   # the default: reordered for the fixed schedule only


def test_option_names_and_ranges():
    c = api.Context(0)
    for name in ("matcher", "matcher_fused", "score", "score_stats", "prune", "side_by_side", "reorder", "bail", "stamps", "stream_copy_kernels",
                 "model_room_mib") + tuple("debug." + k for k in SHAPE_KNOBS):
        v = c.get_option(name)
        c.set_option(name, v)   # every default is a legal value
    for name, bad in (("debug.prefix", 100), ("debug.prefix", 320), ("debug.list_g2", 65), ("debug.reorder_top", 17), ("prune", 3),
                      ("side_by_side", 17), ("side_by_side", -1), ("no_such_option", 0), ("list_g2", 4), ("msplit", 0), ("debug.prune", 1)) + \
            tuple(("debug." + k, 4) for k in RETIRED_KNOBS):   # (the two families do not answer to each other's names; retired knobs are gone)
        with pytest.raises(api.PsError):
            c.set_option(name, bad)
    c.close()


# ---- the staged scoring behind the host-pointer and streaming entry points (one pair, very many hypotheses) ----
def _one_pair(n, index, frac=0.6, noise=0.004):
    a, b = synth.make_pair(n, config=2, index=index, inlier_frac=frac, noise=noise)
    return a, b


@pytest.mark.parametrize("mode,est,H", [
    (EUCLIDEAN_ERROR, EST_FIXED, 70000),          # staged from ceil(H/256) - 1 >= 256 on (Euclidean kernels)
    (ADAPTIVE_ERROR, EST_USAC, 850000),           # USAC's own default maxHypotheses (USAC_wrapper.cpp:70)
    (REPROJECTION_ERROR, EST_FIXED, 200000),      # ... >= 768 (reprojection kernels, fixed schedule)
    (REPROJECTION_ERROR, EST_USAC, 850000),
])
def test_host_entry_with_staged_scoring_two_calls_one_context(oracle, mode, est, H):
    """ps_ransac_rigid3d on ONE context, twice (the second call meets the survivor counters and lists the first left), and a
    smaller call in between: staged == complete == oracle.  Round 3's host entry never cleared the counters."""
    a, b = _one_pair(260, 31)
    a2, b2 = _one_pair(420, 32, frac=0.35, noise=0.01)
    prm = default_ransac_params(mode)
    prm.minimalInlierRatioThreshold = 0.05
    K = TUM_FR1_K
    staged, full = api.Context(0), api.Context(0)
    staged.set_option("reorder", 1)
    staged.set_option("prune", 2)
    full.set_option("prune", 0)
    for rep, (x, y, seed) in enumerate(((a, b, 5), (a2, b2, 6), (a, b, 5), (a2, b2, 7))):
        cfg, _ = make_config(est, H, seed=seed)
        m = oracle.match_hamming256(x["desc"], y["desc"])
        g = staged.ransac_rigid3d(prm, cfg, K, x["pts"], y["pts"], m)
        f = full.ransac_rigid3d(prm, cfg, K, x["pts"], y["pts"], m)
        o = oracle.ransac_rigid3d(prm, cfg, K, x["pts"], y["pts"], m)
        for r in (f, o):
            assert np.array_equal(g["mask"], r["mask"]) and g["pose"].tobytes() == r["pose"].tobytes(), (rep, mode, est)
            for fld in STAT_FIELDS:
                u, v = g["stats"][fld], r["stats"][fld]
                assert u == v or (np.isnan(u) and np.isnan(v)), (rep, fld, u, v)
    staged.close()
    full.close()


def test_debug_counts_are_complete_counts_whatever_the_size(oracle):
    """ps_debug_ransac_counts returns every hypothesis's own count: never the staged scoring's lower bounds."""
    a, b = _one_pair(200, 41)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 70000, seed=9)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    c = api.Context(0)
    c.set_option("reorder", 1)
    c.set_option("prune", 2)
    c.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)          # a staged call first
    cg = c.debug_ransac_counts(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
    cc, _ = oracle.hypothesis_counts(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
    assert np.array_equal(cg, cc[: len(cg)])
    assert np.array_equal(c.stage_survivors(1), np.zeros((2, 1), np.int32))   # that call was not staged
    c.close()


@pytest.mark.parametrize("mode,est,H", [(EUCLIDEAN_ERROR, EST_FIXED, 70000), (REPROJECTION_ERROR, EST_USAC, 850000),
                                        (REPROJECTION_ERROR, EST_FIXED, 200000)])
def test_stream_push_with_staged_scoring(oracle, mode, est, H):
    """ps_vo_stream_push (one pair per push; hipGraph replay from the third push on) with the staged scoring on, while
    another call on the SAME context grows staged-scoring blocks in between (round 3's graph key did not see them)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(7, 300, config=3, index=515, inlier_frac=0.6, noise=0.004)
    big = synth.make_sequence(40, 300, config=3, index=516, inlier_frac=0.6, noise=0.004)
    prm = default_ransac_params(mode)
    prm.minimalInlierRatioThreshold = 0.05
    c = api.Context(0)
    c.set_option("reorder", 1)
    c.set_option("prune", 2)
    vs = api.VoStream(c, 300)
    for f in range(7):
        cfg, _ = make_config(est, H, seed=100 + f)
        r = vs.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f])
        if f == 0:
            assert r is None
            continue
        if f == 4:   # a batch with more pairs on the same context: validMask / survN / prefInfo / permBuf grow
            cfgb, _ = make_config(EST_FIXED, 4096, seed=1)
            fs = FrameSetDevice(big["desc"], big["pts"], big["nkpts"])
            pb = PairBatchDevice(big["pairs"], fs.max_kpts)
            run_pairs(c, prm, cfgb, TUM_FR1_K, fs, pb)
            pb.download()
        o = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][f - 1:f], threads=1)
        n = int(o["numMatches"][0])
        assert len(r["matches"]) == n and np.array_equal(r["mask"], o["inlierMask"][0, :n]), f
        assert r["pose"].tobytes() == np.ascontiguousarray(o["pose"][0].reshape(4, 4).T).tobytes(), f
        for fld in STAT_FIELDS:
            u, v = r["stats"][fld], o["stats"][0][fld]
            assert u == v or (np.isnan(u) and np.isnan(v)), (f, fld, u, v)
    vs.close()
    c.close()


def test_stage_diagnostics_reject_another_batch_shape():
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(30, 400, config=3, index=61, inlier_frac=0.7, noise=0.004)
    c = api.Context(0)
    c.set_option("prune", 2)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 4096, seed=3)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
    pb.download()
    P = len(seq["pairs"])
    assert c.stage_survivors(P).shape == (2, P)
    c.stage_order(P, fs.max_kpts)
    with pytest.raises(api.PsError):
        c.stage_survivors(P - 1)
    with pytest.raises(api.PsError):
        c.stage_order(P - 1, fs.max_kpts)
    c.close()


def test_nothing_to_gain_policy_switches_to_complete_scoring_and_probes(oracle):
    """Option "bail" (Euclidean metrics, batched calls): on data whose prefix leaves nothing to abandon the context scores the
    next calls completely and looks again with the staged form every 16th call; on good data it stays staged.  Every call's
    outputs are identical (selection rule RANSAC.cpp:438-455 untouched: the policy only chooses between two bit-identical forms)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    bad = synth.make_sequence(41, 500, config=9, index=15, inlier_frac=0.12, noise=0.02)
    good = synth.make_sequence(41, 500, config=9, index=70, inlier_frac=0.7, noise=0.004)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 4096, seed=11)
    c = api.Context(0)
    c.set_option("prune", 2)

    def call(seq):
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()       # (synchronises: the observation of this call has landed before the next one is planned)
        return g, c.get_option("last_staged_pairs") > 0

    ref_bad = _run(bad, prm, cfg, 0)
    ref_good = _run(good, prm, cfg, 0)
    staged = []
    for i in range(40):
        g, was_staged = call(bad)
        _same(g, ref_bad, len(bad["pairs"]))
        staged.append(was_staged)
    assert staged[0] and c.get_option("hopeless") == 1
    assert sum(staged) <= 5 and any(staged[2:])          # complete scoring except the first call(s) and the periodic probes
    staged = []
    for i in range(40):
        g, was_staged = call(good)
        _same(g, ref_good, len(good["pairs"]))
        staged.append(was_staged)
    assert c.get_option("hopeless") == 0 and all(staged[-16:])   # a probe saw the good data: staged again, and stays so
    c.set_option("bail", 0)
    c2 = api.Context(0)
    c2.set_option("bail", 0)
    c2.set_option("prune", 2)
    c.close()
    c, staged = c2, []
    for i in range(6):
        g, was_staged = call(bad)
        _same(g, ref_bad, len(bad["pairs"]))
        staged.append(was_staged)
    assert all(staged)                                    # policy off: always the staged form
    c.close()


def test_nothing_to_gain_policy_is_kept_per_kind_of_call(oracle):
    """one context alternating a mostly failing batch (loop-closure candidates) with a good VO batch.  The
    policy's state belongs to the kind of call (metric, schedule, H, batch-size class, frame set): the good batches stay staged
    on every call, the bad ones go to complete scoring; an adaptive schedule never leaves the staged form; setting the option
    resets what was observed."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    bad = synth.make_sequence(41, 500, config=9, index=15, inlier_frac=0.12, noise=0.02)
    good = synth.make_sequence(41, 500, config=9, index=70, inlier_frac=0.7, noise=0.004)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 4096, seed=11)
    c = api.Context(0)
    c.set_option("prune", 2)
    sets = {}
    for name, seq in (("bad", bad), ("good", good)):       # two frame sets that stay resident: two kinds of call
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        sets[name] = (seq, fs, PairBatchDevice(seq["pairs"], fs.max_kpts), _run(seq, prm, cfg, 0))

    def call(name, cfg_=cfg):
        seq, fs, pb, ref = sets[name]
        run_pairs(c, prm, cfg_, TUM_FR1_K, fs, pb)
        g = pb.download()
        if cfg_ is cfg:
            _same(g, ref, len(seq["pairs"]))
        return c.get_option("last_staged_pairs") > 0

    staged = {"bad": [], "good": []}
    for i in range(24):
        for name in ("bad", "good"):
            staged[name].append(call(name))
    assert all(staged["good"])                                   # never dragged into complete scoring by its neighbour
    assert staged["bad"][0] and sum(staged["bad"]) <= 4          # first call + the periodic probe
    c.set_option("bail", 1)                                      # forgets the observations
    assert call("bad") and c.get_option("hopeless") == 0
    # an adaptive schedule with reordering forced on: the policy does not apply, always staged (long caps: complete scoring
    # would run for minutes)
    c.set_option("reorder", 1)
    cfg_u, _ = make_config(EST_USAC, 3000, seed=11)
    assert all(call("bad", cfg_u) for _ in range(5))
    c.close()


def test_batch_under_the_reference_usac_cap_in_one_call(oracle):
    """USAC's cap of 850 000 (USAC_wrapper.cpp:70) with 212 pairs: round 4 parked one model per pair and cap entry (48 bytes:
    8.6 GB here) and took such a batch in slices.  Models are parked in proportion to the work now -- the leading hypotheses of
    every pair get a slot, a later one is swept in one piece and rebuilt if it wins -- so the batch is ONE staged call whose
    arena stays small; every pair equals the oracle (sample streams seeded with seed + pair index)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(213, 260, config=3, index=41, inlier_frac=0.6, noise=0.004)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_USAC, 850000, seed=0xABCDEF)
    c = api.Context(0)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    assert c.get_option("last_staged_pairs") == len(seq["pairs"])           # one call, staged
    slots = c.get_option("last_model_slots")
    assert 256 <= slots < 850000 and slots * 212 * 48 <= (256 << 20)         # models: 256 MB at most, not 8.6 GB
    assert c.get_option("arena_mib") < 1100                                  # counts (4 B x 212 x 850 000 = 721 MB) + models + records
    c.close()
    P = len(seq["pairs"])
    assert P == 212
    for p in (0, 1, 100, 208, 209, 210, 211):
        cfgp, _ = make_config(EST_USAC, 850000, seed=0xABCDEF + p)
        cp = oracle.vo_pairs(prm, cfgp, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][p:p + 1], threads=1)
        n = int(cp["numMatches"][0])
        assert int(g["numMatches"][p]) == n, p
        assert np.array_equal(g["inlierMask"][p, :n], cp["inlierMask"][0, :n]), p
        assert g["pose"][p].tobytes() == cp["pose"][0].tobytes(), p
        for f in STAT_FIELDS:
            x, y = g["stats"][p][f], cp["stats"][0][f]
            assert x == y or (np.isnan(x) and np.isnan(y)), (p, f, x, y)


@pytest.mark.parametrize("mode,est,H,kpts,below,above", [
    # threshold = base + perRow x capacity in units of pairs x (ceil(H / 256) - 1) x capacity (ps_capi.hip: kStagedFrom*)
    (EUCLIDEAN_ERROR, EST_FIXED, 4096, 500, 100, 140),       # 7.8e5 + 250 x 500 = 9.05e5: 121 pairs
    (REPROJECTION_ERROR, EST_FIXED, 4096, 400, 230, 270),    # 1.3e6 + 500 x 400 = 1.5e6: 250 pairs
    (EUCLIDEAN_ERROR, EST_RANSAC, 1157, 300, 40, 60),        # 6.0e4: 50 pairs (hb - 1 = 4)
    (REPROJECTION_ERROR, EST_USAC, 3000, 300, 10, 18),       # 4.5e4: 14 pairs (hb - 1 = 11)
])
def test_cost_model_picks_the_form_by_batch_size_and_both_forms_agree(mode, est, H, kpts, below, above):
    """Round 5: the default (option prune = 1) takes the staged form from the cost model's batch size on -- pairs x work-groups of
    hypotheses x frame capacity against base + perRow x capacity (profiles/r05c/staged_crossover.txt) -- and complete scoring
    below it.  Just below and just above the point: the form taken is the predicted one and the outputs are those of the other
    form (selection rule RANSAC.cpp:438-455 untouched either way)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    prm = default_ransac_params(mode, lc=(H == 1157))
    cfg, _ = make_config(est, H, seed=5150)
    for P, expect_staged in ((below, False), (above, True)):
        seq = synth.make_sequence(P + 1, kpts, config=3, index=8800 + P, inlier_frac=0.6, noise=0.005)
        c = api.Context(0)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        assert (c.get_option("last_staged_pairs") > 0) == expect_staged, (P, c.get_option("last_staged_pairs"))
        c.close()
        _same(g, _run(seq, prm, cfg, 0 if expect_staged else 1), P)


@pytest.mark.parametrize("mode,est,H,kpts,sbs,below,above", [
    # a context that shares the chip with other launch chains (option "side_by_side": the chains of a PsBatchQueue, the lanes of
    # the pipelined stream) takes the staged form from far smaller batches on (ps_capi.hip: kStagedFrom*Sbs); with TWO chains the
    # point lies half way, on the logarithmic scale
    (EUCLIDEAN_ERROR, EST_FIXED, 4096, 500, 4, 56, 74),      # 4.5e5 + 75 x 500 = 4.875e5: 65 pairs (alone: 121)
    (REPROJECTION_ERROR, EST_FIXED, 4096, 400, 4, 48, 64),   # 3.0e5 + 90 x 400 = 3.36e5: 56 pairs (alone: 250)
    (REPROJECTION_ERROR, EST_FIXED, 4096, 400, 2, 104, 134), # sqrt(3.36e5 x 1.5e6) = 7.1e5: 119 pairs
    (EUCLIDEAN_ERROR, EST_RANSAC, 1157, 300, 3, 14, 26),     # 2.4e4: 20 pairs (hb - 1 = 4; alone: 50)
])
def test_cost_model_side_by_side_takes_the_staged_form_earlier(mode, est, H, kpts, sbs, below, above):
    """Round 6: chains that run side by side hide one another's launch gaps, so the staged form's dependent launches cost throughput
    little and it pays from 2 - 5 times smaller batches on (profiles/r06u/concurrent_crossover.txt, bench_data_crossover.txt).  The form taken just below /
    just above the point is the predicted one, a lone context still takes complete scoring there, and the outputs are those of the
    other form."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    prm = default_ransac_params(mode, lc=(H == 1157))
    cfg, _ = make_config(est, H, seed=6160)
    for P, expect_staged in ((below, False), (above, True)):
        seq = synth.make_sequence(P + 1, kpts, config=3, index=9900 + P, inlier_frac=0.6, noise=0.005)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        c = api.Context(0)
        c.set_option("side_by_side", sbs)
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        assert (c.get_option("last_staged_pairs") > 0) == expect_staged, (P, c.get_option("last_staged_pairs"))
        c.set_option("side_by_side", 0)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        pb.download()
        assert c.get_option("last_staged_pairs") == 0, P     # alone: complete scoring on either side of this point
        c.close()
        _same(g, _run(seq, prm, cfg, 0 if expect_staged else 2), P)


def test_queue_chains_know_they_run_side_by_side():
    """The chains of a PsBatchQueue carry side_by_side = their number (the parent's options otherwise), a queue of one chain does not."""
    c = api.Context(0)
    q = api.BatchQueue(c, 0)
    assert q.chains == 4                                    # the library's default (profiles/r06u/small_batch_chains.txt)
    assert [x.get_option("side_by_side") for x in q.contexts] == [4] * 4 and c.get_option("side_by_side") == 0
    q.close()
    q = api.BatchQueue(c, 1)
    assert q.contexts[0].get_option("side_by_side") == 0
    q.close()
    c.close()
