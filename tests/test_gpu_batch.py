"""Device-resident batched path (ps_vo_pairs_device) vs the oracle's Matcher::match data flow."""
import numpy as np
import pytest

from putslam_amd import synth
from putslam_amd._abi import (EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K,
                              default_ransac_params, make_config)

pytestmark = pytest.mark.gpu

STAT_FIELDS = ("numMatchesIn", "numMatchesValid", "bestHypothesis", "bestInlierCount", "iterationsRun", "numInliers",
               "accepted", "bestInlierRatio", "pointInlierRatio")


def _compare(g, c, P):
    assert np.array_equal(g["numMatches"], c["numMatches"])
    for p in range(P):
        n = int(c["numMatches"][p])
        assert g["matches"][p, :n].tobytes() == c["matches"][p, :n].tobytes(), p
        assert np.array_equal(g["inlierMask"][p, :n], c["inlierMask"][p, :n]), p
        for f in STAT_FIELDS:
            a, b = g["stats"][p][f], c["stats"][p][f]
            assert a == b or (np.isnan(a) and np.isnan(b)), (p, f, a, b)
    assert np.abs(g["pose"] - c["pose"]).max() <= 1e-5
    assert g["pose"].tobytes() == c["pose"].tobytes()


@pytest.mark.parametrize("mode,est,H", [(EUCLIDEAN_ERROR, EST_RANSAC, 487), (REPROJECTION_ERROR, EST_RANSAC, 487),
                                        (REPROJECTION_ERROR, EST_FIXED, 1024), (EUCLIDEAN_ERROR, EST_USAC, 800)])
def test_sequence_batch(ctx, oracle, mode, est, H):
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(9, 600, config=3, index=mode * 10 + est)
    prm = default_ransac_params(mode)
    cfg, _ = make_config(est, H, seed=1234)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=4)
    _compare(g, c, len(seq["pairs"]))
    # poses must also be close to the generator's ground truth (sanity of the whole path)
    for p in range(len(seq["pairs"])):
        if g["stats"][p]["accepted"]:
            T = g["pose"][p].reshape(4, 4).T
            assert np.abs(T - seq["gt"][p]).max() < (3e-2 if est == EST_USAC else 5e-3)  # USAC: minimal-sample pose, no refit


@pytest.mark.parametrize("frames", [2, 3, 4])
def test_smallest_batches(ctx, oracle, frames):
    """One pair runs its scoring as one launch, two pairs and more generate the models in a launch of their own and sweep in a
    second (Plan::genPlain): the boundary, every metric family, fixed and adaptive schedules."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    from putslam_amd._abi import ADAPTIVE_ERROR, EUCLIDEAN_AND_REPROJECTION_ERROR
    seq = synth.make_sequence(frames, 700, config=3, index=300 + frames)
    for mode, est, H in ((EUCLIDEAN_ERROR, EST_FIXED, 2000), (REPROJECTION_ERROR, EST_FIXED, 1500), (ADAPTIVE_ERROR, EST_RANSAC, 487),
                         (EUCLIDEAN_AND_REPROJECTION_ERROR, EST_USAC, 900)):
        prm = default_ransac_params(mode)
        cfg, _ = make_config(est, H, seed=99)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=2)
        _compare(g, c, len(seq["pairs"]))


@pytest.mark.parametrize("cap,pad", [(600, 0), (333, 0), (333, 4096), (2000, 0)])
@pytest.mark.parametrize("matcher", [0, 1])
def test_packed_frame_set_equals_dense(oracle, cap, pad, matcher):
    """PsFrameSet strides (ABI 2): frames that keep descriptors and points together -- [cap x 32 B][cap x 12 B] per frame, any
    stride that is a multiple of 16 -- through both matcher kernels and kernel 2: the oracle's bytes, which are the dense frame
    set's; strides the kernels cannot take are refused."""
    from putslam_amd import api
    from putslam_amd.device_batch import PackedFrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(7, cap, config=3, index=8800 + cap)
    nk = seq["nkpts"].copy()
    nk[2], nk[5] = cap // 3, 0
    pairs = np.array([[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 6], [6, 0], [3, 1]], np.int32)
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 700, seed=31)
    c2 = api.Context(0)
    c2.set_option("matcher", matcher)
    stride = (cap * 44 + 15) // 16 * 16 + pad
    fs = PackedFrameSetDevice(seq["desc"], seq["pts"], nk, stride=stride)
    pb = PairBatchDevice(pairs, cap)
    run_pairs(c2, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], nk, pairs, threads=4)
    _compare(g, c, len(pairs))
    v = fs.view()
    for ds, ps in ((stride + 8, stride), (cap * 32 - 16, stride), (stride, stride + 2), (stride, cap * 12 - 4)):
        bad = api.DeviceFrames(v.desc_ptr, v.pts_ptr, v.nkpts_ptr, v.num_frames, v.max_kpts, ds, ps)
        with pytest.raises(api.PsError) as e:
            c2.vo_pairs_device(prm, cfg, TUM_FR1_K, bad, pb.pairs.data_ptr(), len(pairs), pb.view())
        assert e.value.code == -1 and "FrameStride" in str(e.value)
    c2.close()


def test_ragged_frames_and_arbitrary_pairs(ctx, oracle):
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(6, 512, config=3, index=77)
    nk = np.array([512, 300, 511, 1, 0, 450], np.int32)  # ragged, a single-keypoint and an empty frame
    pairs = np.array([[0, 1], [1, 2], [2, 0], [3, 2], [4, 5], [5, 4], [0, 0], [5, 3]], np.int32)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_RANSAC, 487, seed=5)
    fs = FrameSetDevice(seq["desc"], seq["pts"], nk)
    pb = PairBatchDevice(pairs, fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], nk, pairs, threads=2)
    _compare(g, c, len(pairs))


def test_full_size_properties(ctx, oracle):
    """BASELINE config sizes (2000 kpts, H = 4096): oracle on a sample of pairs + size-independent properties."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(33, 2000, config=3, index=1)
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 4096, seed=42)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    P = len(seq["pairs"])
    # idempotence: a second run over the same inputs is bit-identical
    pb2 = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb2)
    g2 = pb2.download()
    assert g["pose"].tobytes() == g2["pose"].tobytes() and np.array_equal(g["inlierMask"], g2["inlierMask"])
    for p in range(P):
        n = int(g["numMatches"][p])
        m = g["matches"][p, :n]
        assert np.all(np.diff(m["queryIdx"]) > 0)                 # ascending queryIdx, unique
        assert len(np.unique(m["trainIdx"])) == n                 # each train row chose exactly one query
        assert np.all((m["distance"] >= 0) & (m["distance"] <= 256))
        st = g["stats"][p]
        assert st["numMatchesIn"] == n and st["numInliers"] == int(g["inlierMask"][p, :n].sum())
        assert st["numInliers"] <= st["bestInlierCount"] <= st["numMatchesValid"] <= n
        T = g["pose"][p].reshape(4, 4).T.astype(np.float64)
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-5 and np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-5
        assert np.abs(T - seq["gt"][p]).max() < 5e-3
    sample = [0, 7, 31]
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][sample], threads=3)
    gs = {k: v[sample] for k, v in g.items()}
    # hypothesis streams are seeded with seed + pair index: re-run the sampled pairs at their own index
    for j, p in enumerate(sample):
        cfgp, _ = make_config(EST_FIXED, 4096, seed=42 + p)
        n = int(g["numMatches"][p])
        cc = oracle.ransac_rigid3d(prm, cfgp, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1],
                                   g["matches"][p, :n])
        assert c["matches"][j, :n].tobytes() == g["matches"][p, :n].tobytes()
        assert np.array_equal(cc["mask"], g["inlierMask"][p, :n])
        assert cc["pose"].T.astype(np.float32).tobytes() == g["pose"][p].tobytes()


def test_stress_config_sizes(ctx, oracle):
    """BASELINE configs[4]: 5000 keypoints, 256-bit descriptors, 100 000 hypotheses (adaptive stop disabled)."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(3, 5000, config=5, index=0)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    # Euclidean mode: the oracle scores all 100 000 hypotheses in a few seconds -> full bit-exact comparison
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 100000, seed=9)
    pb = PairBatchDevice(seq["pairs"][:1], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][:1], threads=1)
    _compare(g, c, 1)
    # reprojection mode: size-independent properties + the selected hypothesis re-scored by the oracle
    prm = default_ransac_params(REPROJECTION_ERROR)
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    g = pb.download()
    for p in range(2):
        st = g["stats"][p]
        n = int(g["numMatches"][p])
        assert st["iterationsRun"] == 100000 and 0 <= st["bestHypothesis"] < 100000
        assert st["numInliers"] == int(g["inlierMask"][p, :n].sum()) and st["accepted"] == 1
        T = g["pose"][p].reshape(4, 4).T.astype(np.float64)
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-5 and np.abs(T - seq["gt"][p]).max() < 5e-3
        # the winner's count must be what the oracle counts for that very sample (explicit one-sample stream)
        cfgp, _ = make_config(EST_FIXED, int(st["bestHypothesis"]) + 1, seed=9 + p)
        cnt, M = oracle.hypothesis_counts(prm, cfgp, TUM_FR1_K, seq["pts"][p], seq["pts"][p + 1], g["matches"][p, :n])
        assert M == st["numMatchesValid"] and cnt[-1] == st["bestInlierCount"] and cnt.max() == cnt[-1]


def test_streaming_match_state(ctx, oracle):
    """ps_vo_stream_*: Matcher::match with the previous frame kept on the GPU; ragged frame sizes."""
    from putslam_amd import api
    seq = synth.make_sequence(7, 700, config=3, index=9)
    nk = [700, 650, 700, 300, 700, 1, 700]
    prm = default_ransac_params(REPROJECTION_ERROR)
    st = api.VoStream(ctx, 700)
    prev = None
    for f in range(7):
        cfg, _ = make_config(EST_RANSAC, 487, seed=100 + f)
        d, p3 = seq["desc"][f][: nk[f]], seq["pts"][f][: nk[f]]
        r = st.push(prm, cfg, TUM_FR1_K, d, p3)
        if f == 0:
            assert r is None
        else:
            m = oracle.match_hamming256(prev[0], d)
            c = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, prev[1], p3, m)
            assert r["matches"].tobytes() == m.tobytes()
            assert np.array_equal(r["mask"], c["mask"]) and r["pose"].tobytes() == c["pose"].tobytes()
            for fld in STAT_FIELDS:
                a, b = r["stats"][fld], c["stats"][fld]
                assert a == b or (np.isnan(a) and np.isnan(b)), (f, fld, a, b)
        prev = (d, p3)
    st.close()


@pytest.mark.parametrize("S,est,H", [(2, EST_FIXED, 1024), (3, EST_RANSAC, 487), (2, EST_USAC, 600)])
def test_split_over_streams_equals_single_call(ctx, S, est, H):
    """bench.py's default submission (sub-batches on several HIP streams, one context each) must be bit-identical
    to the single ps_vo_pairs_device call: pair p draws from seed + p wherever its sub-batch starts."""
    import torch
    from putslam_amd import api
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs, run_pairs_split
    seq = synth.make_sequence(24, 700, config=3, index=200 + S)
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(est, H, seed=0xABC)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    one = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, one)
    a = one.download()
    ctxs = [ctx] + [api.Context(0) for _ in range(S - 1)]
    side = [torch.cuda.Stream() for _ in range(S)]
    for join in (True, False):
        two = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs_split(ctxs, side, prm, est, H, cfg.seed, TUM_FR1_K, fs, two, join=join)
        b = two.download()                       # synchronises the device, hence every stream
        _compare(b, a, len(seq["pairs"]))
    for c in ctxs[1:]:
        c.close()
    ctx.set_stream(0)   # back to the context's private stream


def test_streaming_graph_replay_long_sequence(ctx, oracle):
    """ps_vo_stream_push replays a captured hipGraph from the third push on (one graph per frame slot).  Every push
    must still equal the oracle: new seed every frame, ragged frame sizes, a parameter change in the middle (graphs
    are rebuilt), other calls on the same context in between (they resize and rewrite the shared scratch/tables)."""
    from putslam_amd import api
    seq = synth.make_sequence(16, 600, config=3, index=31)
    nk = [600, 600, 590, 600, 420, 600, 600, 1, 600, 600, 333, 600, 600, 600, 0, 600]
    st = api.VoStream(ctx, 600)
    prev = None
    for f in range(16):
        mode = REPROJECTION_ERROR if f < 9 else EUCLIDEAN_ERROR            # parameter change at frame 9
        est, H = (EST_RANSAC, 487) if f < 12 else (EST_FIXED, 700)         # estimator / H change at frame 12
        prm = default_ransac_params(mode)
        cfg, _ = make_config(est, H, seed=1000 + 17 * f)
        d, p3 = seq["desc"][f][: nk[f]], seq["pts"][f][: nk[f]]
        r = st.push(prm, cfg, TUM_FR1_K, d, p3)
        if f == 0:
            assert r is None
        else:
            m = oracle.match_hamming256(prev[0], d)
            c = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, prev[1], p3, m)
            assert r["matches"].tobytes() == m.tobytes(), f
            assert np.array_equal(r["mask"], c["mask"]) and r["pose"].tobytes() == c["pose"].tobytes(), f
            for fld in STAT_FIELDS:
                a, b = r["stats"][fld], c["stats"][fld]
                assert a == b or (np.isnan(a) and np.isnan(b)), (f, fld, a, b)
        if f in (5, 6):   # a foreign call on the same context: bigger scratch, other stop tables
            big = synth.make_pair(1500, config=2, index=f)
            mm = ctx.match_hamming256(big[0]["desc"], big[1]["desc"])
            c2, _ = make_config(EST_USAC, 2000, seed=3)
            ctx.ransac_rigid3d(default_ransac_params(EUCLIDEAN_ERROR), c2, TUM_FR1_K, big[0]["pts"], big[1]["pts"], mm)
        prev = (d, p3)
    st.close()


def test_run_pairs_stream_ordering(ctx):
    """run_pairs is ordered like a torch op: queued behind the work of the current stream and visible to it afterwards,
    whether the current stream is the legacy default stream (handle 0, which the C ABI cannot take: a side stream is
    forked and joined) or an explicit one."""
    import torch
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(12, 500, config=3, index=61)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_RANSAC, 487, seed=77)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    ref = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, ref)                       # default stream
    pose_ref = ref.pose.clone()                                         # a default-stream op: must see the results
    torch.cuda.synchronize()
    assert pose_ref.abs().sum().item() > 0
    s = torch.cuda.Stream()
    other = PairBatchDevice(seq["pairs"], fs.max_kpts)
    with torch.cuda.stream(s):
        other.pose.fill_(123.0)                                         # queued before: must not survive
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, other)
        pose_s = other.pose.clone()
    s.synchronize()
    assert torch.equal(pose_s, pose_ref)
    _compare(other.download(), ref.download(), len(seq["pairs"]))
    ctx.set_stream(0)


def test_full_length_sequence_sampled_against_oracle(ctx, oracle):
    """BASELINE configs[2] at full length -- the 500-frame x 2000-keypoint sequence bench.py times, H = 4096 fixed,
    reprojection error -- submitted as three chains on three streams (the bench default until round 5; it is two now): 20 sampled pairs are compared
    field by field with the oracle (matches, mask, pose bytes, statistics), every pair is checked for the
    size-independent properties, and the composed trajectory stays next to the ground truth."""
    import torch
    from putslam_amd import api, sharding
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_split
    seq = synth.make_sequence(500, 2000, config=3, index=0)
    P = len(seq["pairs"])
    prm = default_ransac_params(REPROJECTION_ERROR)
    H, seed = 4096, 0xB0B0
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    ctxs = [ctx, api.Context(0), api.Context(0)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    run_pairs_split(ctxs, streams, prm, EST_FIXED, H, seed, TUM_FR1_K, fs, pb, join=False)
    g = pb.download()
    for c in ctxs[1:]:
        c.close()
    ctx.set_stream(0)
    assert int(g["stats"]["accepted"].sum()) == P
    for p in range(P):
        n = int(g["numMatches"][p])
        st = g["stats"][p]
        assert st["numMatchesIn"] == n and st["numInliers"] == int(g["inlierMask"][p, :n].sum())
        assert st["numInliers"] <= st["bestInlierCount"] <= st["numMatchesValid"] <= n and st["iterationsRun"] == H
        assert np.abs(g["pose"][p].reshape(4, 4).T - seq["gt"][p]).max() < 5e-3
    sample = sorted(set(np.random.default_rng(7).integers(0, P, 19).tolist() + [0, P - 1]))[:20]
    for p in sample:
        f0, f1 = seq["pairs"][p]
        m = oracle.match_hamming256(seq["desc"][f0], seq["desc"][f1])
        n = int(g["numMatches"][p])
        assert n == len(m) and g["matches"][p, :n].tobytes() == m.tobytes(), p
        cfgp, _ = make_config(EST_FIXED, H, seed=seed + p)          # pair p draws from seed + p
        c = oracle.ransac_rigid3d(prm, cfgp, TUM_FR1_K, seq["pts"][f0], seq["pts"][f1], m)
        assert np.array_equal(c["mask"], g["inlierMask"][p, :n]), p
        assert c["pose"].T.astype(np.float32).tobytes() == g["pose"][p].tobytes(), p
        for fld in STAT_FIELDS:
            a, b = g["stats"][p][fld], c["stats"][fld]
            assert a == b or (np.isnan(a) and np.isnan(b)), (p, fld, a, b)
    traj = sharding.compose_trajectory(g["pose"].reshape(P, 4, 4).transpose(0, 2, 1))
    gt = sharding.compose_trajectory(seq["gt"].astype(np.float32))
    assert np.abs(traj[-1][:3, 3] - gt[-1][:3, 3]).max() < 0.05


def test_stream_push_failure_leaves_the_resident_frame_alone(ctx, oracle):
    """A push that fails (unsupported parameters, hypothesis count out of range) must not advance the stream: the next
    good push still matches against the last frame that was really uploaded, exactly like the uninterrupted sequence."""
    from putslam_amd import api
    seq = synth.make_sequence(5, 500, config=3, index=77)
    prm = default_ransac_params(EUCLIDEAN_ERROR)

    def run(inject):
        st = api.VoStream(ctx, 500)
        out = []
        for f in range(5):
            cfg, _ = make_config(EST_RANSAC, 487, seed=9 + f)
            if inject and f in (2, 3):
                bad = default_ransac_params(EUCLIDEAN_ERROR)
                bad.usedPairs = 4                                     # PS_ERR_UNSUPPORTED
                with pytest.raises(api.PsError):
                    st.push(bad, cfg, TUM_FR1_K, seq["desc"][4], seq["pts"][4])   # a different frame: must not stick
                cbad, _ = make_config(EST_RANSAC, 0, seed=1)          # numHypotheses out of range
                with pytest.raises(api.PsError):
                    st.push(prm, cbad, TUM_FR1_K, seq["desc"][0], seq["pts"][0])
            out.append(st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f]))
        st.close()
        return out

    a, b = run(False), run(True)
    assert a[0] is None and b[0] is None
    for f in range(1, 5):
        assert a[f]["matches"].tobytes() == b[f]["matches"].tobytes(), f
        assert a[f]["pose"].tobytes() == b[f]["pose"].tobytes() and np.array_equal(a[f]["mask"], b[f]["mask"])
        m = oracle.match_hamming256(seq["desc"][f - 1], seq["desc"][f])
        assert b[f]["matches"].tobytes() == m.tobytes()
    # the first frame failing leaves the stream empty: the next push is still a first frame
    st = api.VoStream(ctx, 500)
    bad = default_ransac_params(EUCLIDEAN_ERROR)
    bad.usedPairs = 5
    cfg, _ = make_config(EST_RANSAC, 487, seed=1)
    assert st.push(bad, cfg, TUM_FR1_K, seq["desc"][0], seq["pts"][0]) is None   # first frames are only stored
    assert st.push(prm, cfg, TUM_FR1_K, seq["desc"][1], seq["pts"][1]) is not None
    st.close()


def test_one_context_called_from_two_streams_is_ordered(oracle):
    """ps_context_set_stream orders the new stream behind the work queued on the previous one: two run_pairs calls on ONE
    context from two different torch streams share the scratch arena and must still both be right."""
    import torch
    from putslam_amd import api
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    c = api.Context(0)
    seqs = [synth.make_sequence(40, 900, config=3, index=300 + i) for i in range(2)]
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 2048, seed=5)
    fss = [FrameSetDevice(s["desc"], s["pts"], s["nkpts"]) for s in seqs]
    ref = []
    for s, fs in zip(seqs, fss):
        pb = PairBatchDevice(s["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        ref.append(pb.download())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        pbs = [PairBatchDevice(s["pairs"], fs.max_kpts) for s, fs in zip(seqs, fss)]
        for i in (0, 1, 0, 1):                                        # alternate streams without synchronising
            with torch.cuda.stream(streams[i]):
                run_pairs(c, prm, cfg, TUM_FR1_K, fss[i], pbs[i])
        torch.cuda.synchronize()
        for i in range(2):
            _compare(pbs[i].download(), ref[i], len(seqs[i]["pairs"]))
    c.close()


def test_maximum_keypoints_batch_with_coresident_workgroups(ctx, oracle):
    """PS_MAX_KPTS rows per frame in a batch large enough (299 pairs on 256 CUs) that two cross-check work-groups, 64 KiB
    of LDS each, share a CU: match lists identical run to run and equal to the oracle's on sampled pairs."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    n, F = 16384, 300
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc = np.empty((F, n, 32), np.uint8)
    for f in range(F):
        d = base[rng.permutation(n)] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1)
        d[n // 2:] = rng.integers(0, 256, (n - n // 2, 32), dtype=np.uint8)
        desc[f] = d
    pts = rng.uniform(-1, 1, (F, n, 3)).astype(np.float32)
    pts[..., 2] = rng.uniform(0.5, 5.0, (F, n)).astype(np.float32)
    nk = np.full(F, n, np.int32)
    pairs = np.array([[i, i + 1] for i in range(F - 1)], np.int32)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_FIXED, 256, seed=1)
    fs = FrameSetDevice(desc, pts, nk)
    outs = []
    for _ in range(2):
        pb = PairBatchDevice(pairs, fs.max_kpts)
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
        outs.append(pb.download())
    assert np.array_equal(outs[0]["numMatches"], outs[1]["numMatches"])
    for p in range(len(pairs)):
        k = int(outs[0]["numMatches"][p])
        assert outs[0]["matches"][p, :k].tobytes() == outs[1]["matches"][p, :k].tobytes(), p
    for p in (0, 150, 298):
        c = oracle.match_hamming256(desc[p], desc[p + 1])
        k = int(outs[0]["numMatches"][p])
        assert k == len(c) and outs[0]["matches"][p, :k].tobytes() == c.tobytes(), p


def test_stream_graph_key_separates_kernel_variants(oracle):
    """The streaming call replays a captured hipGraph while its key (parameters, kernel variants, arena) is unchanged.
    Changing a kernel-variant option after warm-up must change the key: the next push is enqueued afresh with the newly
    selected kernels ("matcher_used" is written when a push is ENQUEUED, not when a graph is replayed -- with the key's
    variant fields overlapping, as in round 2, it kept reporting the old kernel), and poses stay the oracle's."""
    from putslam_amd import api
    c = api.Context(0)
    c.set_option("matcher", 1)
    seq = synth.make_sequence(9, 700, config=3, index=55)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    st = api.VoStream(c, 700)
    poses = []
    for f in range(9):
        if f == 5:
            c.set_option("matcher", 0)          # after the graphs of both slots were captured and replayed
            c.set_option("score", 0)
        cfg, _ = make_config(EST_RANSAC, 487, seed=900 + f)
        r = st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f])
        if f in (4, 8):
            assert c.get_option("matcher_used") == (1 if f == 4 else 0), f
        if f > 0:
            poses.append(r["pose"])
            m = oracle.match_hamming256(seq["desc"][f - 1], seq["desc"][f])
            want = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, seq["pts"][f - 1], seq["pts"][f], m)
            assert r["pose"].tobytes() == want["pose"].tobytes(), f
    st.close()
    c.close()
