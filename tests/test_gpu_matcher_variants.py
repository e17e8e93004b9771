"""Kernel 1 triplets: the FP4 matrix-core sweep with the query tiles expanded by the work-group itself
(ps_hamming_mfma_fused, default), the same sweep reading an FP4 image written by a launch of its own (ps_hamming_mfma,
round 2's form) and the integer VALU sweep (ps_hamming_nn) must all reproduce the oracle's BFMatcher(NORM_HAMMING, crossCheck=true) list bit for bit
(reference src/Matcher/matcherOpenCV.cpp:100-105,198-206)."""
import numpy as np
import pytest

from putslam_amd import api, synth

pytestmark = pytest.mark.gpu

VARIANTS = [("mfma-fused", 1, 1), ("mfma-image", 1, 0), ("valu", 0, 0)]


@pytest.fixture(scope="module", params=VARIANTS, ids=[v[0] for v in VARIANTS])
def vctx(request):
    c = api.Context(0)
    c.set_option("matcher", request.param[1])
    c.set_option("matcher_fused", request.param[2])
    assert c.get_option("matcher") == request.param[1] and c.get_option("matcher_fused") == request.param[2]
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 500, 1999, 2000, 2017, 5000])
def test_variant_match_bit_exact(vctx, oracle, n):
    a, b = synth.make_pair(n, config=2, index=1000 + n)
    assert vctx.match_hamming256(a["desc"], b["desc"]).tobytes() == oracle.match_hamming256(a["desc"], b["desc"]).tobytes()


@pytest.mark.parametrize("nq,nt", [(300, 1000), (1000, 300), (1, 700), (700, 1), (513, 511), (33, 4097), (4097, 33)])
def test_variant_match_ragged(vctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 11 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    k = min(nq, nt) // 2
    if k:
        t[:k] = q[rng.permutation(nq)[:k]] ^ np.packbits(rng.random((k, 256)) < 0.05, axis=1)
    assert vctx.match_hamming256(q, t).tobytes() == oracle.match_hamming256(q, t).tobytes()


def test_variant_extreme_distances_and_ties(vctx, oracle):
    # distance 0, distance 256, every row identical (ties resolve to the lowest index in both passes), and a
    # block of rows that differ from their query in exactly one bit at every bit position (0..255)
    z, o = np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8)
    assert vctx.match_hamming256(z, o)[0]["distance"] == 256.0
    assert vctx.match_hamming256(o, o)[0]["distance"] == 0.0
    q = np.tile(np.arange(32, dtype=np.uint8), (70, 1))
    t = np.tile(np.arange(32, dtype=np.uint8), (45, 1))
    g = vctx.match_hamming256(q, t)
    assert g.tobytes() == oracle.match_hamming256(q, t).tobytes() and len(g) == 1
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (256, 32), dtype=np.uint8)
    flip = np.zeros((256, 256), np.uint8)
    flip[np.arange(256), np.arange(256)] = 1
    t = base ^ np.packbits(flip, axis=1)
    g = vctx.match_hamming256(base, t)
    assert g.tobytes() == oracle.match_hamming256(base, t).tobytes()
    assert np.all(g["distance"] == 1.0) and np.array_equal(g["queryIdx"], g["trainIdx"])


def test_variant_low_entropy_descriptors(vctx, oracle):
    # few distinct bytes -> many equal distances: the tie rule (lowest query, then lowest train) decides almost everything
    rng = np.random.default_rng(77)
    q = rng.integers(0, 2, (900, 32), dtype=np.uint8) * 255
    t = rng.integers(0, 2, (1100, 32), dtype=np.uint8) * 255
    assert vctx.match_hamming256(q, t).tobytes() == oracle.match_hamming256(q, t).tobytes()


@pytest.mark.parametrize("qsplit", [1, 2, 5])
def test_variant_ties_across_query_tiles(vctx, oracle, qsplit):
    """One work-group sweeps many 32-row query tiles: equal distances in different tiles must resolve to the lowest
    query index whatever the row inside the tile (a later tile's lower row must not win)."""
    rng = np.random.default_rng(2024 + qsplit)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    q = base[rng.integers(0, 40, 1500)]                      # 1500 queries drawn from 40 distinct rows: many exact ties
    t = base[rng.integers(0, 40, 1300)] ^ np.packbits(rng.random((1300, 256)) < 0.02, axis=1)
    vctx.set_option("debug.qsplit", qsplit)
    try:
        g = vctx.match_hamming256(q, t)
    finally:
        vctx.set_option("debug.qsplit", 0)
    assert g.tobytes() == oracle.match_hamming256(q, t).tobytes()


def test_variants_agree_on_a_batch(oracle):
    """ps_vo_pairs_device with either matcher: identical matches / masks / poses / stats for a 9-pair batch with
    ragged keypoint counts; three sampled pairs are also compared with the oracle."""
    from putslam_amd._abi import EST_RANSAC, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(10, 700, config=3, index=5)
    seq["nkpts"][3] = 517
    seq["nkpts"][7] = 64
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_RANSAC, 487, seed=4242)
    outs = []
    for kind, fused in ((1, 1), (0, 0), (1, 0)):
        c = api.Context(0)
        c.set_option("matcher", kind)
        c.set_option("matcher_fused", fused)
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb, use_torch_stream=False)
        c.synchronize()
        outs.append(pb.download())
        c.close()
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["numMatches"], b["numMatches"])
        for p in range(len(seq["pairs"])):
            n = int(a["numMatches"][p])
            assert a["matches"][p][:n].tobytes() == b["matches"][p][:n].tobytes()
            assert np.array_equal(a["inlierMask"][p][:n], b["inlierMask"][p][:n])
        assert a["pose"].tobytes() == b["pose"].tobytes()
        assert a["stats"].tobytes() == b["stats"].tobytes()
    for p in (0, 3, 7):
        f0, f1 = seq["pairs"][p]
        n0, n1 = int(seq["nkpts"][f0]), int(seq["nkpts"][f1])
        m = oracle.match_hamming256(seq["desc"][f0][:n0], seq["desc"][f1][:n1])
        assert a["matches"][p][:len(m)].tobytes() == m.tobytes() and int(a["numMatches"][p]) == len(m)


def test_keys_block_is_all_ones_at_rest(oracle):
    """The matcher forms that merge their query splits with atomicMin start from an all-ones keys block and no longer clear
    it themselves: kernel 2 puts kNoKey back into every entry it has read.  One context, every matcher form in turn, shrinking
    and growing frames, forced and automatic query splits, single pairs and a batch in between: every list equals the oracle's
    (a stale key left by an earlier call would win an atomicMin and show up as a wrong match)."""
    from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    c = api.Context(0)
    rng = np.random.default_rng(991)
    forms = [(0, 0, 0), (1, 1, 3), (0, 0, 7), (1, 0, 2), (1, 1, 1), (0, 0, 0), (1, 1, 0), (0, 0, 1), (0, 0, 64)]
    sizes = [(2000, 1900), (700, 2000), (1999, 64), (33, 1500), (1200, 1200), (5, 2000), (2000, 2000), (300, 40), (1500, 1700)]
    seq = synth.make_sequence(8, 900, config=3, index=17)
    seq["nkpts"][2] = 401
    prm = default_ransac_params(0)
    cfg, _ = make_config(EST_RANSAC, 487, seed=99)
    try:
        for rnd, ((kind, fused, qsplit), (nq, nt)) in enumerate(zip(forms, sizes)):
            c.set_option("matcher", kind)
            c.set_option("matcher_fused", fused)
            c.set_option("debug.qsplit", qsplit)
            q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            k = min(nq, nt) // 2
            t[:k] = q[rng.permutation(nq)[:k]] ^ np.packbits(rng.random((k, 256)) < 0.04, axis=1)
            assert c.match_hamming256(q, t).tobytes() == oracle.match_hamming256(q, t).tobytes(), (rnd, kind, fused, qsplit)
            if rnd % 3 == 1:  # a batch on the same context in between (its own pair count and row stride)
                fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
                pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
                run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb, use_torch_stream=False)
                c.synchronize()
                got = pb.download()
                for p in (0, 1, 2, 6):
                    f0, f1 = seq["pairs"][p]
                    n0, n1 = int(seq["nkpts"][f0]), int(seq["nkpts"][f1])
                    m = oracle.match_hamming256(seq["desc"][f0][:n0], seq["desc"][f1][:n1])
                    assert int(got["numMatches"][p]) == len(m) and got["matches"][p][:len(m)].tobytes() == m.tobytes(), (rnd, p)
    finally:
        c.close()


def test_keys_block_is_all_ones_at_rest_across_shapes_variants_and_failed_calls():
    """the matcher forms that merge their query splits with atomicMin take no clearing launch -- the keys block
    is all-ones at rest because kernel 2 puts kNoKey back into every entry it read.  One context alternating batch sizes,
    frame capacities, ragged row counts, both matchers, query splits, streamed pushes and a failed call: after each of them
    ps_debug_keys_clean finds no other word, and results stay those of a fresh context."""
    from putslam_amd import api, synth
    from putslam_amd._abi import EST_RANSAC, EUCLIDEAN_ERROR, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    c = api.Context(0)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_RANSAC, 487, seed=5)
    rng = np.random.default_rng(3)
    shapes = [(2, 300), (9, 1200), (3, 2000), (30, 257), (2, 4100), (12, 640), (2, 31)]
    for it in range(14):
        frames, kpts = shapes[it % len(shapes)]
        seq = synth.make_sequence(frames, kpts, config=3, index=900 + it)
        if it % 3 == 1:
            seq["nkpts"][:] = rng.integers(0, kpts + 1, frames)
        c.set_option("matcher", int(rng.integers(0, 3)))
        c.set_option("debug.qsplit", int(rng.choice([0, 0, 1, 2, 5, 16])))
        fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
        g = pb.download()
        assert c.debug_keys_clean() == 0, it
        ref = api.Context(0)
        pb2 = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(ref, prm, cfg, TUM_FR1_K, fs, pb2)
        r = pb2.download()
        ref.close()
        assert np.array_equal(g["numMatches"], r["numMatches"]) and g["matches"].tobytes() == r["matches"].tobytes(), it
        if it % 4 == 2:   # a host-pointer call and a streamed push in between
            a, b = synth.make_pair(int(rng.integers(50, 3000)), config=2, index=it)
            c.match_hamming256(a["desc"], b["desc"])
            assert c.debug_keys_clean() == 0
            st = api.VoStream(c, 700)
            for f in range(4):
                st.push(prm, cfg, TUM_FR1_K, seq["desc"][f % frames][:min(kpts, 700)], seq["pts"][f % frames][:min(kpts, 700)])
            st.close()
            assert c.debug_keys_clean() == 0
        if it % 5 == 4:   # a failed call (bad hypothesis count) must leave the invariant alone
            badcfg, _ = make_config(EST_RANSAC, 0, seed=1)
            with pytest.raises(api.PsError):
                run_pairs(c, prm, badcfg, TUM_FR1_K, fs, pb)
            assert c.debug_keys_clean() == 0
    c.close()
