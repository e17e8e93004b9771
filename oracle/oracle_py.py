"""ctypes binding of the CPU oracle (oracle/_build/libputslam_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under putslam_amd/.  PARITY UNPINNED (see
oracle/putslam_oracle.h): the reference cannot be compiled here (needs OpenCV + Eigen).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

from putslam_amd._abi import (DMATCH_DTYPE, STATS_DTYPE, PsDMatch, PsFrameSet, PsPairResults,
                              PsRansacConfig, PsRansacParams, PsRansacStats)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PUTSLAM_ORACLE_LIB: another build of the same source (tests/test_oracle_sanitized.py runs the known-answer workload through an
# AddressSanitizer + UndefinedBehaviorSanitizer build)
_SO = os.environ.get("PUTSLAM_ORACLE_LIB") or os.path.join(_HERE, "_build", "libputslam_oracle.so")


def _cpu_stamp():
    """-march=native code must not travel between hosts with different CPUs: key the build on the flags."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    """Compile the oracle with gcc (seconds). Building the checker is not using it."""
    if os.environ.get("PUTSLAM_ORACLE_LIB"):
        return _SO      # (a build somebody else made: used as it is)
    srcs = [os.path.join(_HERE, f) for f in ("putslam_oracle.c", "putslam_oracle.h", "po_svd.inc")]
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    stamp_file = os.path.join(_HERE, "_build", "cpu.stamp")
    stamp = _cpu_stamp()
    # (check + build under a file lock: several processes starting together -- the soak's workers on a fresh box, whose CPU
    # stamp differs from the build host's -- would otherwise rebuild at once and load each other's half-written library)
    import fcntl
    with open(os.path.join(_HERE, "_build", "build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        fresh = (os.path.exists(_SO) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp
                 and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs))
        if force or not fresh:
            subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
            with open(stamp_file, "w") as f:
                f.write(stamp)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.po_hamming256.restype = C.c_int
        L.po_round_size.restype = C.c_int
        L.po_round_size.argtypes = [C.c_double, C.c_int]
        L.po_ransac_iterations.restype = C.c_int
        L.po_ransac_iterations.argtypes = [C.c_double, C.c_double, C.c_int]
        L.po_usac_stopping.restype = C.c_uint
        L.po_usac_stopping.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        L.po_draw31.restype = C.c_uint32
        L.po_draw31.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.po_point_inlier_ratio.restype = C.c_double
        L.po_umeyama_f32.restype = C.c_int
        L.po_is_inlier.restype = C.c_int
        L.po_is_inlier.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_double]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().po_hamming256(_p(a), _p(b))


def match_hamming256(query, train):
    """query (nq, >=32) / train (nt, >=32) uint8 arrays (row pitch taken from strides)."""
    assert query.dtype == np.uint8 and train.dtype == np.uint8
    nq, nt = query.shape[0], train.shape[0]
    out = np.zeros(max(nq, 1), DMATCH_DTYPE)
    n = C.c_int(0)
    qs = query.strides[0] if nq else 32
    ts = train.strides[0] if nt else 32
    rc = lib().po_match_hamming256(_p(query), nq, C.c_size_t(qs), _p(train), nt, C.c_size_t(ts), _p(out), C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def set_matcher_simd(on):
    """Timed-baseline switch: SIMD popcount sweep (default when the host has AVX2) or the scalar popcnt loop."""
    lib().po_set_matcher_simd(int(bool(on)))


def matcher_simd_kind():
    """0 = scalar only, 1 = AVX2 nibble lookup, 2 = AVX-512 VPOPCNTQ (what -march=native gave this build)."""
    return {0: "scalar", 1: "avx2-lut", 2: "avx512-vpopcnt"}[lib().po_matcher_simd_kind()]


def round_size(x, size):
    return lib().po_round_size(float(x), int(size))


def keypoints2Dto3D(xy, depth, K, scale):
    xy = np.ascontiguousarray(xy, np.float32)
    K = np.ascontiguousarray(K, np.float32)
    assert depth.dtype == np.uint16
    out = np.zeros((xy.shape[0], 3), np.float32)
    lib().po_keypoints2Dto3D(_p(xy), xy.shape[0], _p(depth), depth.shape[0], depth.shape[1],
                             C.c_size_t(depth.strides[0]), _p(K), C.c_double(scale), _p(out))
    return out


def points3Dto2D(xyz, K):
    xyz = np.ascontiguousarray(xyz, np.float32)
    K = np.ascontiguousarray(K, np.float32)
    uv = np.zeros((xyz.shape[0], 2), np.float32)
    lib().po_points3Dto2D(_p(xyz), xyz.shape[0], _p(K), _p(uv))
    return uv


def umeyama_f32(src, dst):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    T = np.zeros(16, np.float32)
    ok = lib().po_umeyama_f32(_p(src), _p(dst), src.shape[0], _p(T))
    return T.reshape(4, 4).T.copy(), bool(ok)  # returned as a normal (row, col) matrix


def jacobi_svd3(A, dtype=np.float32):
    A = np.ascontiguousarray(A, dtype)
    U = np.zeros((3, 3), dtype)
    S = np.zeros(3, dtype)
    V = np.zeros((3, 3), dtype)
    fn = lib().po_jacobi_svd3_f32 if dtype == np.float32 else lib().po_jacobi_svd3_f64
    fn(_p(A), _p(U), _p(S), _p(V))
    return U, S, V


def inverse4_f32(T):
    Tc = np.ascontiguousarray(np.asarray(T, np.float32).T)  # column-major storage
    R = np.zeros(16, np.float32)
    lib().po_inverse4_f32(_p(Tc), _p(R))
    return R.reshape(4, 4).T.copy()


def ransac_iterations(r, p=0.98, n=3):
    return lib().po_ransac_iterations(r, p, n)


def usac_stopping(inl, tot, s=3):
    return lib().po_usac_stopping(inl, tot, s)


def usac_replay(valid, counts, H, M):
    """(iterations, best count, best hypothesis) of the USAC loop over replayed per-hypothesis outcomes (po_usac_replay)."""
    v = np.ascontiguousarray(valid, np.int32)
    c = np.ascontiguousarray(counts, np.int32)
    it, bc, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    lib().po_usac_replay(_p(v), _p(c), int(len(c)), int(H), int(M), C.byref(it), C.byref(bc), C.byref(b))
    return int(it.value), int(bc.value), int(b.value)


def draw31(seed, h, j):
    return int(lib().po_draw31(C.c_uint64(seed), int(h), int(j)))


def sample_triplet(cfg, h, M):
    idx = (C.c_int * 3)()
    lib().po_sample_triplet(C.byref(cfg), C.c_uint64(cfg.seed), int(h), int(M), idx)
    return list(idx)


def is_inlier(mode, T, K, prev_pt, cur_pt, thrE, thrR):
    Tc = np.ascontiguousarray(np.asarray(T, np.float32).T)
    Ti = np.zeros(16, np.float32)
    lib().po_inverse4_f32(_p(Tc), _p(Ti))
    K = np.ascontiguousarray(K, np.float32)
    pp = np.ascontiguousarray(prev_pt, np.float32)
    cp = np.ascontiguousarray(cur_pt, np.float32)
    return lib().po_is_inlier(int(mode), _p(Tc), _p(Ti), _p(K), _p(pp), _p(cp), float(thrE), float(thrR))


def eval_errors(T, K, prev_pt, cur_pt):
    """(Euclid norm, reprojection error new, reprojection error old) of one match under model T (4x4), as the
    reference computes them (RANSAC.cpp:266-272,346-366)."""
    Tc = np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))
    Ti = np.zeros(16, np.float32)
    lib().po_inverse4_f32(_p(Tc), _p(Ti))
    K = np.ascontiguousarray(K, np.float32)
    pp = np.ascontiguousarray(prev_pt, np.float32)
    cp = np.ascontiguousarray(cur_pt, np.float32)
    err = np.zeros(3, np.float64)
    lib().po_eval_errors(_p(Tc), _p(Ti), _p(K), _p(pp), _p(cp), _p(err))
    return err


def hypothesis_model(cfg, prev, cur, matches, h):
    """Model of hypothesis h as hypothesis_counts builds it: (T 4x4 or None when invalid, valid-match indices, sample)."""
    prev = np.ascontiguousarray(prev, np.float32)
    cur = np.ascontiguousarray(cur, np.float32)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    m = matches.shape[0]
    T = np.zeros(16, np.float32)
    valid = np.zeros(max(m, 1), np.int32)
    M = C.c_int(0)
    idx = np.zeros(3, np.int32)
    ok = lib().po_hypothesis_model(C.byref(cfg), _p(prev), _p(cur), _p(matches), m, int(h), _p(T), _p(valid),
                                   C.byref(M), _p(idx))
    return (T.reshape(4, 4).T.copy() if ok else None), valid[: M.value].copy(), idx


def ransac_rigid3d(params, cfg, K, prev, cur, matches, want_counts=False):
    prev = np.ascontiguousarray(prev, np.float32)
    cur = np.ascontiguousarray(cur, np.float32)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    K = None if K is None else np.ascontiguousarray(K, np.float32)
    m = matches.shape[0]
    pose = np.zeros(16, np.float32)
    inl = np.zeros(max(m, 1), DMATCH_DTYPE)
    ninl = C.c_int(0)
    mask = np.zeros(max(m, 1), np.uint8)
    stats = np.zeros(1, STATS_DTYPE)
    counts = np.zeros(max(cfg.numHypotheses, 1), np.int32) if want_counts else None
    rc = lib().po_ransac_rigid3d(C.byref(params), C.byref(cfg), _p(K), _p(prev), prev.shape[0], _p(cur),
                                 cur.shape[0], _p(matches), m, _p(pose), _p(inl), C.byref(ninl), _p(mask),
                                 _p(stats), _p(counts))
    assert rc == 0
    res = dict(pose=pose.reshape(4, 4).T.copy(), inliers=inl[: ninl.value].copy(), mask=mask[:m].copy(),
               stats=stats[0].copy())
    if want_counts:
        res["counts"] = counts[: cfg.numHypotheses]
    return res


def hypothesis_counts(params, cfg, K, prev, cur, matches):
    prev = np.ascontiguousarray(prev, np.float32)
    cur = np.ascontiguousarray(cur, np.float32)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    K = None if K is None else np.ascontiguousarray(K, np.float32)
    counts = np.zeros(max(cfg.numHypotheses, 1), np.int32)
    M = C.c_int(0)
    lib().po_hypothesis_counts(C.byref(params), C.byref(cfg), _p(K), _p(prev), prev.shape[0], _p(cur),
                               cur.shape[0], _p(matches), matches.shape[0], _p(counts), C.byref(M))
    return counts[: cfg.numHypotheses], M.value


def point_inlier_ratio(inliers, allm):
    inliers = np.ascontiguousarray(inliers, DMATCH_DTYPE)
    allm = np.ascontiguousarray(allm, DMATCH_DTYPE)
    return lib().po_point_inlier_ratio(_p(inliers), inliers.shape[0], _p(allm), allm.shape[0])


def kabsch_f64(A, B):
    """A, B: (n,3) arrays. Returns the 4x4 transform mapping A onto B."""
    A = np.asfortranarray(A, np.float64)
    B = np.asfortranarray(B, np.float64)
    n = A.shape[0]
    T = np.zeros(16, np.float64)
    lib().po_kabsch_f64(_p(A), _p(B), n, max(n, 1), _p(T))
    return T.reshape(4, 4).T.copy()


def vo_pairs(params, cfg, K, desc, pts, nkpts, pairs, threads=1):
    """Host-memory batch: desc (F,cap,32) u8, pts (F,cap,3) f32, nkpts (F,) i32, pairs (P,2) i32."""
    desc = np.ascontiguousarray(desc, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32)
    nkpts = np.ascontiguousarray(nkpts, np.int32)
    pairs = np.ascontiguousarray(pairs, np.int32)
    K = np.ascontiguousarray(K, np.float32)
    F, cap = desc.shape[0], desc.shape[1]
    P = pairs.shape[0]
    fs = PsFrameSet(_p(desc).value, _p(pts).value, _p(nkpts).value, F, cap)
    out = dict(matches=np.zeros((P, cap), DMATCH_DTYPE), numMatches=np.zeros(P, np.int32),
               inlierMask=np.zeros((P, cap), np.uint8), pose=np.zeros((P, 16), np.float32),
               stats=np.zeros(P, STATS_DTYPE))
    res = PsPairResults(_p(out["matches"]).value, _p(out["numMatches"]).value, _p(out["inlierMask"]).value,
                        _p(out["pose"]).value, _p(out["stats"]).value)
    rc = lib().po_vo_pairs(C.byref(params), C.byref(cfg), _p(K), C.byref(fs), _p(pairs), P, C.byref(res),
                           int(threads))
    assert rc == 0
    return out


def predicted_level(octave, det_dist, cur_dist):
    lib().po_predicted_level.restype = C.c_int
    lib().po_predicted_level.argtypes = [C.c_int, C.c_double, C.c_double]
    return lib().po_predicted_level(int(octave), float(det_dist), float(cur_dist))


def satdiff_hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().po_satdiff_hamming256(_p(a), _p(b))


def match_xyz(map_pos, map_desc, map_level, cur_pos, cur_desc, cur_level, radius, ratio):
    map_pos = np.ascontiguousarray(map_pos, np.float32)
    cur_pos = np.ascontiguousarray(cur_pos, np.float32)
    map_desc = np.ascontiguousarray(map_desc, np.uint8)
    cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    map_level = np.ascontiguousarray(map_level, np.int32)
    cur_level = np.ascontiguousarray(cur_level, np.int32)
    nmap, ncur = map_pos.shape[0], cur_pos.shape[0]
    cap = max(1, nmap * 8)
    while True:
        out = np.zeros(cap, DMATCH_DTYPE)
        n = C.c_int(0)
        lib().po_match_xyz(_p(map_pos), _p(map_desc), C.c_size_t(32), _p(map_level), nmap, _p(cur_pos), _p(cur_desc),
                           C.c_size_t(32), _p(cur_level), ncur, C.c_double(radius), C.c_double(ratio), _p(out), cap,
                           C.byref(n))
        if n.value <= cap:
            return out[: n.value].copy()
        cap = n.value


def remove_image_distortion(xy, K, dist5):
    xy = np.ascontiguousarray(xy, np.float32)
    K = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist5, np.float64)
    out = np.zeros_like(xy)
    lib().po_remove_image_distortion(_p(xy), xy.shape[0], _p(K), _p(d), _p(out))
    return out
