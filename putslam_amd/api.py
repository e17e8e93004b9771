"""Thin Python access to the C ABI (include/putslam_hip.h) for tests, the bench and Python callers.

Every function here ends in a HIP kernel launch inside libputslam_hip.so; nothing is computed
in Python/numpy and nothing falls back to the CPU.  The C++ drop-in classes that mirror the
reference's Matcher / RANSAC / TransformEst surface live in putslam_amd/csrc/dropin/.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._abi import (DMATCH_DTYPE, STATS_DTYPE, PS_ERR_BUSY, PS_OK, PsFrameSet, PsHostPairResults, PsPairResults,  # noqa: F401
                   PsRansacConfig, PsRansacParams, default_ransac_params, make_config)


class PsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"putslam_hip error {code}: {msg}")
        self.code = code


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One HIP stream + scratch arena (PsContext).  Not shared between threads."""

    def __init__(self, device=0, lib=None):
        self._L = _lib.load() if lib is None else _lib.load_path(lib)   # (lib: another build, A/B timing only)
        h = C.c_void_p()
        rc = self._L.ps_context_create(int(device), C.byref(h))
        if rc != PS_OK:
            raise PsError(rc, "ps_context_create failed (no usable HIP device? there is no CPU fallback)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.ps_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != PS_OK:
            raise PsError(rc, self._L.ps_last_error(self._h).decode())

    @property
    def arch(self):
        return self._L.ps_device_arch(self._h).decode()

    def pack_records(self, pose_ptr, stats_ptr, valid, pairs, records_ptr, stream_ptr=0):
        """ps_pack_records_device: the 72-byte per-pair records of the multi-GPU gather (pose + numInliers + numMatchesIn as 18 floats),
        one launch on `stream_ptr` (0: the context's stream); rows [valid, pairs) are zero-filled.  Device pointers."""
        self._chk(self._L.ps_pack_records_device(self._h, C.c_void_p(stream_ptr), C.c_void_p(pose_ptr), C.c_void_p(stats_ptr),
                                                 int(valid), int(pairs), C.c_void_p(records_ptr)))

    def set_stream(self, stream_ptr):
        self._chk(self._L.ps_context_set_stream(self._h, C.c_void_p(stream_ptr)))

    @property
    def stream_ptr(self):
        """The hipStream_t the context's calls are queued on (ps_context_stream)."""
        return int(self._L.ps_context_stream(self._h) or 0)

    def synchronize(self):
        self._chk(self._L.ps_context_synchronize(self._h))

    def set_option(self, name, value):
        """Kernel variant switches (include/putslam_hip.h: ps_context_set_option), e.g. ("matcher", 0 | 1)."""
        self._chk(self._L.ps_context_set_option(self._h, name.encode(), int(value)))

    def score_stats(self):
        """(parked, evaluations) of the last fast scoring launch (needs set_option("score_stats", 1))."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.ps_debug_score_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def score_stats_ex(self):
        """All eight counters of the last scoring step (ps_debug_score_stats_ex): parked, evaluations made, reserved x 6."""
        out = (C.c_uint64 * 8)()
        self._chk(self._L.ps_debug_score_stats_ex(self._h, out))
        return [int(v) for v in out]

    def stage_survivors(self, P):
        """(2, P) hypotheses of every pair that survived stages 1 and 2 of the last staged scoring step."""
        out = np.zeros((2, int(P)), np.int32)
        self._chk(self._L.ps_debug_stage_survivors(self._h, int(P), out.ctypes.data))
        return out

    def stage_order(self, P, cap):
        """(perm (P, cap), front (P,)): the order stages 1+ of the last staged scoring step swept the matches in."""
        perm = np.zeros((int(P), int(cap)), np.int32)
        front = np.zeros(int(P), np.int32)
        self._chk(self._L.ps_debug_stage_order(self._h, int(P), int(cap), perm.ctypes.data, front.ctypes.data))
        return perm, front

    def stamps(self):
        """Shader-clock stamps of kernels 2 and 4 of the last call (needs set_option("stamps", 1)); ps_debug_stamps."""
        out = (C.c_uint64 * 16)()
        self._chk(self._L.ps_debug_stamps(self._h, out))
        return [int(v) for v in out]

    def get_option(self, name):
        v = self._L.ps_context_get_option(self._h, name.encode())
        if v < 0:
            raise ValueError(f"unknown option {name!r}")
        return v

    def enable_timing(self, on=True):
        self._chk(self._L.ps_context_enable_timing(self._h, 1 if on else 0))

    def last_kernel_times_ms(self):
        ms = np.zeros(8, np.float32)
        n = self._L.ps_last_kernel_times_ms(self._h, _p(ms))
        if n < 0:
            self._chk(n)
        names = kernel_names()
        return {names[i] if i < len(names) else f"k{i}": float(ms[i]) for i in range(n)}

    def kernel_time_totals(self):
        """{kernel name: (sum of launch durations in ms, launches)} over the calls since enable_timing()."""
        sums = np.zeros(8, np.float64)
        cnt = np.zeros(8, np.int32)
        n = self._L.ps_kernel_time_totals(self._h, _p(sums), _p(cnt))
        if n < 0:
            self._chk(n)
        names = kernel_names()
        return {names[i]: (float(sums[i]), int(cnt[i])) for i in range(n) if cnt[i] > 0}

    # ---- A1 ----
    def match_hamming256(self, query, train):
        """MatcherOpenCV::performMatching (matcherOpenCV.cpp:198-206): query=prev rows, train=cur rows (uint8, 32 cols)."""
        assert query.dtype == np.uint8 and train.dtype == np.uint8
        nq, nt = query.shape[0], train.shape[0]
        out = np.zeros(max(nq, 1), DMATCH_DTYPE)
        n = C.c_int(0)
        qs = query.strides[0] if nq else 32
        ts = train.strides[0] if nt else 32
        self._chk(self._L.ps_match_hamming256(self._h, _p(query), nq, qs, _p(train), nt, ts, _p(out), C.byref(n)))
        return out[: n.value].copy()

    # ---- A4..A9, A11 ----
    def ransac_rigid3d(self, params, cfg, K, prev, cur, matches):
        """RANSAC::estimateTransformation (RANSAC.cpp:50-174) / RANSAC_USAC (USAC_wrapper.cpp:104-151)."""
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
        self._chk(self._L.ps_ransac_rigid3d(self._h, C.byref(params), C.byref(cfg), _p(K), _p(prev), prev.shape[0],
                                            _p(cur), cur.shape[0], _p(matches), m, _p(pose), _p(inl), C.byref(ninl),
                                            _p(mask), _p(stats)))
        return dict(pose=pose.reshape(4, 4).T.copy(), inliers=inl[: ninl.value].copy(), mask=mask[:m].copy(),
                    stats=stats[0].copy())

    def debug_ransac_counts(self, params, cfg, K, prev, cur, matches):
        prev = np.ascontiguousarray(prev, np.float32)
        cur = np.ascontiguousarray(cur, np.float32)
        matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
        K = None if K is None else np.ascontiguousarray(K, np.float32)
        counts = np.zeros(max(cfg.numHypotheses, 1), np.int32)
        n = C.c_int(0)
        self._chk(self._L.ps_debug_ransac_counts(self._h, C.byref(params), C.byref(cfg), _p(K), _p(prev),
                                                 prev.shape[0], _p(cur), cur.shape[0], _p(matches),
                                                 matches.shape[0], _p(counts), C.byref(n)))
        return counts[: n.value].copy()

    def debug_keys_clean(self):
        """Words of the keys block that are not all-ones at rest (ps_debug_keys_clean): must be 0."""
        bad = C.c_uint64(0)
        self._chk(self._L.ps_debug_keys_clean(self._h, C.byref(bad)))
        return int(bad.value)

    def debug_limits(self, estimator, min_ratio, H, M):
        out = np.zeros(M, np.int32)
        self._chk(self._L.ps_debug_limits(self._h, int(estimator), float(min_ratio), int(H), int(M), _p(out)))
        return out

    def debug_fastdiv(self, seed=1, blocks=2048, per_thread=2048):
        bad, n = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.ps_debug_fastdiv(self._h, int(seed), int(blocks), int(per_thread), C.byref(bad), C.byref(n)))
        return bad.value, n.value

    def debug_mathcheck(self, mode, elements, seed=1):
        """(mismatches, tested): ps_debug_mathcheck -- the prologue's exact short sqrt / reciprocal / quotient forms vs the operators."""
        bad, n = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.ps_debug_mathcheck(self._h, int(mode), int(seed), int(elements), C.byref(bad), C.byref(n)))
        return bad.value, n.value

    # ---- A7 ----
    def umeyama_f32(self, src, dst):
        """src, dst: (nsets, k, 3) or (k, 3). Returns (T (nsets,4,4) row/col matrices, valid (nsets,))."""
        src = np.ascontiguousarray(src, np.float32)
        dst = np.ascontiguousarray(dst, np.float32)
        single = src.ndim == 2
        if single:
            src, dst = src[None], dst[None]
        nsets, k = src.shape[0], src.shape[1]
        T = np.zeros((nsets, 16), np.float32)
        valid = np.zeros(nsets, np.int32)
        self._chk(self._L.ps_umeyama_f32(self._h, _p(src), _p(dst), k, nsets, _p(T), _p(valid)))
        T = T.reshape(nsets, 4, 4).transpose(0, 2, 1).copy()
        return (T[0], bool(valid[0])) if single else (T, valid.astype(bool))

    # ---- A10 ----
    def kabsch_f64(self, A, B):
        """KabschEst::computeTransformation (kabschEst.cpp:24-68). A, B (n,3). Returns 4x4 mapping A onto B."""
        A = np.asfortranarray(A, np.float64)
        B = np.asfortranarray(B, np.float64)
        n = A.shape[0]
        T = np.zeros(16, np.float64)
        self._chk(self._L.ps_kabsch_f64(self._h, _p(A), _p(B), n, max(n, 1), _p(T)))
        return T.reshape(4, 4).T.copy()

    # ---- A3 ----
    def keypoints2Dto3D(self, xy, depth, K, scale):
        xy = np.ascontiguousarray(xy, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        assert depth.dtype == np.uint16 and depth.ndim == 2
        out = np.zeros((xy.shape[0], 3), np.float32)
        self._chk(self._L.ps_keypoints2Dto3D(self._h, _p(xy), xy.shape[0], _p(depth), depth.shape[0], depth.shape[1],
                                             depth.strides[0], _p(K), float(scale), _p(out)))
        return out

    def remove_image_distortion(self, xy, K, dist5):
        """RGBD::removeImageDistortion (RGBD.cpp:254-314)."""
        xy = np.ascontiguousarray(xy, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        d = np.ascontiguousarray(dist5, np.float64)
        out = np.zeros_like(xy)
        self._chk(self._L.ps_remove_image_distortion(self._h, _p(xy), xy.shape[0], _p(K), _p(d), _p(out)))
        return out

    def points3Dto2D(self, xyz, K):
        xyz = np.ascontiguousarray(xyz, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        uv = np.zeros((xyz.shape[0], 2), np.float32)
        self._chk(self._L.ps_points3Dto2D(self._h, _p(xyz), xyz.shape[0], _p(K), _p(uv)))
        return uv

    # ---- N2 ----
    def match_xyz(self, map_pos, map_desc, map_level, cur_pos, cur_desc, cur_level, radius=0.12, ratio=0.55):
        """Guided map matching core of Matcher::matchXYZ (matcher.cpp:694-746)."""
        map_pos = np.ascontiguousarray(map_pos, np.float32)
        cur_pos = np.ascontiguousarray(cur_pos, np.float32)
        map_desc = np.ascontiguousarray(map_desc, np.uint8)
        cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
        map_level = np.ascontiguousarray(map_level, np.int32)
        cur_level = np.ascontiguousarray(cur_level, np.int32)
        nmap, ncur = map_pos.shape[0], cur_pos.shape[0]
        cap = max(1, 4 * nmap)
        while True:
            out = np.zeros(cap, DMATCH_DTYPE)
            n = C.c_int(0)
            rc = self._L.ps_match_xyz(self._h, _p(map_pos), _p(map_desc), 32, _p(map_level), nmap, _p(cur_pos),
                                      _p(cur_desc), 32, _p(cur_level), ncur, float(radius), float(ratio), _p(out), cap,
                                      C.byref(n))
            if rc == PS_OK:
                return out[: n.value].copy()
            if n.value > cap:
                cap = n.value
                continue
            self._chk(rc)

    def predicted_level(self, octave, det_dist, cur_dist):
        return self._L.ps_predicted_level(int(octave), float(det_dist), float(cur_dist))

    # ---- A2 / A12: device-resident batch ----
    def vo_pairs_device(self, params, cfg, K, frames: "DeviceFrames", pairs_dev_ptr, P, out: "DeviceResults"):
        K = np.ascontiguousarray(K, np.float32)
        fs = PsFrameSet(frames.desc_ptr, frames.pts_ptr, frames.nkpts_ptr, frames.num_frames, frames.max_kpts,
                        frames.desc_stride, frames.pts_stride)
        res = PsPairResults(out.matches_ptr, out.num_matches_ptr, out.mask_ptr, out.pose_ptr, out.stats_ptr)
        self._chk(self._L.ps_vo_pairs_device(self._h, C.byref(params), C.byref(cfg), _p(K), C.byref(fs),
                                             C.c_void_p(pairs_dev_ptr), int(P), C.byref(res)))


class _ChainContext(Context):
    """A chain context of a BatchQueue: owned by the queue (never destroyed from here)."""

    def __init__(self, L, handle):
        self._L = L
        self._h = C.c_void_p(handle)

    def close(self):
        self._h = None


class BatchQueue:
    """PsBatchQueue: ps_vo_pairs_device through launch chains that are never joined (chains = 0: the library's default, four;
    whole batches in turn: consecutive batches run side by side and need output blocks of their own, `chains` of them in turn)."""

    def __init__(self, ctx: Context, chains=0):
        self._ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx._L.ps_batch_queue_create(ctx._h, int(chains), C.byref(h)))
        self._h = h
        self.chains = ctx._L.ps_batch_queue_chains(h)
        self.contexts = [_ChainContext(ctx._L, ctx._L.ps_batch_queue_context(h, i)) for i in range(self.chains)]

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self._ctx, "_h", None):
                self._ctx._L.ps_batch_queue_destroy(self._h)
            self._h = None
            for c in self.contexts:
                c.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, params, cfg, K, frames: "DeviceFrames", pairs_dev_ptr, P, out: "DeviceResults"):
        """Asynchronous; returns the batch's ticket."""
        K = np.ascontiguousarray(K, np.float32)
        fs = PsFrameSet(frames.desc_ptr, frames.pts_ptr, frames.nkpts_ptr, frames.num_frames, frames.max_kpts,
                        frames.desc_stride, frames.pts_stride)
        res = PsPairResults(out.matches_ptr, out.num_matches_ptr, out.mask_ptr, out.pose_ptr, out.stats_ptr)
        t = C.c_int64(-1)
        self._ctx._chk(self._ctx._L.ps_batch_queue_submit(self._h, C.byref(params), C.byref(cfg), _p(K), C.byref(fs),
                                                          C.c_void_p(pairs_dev_ptr), int(P), C.byref(res), C.byref(t)))
        return int(t.value)

    def wait(self, ticket):
        self._ctx._chk(self._ctx._L.ps_batch_queue_wait(self._h, int(ticket)))

    def query(self, ticket):
        r = self._ctx._L.ps_batch_queue_query(self._h, int(ticket))
        if r < 0:
            self._ctx._chk(r)
        return bool(r)

    def wait_on_stream(self, ticket, stream_ptr):
        self._ctx._chk(self._ctx._L.ps_batch_queue_wait_on_stream(self._h, int(ticket), C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ctx._chk(self._ctx._L.ps_batch_queue_synchronize(self._h))

    def last_split(self):
        b = np.zeros(self.chains + 1, np.int32)
        self._ctx._L.ps_batch_queue_last_split(self._h, b.ctypes.data)
        return [int(x) for x in b]


class VoStream:
    """Streaming Matcher::match (matcher.cpp:452-516): previous frame resident in HBM (ps_vo_stream_*)."""

    def __init__(self, ctx: Context, max_kpts):
        self._ctx = ctx
        self._cap = int(max_kpts)
        h = C.c_void_p()
        ctx._chk(ctx._L.ps_vo_stream_create(ctx._h, self._cap, C.byref(h)))
        self._h = h
        # push()'s output block, allocated once with its pointers (four arrays and five pointer objects per call were a
        # quarter of a pushed frame's 0.097 ms; the C call itself takes 0.07 - 0.08, demos/cpp/demo_latency)
        self._matches = np.zeros(max(self._cap, 1), DMATCH_DTYPE)
        self._mask = np.zeros(max(self._cap, 1), np.uint8)
        self._pose = np.zeros(16, np.float32)
        self._stats = np.zeros(1, STATS_DTYPE)
        self._nm = C.c_int(0)
        self._out_ptrs = (_p(self._matches), C.byref(self._nm), _p(self._mask), _p(self._pose), _p(self._stats))

    def close(self):
        if getattr(self, "_h", None):
            # (a stream outliving its context -- a test that failed before closing it, collected after the context fixture -- is
            # leaked, not destroyed through a dangling context)
            if getattr(self._ctx, "_h", None):
                self._ctx._L.ps_vo_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, params, cfg, K, desc, pts):
        """Returns None for the first frame, else dict(matches, mask, pose, stats)."""
        desc = np.ascontiguousarray(desc, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32)
        K = np.ascontiguousarray(K, np.float32)
        pm, pn, pk, pp, ps = self._out_ptrs
        rc = self._ctx._L.ps_vo_stream_push(self._h, C.byref(params), C.byref(cfg), C.c_void_p(K.ctypes.data), C.c_void_p(desc.ctypes.data),
                                            32, C.c_void_p(pts.ctypes.data), desc.shape[0], pm, pn, pk, pp, ps)
        if rc:
            self._ctx._chk(rc)
        nm = self._nm.value
        if nm < 0:
            return None
        return dict(matches=self._matches[:nm].copy(), mask=self._mask[:nm].copy(),
                    pose=self._pose.reshape(4, 4).T.copy(), stats=self._stats[0].copy())

    # ---- pipelined form (ps_vo_stream_configure_async ...): results come back with a lag, in pair order ----
    def configure_async(self, params, cfg, K, chunk_frames=0, lanes=0, results=0, packed=False):
        """results: 0 = everything (PS_RESULTS_FULL), 1 = inlier matches + pose + stats, 2 = pose + stats.
        packed: PS_FRAMES_PACKED -- every frame one block [cap x 32 B][cap x 12 B] of `packed_stride` bytes, on the host and in
        the ring: one upload per chunk (push_many_packed)."""
        K = None if K is None else np.ascontiguousarray(K, np.float32)
        self._ctx._chk(self._ctx._L.ps_vo_stream_set_result_mode(self._h, int(results)))
        self._ctx._chk(self._ctx._L.ps_vo_stream_set_frame_layout(self._h, 1 if packed else 0))
        self._ctx._chk(self._ctx._L.ps_vo_stream_configure_async(self._h, C.byref(params), C.byref(cfg), _p(K),
                                                                 int(chunk_frames), int(lanes)))

    def _rc(self, rc):
        """PS_ERR_BUSY is flow control, not a failure: returns False for it, True for PS_OK, raises otherwise."""
        if rc == PS_ERR_BUSY:
            return False
        self._ctx._chk(rc)
        return True

    def push_async(self, desc, pts):
        """One frame (matcher.cpp:452-516's call shape); False = no room (every lane busy and the upload-ahead queue full), pop first."""
        desc = np.ascontiguousarray(desc, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32)
        return self._rc(self._ctx._L.ps_vo_stream_push_async(self._h, _p(desc), 32, _p(pts), desc.shape[0]))

    def push_many(self, desc, pts, nkpts):
        """desc (F, cap, 32) u8, pts (F, cap, 3) f32, nkpts (F,) i32 -- numpy arrays (pinned ones are read in place: keep them
        untouched until their results have been popped) or raw (address, address, array) for pre-sliced pinned blocks."""
        nk = np.ascontiguousarray(nkpts, np.int32)
        if isinstance(desc, int):
            dp, pp = C.c_void_p(desc), C.c_void_p(pts)
        else:
            assert desc.flags.c_contiguous and pts.flags.c_contiguous and desc.dtype == np.uint8 and pts.dtype == np.float32
            assert desc.shape[1:] == (self._cap, 32) and pts.shape[1:] == (self._cap, 3)
            dp, pp = _p(desc), _p(pts)
        return self._rc(self._ctx._L.ps_vo_stream_push_many(self._h, dp, pp, _p(nk), nk.shape[0]))

    @property
    def packed_stride(self):
        """Bytes per frame of the packed layout (cap x 44 rounded up to a multiple of 16)."""
        return int(self._ctx._L.ps_vo_stream_packed_stride(self._h))

    def push_many_packed(self, frames, nkpts):
        """frames: (F, packed_stride) u8, every row [cap x 32 B descriptors][cap x 12 B points][padding] -- a numpy array
        (pinned: read in place) or a raw address; nkpts (F,) i32."""
        nk = np.ascontiguousarray(nkpts, np.int32)
        if isinstance(frames, int):
            fp = C.c_void_p(frames)
        else:
            assert frames.flags.c_contiguous and frames.dtype == np.uint8 and frames.shape[1:] == (self.packed_stride,)
            fp = _p(frames)
        return self._rc(self._ctx._L.ps_vo_stream_push_many_packed(self._h, fp, self.packed_stride, _p(nk), nk.shape[0]))

    def flush(self):
        return self._rc(self._ctx._L.ps_vo_stream_flush(self._h))

    def reset(self):
        return self._rc(self._ctx._L.ps_vo_stream_reset(self._h))

    def pending(self):
        return self._ctx._L.ps_vo_stream_pending(self._h)

    def graph_launches(self):
        """Pushes / small chunks replayed from a captured hipGraph so far."""
        return int(self._ctx._L.ps_vo_stream_graph_launches(self._h))

    def pop_many(self, wait=True, copy=True):
        """Results of the oldest chunk in flight: None if nothing is ready, else dict(first_pair, epoch, matches (n, cap),
        numMatches, inlierMask, pose (n, 16), stats).  copy=False: views of the pinned block, valid until the next pop."""
        v = PsHostPairResults()
        self._ctx._chk(self._ctx._L.ps_vo_stream_pop_many(self._h, 1 if wait else 0, C.byref(v)))
        if v.count == 0:
            return None
        n, cap = v.count, v.maxKpts

        def arr(ptr, nbytes, dtype, shape):
            a = np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype).reshape(shape)
            return a.copy() if copy else a

        return dict(first_pair=int(v.firstPair), epoch=int(v.epoch), count=n, result_mode=int(v.resultMode),
                    matches=arr(v.matches, n * cap * 16, DMATCH_DTYPE, (n, cap)) if v.matches else None,
                    numMatches=arr(v.numMatches, n * 4, np.int32, (n,)),
                    inlierMask=arr(v.inlierMask, n * cap, np.uint8, (n, cap)) if v.inlierMask else None,
                    pose=arr(v.pose, n * 64, np.float32, (n, 16)),
                    stats=arr(v.stats, n * STATS_DTYPE.itemsize, STATS_DTYPE, (n,)))

    def pop(self, wait=True):
        """One pair, copied out: None if nothing is ready / in flight, else the dict `push` returns."""
        matches = np.zeros(max(self._cap, 1), DMATCH_DTYPE)
        mask = np.zeros(max(self._cap, 1), np.uint8)
        pose = np.zeros(16, np.float32)
        stats = np.zeros(1, STATS_DTYPE)
        nm = C.c_int(0)
        self._ctx._chk(self._ctx._L.ps_vo_stream_pop(self._h, 1 if wait else 0, _p(matches), C.byref(nm), _p(mask), _p(pose),
                                                     _p(stats)))
        if nm.value < 0:
            return None
        return dict(matches=matches[: nm.value].copy(), mask=mask[: nm.value].copy(),
                    pose=pose.reshape(4, 4).T.copy(), stats=stats[0].copy())


class PinnedBuffer:
    """Page-locked host memory from ps_host_alloc, as a numpy array (frames handed to VoStream.push_many in place)."""

    def __init__(self, shape, dtype):
        self._L = _lib.load()
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(shape)) * self.dtype.itemsize
        self.ptr = self._L.ps_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MemoryError("ps_host_alloc failed")
        self.array = np.frombuffer((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr), dtype=self.dtype,
                                   count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._L.ps_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceFrames:
    """Raw device pointers of a frame set (PsFrameSet)."""

    def __init__(self, desc_ptr, pts_ptr, nkpts_ptr, num_frames, max_kpts, desc_stride=0, pts_stride=0):
        self.desc_ptr, self.pts_ptr, self.nkpts_ptr = desc_ptr, pts_ptr, nkpts_ptr
        self.num_frames, self.max_kpts = int(num_frames), int(max_kpts)
        self.desc_stride, self.pts_stride = int(desc_stride), int(pts_stride)      # bytes between frames; 0 = dense


class DeviceResults:
    """Raw device pointers of per-pair outputs (PsPairResults)."""

    def __init__(self, matches_ptr, num_matches_ptr, mask_ptr, pose_ptr, stats_ptr):
        self.matches_ptr, self.num_matches_ptr, self.mask_ptr = matches_ptr, num_matches_ptr, mask_ptr
        self.pose_ptr, self.stats_ptr = pose_ptr, stats_ptr


def kernel_names():
    L = _lib.load()
    raw = L.ps_kernel_names()
    names, cur, i = [], b"", 0
    while True:
        ch = raw[i]
        i += 1
        if ch == b"\x00":
            if not cur:
                break
            names.append(cur.decode())
            cur = b""
        else:
            cur += ch
    return names


def algorithmic_bytes(nkpts, matches_in, matches_valid, H):
    return int(_lib.load().ps_algorithmic_bytes(int(nkpts), int(matches_in), int(matches_valid), int(H)))
