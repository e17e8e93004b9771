"""Error behaviour of the C ABI (include/putslam_hip.h: negative PsStatus + ps_last_error, outputs reset to the
reference's failure values: identity pose, no inliers) and that a failed call leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

from putslam_amd import _lib, synth
from putslam_amd._abi import (DMATCH_DTYPE, EST_RANSAC, PsRansacStats, TUM_FR1_K, default_ransac_params, make_config)

pytestmark = pytest.mark.gpu

OK, BAD_ARG, NO_DEVICE, HIP, ALLOC, UNSUPPORTED = 0, -1, -2, -3, -4, -5


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture()
def raw():
    L = _lib.load()
    h = C.c_void_p()
    assert L.ps_context_create(0, C.byref(h)) == OK
    yield L, h
    L.ps_context_destroy(h)


def test_context_create_rejects_bad_device():
    L = _lib.load()
    h = C.c_void_p()
    assert L.ps_context_create(-1, C.byref(h)) == BAD_ARG and not h.value
    assert L.ps_context_create(4096, C.byref(h)) == BAD_ARG and not h.value
    assert L.ps_context_create(0, None) == BAD_ARG
    assert L.ps_last_error(None) == b"null context"
    L.ps_context_destroy(None)          # harmless


def test_match_argument_errors_then_success(raw):
    L, h = raw
    r = np.random.default_rng(1)
    q = r.integers(0, 256, (100, 32), dtype=np.uint8)
    t = r.integers(0, 256, (90, 32), dtype=np.uint8)
    out = np.zeros(100, DMATCH_DTYPE)
    n = C.c_int(7)
    assert L.ps_match_hamming256(h, None, 100, 32, _p(t), 90, 32, _p(out), C.byref(n)) == BAD_ARG and n.value == 0
    assert b"bad argument" in L.ps_last_error(h)
    assert L.ps_match_hamming256(h, _p(q), 100, 32, _p(t), 90, 32, None, C.byref(n)) == BAD_ARG
    assert L.ps_match_hamming256(h, _p(q), -1, 32, _p(t), 90, 32, _p(out), C.byref(n)) == BAD_ARG
    assert L.ps_match_hamming256(h, _p(q), 100, 16, _p(t), 90, 32, _p(out), C.byref(n)) == UNSUPPORTED  # short rows
    assert b"32 bytes" in L.ps_last_error(h)
    assert L.ps_match_hamming256(h, _p(q), 20000, 32, _p(t), 90, 32, _p(out), C.byref(n)) == UNSUPPORTED
    # empty sides are not errors: BFMatcher returns no matches
    assert L.ps_match_hamming256(h, _p(q), 0, 32, _p(t), 90, 32, _p(out), C.byref(n)) == OK and n.value == 0
    assert L.ps_match_hamming256(h, _p(q), 100, 32, _p(t), 0, 32, _p(out), C.byref(n)) == OK and n.value == 0
    # the context still works, and a successful call clears the message
    assert L.ps_match_hamming256(h, _p(q), 100, 32, _p(t), 90, 32, _p(out), C.byref(n)) == OK and n.value > 0
    assert L.ps_last_error(h) == b""


def test_ransac_argument_errors_reset_outputs(raw, oracle):
    L, h = raw
    a, b = synth.make_pair(300, config=2, index=3)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(1)
    cfg, _ = make_config(EST_RANSAC, 487, seed=9)
    K = np.ascontiguousarray(TUM_FR1_K, np.float32)
    prev, cur = np.ascontiguousarray(a["pts"], np.float32), np.ascontiguousarray(b["pts"], np.float32)
    pose = np.full(16, 7.0, np.float32)
    inl = np.zeros(len(m), DMATCH_DTYPE)
    ninl = C.c_int(99)
    mask = np.full(len(m), 5, np.uint8)
    st = PsRansacStats()

    def call(prm_=prm, cfg_=cfg, matches=m, nprev=len(prev), ncur=len(cur), pose_=pose):
        return L.ps_ransac_rigid3d(h, C.byref(prm_) if prm_ is not None else None,
                                   C.byref(cfg_) if cfg_ is not None else None, _p(K), _p(prev), nprev, _p(cur), ncur,
                                   _p(matches), len(matches), _p(pose_) if pose_ is not None else None, _p(inl),
                                   C.byref(ninl), _p(mask), C.byref(st))

    def is_reset():
        return np.array_equal(pose.reshape(4, 4), np.eye(4, dtype=np.float32)) and ninl.value == 0

    assert call(prm_=None) == BAD_ARG and is_reset()
    pose[:] = 7.0
    bad = default_ransac_params(1)
    bad.usedPairs = 4
    assert call(prm_=bad) == UNSUPPORTED and is_reset() and b"usedPairs" in L.ps_last_error(h)
    for H in (0, -3, (1 << 20) + 1):
        c2, _ = make_config(EST_RANSAC, 487, seed=9)
        c2.numHypotheses = H
        pose[:] = 7.0
        assert call(cfg_=c2) == BAD_ARG and is_reset()
    c3, _ = make_config(EST_RANSAC, 487, seed=9)
    c3.estimator = 17
    assert call(cfg_=c3) == BAD_ARG and b"estimator" in L.ps_last_error(h)
    m2 = m.copy()
    m2["trainIdx"][5] = len(cur)                       # one index past the end
    pose[:] = 7.0
    assert call(matches=m2) == BAD_ARG and is_reset() and b"out of range" in L.ps_last_error(h)
    m2 = m.copy()
    m2["queryIdx"][0] = -1
    assert call(matches=m2) == BAD_ARG
    assert call(pose_=None) == BAD_ARG
    # and the same context then produces the oracle's answer
    assert call() == OK
    c = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
    assert np.array_equal(pose.reshape(4, 4).T.view(np.uint32), np.asarray(c["pose"], np.float32).view(np.uint32))  # ABI: column-major
    assert ninl.value == len(c["inliers"])


def test_null_context_is_rejected_everywhere():
    L = _lib.load()
    n = C.c_int(0)
    buf = np.zeros(64, np.uint8)
    assert L.ps_context_synchronize(None) == BAD_ARG
    assert L.ps_match_hamming256(None, _p(buf), 1, 32, _p(buf), 1, 32, _p(buf), C.byref(n)) == BAD_ARG
    assert L.ps_context_enable_timing(None, 1) == BAD_ARG
    assert L.ps_context_set_stream(None, None) == BAD_ARG


def test_minutes_long_batch_is_refused_before_anything_is_allocated():
    """complete scoring (option "prune" = 0) of a batch under USAC's cap -- 20 000 pairs x 850 000 hypotheses x
    16 384 matches > 2e14 evaluations -- is refused with PS_ERR_UNSUPPORTED.  The refusal used to stand BEHIND the request for the
    per-hypothesis counts block (4 bytes x pairs x hypotheses = 68 GB here, 400 GB for larger batches: an allocation failure or a
    huge allocation instead of the message); now it comes first and the context's arena stays as small as it was."""
    import torch
    from putslam_amd import api
    from putslam_amd._abi import EST_USAC
    c = api.Context(0)
    c.set_option("prune", 0)
    before = c.get_option("arena_mib")
    dummy = torch.zeros(4096, dtype=torch.uint8, device="cuda")          # never touched: the call is refused before any launch
    p = dummy.data_ptr()
    prm = default_ransac_params(0)
    cfg, _ = make_config(EST_USAC, 850000, seed=1)
    with pytest.raises(api.PsError) as e:
        c.vo_pairs_device(prm, cfg, TUM_FR1_K, api.DeviceFrames(p, p, p, 20001, 16384), p, 20000, api.DeviceResults(p, p, p, p, p))
    assert e.value.code == UNSUPPORTED and "run for minutes" in str(e.value)
    assert c.get_option("arena_mib") <= before + 16                       # (the stop table of the cap: 850 000 doubles)
    c.set_option("prune", 1)
    c.close()
