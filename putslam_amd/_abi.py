"""ctypes / numpy mirrors of the PODs in include/putslam_hip.h.

Field order and sizes follow the header exactly; tests/test_abi_layout.py checks
the sizes against the compiled library (ps_abi_sizeof_*).
"""
import ctypes as C

import numpy as np

PS_OK = 0
PS_DESC_BYTES = 32
PS_MAX_KPTS = 16384

# RANSAC::ERROR_VERSION (reference include/putslam/TransformEst/RANSAC.h:22)
EUCLIDEAN_ERROR = 0
REPROJECTION_ERROR = 1
EUCLIDEAN_AND_REPROJECTION_ERROR = 2
MAHALANOBIS_ERROR = 3
ADAPTIVE_ERROR = 4

EST_RANSAC = 0
EST_USAC = 1
EST_FIXED = 2


class PsDMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32),
                ("distance", C.c_float)]


class PsRansacParams(C.Structure):
    _fields_ = [("verbose", C.c_int32),
                ("errorVersion", C.c_int32), ("errorVersionVO", C.c_int32), ("errorVersionMap", C.c_int32),
                ("inlierThresholdEuclidean", C.c_double), ("inlierThresholdReprojection", C.c_double),
                ("inlierThresholdMahalanobis", C.c_double),
                ("minimalInlierRatioThreshold", C.c_double),
                ("minimalNumberOfMatches", C.c_int32), ("usedPairs", C.c_int32),
                ("iterationCount", C.c_int32)]


class PsRansacConfig(C.Structure):
    _fields_ = [("estimator", C.c_int32), ("numHypotheses", C.c_int32), ("seed", C.c_uint64),
                ("sampleIdx", C.POINTER(C.c_uint32))]


class PsRansacStats(C.Structure):
    _fields_ = [("numMatchesIn", C.c_int32), ("numMatchesValid", C.c_int32),
                ("bestHypothesis", C.c_int32), ("bestInlierCount", C.c_int32),
                ("iterationsRun", C.c_int32), ("numInliers", C.c_int32), ("accepted", C.c_int32),
                ("bestInlierRatio", C.c_float), ("pointInlierRatio", C.c_double)]


class PsFrameSet(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("pts", C.c_void_p), ("nkpts", C.c_void_p),
                ("numFrames", C.c_int32), ("maxKpts", C.c_int32),
                ("descFrameStride", C.c_size_t), ("ptsFrameStride", C.c_size_t)]     # ABI 2: 0 = dense frames


class PsPairResults(C.Structure):
    _fields_ = [("matches", C.c_void_p), ("numMatches", C.c_void_p), ("inlierMask", C.c_void_p),
                ("pose", C.c_void_p), ("stats", C.c_void_p)]


class PsHostPairResults(C.Structure):
    _fields_ = [("matches", C.c_void_p), ("numMatches", C.c_void_p), ("inlierMask", C.c_void_p),
                ("pose", C.c_void_p), ("stats", C.c_void_p), ("firstPair", C.c_int64), ("count", C.c_int32),
                ("maxKpts", C.c_int32), ("epoch", C.c_int32), ("resultMode", C.c_int32)]


PS_ERR_BUSY = -6

DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
STATS_DTYPE = np.dtype([("numMatchesIn", "<i4"), ("numMatchesValid", "<i4"), ("bestHypothesis", "<i4"),
                        ("bestInlierCount", "<i4"), ("iterationsRun", "<i4"), ("numInliers", "<i4"),
                        ("accepted", "<i4"), ("bestInlierRatio", "<f4"), ("pointInlierRatio", "<f8")])
assert DMATCH_DTYPE.itemsize == C.sizeof(PsDMatch) == 16
assert STATS_DTYPE.itemsize == C.sizeof(PsRansacStats) == 40

# Camera intrinsics of the reference's default dataset config
# (resources/datasetConfig/freiburg1_desk.xml:5-6,20; the same constants are hard-coded at RGBD.cpp:22-23).
TUM_FR1_K = np.array([517.3, 0.0, 318.6, 0.0, 516.5, 255.3, 0.0, 0.0, 1.0], dtype=np.float32)
TUM_DEPTH_SCALE = 5000.0
# rgbDistortion (k1, k2, p1, p2, k3) of the same file, :7 -- what RGBD::removeImageDistortion is given (RGBD.cpp:254-314)
TUM_FR1_DIST = (-0.0410, 0.3286, 0.0087, 0.0051, -0.5643)


def default_ransac_params(error_version=EUCLIDEAN_ERROR, lc=False):
    """Shipped defaults: resources/putslammatcherOpenCVParameters.xml:29-37 (LC variant: ...LC.xml:30)."""
    p = PsRansacParams()
    p.verbose = 0
    p.errorVersion = error_version
    p.errorVersionVO = 0
    p.errorVersionMap = 0
    p.inlierThresholdEuclidean = 0.04
    p.inlierThresholdReprojection = 2.0
    p.inlierThresholdMahalanobis = 0.0002
    p.minimalInlierRatioThreshold = 0.15 if lc else 0.2
    p.minimalNumberOfMatches = 10 if lc else 15
    p.usedPairs = 3
    p.iterationCount = 0
    return p


def make_config(estimator=EST_RANSAC, num_hypotheses=487, seed=1, sample_idx=None):
    """Returns (cfg, keepalive). sample_idx: optional (H,3) uint32 array of raw draws."""
    cfg = PsRansacConfig()
    cfg.estimator = estimator
    cfg.numHypotheses = int(num_hypotheses)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep = None
    if sample_idx is not None:
        keep = np.ascontiguousarray(sample_idx, dtype=np.uint32)
        assert keep.shape == (cfg.numHypotheses, 3)
        cfg.sampleIdx = keep.ctypes.data_as(C.POINTER(C.c_uint32))
    else:
        cfg.sampleIdx = None
    return cfg, keep
