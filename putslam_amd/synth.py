"""Synthetic RGB-D frame pairs / sequences.

Replaces the out-of-scope detect / describe / undistort stages of Matcher::match
(reference src/Matcher/matcher.cpp:457-467,474): it hands the hot path exactly what those
stages hand it -- N x 32-byte binary descriptors and N back-projected 3-D points per frame.
Used by tests and bench.py on both the HIP path and the CPU oracle; it contains no part of the
path under test.

640x480, fx 517.3 fy 516.5 cx 318.6 cy 255.3, depth scale 5000
(reference resources/datasetConfig/freiburg1_desk.xml:5-6,20).
"""
import numpy as np

from ._abi import TUM_DEPTH_SCALE, TUM_FR1_K

SEED_BASE = 0x5055_5453_4C41_4D00
W, H = 640, 480
FX, FY, CX, CY = (np.float32(TUM_FR1_K[0]), np.float32(TUM_FR1_K[4]), np.float32(TUM_FR1_K[2]),
                  np.float32(TUM_FR1_K[5]))


def _rng(config, index):
    return np.random.Generator(np.random.PCG64((SEED_BASE ^ (int(config) << 32) ^ int(index)) & (2 ** 64 - 1)))


def backproject(xy, depth_u16, scale=TUM_DEPTH_SCALE):
    """RGBD::point2Dto3D arithmetic (reference src/RGBD/RGBD.cpp:55-64) for per-keypoint depth samples."""
    xy = np.asarray(xy, np.float32)
    Z = (depth_u16.astype(np.float64) / np.float64(scale)).astype(np.float32)
    u = (xy[:, 0] - CX) / FX
    v = (xy[:, 1] - CY) / FY
    return np.stack([u * Z, v * Z, Z], axis=1).astype(np.float32)


def _rand_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def _new_points(rng, n):
    xy = np.stack([rng.uniform(16, 624, n), rng.uniform(16, 464, n)], axis=1).astype(np.float32)
    z = rng.uniform(0.5, 5.5, n)
    d = np.round(z * TUM_DEPTH_SCALE).astype(np.uint16)
    return xy, d


def first_frame(rng, n, missing=0.10):
    xy, d = _new_points(rng, n)
    d[rng.random(n) < missing] = 0
    return dict(xy=xy, depth=d, pts=backproject(xy, d), desc=_rand_desc(rng, n))


def random_motion(rng, max_angle_deg=2.0, max_trans=0.05):
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, max_angle_deg))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)
    tdir = rng.standard_normal(3)
    tdir /= np.linalg.norm(tdir)
    t = tdir * rng.uniform(0, max_trans)
    return R, t


def next_frame(rng, prev, n, inlier_frac=0.70, noise=0.004, flip=0.08, missing=0.10):
    """Current frame derived from `prev`: p_prev = R p_cur + t for the true correspondences."""
    R, t = random_motion(rng)
    npv = prev["pts"].shape[0]
    n_true = min(int(round(inlier_frac * n)), npv)
    src = rng.permutation(npv)[:n_true]
    xy, d = _new_points(rng, n)
    desc = _rand_desc(rng, n)
    truth = np.full(n, -1, np.int64)
    p_prev = prev["pts"][src].astype(np.float64)
    valid = prev["depth"][src] > 0
    p_cur = (p_prev - t) @ R + rng.normal(0.0, noise, (n_true, 3))  # R^T (p - t)
    z = p_cur[:, 2]
    okz = valid & (z > 0.3)
    u = np.where(okz, p_cur[:, 0] * float(FX) / np.where(okz, z, 1.0) + float(CX), -1.0)
    v = np.where(okz, p_cur[:, 1] * float(FY) / np.where(okz, z, 1.0) + float(CY), -1.0)
    inside = okz & (u >= 1) & (u < W - 1) & (v >= 1) & (v < H - 1) & (z < 6.5)
    # true correspondences with usable geometry: re-projected and re-quantised through the depth grid
    k = np.nonzero(inside)[0]
    xy[k, 0] = u[k].astype(np.float32)
    xy[k, 1] = v[k].astype(np.float32)
    d[k] = np.clip(np.round(z[k] * TUM_DEPTH_SCALE), 0, 65535).astype(np.uint16)
    # correspondences whose previous depth was missing keep their descriptor link but have no depth
    k0 = np.nonzero(~valid)[0]
    d[k0] = 0
    linked = np.concatenate([k, k0])
    fl = rng.random((linked.size, 256)) < flip
    desc[linked] = prev["desc"][src[linked]] ^ np.packbits(fl, axis=1)
    truth[linked] = src[linked]
    d[rng.random(n) < missing] = 0
    perm = rng.permutation(n)
    xy, d, desc, truth = xy[perm], d[perm], desc[perm], truth[perm]
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return dict(xy=xy, depth=d, pts=backproject(xy, d), desc=desc, truth=truth, T_prev_from_cur=T)


def make_pair(n, config=2, index=0, **kw):
    rng = _rng(config, index)
    a = first_frame(rng, n)
    b = next_frame(rng, a, n, **kw)
    return a, b


def make_sequence(num_frames, n, config=3, index=0, **kw):
    """Frame set in the batched layout: desc (F,n,32) u8, pts (F,n,3) f32, nkpts (F,), pairs (F-1,2), gt (F-1,4,4)."""
    rng = _rng(config, index)
    frames = [first_frame(rng, n)]
    for _ in range(num_frames - 1):
        frames.append(next_frame(rng, frames[-1], n, **kw))
    desc = np.stack([f["desc"] for f in frames])
    pts = np.stack([f["pts"] for f in frames])
    nk = np.full(num_frames, n, np.int32)
    pairs = np.stack([np.arange(num_frames - 1), np.arange(1, num_frames)], axis=1).astype(np.int32)
    gt = np.stack([f["T_prev_from_cur"] for f in frames[1:]]) if num_frames > 1 else np.zeros((0, 4, 4))
    return dict(desc=desc, pts=pts, nkpts=nk, pairs=pairs, gt=gt, frames=frames)
