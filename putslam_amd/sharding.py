"""Multi-GPU sharding of independent frame pairs + the host-side VO driver.

The path shards with no data-path collective: each rank runs match -> RANSAC -> refit on its own
pairs.  The only exchange is a gather of fixed-size per-pair records (pose + counts, 72 B) to
rank 0 -- torch.distributed with the "nccl" backend (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
After the gather the single sequential step of the reference, VO_k = VO_{k-1} * inc_k with the 0.1 m
gate (reference src/PUTSLAM/PUTSLAM.cpp:735-740), runs on the host.
"""
import numpy as np

RECORD_FLOATS = 18  # pose[16] + numInliers + numMatches  (72 bytes per frame pair)


def shard_range(total, world, rank):
    """Contiguous chunk [lo, hi) of `total` units for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sequence(num_frames, world, rank):
    """One long sequence split over ranks: pairs [lo, hi) and the frames they touch (one-frame halo:
    frame `hi` is uploaded by this rank and by the next one)."""
    lo, hi = shard_range(num_frames - 1, world, rank)
    frames = (lo, hi + 1) if hi > lo else (lo, lo)
    return dict(pair_lo=lo, pair_hi=hi, frame_lo=frames[0], frame_hi=frames[1])


def pack_records(pose, num_inliers, num_matches):
    """pose (P,16) column-major floats + counts -> (P,18) float32 records (torch tensors or numpy arrays)."""
    import torch
    if isinstance(pose, np.ndarray):
        pose = torch.from_numpy(np.ascontiguousarray(pose, np.float32))
        num_inliers = torch.from_numpy(np.asarray(num_inliers))
        num_matches = torch.from_numpy(np.asarray(num_matches))
    rec = torch.empty((pose.shape[0], RECORD_FLOATS), dtype=torch.float32, device=pose.device)
    rec[:, :16] = pose
    rec[:, 16] = num_inliers.to(torch.float32)
    rec[:, 17] = num_matches.to(torch.float32)
    return rec


def pack_records_device(ctx, pose, stats_i32, pad_to=None, stream=None):
    """The same records from a batch's device-resident results in ONE launch of the library (ps_pack_records_device) on `stream`
    (a torch stream; None: the current one): pose (n, 16) float32, stats_i32 (n, 10) int32 view of PsRansacStats, both torch CUDA
    tensors (contiguous rows); rows [n, pad_to) are zero-filled.  Returns the (pad_to or n, 18) float32 record block."""
    import torch
    n = int(pose.shape[0])
    rows = n if pad_to is None else int(pad_to)
    st = stream if stream is not None else torch.cuda.current_stream(pose.device)
    with torch.cuda.stream(st):
        rec = torch.empty((rows, RECORD_FLOATS), dtype=torch.float32, device=pose.device)
        ctx.pack_records(pose.data_ptr() if n else 0, stats_i32.data_ptr() if n else 0, n, rows, rec.data_ptr(), st.cuda_stream)
    return rec


def gather_records(rec, dst=0, group=None, out=None, async_op=False):
    """Gather equally sized per-rank record blocks on `dst` (RCCL/gloo gather).
    Blocking form: returns the list on dst, else None.  async_op=True: returns (work, list-or-None); the caller
    waits on `work` before reading the list or reusing `rec` / `out` (lets the gather of one step run beside the
    kernels of the next)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = [torch.empty_like(rec) for _ in range(world)]
        work = dist.gather(rec, out, dst=dst, group=group, async_op=async_op)
    else:
        out = None
        work = dist.gather(rec, None, dst=dst, group=group, async_op=async_op)
    return (work, out) if async_op else out


def broadcast_params(params, K, estimator, num_hypotheses, seed, src=0, group=None, device="cpu"):
    """The parameter block of the run -- PsRansacParams, the camera matrix, estimator / hypotheses / base seed --
    from `src` to every rank (~120 B, one broadcast at start; rank `src` is authoritative).
    Returns (params, K, estimator, num_hypotheses, seed) as received."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    from ._abi import PsRansacParams
    nb = C.sizeof(PsRansacParams)
    blob = np.zeros(nb + 36 + 16, np.uint8)
    blob[:nb] = np.frombuffer(bytes(params), np.uint8)
    blob[nb:nb + 36] = np.frombuffer(np.ascontiguousarray(K, np.float32).tobytes(), np.uint8)
    blob[nb + 36:] = np.frombuffer(np.array([estimator, num_hypotheses], np.int32).tobytes() +
                                   np.array([seed], np.uint64).tobytes(), np.uint8)
    t = torch.from_numpy(blob).to(device)
    dist.broadcast(t, src=src, group=group)
    got = t.cpu().numpy()
    out = PsRansacParams.from_buffer_copy(got[:nb].tobytes())
    K2 = np.frombuffer(got[nb:nb + 36].tobytes(), np.float32).reshape(3, 3).copy()
    est, H = (int(v) for v in np.frombuffer(got[nb + 36:nb + 44].tobytes(), np.int32))
    seed2 = int(np.frombuffer(got[nb + 44:nb + 52].tobytes(), np.uint64)[0])
    return out, K2, est, H, seed2


def _mul4_f32(A, B):
    """Eigen fixed-size 4x4 float product, coefficient-wise with the 4-term sum as (p0+p1)+(p2+p3)."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    C = np.empty((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            p = A[i, :] * B[:, j]
            C[i, j] = np.float32(np.float32(p[0] + p[1]) + np.float32(p[2] + p[3]))
    return C


def compose_trajectory(increments, gate=0.1):
    """VO pose composition of PUTSLAM::startProcessing (PUTSLAM.cpp:735-740).
    increments: (P,4,4) float32 normal (row, col) matrices.  Returns (P+1,4,4): VO_0 = I."""
    inc = np.asarray(increments, np.float32)
    out = np.empty((inc.shape[0] + 1, 4, 4), np.float32)
    out[0] = np.eye(4, dtype=np.float32)
    for k in range(inc.shape[0]):
        T = inc[k]
        tr = np.sqrt(float(T[0, 3]) ** 2 + float(T[1, 3]) ** 2 + float(T[2, 3]) ** 2)  # pow(double) + sqrt
        if tr > gate:
            T = np.eye(4, dtype=np.float32)
        out[k + 1] = _mul4_f32(out[k], T)
    return out


def rotation_to_quaternion_f32(R):
    """Eigen::Quaternion<float>(Matrix3f) (used at PUTSLAM.cpp:1011). Returns (x, y, z, w)."""
    m = np.asarray(R, np.float32)
    f = np.float32
    t = f(f(m[0, 0] + m[1, 1]) + m[2, 2])
    if t > f(0):
        t = np.sqrt(f(t + f(1.0)), dtype=np.float32)
        w = f(f(0.5) * t)
        t = f(f(0.5) / t)
        x = f(f(m[2, 1] - m[1, 2]) * t)
        y = f(f(m[0, 2] - m[2, 0]) * t)
        z = f(f(m[1, 0] - m[0, 1]) * t)
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(f(f(f(m[i, i] - m[j, j]) - m[k, k]) + f(1.0)), dtype=np.float32)
        q = [f(0)] * 3
        q[i] = f(f(0.5) * t)
        t = f(f(0.5) / t)
        w = f(f(m[k, j] - m[j, k]) * t)
        q[j] = f(f(m[j, i] + m[i, j]) * t)
        q[k] = f(f(m[k, i] + m[i, k]) * t)
        x, y, z = q
    return x, y, z, w


def format_tum_line(timestamp, T):
    """One line of saveTrajectoryFreiburgFormat (PUTSLAM.cpp:1006-1016): timestamp with precision 17,
    pose entries with the default stream precision (6 significant digits)."""
    x, y, z, w = rotation_to_quaternion_f32(np.asarray(T)[:3, :3])
    vals = [T[0][3], T[1][3], T[2][3], x, y, z, w]
    return ("%.17g" % float(timestamp)) + " " + " ".join("%g" % float(np.float32(v)) for v in vals)


def write_tum_trajectory(path, timestamps, poses):
    with open(path, "w") as f:
        for ts, T in zip(timestamps, poses):
            f.write(format_tum_line(ts, T) + "\n")
