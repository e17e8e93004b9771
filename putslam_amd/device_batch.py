"""Device-resident frame sets and pair batches for ps_vo_pairs_device.

PyTorch is used here only as the HBM allocator / stream owner (torch.cuda on ROCm); all
compute happens in libputslam_hip.so behind the C ABI.
"""
import numpy as np
import torch

from . import api
from ._abi import DMATCH_DTYPE, STATS_DTYPE


class FrameSetDevice:
    """desc (F,cap,32) u8, pts (F,cap,3) f32, nkpts (F,) i32 resident in HBM."""

    def __init__(self, desc, pts, nkpts, device="cuda:0"):
        desc = np.ascontiguousarray(desc, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32)
        nkpts = np.ascontiguousarray(nkpts, np.int32)
        assert desc.ndim == 3 and desc.shape[2] == 32 and pts.shape == desc.shape[:2] + (3,)
        self.device = torch.device(device)
        self.desc = torch.from_numpy(desc).to(self.device)
        self.pts = torch.from_numpy(pts).to(self.device)
        self.nkpts = torch.from_numpy(nkpts).to(self.device)
        self.num_frames, self.max_kpts = desc.shape[0], desc.shape[1]

    def view(self):
        return api.DeviceFrames(self.desc.data_ptr(), self.pts.data_ptr(), self.nkpts.data_ptr(), self.num_frames,
                                self.max_kpts)


def pack_frames(desc, pts, stride=None):
    """(F, cap, 32) u8 + (F, cap, 3) f32 -> (F, stride) u8: every frame one block [cap x 32 B descriptors][cap x 12 B points]
    padded to `stride` bytes (default: cap x 44 rounded up to 16 -- PS_FRAMES_PACKED / ps_vo_stream_packed_stride)."""
    F, cap = desc.shape[:2]
    stride = ((cap * 44 + 15) // 16) * 16 if stride is None else int(stride)
    out = np.zeros((F, stride), np.uint8)
    out[:, :cap * 32] = np.ascontiguousarray(desc, np.uint8).reshape(F, cap * 32)
    out[:, cap * 32:cap * 44] = np.ascontiguousarray(pts, np.float32).reshape(F, cap * 3).view(np.uint8)
    return out


class PackedFrameSetDevice:
    """The same frames as FrameSetDevice with every frame's descriptors and points in ONE block (PsFrameSet strides, ABI 2)."""

    def __init__(self, desc, pts, nkpts, device="cuda:0", stride=None):
        self.device = torch.device(device)
        packed = pack_frames(desc, pts, stride)
        self.num_frames, self.max_kpts = desc.shape[0], desc.shape[1]
        self.stride = packed.shape[1]
        self.blocks = torch.from_numpy(packed).to(self.device)
        self.nkpts = torch.from_numpy(np.ascontiguousarray(nkpts, np.int32)).to(self.device)

    def view(self):
        base = self.blocks.data_ptr()
        return api.DeviceFrames(base, base + self.max_kpts * 32, self.nkpts.data_ptr(), self.num_frames, self.max_kpts,
                                self.stride, self.stride)


class PairBatchDevice:
    """Pairs (P,2) i32 and the per-pair outputs, all in HBM."""

    def __init__(self, pairs, max_kpts, device="cuda:0"):
        pairs = np.ascontiguousarray(pairs, np.int32)
        self.device = torch.device(device)
        self.P = pairs.shape[0]
        self.cap = int(max_kpts)
        self.pairs = torch.from_numpy(pairs).to(self.device)
        P, cap = max(self.P, 1), self.cap
        self.matches = torch.zeros((P, cap, 16), dtype=torch.uint8, device=self.device)
        self.num_matches = torch.zeros(P, dtype=torch.int32, device=self.device)
        self.mask = torch.zeros((P, cap), dtype=torch.uint8, device=self.device)
        self.pose = torch.zeros((P, 16), dtype=torch.float32, device=self.device)
        self.stats = torch.zeros((P, STATS_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        # torch fills the blocks on its current stream; the library's chains run on streams of their own that are NOT ordered with
        # it (non-blocking streams): a fill still pending when a chain writes results would zero them afterwards (found by the queue
        # fuzz on batches of three pairs).  The blocks are ready when the constructor returns.
        torch.cuda.current_stream(self.device).synchronize()

    def view(self):
        return api.DeviceResults(self.matches.data_ptr(), self.num_matches.data_ptr(), self.mask.data_ptr(),
                                 self.pose.data_ptr(), self.stats.data_ptr())

    def download(self):
        torch.cuda.synchronize(self.device)
        P = self.P
        return dict(matches=self.matches.cpu().numpy().view(DMATCH_DTYPE).reshape(max(P, 1), self.cap)[:P],
                    numMatches=self.num_matches.cpu().numpy()[:P],
                    inlierMask=self.mask.cpu().numpy()[:P],
                    pose=self.pose.cpu().numpy()[:P],
                    stats=self.stats.cpu().numpy().view(STATS_DTYPE).reshape(-1)[:P])


_SIDE_STREAMS = {}


def _work_stream(device):
    """A non-default torch stream for the calling device.  The C ABI takes a hipStream_t and reads NULL as "the
    context's private stream", so torch's legacy default stream (handle 0) cannot be handed over: work submitted
    from the default stream runs on this side stream instead, forked from and joined back to the default stream."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def run_pairs(ctx, params, cfg, K, frames: FrameSetDevice, batch: PairBatchDevice, use_torch_stream=True):
    """Asynchronous: match -> cross-check -> RANSAC -> refit for every pair of the batch, ordered after the work
    already queued on torch's current stream; the current stream waits for the results."""
    if not use_torch_stream:
        ctx.vo_pairs_device(params, cfg, K, frames.view(), batch.pairs.data_ptr(), batch.P, batch.view())
        return
    cur = torch.cuda.current_stream(frames.device)
    if cur.cuda_stream != 0:
        ctx.set_stream(cur.cuda_stream)
        ctx.vo_pairs_device(params, cfg, K, frames.view(), batch.pairs.data_ptr(), batch.P, batch.view())
        return
    st = _work_stream(frames.device)
    st.wait_stream(cur)
    ctx.set_stream(st.cuda_stream)
    ctx.vo_pairs_device(params, cfg, K, frames.view(), batch.pairs.data_ptr(), batch.P, batch.view())
    cur.wait_stream(st)


def run_pairs_split(ctxs, streams, params, estimator, num_hypotheses, seed, K, frames: FrameSetDevice,
                    batch: PairBatchDevice, bounds=None, join=True):
    """The same batch as `run_pairs`, submitted as len(ctxs) sub-batches on len(ctxs) HIP streams (`streams`: one
    non-default torch stream per context); every context owns its scratch arena, results land in disjoint slices of
    `batch`.  Pair p keeps its hypothesis stream (seed + p), so the outputs are bit-identical to the single call.
    join=True: every stream first waits for the current stream and the current stream waits for all of them at the
    end (the call is then ordered like `run_pairs`).  join=False: the sub-batch chains are only ordered within their
    own stream -- consecutive calls pipeline into each other and the caller synchronises before reading results."""
    from . import api
    from ._abi import make_config
    S = len(ctxs)
    assert len(streams) >= S and all(st.cuda_stream != 0 for st in streams[:S])
    P = batch.P
    if bounds is None:
        # (two chains: 45 % / 55 % -- unequal sub-batches stay out of step, one chain's matrix-core Hamming sweep beside the
        # other's vector scoring sweep; equal ones march in lockstep and lose 2 - 3 %, profiles/r05k/chains_ab.txt)
        bounds = [0, int(P * 0.45), P] if (S == 2 and P >= 20) else [P * i // S for i in range(S + 1)]
    cur = torch.cuda.current_stream(frames.device)
    if join:
        for st in streams[:S]:
            st.wait_stream(cur)
    keep = []
    for i in range(S):
        lo, hi = bounds[i], bounds[i + 1]
        if hi <= lo:
            continue
        ctxs[i].set_stream(streams[i].cuda_stream)
        ci, k = make_config(estimator, num_hypotheses, seed=seed + lo)
        keep.append(k)
        view = api.DeviceResults(batch.matches[lo:].data_ptr(), batch.num_matches[lo:].data_ptr(),
                                 batch.mask[lo:].data_ptr(), batch.pose[lo:].data_ptr(), batch.stats[lo:].data_ptr())
        ctxs[i].vo_pairs_device(params, ci, K, frames.view(), batch.pairs[lo:].data_ptr(), hi - lo, view)
    if join:
        for st in streams[:S]:
            cur.wait_stream(st)


def run_pairs_queue(queue, params, cfg, K, frames: FrameSetDevice, batch: PairBatchDevice):
    """The same batch as `run_pairs` through a PsBatchQueue (api.BatchQueue): the library hands the batches to its launch chains
    (two) in turn, whole; the chains are never joined, so consecutive batches run side by side -- give them output blocks of their
    own (two `PairBatchDevice`s used in turn).  Asynchronous; returns the batch's ticket -- `queue.wait(ticket)` (host),
    `queue.wait_on_stream(ticket, stream)` (a stream of the caller's) or `queue.synchronize()` before the results are read."""
    return queue.submit(params, cfg, K, frames.view(), batch.pairs.data_ptr(), batch.P, batch.view())
