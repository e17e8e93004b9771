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


def run_pairs(ctx, params, cfg, K, frames: FrameSetDevice, batch: PairBatchDevice, use_torch_stream=True):
    """Asynchronous: match -> cross-check -> RANSAC -> refit for every pair of the batch."""
    if use_torch_stream:
        ctx.set_stream(torch.cuda.current_stream(frames.device).cuda_stream)
    ctx.vo_pairs_device(params, cfg, K, frames.view(), batch.pairs.data_ptr(), batch.P, batch.view())


def run_pairs_split(ctxs, side_streams, params, estimator, num_hypotheses, seed, K, frames: FrameSetDevice,
                    batch: PairBatchDevice, bounds=None, join=True):
    """The same batch as `run_pairs`, submitted as len(ctxs) sub-batches on len(ctxs) HIP streams (ctxs[0] on the
    current torch stream, ctxs[i] on side_streams[i-1]); every context owns its scratch arena, results land in
    disjoint slices of `batch`.  Pair p keeps its hypothesis stream (seed + p), so the outputs are bit-identical
    to the single call.  With join=True the current stream waits for the side streams before returning."""
    from . import api
    from ._abi import make_config
    S = len(ctxs)
    P = batch.P
    if bounds is None:
        bounds = [P * i // S for i in range(S + 1)]
    cur = torch.cuda.current_stream(frames.device)
    keep = []
    for i in range(S):
        lo, hi = bounds[i], bounds[i + 1]
        if hi <= lo:
            continue
        st = cur if i == 0 else side_streams[i - 1]
        if i > 0 and join:
            st.wait_stream(cur)
        ctxs[i].set_stream(st.cuda_stream)
        ci, k = make_config(estimator, num_hypotheses, seed=seed + lo)
        keep.append(k)
        view = api.DeviceResults(batch.matches[lo:].data_ptr(), batch.num_matches[lo:].data_ptr(),
                                 batch.mask[lo:].data_ptr(), batch.pose[lo:].data_ptr(), batch.stats[lo:].data_ptr())
        ctxs[i].vo_pairs_device(params, ci, K, frames.view(), batch.pairs[lo:].data_ptr(), hi - lo, view)
    if join:
        for st in side_streams[:S - 1]:
            cur.wait_stream(st)
