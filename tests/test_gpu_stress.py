"""Race / determinism net for the staged scoring (ps_score_fast.h): seven dependent launches per step, survivor lists appended
by per-wavefront atomics, LDS meeting slots, (B0, L0) published between launches.  A data race there would show as a result
that differs from run to run once in thousands of steps -- so the SAME staged batch is run many times on three concurrent
chains (three contexts, three HIP streams, the bench's submission) while a fourth context keeps the chip busy with a different
batch, and every output byte of every run must equal the first run's and the complete sweep's (prune = 0).  What has to be
preserved is the reference's sequential selection rule, RANSAC.cpp:438-455, whatever the interleaving of work-groups."""
import numpy as np
import pytest

from putslam_amd import api, synth
from putslam_amd._abi import (EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_AND_REPROJECTION_ERROR, EUCLIDEAN_ERROR,
                              REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params)

pytestmark = pytest.mark.gpu

REPS = 200
KPTS = 2000
FRAMES = 241   # 240 pairs = 3 chains x 80: above the staged threshold of every case below (P (ceil(H/256) - 1) >= 256 / 768)


@pytest.fixture(scope="module")
def rig():
    import torch
    from putslam_amd.device_batch import FrameSetDevice
    seq = synth.make_sequence(FRAMES, KPTS, config=3, index=9001, inlier_frac=0.7, noise=0.004)
    # a different batch for the fourth context: fewer keypoints, harder data, other sizes
    oth = synth.make_sequence(90, 700, config=3, index=9002, inlier_frac=0.35, noise=0.01)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    fo = FrameSetDevice(oth["desc"], oth["pts"], oth["nkpts"])
    ctxs = [api.Context(0) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    bg_ctx, bg_stream = api.Context(0), torch.cuda.Stream()
    yield dict(seq=seq, oth=oth, fs=fs, fo=fo, ctxs=ctxs, streams=streams, bg_ctx=bg_ctx, bg_stream=bg_stream, torch=torch)
    torch.cuda.synchronize()
    for c in ctxs + [bg_ctx]:
        c.close()


def _tensors(pb):
    return (pb.matches, pb.num_matches, pb.mask, pb.pose, pb.stats)


@pytest.mark.parametrize("mode,est,H", [
    (EUCLIDEAN_ERROR, EST_FIXED, 4096),
    (REPROJECTION_ERROR, EST_FIXED, 4096),
    (EUCLIDEAN_ERROR, EST_RANSAC, 1157),
    (REPROJECTION_ERROR, EST_USAC, 3000),
    (EUCLIDEAN_AND_REPROJECTION_ERROR, EST_FIXED, 4096),
])
@pytest.mark.parametrize("reorder", [2, 1])
def test_staged_batch_is_deterministic_under_contention(rig, mode, est, H, reorder):
    torch = rig["torch"]
    from putslam_amd._abi import make_config
    from putslam_amd.device_batch import PairBatchDevice, run_pairs, run_pairs_split
    seq, fs, ctxs, streams = rig["seq"], rig["fs"], rig["ctxs"], rig["streams"]
    prm = default_ransac_params(mode, lc=(H == 1157))
    P = len(seq["pairs"])

    def submit(prune, join=True):
        for c in ctxs:
            c.set_option("prune", 2 if prune else 0)   # (2: the staged form whatever the cost model says of this batch size)
            c.set_option("reorder", reorder)
        pb = PairBatchDevice(seq["pairs"], fs.max_kpts)   # fresh zeroed outputs: a missing write shows too
        run_pairs_split(ctxs, streams, prm, est, H, 4242, TUM_FR1_K, fs, pb, join=join)
        return pb

    # background: another batch, another mode, queued again and again on a fourth context / stream, never joined until the end
    bg_ctx, bg_stream, fo, oth = rig["bg_ctx"], rig["bg_stream"], rig["fo"], rig["oth"]
    bg_prm = default_ransac_params(REPROJECTION_ERROR if mode == EUCLIDEAN_ERROR else EUCLIDEAN_ERROR)
    bg_cfg, _ = make_config(EST_FIXED, 2048, seed=77)
    bg_ctx.set_option("reorder", 1)
    bg_first = None
    bg_runs = []

    full = submit(0)
    torch.cuda.synchronize()
    assert all(c.get_option("last_staged_pairs") == 0 for c in ctxs)
    ref = [t.clone() for t in _tensors(full)]
    assert int(ref[1].sum().item()) > 0
    for rep in range(REPS):
        with torch.cuda.stream(bg_stream):
            for _ in range(2):
                pbo = PairBatchDevice(oth["pairs"], fo.max_kpts)
                run_pairs(bg_ctx, bg_prm, bg_cfg, TUM_FR1_K, fo, pbo)
                bg_runs.append(pbo)
        pb = submit(1)
        if rep % 2:   # every other repetition a second submission right behind the first, chains only ordered within their
            pb2 = submit(1, join=False)   # own stream: consecutive steps pipeline into each other (the bench's form)
        torch.cuda.synchronize()
        assert all(c.get_option("last_staged_pairs") > 0 for c in ctxs)   # the staged form did run
        for name, a, b in zip(("matches", "numMatches", "mask", "pose", "stats"), _tensors(pb), ref):
            assert torch.equal(a, b), (rep, name, mode, est, H, reorder)
        if rep % 2:
            for name, a, b in zip(("matches", "numMatches", "mask", "pose", "stats"), _tensors(pb2), ref):
                assert torch.equal(a, b), (rep, name, "second submission")
        if len(bg_runs) >= 8:
            torch.cuda.synchronize()
            for r in bg_runs:
                cur = [t.clone() for t in _tensors(r)]
                if bg_first is None:
                    bg_first = cur
                for a, b in zip(cur, bg_first):
                    assert torch.equal(a, b), (rep, "background batch")
            bg_runs = []
    torch.cuda.synchronize()
    for c in ctxs:
        c.set_option("prune", 1)
        c.set_option("reorder", 2)
