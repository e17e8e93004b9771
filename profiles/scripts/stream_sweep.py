#!/usr/bin/env python3
"""Pipelined streaming (ps_vo_stream_push_many / pop_many) on the bench workload: pairs/s over chunk size x lanes.
    python profiles/scripts/stream_sweep.py [--ev 1] [--est fixed] [--hyp 4096] [--frames 500] [--kpts 2000]
Frames start in pinned host memory; every step uploads all of them and downloads every pair's results."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ev", type=int, default=1)
    ap.add_argument("--est", default="fixed")
    ap.add_argument("--hyp", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warm", type=float, default=0.5, help="seconds of untimed steps before every row")
    ap.add_argument("--results", type=int, default=0, help="0 = everything, 1 = inlier matches + pose + stats, 2 = pose + stats")
    ap.add_argument("--packed", type=int, default=0, help="1 = PS_FRAMES_PACKED: one block per frame, one upload per chunk (push_many_packed)")
    ap.add_argument("--idle-contexts", type=int, default=0, help="contexts (stream + arena each) created before the sweep and left idle")
    ap.add_argument("--idle-queue", type=int, default=0, help="a PsBatchQueue of that many chains, warmed with one batch, left idle")
    ap.add_argument("--grid", default="125x4,125x6,125x8,166x4,166x6,250x4,250x6,64x8,32x8")
    a = ap.parse_args()
    from putslam_amd import api, synth
    from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config
    est = {"fixed": EST_FIXED, "ransac": EST_RANSAC, "usac": EST_USAC}[a.est]
    seq = synth.make_sequence(a.frames, a.kpts, config=3, index=0)
    F, cap = seq["desc"].shape[:2]
    hd, hp = api.PinnedBuffer((F, cap, 32), np.uint8), api.PinnedBuffer((F, cap, 3), np.float32)
    hd.array[:] = seq["desc"]
    hp.array[:] = seq["pts"]
    nk = np.ascontiguousarray(seq["nkpts"], np.int32)
    if a.packed:
        from putslam_amd.device_batch import pack_frames
        hpk = api.PinnedBuffer((F, (cap * 44 + 15) // 16 * 16), np.uint8)
        hpk.array[:] = pack_frames(seq["desc"], seq["pts"], hpk.array.shape[1])
    prm = default_ransac_params(a.ev)
    cfg, _ = make_config(est, a.hyp, seed=0xB0B0)
    ctx = api.Context(0)
    idle = [api.Context(0) for _ in range(a.idle_contexts)]
    if a.idle_queue:
        from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue
        iq = api.BatchQueue(ctx, a.idle_queue)
        ifs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
        ipb = [PairBatchDevice(seq["pairs"], ifs.max_kpts) for _ in range(a.idle_queue)]
        for k in range(2 * a.idle_queue):
            run_pairs_queue(iq, prm, cfg, TUM_FR1_K, ifs, ipb[k % a.idle_queue])
        iq.synchronize()
    for item in a.grid.split(","):
        ahead = 2                                   # "125x4a0": chunk x lanes, upload-ahead depth (option stream_ahead)
        if "a" in item:
            item, ahead = item.split("a")[0], int(item.split("a")[1])
        chunk, lanes = (int(v) for v in item.split("x"))
        ctx.set_option("stream_ahead", ahead)
        st = api.VoStream(ctx, cap)
        st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=chunk, lanes=lanes, results=a.results, packed=bool(a.packed))
        done = [0]

        def take(wait):
            b = st.pop_many(wait=wait, copy=False)
            if b is None:
                return False
            done[0] += b["count"]
            return True

        def step():
            while not st.reset():
                take(True)
            f = 0
            while f < F:
                n = min(chunk, F - f)
                if (st.push_many_packed(hpk.array[f:f + n], nk[f:f + n]) if a.packed else st.push_many(hd.array[f:f + n], hp.array[f:f + n], nk[f:f + n])):
                    f += n
                    while take(False):
                        pass
                else:
                    take(True)

        tw = time.perf_counter()
        while time.perf_counter() - tw < a.warm:          # (the first rows of an un-warmed sweep ran on a chip at a low clock)
            step()
        while take(True):
            pass
        d0 = done[0]
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        while take(True):
            pass
        el = time.perf_counter() - t0
        print(f"chunk {chunk:4d} lanes {lanes} ahead {ahead}: {(done[0] - d0) / el:10.0f} pairs/s  {el / a.steps * 1e3:7.3f} ms/step  "
              f"H2D {a.steps * F * cap * 44 / el / 1e9:5.1f} GB/s", flush=True)
        st.close()


if __name__ == "__main__":
    main()
