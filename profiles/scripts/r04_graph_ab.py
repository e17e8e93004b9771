#!/usr/bin/env python3
"""ps_vo_pairs_device with and without the replay of repeated calls as a captured hipGraph (option "graph", round 4 trial):
one 2000-keypoint pair per call (synchronised after every call: the latency shape) and 499 pairs per call (pipelined)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
for frames, sync_each in ((2, True), (500, False)):
    seq = synth.make_sequence(frames, 2000, config=3, index=0)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    for ev, est, H in ((0, EST_RANSAC, 487), (1, EST_FIXED, 4096)):
        prm = default_ransac_params(ev); cfg, _ = make_config(est, H, seed=3)
        res = {}
        ctxs = {g: api.Context(0) for g in (0, 1)}
        for g, c in ctxs.items():
            c.set_option("graph", g)
        for rnd in range(5):
            for g, c in ctxs.items():
                for _ in range(10):
                    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
                torch.cuda.synchronize()
                n = 200 if sync_each else 40
                t0 = time.perf_counter()
                for _ in range(n):
                    run_pairs(c, prm, cfg, TUM_FR1_K, fs, pb)
                    if sync_each:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                res.setdefault(g, []).append((time.perf_counter() - t0) / n * 1e6)
        print(f"pairs/call {frames - 1} E{ev} H{H}: graph off {np.median(res[0]):.1f} us/call, graph on {np.median(res[1]):.1f} us/call "
              f"(replays {ctxs[1].get_option('graph_launches')})", flush=True)
        for c in ctxs.values():
            c.close()
