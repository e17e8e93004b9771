#!/usr/bin/env python3
"""Alternating A/B of the pipelined stream's shapes on ONE box in ONE process: every shape `--turns` times, round robin (box to box
and warm-up differences are larger than most of the effects looked for).  Shapes: chunk x lanes a ahead, e.g. 125x6a0 125x4a4.
    python profiles/scripts/stream_ab.py 125x6a0 125x4a0 125x4a4 125x3a3 125x3a6 [--turns 4] [--results 0]"""
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
    ap.add_argument("shapes", nargs="+")
    ap.add_argument("--turns", type=int, default=4)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--results", type=int, default=0)
    ap.add_argument("--ev", type=int, default=1)
    a = ap.parse_args()
    from putslam_amd import api, synth
    from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
    seq = synth.make_sequence(500, 2000, config=3, index=0)
    F, cap = seq["desc"].shape[:2]
    hd, hp = api.PinnedBuffer((F, cap, 32), np.uint8), api.PinnedBuffer((F, cap, 3), np.float32)
    hd.array[:] = seq["desc"]
    hp.array[:] = seq["pts"]
    nk = np.ascontiguousarray(seq["nkpts"], np.int32)
    prm = default_ransac_params(a.ev)
    cfg, _ = make_config(EST_FIXED if a.ev else EST_RANSAC, 4096 if a.ev else 487, seed=0xB0B0)
    ctx = api.Context(0)
    rates = {s: [] for s in a.shapes}
    for turn in range(a.turns + 1):                      # (turn 0 warms the chip and is dropped)
        for shape in a.shapes:
            item, ahead = shape.split("a")
            chunk, lanes = (int(v) for v in item.split("x"))
            ctx.set_option("stream_ahead", int(ahead))
            st = api.VoStream(ctx, cap)
            st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=chunk, lanes=lanes, results=a.results)
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
                    if st.push_many(hd.array[f:f + n], hp.array[f:f + n], nk[f:f + n]):
                        f += n
                        while take(False):
                            pass
                    else:
                        take(True)

            for _ in range(4):
                step()
            while take(True):
                pass
            d0, t0 = done[0], time.perf_counter()
            for _ in range(a.steps):
                step()
            while take(True):
                pass
            if turn:
                rates[shape].append((done[0] - d0) / (time.perf_counter() - t0))
            st.close()
    for shape in a.shapes:
        r = np.array(rates[shape])
        print(f"{shape:>10s}: median {np.median(r):9.0f} pairs/s   min {r.min():9.0f}  max {r.max():9.0f}   ({len(r)} turns)")


if __name__ == "__main__":
    main()
