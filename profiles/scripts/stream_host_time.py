#!/usr/bin/env python3
"""Where the host thread of the pipelined stream spends its time: seconds inside push_many, inside blocking pops, inside polls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from putslam_amd import api, synth  # noqa: E402
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config  # noqa: E402

ev, est, hyp = (int(sys.argv[1]), EST_FIXED, 4096) if len(sys.argv) > 1 and sys.argv[1] == "1" else (0, EST_RANSAC, 487)
seq = synth.make_sequence(500, 2000, config=3, index=0)
F, cap = seq["desc"].shape[:2]
hd, hp = api.PinnedBuffer((F, cap, 32), np.uint8), api.PinnedBuffer((F, cap, 3), np.float32)
hd.array[:] = seq["desc"]
hp.array[:] = seq["pts"]
nk = np.ascontiguousarray(seq["nkpts"], np.int32)
prm = default_ransac_params(ev)
cfg, _ = make_config(est, hyp, seed=0xB0B0)
ctx = api.Context(0)
for chunk, lanes in ((125, 0), (125, 4), (250, 0)):
    st = api.VoStream(ctx, cap)
    st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=chunk, lanes=lanes)
    T = {"push": 0.0, "push_busy": 0.0, "wait": 0.0, "poll": 0.0}
    N = {"push": 0, "push_busy": 0, "wait": 0, "poll": 0}

    def timed(key, fn):
        t = time.perf_counter()
        r = fn()
        T[key] += time.perf_counter() - t
        N[key] += 1
        return r

    def step():
        while not st.reset():
            timed("wait", lambda: st.pop_many(wait=True, copy=False))
        f = 0
        while f < F:
            n = min(chunk, F - f)
            t = time.perf_counter()
            ok = st.push_many(hd.array[f:f + n], hp.array[f:f + n], nk[f:f + n])
            dt = time.perf_counter() - t
            T["push" if ok else "push_busy"] += dt
            N["push" if ok else "push_busy"] += 1
            if ok:
                f += n
                while timed("poll", lambda: st.pop_many(wait=False, copy=False)) is not None:
                    pass
            else:
                timed("wait", lambda: st.pop_many(wait=True, copy=False))

    for _ in range(3):
        step()
    while st.pop_many(wait=True, copy=False) is not None:
        pass
    for k in T:
        T[k] = 0.0
        N[k] = 0
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    while timed("wait", lambda: st.pop_many(wait=True, copy=False)) is not None:
        pass
    el = time.perf_counter() - t0
    print(f"E{ev} chunk {chunk} lanes {lanes}: {30 * (F - 1) / el:9.0f} pairs/s, wall {el * 1e3:7.1f} ms: " +
          ", ".join(f"{k} {T[k] * 1e3:6.1f} ms / {N[k]} = {T[k] / max(N[k], 1) * 1e6:6.1f} us" for k in T), flush=True)
    st.close()
