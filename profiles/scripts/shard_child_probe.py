#!/usr/bin/env python3
"""What in a parent process slows the native shard demo run as its child (bench.py: 590 k before its streamed legs, 540 k after)?
The child is run (a) first, (b) after the parent made pinned copies on a stream of its own, (c) after it ran a pipelined stream,
(d) after that stream was closed.    python profiles/scripts/shard_child_probe.py"""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402
from putslam_amd import api, synth  # noqa: E402
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config  # noqa: E402

env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}


def child(tag):
    out = []
    for exe, args, pat in (("demo_sequences_multi_gpu", ["--gpus", "1", "--steps", "20", "--repeats", "5", "--warm-seconds", "0.5"], r"median ([0-9.]+) frame-pairs/s in all"),
                           ("demo_sequences_multi_gpu", ["--gpus", "1", "--steps", "20", "--repeats", "5", "--warm-seconds", "0.5", "--blocking"], r"median ([0-9.]+) frame-pairs/s in all"),
                           ("demo_batch_queue", ["--frames", "500", "--steps", "20", "--repeats", "5", "--warm-seconds", "0.5"], r"median ([0-9.]+) frame-pairs/s")):
        p = subprocess.run([os.path.join(ROOT, "demos", "cpp", exe)] + args, capture_output=True, text=True, env=env, timeout=300)
        m = re.search(pat, p.stdout)
        out.append("%s%s %s" % (exe[5:14], " blocking" if "--blocking" in args else "", m.group(1) if m else "?"))
    print("%-52s %s" % (tag, "   ".join(out)), flush=True)


torch.zeros(1, device="cuda")
child("(a) parent: torch initialised")
s = torch.cuda.Stream()
h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
with torch.cuda.stream(s):
    for _ in range(20):
        d.copy_(h, non_blocking=True)
        h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
child("(b) + pinned copies both ways on a stream")
seq = synth.make_sequence(500, 2000, config=3, index=0)
F, cap = seq["desc"].shape[:2]
hd, hp = api.PinnedBuffer((F, cap, 32), np.uint8), api.PinnedBuffer((F, cap, 3), np.float32)
hd.array[:] = seq["desc"]
hp.array[:] = seq["pts"]
nk = np.ascontiguousarray(seq["nkpts"], np.int32)
prm = default_ransac_params(1)
cfg, _ = make_config(EST_FIXED, 4096, seed=1)
ctx = api.Context(0)
st = api.VoStream(ctx, cap)
st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=250, lanes=0)
for _ in range(10):
    f = 0
    while f < F:
        n = min(250, F - f)
        if st.push_many(hd.array[f:f + n], hp.array[f:f + n], nk[f:f + n]):
            f += n
        else:
            st.pop_many(wait=True, copy=False)
    while st.pop_many(wait=True, copy=False) is not None:
        pass
    st.reset()
child("(c) + a pipelined stream ran (still configured)")
import time  # noqa: E402


def busy(tag):
    t0, w0 = time.process_time(), time.perf_counter()
    time.sleep(1.0)
    print("    parent CPU while idle %s: %.2f core-seconds per second" % (tag, (time.process_time() - t0) / (time.perf_counter() - w0)), flush=True)


busy("after the chunk-250 stream")
hpk = api.PinnedBuffer((F, (cap * 44 + 15) // 16 * 16), np.uint8)
from putslam_amd.device_batch import pack_frames  # noqa: E402
hpk.array[:] = pack_frames(seq["desc"], seq["pts"], hpk.array.shape[1])
for chunk in (1, 4):
    st2 = api.VoStream(ctx, cap)
    st2.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=chunk, lanes=0, packed=True)
    for _ in range(3):
        f = 0
        while f < F:
            n = min(chunk, F - f)
            if st2.push_many_packed(hpk.array[f:f + n], nk[f:f + n]):
                f += n
                while st2.pop_many(wait=False, copy=False) is not None:
                    pass
            else:
                st2.pop_many(wait=True, copy=False)
        while st2.pop_many(wait=True, copy=False) is not None:
            pass
        st2.reset()
    st2.close()
    busy("after the chunk-%d stream (closed)" % chunk)
    child("(c%d) + a mini stream of %d frame(s) per chunk ran and was closed" % (chunk, chunk))
hpk.close()
st.close()
ctx.close()
hd.close()
hp.close()
child("(d) + that stream, its context and buffers closed")
