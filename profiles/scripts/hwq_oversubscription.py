#!/usr/bin/env python3
"""Does a parent process that holds many (idle) hardware queues slow a child process' chains?  The parent creates K torch streams
(GPU_MAX_HW_QUEUES=16: up to 16 distinct hardware queues), runs a kernel on each, then starts demos/cpp/demo_sequences_multi_gpu
--gpus 1 (4 chains + comm stream + RCCL's) and demo_batch_queue (4 chains) as children.
    python profiles/scripts/hwq_oversubscription.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
held = []
for K in (0, 4, 8, 12, 16, 24):
    while len(held) < K:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x = torch.zeros(1024, device="cuda") + 1
        held.append((s, x))
    torch.cuda.synchronize()
    out = []
    for exe, args, pat in (("demo_sequences_multi_gpu", ["--gpus", "1", "--steps", "20", "--repeats", "5", "--warm-seconds", "0.5"], r"median ([0-9.]+) frame-pairs/s in all"),
                           ("demo_batch_queue", ["--frames", "500", "--steps", "20", "--repeats", "5", "--warm-seconds", "0.5"], r"median ([0-9.]+) frame-pairs/s")):
        p = subprocess.run([os.path.join(ROOT, "demos", "cpp", exe)] + args, capture_output=True, text=True, env=env, timeout=300)
        m = re.search(pat, p.stdout)
        out.append("%s %s" % (exe, m.group(1) if m else "?"))
    print("parent holds %2d streams: %s" % (K, "   ".join(out)), flush=True)
