#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace --stats --output-format csv run: calls, mean and median duration,
total per `steps` steps.  usage: trace_stats.py <dir with *_kernel_trace.csv> <steps in the run>"""
import csv, glob, os, sys
from collections import defaultdict
import numpy as np
d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
dur = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void psdev::", "").replace("void ", "")
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if not n.startswith("ps_"):
        continue
    v = np.array(v)
    # the median launch x launches per step: robust against the few slow launches of the warm-up / statistics passes
    per_step = np.median(v) * round(len(v) / steps)
    tot += per_step
    print(f"{n:42s} launches/step {len(v) / steps:5.2f}  median {np.median(v):8.1f} us  mean {v.mean():8.1f} us  per step {per_step:8.1f} us")
print(f"sum per step {tot:.1f} us")
