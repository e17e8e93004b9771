#!/bin/bash
# rocprofv3 kernel timeline of the synchronous per-frame call (ps_vo_stream_push) and of the single-pair device call:
# gpurun_out/<tag>/push_trace/timeline.txt = the last pushes, kernel by kernel, with the idle time in front of each
tag=${1:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag/push_trace
mkdir -p "$out"
cat > /tmp/push_loop.py <<'P'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
ctx = api.Context(0)
seq = synth.make_sequence(40, 2000, config=3, index=5)
prm = default_ransac_params(0)
st = api.VoStream(ctx, 2000)
ts = []
for rep in range(4):
    for f in range(40):
        cfg, _ = make_config(EST_RANSAC, 487, seed=f + 1)
        t0 = time.perf_counter(); st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f]); ts.append(time.perf_counter() - t0)
print("pushed frame median %.1f us" % (np.median(ts[40:]) * 1e6))
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- python3 /tmp/push_loop.py > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'P'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]))
ev.sort()
tail = ev[-6 * 12:]
with open(out + "/timeline.txt", "w") as f:
    prev = tail[0][0]
    for s, e, name in tail:
        f.write(f"idle {(s - prev) / 1e3:8.1f} us   run {(e - s) / 1e3:7.1f} us   {name}\n")
        prev = e
print(open(out + "/timeline.txt").read())
print(open(out + "/run.txt").read()[-300:])
P
