#!/bin/bash
# rocprofv3 timeline (kernels + memory copies) of the pipelined stream: gpurun_out/<tag>/stream_trace/
# usage: stream_trace.sh <tag> [stream_sweep.py arguments]
tag=${1:-r05}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag/stream_trace
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- python3 "$GRAFT_REPO_ROOT/profiles/scripts/stream_sweep.py" --steps 10 "$@" > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
kern = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)
mem = glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True)
ev = []
for f in kern:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in mem:
    rows = list(csv.DictReader(open(f)))
    if rows:
        print("memory copy columns:", list(rows[0].keys()))
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")), r.get("Stream_Id", "")))
ev.sort()
t0 = ev[0][0]
# the last 2 ms of the trace, event by event
tail = [e for e in ev if e[0] > ev[-1][1] - 3_000_000]
with open(out + "/timeline_tail.txt", "w") as f:
    for s, e, k, name, st in tail:
        f.write(f"{(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:9.1f} us  {k} {name} [{st}]\n")
# copy statistics by (direction, size)
agg = collections.defaultdict(list)
for s, e, k, name, st in ev:
    if k == "M":
        agg[name].append((e - s) / 1e3)
with open(out + "/copy_stats.txt", "w") as f:
    for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        try:
            b = int(name.split()[-1])
        except ValueError:
            b = 0
        f.write(f"{name:40s} n={len(d):5d} median {d[len(d)//2]:9.1f} us  p95 {d[int(len(d)*0.95)]:9.1f} us  "
                f"{(b / (d[len(d)//2] * 1e-6) / 1e9 if b else 0):6.1f} GB/s at the median\n")
print(open(out + "/copy_stats.txt").read())
P
rm -rf "$out/raw"
