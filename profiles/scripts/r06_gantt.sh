#!/bin/bash
# Text Gantt of a slice of the pipelined stream's timeline (rocprofv3 kernel + memory-copy trace): every event of a 3-ms slice with
# its queue / stream.  usage (GPU box): bash profiles/scripts/r06_gantt.sh r06u 125x2a2
tag=${1:-r06u}; grid=${2:-125x2a2}
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/$tag/gantt_$grid
mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- python3 "$GRAFT_REPO_ROOT/profiles/scripts/stream_sweep.py" --steps 25 --warm 1.0 --packed 1 --grid $grid > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1)
python3 - "$out" <<'P'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", ""), r["Kernel_Name"].split("(")[0].split("::")[-1][:28]))
for f in glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "?").replace("MEMORY_COPY_", "")))
ev.sort()
end = ev[-1][1]
w0, w1 = end - 9_000_000, end - 6_000_000
with open(out + "/gantt.txt", "w") as f:
    f.write(open(out + "/run.txt").read().strip().splitlines()[-1][:120] + "\n")
    for s, e, q, n in ev:
        if s >= w0 and s <= w1:
            f.write("%8.1f %8.1f  %-5s %s\n" % ((s - w0) / 1e3, (e - s) / 1e3, q, n))
P
rm -rf "$out/raw"
wc -l $out/gantt.txt
