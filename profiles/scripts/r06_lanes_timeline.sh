#!/bin/bash
# rocprofv3 timelines: the pipelined stream at chunks of 125 on 2 / 4 lanes against 125-pair batches through a PsBatchQueue of 2 / 4
# chains (no copies): what the stream's extra lanes do not get.  usage (GPU box): bash profiles/scripts/r06_lanes_timeline.sh r06u
tag=${1:-r06u}
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
run() { # name, command...
  name=$1; shift
  out=gpurun_out/$tag/tl_$name
  mkdir -p "$out"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- "$@" > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1)
  python3 - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id", "")))
for f in glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r.get("Direction", "?"), r.get("Stream_Id", "")))
ev.sort()
end = ev[-1][1]
w0, w1 = end - 22_000_000, end - 2_000_000
win = [e for e in ev if e[0] >= w0 and e[1] <= w1]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)
W = w1 - w0
ker = [(s, e) for s, e, k, n, q in win if k == "K" and "ps_" in n]
with open(out + "/summary.txt", "w") as f:
    f.write(open(out + "/run.txt").read().strip().splitlines()[-1][:160] + "   (under rocprofv3)\n")
    f.write("window %.1f ms, %d path kernels; chip busy %.1f %%; side by side %.2f\n" % (W / 1e6, len(ker), 100.0 * union(ker) / W, sum(e - s for s, e in ker) / max(union(ker), 1)))
    by = collections.defaultdict(list)
    for s, e, k, n, q in win:
        by[(k, n)].append((e - s) / 1e3)
    for (k, n), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        f.write("  %s %-46s n %5d  sum %8.1f us (%4.1f %% of window)  median %7.1f  p90 %7.1f\n" % (k, n, len(d), sum(d), 100.0 * sum(d) / (W / 1e3), d[len(d) // 2], d[int(len(d) * 0.9)]))
    # per queue: busy share and the gaps between consecutive kernels
    q = collections.defaultdict(list)
    for s, e, k, n, qq in win:
        if k == "K": q[qq].append((s, e))
    for qq, iv in sorted(q.items()):
        iv.sort()
        gaps = sorted((iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1))
        f.write("  queue %s: %d kernels, busy %.1f %%, gap median %.1f us p90 %.1f us\n" % (qq, len(iv), 100.0 * union(iv) / W, gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * 0.9)] if gaps else 0))
print(open(out + "/summary.txt").read())
P
  rm -rf "$out/raw"
}
run stream_125x2 python3 "$GRAFT_REPO_ROOT/profiles/scripts/stream_sweep.py" --steps 25 --warm 1.0 --packed 1 --grid 125x2a1
run stream_125x4 python3 "$GRAFT_REPO_ROOT/profiles/scripts/stream_sweep.py" --steps 25 --warm 1.0 --packed 1 --grid 125x4a1
run queue_125x2 "$GRAFT_REPO_ROOT/demos/cpp/demo_batch_queue" --frames 126 --chains 2 --steps 160 --warmup 5 --warm-seconds 1.0 --repeats 3
run queue_125x4 "$GRAFT_REPO_ROOT/demos/cpp/demo_batch_queue" --frames 126 --chains 4 --steps 160 --warmup 5 --warm-seconds 1.0 --repeats 3
