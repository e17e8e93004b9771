#!/bin/bash
# rocprofv3 timeline (kernels + memory copies) of the pipelined stream at chunks of 125, packed frames (one upload per chunk) and
# two arrays (two uploads), with what it says about the three resources: how busy the host link is with uploads, how busy the chip
# is with kernels, how many launch chains run side by side.  usage (GPU box): bash profiles/scripts/r06_stream_timeline.sh r06g
tag=${1:-r06g}
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
for packed in 1 0; do
  out=gpurun_out/$tag/timeline_packed$packed
  mkdir -p "$out"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- python3 "$GRAFT_REPO_ROOT/profiles/scripts/stream_sweep.py" --steps 25 --warm 1.0 --packed $packed --grid 125x0a-1 > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1)
  python3 - "$out" $packed <<'P'
import csv, glob, sys
out, packed = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-50:], r.get("Queue_Id", "")))
for f in glob.glob(out + "/raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r.get("Direction", "?"), r.get("Stream_Id", "")))
ev.sort()
end = ev[-1][1]
w0, w1 = end - 22_000_000, end - 2_000_000            # a 20-ms window inside the timed steps (the last steps but the drain)
win = [e for e in ev if e[0] >= w0 and e[1] <= w1]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)
W = w1 - w0
h2d = [(s, e) for s, e, k, n, q in win if k == "M" and "HOST_TO_DEVICE" in n]
d2h = [(s, e) for s, e, k, n, q in win if k == "M" and "DEVICE_TO_HOST" in n]
ker = [(s, e) for s, e, k, n, q in win if k == "K" and "ps_" in n]
blit = [(s, e) for s, e, k, n, q in win if k == "K" and "copyBuffer" in n]
gaps = sorted((h2d[i + 1][0] - h2d[i][1]) / 1e3 for i in range(len(h2d) - 1))
with open(out + "/summary.txt", "w") as f:
    f.write(open(out + "/run.txt").read().strip().splitlines()[-1] + "   (under rocprofv3)\n")
    f.write("window %.1f ms: %d uploads, %d downloads as copies, %d download blit kernels, %d path kernels\n" % (W / 1e6, len(h2d), len(d2h), len(blit), len(ker)))
    f.write("host link busy with uploads      %5.1f %% of the window (median upload %.1f us; gap between consecutive uploads: median %.1f us, p90 %.1f us)\n"
            % (100.0 * union(h2d) / W, sorted(e - s for s, e in h2d)[len(h2d) // 2] / 1e3, gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * 0.9)] if gaps else 0))
    f.write("chip busy with the path's kernels %5.1f %% of the window (some kernel of kernels 1 - 4 running)\n" % (100.0 * union(ker) / W))
    f.write("kernels running side by side      %5.2f on average while any runs (sum of durations / union)\n" % (sum(e - s for s, e in ker) / max(union(ker), 1)))
    f.write("download blit kernels             %5.1f %% of the window, %.1f us each at the median\n"
            % (100.0 * union(blit) / W, (sorted(e - s for s, e in blit)[len(blit) // 2] / 1e3) if blit else 0))
print(open(out + "/summary.txt").read())
P
  rm -rf "$out/raw"
done
