#!/bin/bash
# rocprofv3 kernel timeline of the pipelined stream at ONE frame per chunk (demos/cpp/demo_latency, leg (d)): ordinary launches
# (the default) against each place's chunk replayed from a captured hipGraph (PUTSLAM_HIP_STREAM_GRAPH=1).  Per chunk: the span
# from its copy-in kernel's start to its last kernel's end, the gaps between its consecutive kernels, and how many chunks'
# kernels run side by side.  usage (GPU box): bash profiles/scripts/r06_mini_timeline.sh r06n
tag=${1:-r06n}
cd "$GRAFT_REPO_ROOT" || exit 1
for graph in 0 1; do
  out=gpurun_out/$tag/mini_graph$graph
  mkdir -p "$out"
  export PUTSLAM_HIP_STREAM_GRAPH=$graph
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -- "$GRAFT_REPO_ROOT/demos/cpp/demo_latency" 2000 0 400 > "$GRAFT_REPO_ROOT/$out/run.txt" 2>&1)
  python3 - "$out" $graph <<'P'
import csv, glob, sys, collections
out, graph = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40], r.get("Queue_Id", r.get("Stream_Id", ""))))
ev.sort()
# the (d0) leg: chunks start with ps_mini_copy_in; take the chunks of the last 40 % of the copy-in launches of the FIRST mode (d0)
starts = [i for i, e in enumerate(ev) if e[2].startswith("ps_mini_copy_in")]
half = len(starts) // 2                       # (d0) then (d1): two equal runs
sel = starts[int(half * 0.5):half]
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
spans, gaps, kcount = [], [], []
per = collections.OrderedDict()               # kernel name -> [durations], [gap in front of it]
pos = {q: {id(e): n for n, e in enumerate(l)} for q, l in byq.items()}
for i in sel:
    s0, e0, n0, q = ev[i]
    lst = byq[q]
    j = pos[q][id(ev[i])]
    per.setdefault(n0, ([], []))[0].append((e0 - s0) / 1e3)
    k = j + 1
    last_end, prev_end, n = e0, e0, 1
    while k < len(lst) and not lst[k][2].startswith("ps_mini_copy_in"):
        g = (lst[k][0] - prev_end) / 1e3
        gaps.append(g)
        d = per.setdefault(lst[k][2], ([], []))
        d[0].append((lst[k][1] - lst[k][0]) / 1e3)
        d[1].append(g)
        prev_end = lst[k][1]
        last_end = lst[k][1]
        n += 1
        k += 1
    spans.append((last_end - s0) / 1e3)
    kcount.append(n)
w0, w1 = ev[sel[0]][0], ev[sel[-1]][1]
win = [(s, e) for s, e, n, q in ev if s >= w0 and e <= w1]
def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)
spans.sort(); gaps.sort()
with open(out + "/summary.txt", "w") as f:
    f.write([l for l in open(out + "/run.txt") if l.startswith("(d0)")][0].strip() + "   (under rocprofv3)\n")
    f.write("graph replay %s: %d chunks analysed, %.1f kernels per chunk\n" % ("ON" if graph == "1" else "off", len(sel), sum(kcount) / len(kcount)))
    f.write("chunk span (copy-in start -> last kernel end): median %.1f us, p90 %.1f us\n" % (spans[len(spans) // 2], spans[int(len(spans) * 0.9)]))
    f.write("gap between consecutive kernels of a chunk: median %.1f us, p90 %.1f us\n" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)]))
    for name, (du, ga) in per.items():
        f.write("   %-42s runs %6.1f us (mean), starts %5.1f us after the kernel in front of it ends\n" % (name, sum(du) / len(du), (sum(ga) / len(ga)) if ga else 0.0))
    f.write("chip busy with some kernel %.1f %% of the window; kernels side by side %.2f on average; chunk rate %.0f /s\n"
            % (100.0 * union(win) / (w1 - w0), sum(e - s for s, e in win) / max(union(win), 1), len(sel) / ((w1 - w0) / 1e9)))
print(open(out + "/summary.txt").read())
P
  rm -rf "$out/raw"
done
