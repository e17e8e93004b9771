#!/bin/bash
# Per-launch durations of one scoring step (kernel trace), E1 and E0, single chain
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-stage_trace}
mkdir -p $out
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
for ev in 1 0; do
  rocprofv3 --kernel-trace --output-format csv -d $out/E$ev -o t -- python3 $ROOT/bench.py --streams 1 --steps 20 --warmup 20 --repeats 1 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E$ev.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv,glob,collections
for ev in (1,0):
    f=glob.glob("$out/E%d/**/t_kernel_trace.csv"%ev, recursive=True)
    rows=[r for r in csv.DictReader(open(f[0])) if 'psdev::' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # last 10 steps: group by step starting at hamming
    names=[r['Kernel_Name'].split('psdev::')[1].split('(')[0][:44] for r in rows]
    idx=[i for i,n in enumerate(names) if n.startswith('ps_hamming')]
    per=collections.OrderedDict()
    cnt=0
    for s,e in zip(idx[-11:-1], idx[-10:]):
        cnt+=1
        for j in range(s,e):
            key=(j-s,names[j])
            per[key]=per.get(key,0)+(int(rows[j]['End_Timestamp'])-int(rows[j]['Start_Timestamp']))/1e3
    print("E%d"%ev)
    tot=0
    for (j,n),v in per.items():
        print("  %d %-46s %8.1f us"%(j,n,v/cnt)); tot+=v/cnt
    # gaps
    s,e=idx[-11],idx[-1]
    wall=(int(rows[e]['Start_Timestamp'])-int(rows[s]['Start_Timestamp']))/1e3/10
    print("  sum %.1f us, wall per step %.1f us"%(tot,wall))
PY
