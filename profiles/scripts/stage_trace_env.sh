#!/bin/bash
# Per-launch durations of one E1 scoring step under knob settings given as NAME:ENV=V,ENV=V ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
out=$ROOT/gpurun_out/stage_trace_env
mkdir -p $out
export GPU_MAX_HW_QUEUES=8
ev=${EV:-1}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( cd /tmp; export TMPDIR=/tmp; IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    rocprofv3 --kernel-trace --output-format csv -d $out/$name -o t -- python3 $ROOT/bench.py --streams 1 --steps 12 --warmup 12 --repeats 1 --error-version $ev --no-cpu-baseline --no-other-modes > $out/$name.log 2>&1 )
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/$name/**/t_kernel_trace.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if 'psdev::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].split('psdev::')[1].split('(')[0][:40] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('ps_hamming')]
per=collections.OrderedDict(); cnt=0
for s,e in zip(idx[-9:-1], idx[-8:]):
    cnt+=1
    for j in range(s,e):
        key=(j-s,names[j]); per[key]=per.get(key,0)+(int(rows[j]['End_Timestamp'])-int(rows[j]['Start_Timestamp']))/1e3
print("$name".ljust(22), " ".join("%6.1f"%(v/cnt) for v in per.values()), " k3=%.1f"%(sum(v for (j,n),v in per.items() if 'score' in n or 'reorder' in n)/cnt))
PY
done
