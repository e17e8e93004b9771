#!/bin/bash
# SQ issue/stall counters of the four kernels (one pass, 8 SQ slots + GRBM), single-stream bench so that
# every kernel has the chip to itself.  Output: gpurun_out/prof_<tag>/sq/
set -u
TAG=${1:-r01e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC \
  --output-format csv -d $OUT/sq2 -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/bench_sq2.log 2>&1
python3 - <<PY
import csv, collections
for d in ("sq", "sq2"):
    try:
        rows = list(csv.DictReader(open("$OUT/%s/bench_counter_collection.csv" % d)))
    except Exception as e:
        print(d, "missing", e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        if "psdev::" not in k: continue
        k = k.split("psdev::")[1].split("(")[0][:28]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
