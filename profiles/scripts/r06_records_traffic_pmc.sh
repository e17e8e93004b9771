#!/bin/bash
# Counters beside profiles/r06d/records_traffic_ab.txt: HBM write traffic (WRITE_SIZE, its own rocprofv3 pass) of kernel 2 with and
# without the diagnostic build's shadow copy of every record.  usage (GPU box): bash profiles/scripts/r06_records_traffic_pmc.sh r06d
tag=${1:-r06d}
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
export PUTSLAM_HIP_LIB=$GRAFT_REPO_ROOT/putslam_amd/libputslam_hip_diag.so
ARGS="--streams 1 --submit python --steps 10 --warmup 10 --warm-seconds 0 --repeats 1 --no-other-modes --no-cpu-baseline"
mkdir -p gpurun_out/$tag
for shadow in 0 1; do
  export PUTSLAM_HIP_DIAG_SHADOW_RECORDS=$shadow
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/pmc_shadow$shadow
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o bench -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $out.log 2>&1)
  python3 - $out $shadow <<'P'
import csv, glob, sys, collections
out, shadow = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE" and "psdev::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("psdev::")[1].split("<")[0].split("(")[0]].append(float(r["Counter_Value"]))
steps = max(len(agg.get("ps_crosscheck_prep", [])), 1)
print("shadow records %s: WRITE_SIZE per step (KB -> MB): " % shadow + ", ".join("%s %.1f MB" % (k, sum(v) / steps / 1024.0) for k, v in sorted(agg.items())))
P
done
