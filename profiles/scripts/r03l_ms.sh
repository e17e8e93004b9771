#!/bin/bash
out=${1:-gpurun_out/r03l_ms}
mkdir -p $out
run() { name=$1; shift
  for ev in 1 0; do
    env "$@" python3 bench.py --streams 1 --steps 8 --warmup 8 --repeats 2 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_$name.json 2>> $out/err.txt
  done
}
for ms in 2 3 4 5 6 7 8 10 14; do run ms$ms PUTSLAM_HIP_MSPLIT=$ms PUTSLAM_HIP_LISTG2=15 PUTSLAM_HIP_REORDER_C2DIV=16; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), "ms/step=%.3f"%d["ms_per_step"], "evals=%.3f"%(d.get("score_evals_frac") or -1), "k3=%.4f"%d["kernel_ms"]["ps_ransac_score"])
    except Exception as e:
        print(f, "ERR", e)
PY
