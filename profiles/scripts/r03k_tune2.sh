#!/bin/bash
# stage 0 of the staged scoring: prefix size and match-range split
out=${1:-gpurun_out/r03k_tune2}
mkdir -p $out
run() { name=$1; shift
  for ev in 1 0; do
    env "$@" python3 bench.py --streams 1 --steps 8 --warmup 8 --repeats 2 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_$name.json 2>> $out/err.txt
  done
}
run base X=1
for pf in 64 128 256; do for ms in 1 2 4; do run pf${pf}_ms$ms PUTSLAM_HIP_PREFIX=$pf PUTSLAM_HIP_MSPLIT=$ms; done; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), "ms/step=%.3f"%d["ms_per_step"], "evals=%.3f"%(d.get("score_evals_frac") or -1), "k3=%.4f"%d["kernel_ms"]["ps_ransac_score"])
    except Exception as e:
        print(f, "ERR", e)
PY
