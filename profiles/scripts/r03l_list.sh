#!/bin/bash
out=${1:-gpurun_out/r03l}
mkdir -p $out
python3 -m pytest tests/test_gpu_prune.py tests/test_gpu_fuzz_slice.py tests/test_gpu_batch.py tests/test_gpu_score_euclid.py tests/test_gpu_score_variants.py tests/test_gpu_band_edges.py -x -q 2>&1 | tail -3
python3 tests/fuzz_gpu.py --batch --iters 200 --procs 4 --seed 2200 2>&1 | tail -2
run() { name=$1; shift
  for ev in 1 0; do
    env "$@" python3 bench.py --streams 1 --steps 8 --warmup 8 --repeats 2 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_$name.json 2>> $out/err.txt
  done
}
for r3 in 1 4 8 16; do for d in 4 16; do run r${r3}_d$d PUTSLAM_HIP_LISTR3=$r3 PUTSLAM_HIP_REORDER_C2DIV=$d PUTSLAM_HIP_LISTG2=15; done; done
run r8_d16_g8 PUTSLAM_HIP_LISTR3=8 PUTSLAM_HIP_REORDER_C2DIV=16 PUTSLAM_HIP_LISTG2=8
run r8_d32_g15 PUTSLAM_HIP_LISTR3=8 PUTSLAM_HIP_REORDER_C2DIV=32 PUTSLAM_HIP_LISTG2=15
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), "ms/step=%.3f"%d["ms_per_step"], "evals=%.3f"%(d.get("score_evals_frac") or -1), "k3=%.4f"%d["kernel_ms"]["ps_ransac_score"])
    except Exception as e:
        print(f, "ERR", e)
PY
bash profiles/scripts/stage_trace.sh stage_trace3 2>&1 | tail -22
