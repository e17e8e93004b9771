# A/B of kernel 3 (errorVersion 1): matrix-core decision-exact kernel vs VALU decision-exact kernel vs value-exact kernel
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_score_variants.py -x -q 2>&1 | tail -15
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 900 python tests/fuzz_gpu.py --iters ${FUZZ_ITERS:-4000} --seed 31337 > gpurun_out/r02f/fuzz_seed31337.txt 2>&1; tail -2 gpurun_out/r02f/fuzz_seed31337.txt
for s in mfma fast exact; do
  PUTSLAM_HIP_SCORE=$s python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r02f/bench_s1_$s.json 2> gpurun_out/r02f/err_s1_$s.txt
  PUTSLAM_HIP_SCORE=$s python bench.py --no-cpu-baseline > gpurun_out/r02f/bench_s3_$s.json 2> gpurun_out/r02f/err_s3_$s.txt
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02f/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), {k:round(v,4) for k,v in d['kernel_ms'].items()}, d.get('score_parked_frac'))
    except Exception as e: print(f, 'ERR', e)
P
