# A/B of kernel 3 (errorVersion 1): decision-exact fast kernel vs value-exact kernel; parity first, then a fuzz soak
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_score_variants.py -x -q 2>&1 | tail -15
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python tests/fuzz_gpu.py --iters ${FUZZ_ITERS:-3000} --seed 20261003 > gpurun_out/r02b/fuzz_seed20261003.txt 2>&1; tail -3 gpurun_out/r02b/fuzz_seed20261003.txt
for s in fast exact; do
  PUTSLAM_HIP_SCORE=$s python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r02b/bench_s1_$s.json 2> gpurun_out/r02b/err_s1_$s.txt
  PUTSLAM_HIP_SCORE=$s python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_s3_$s.json 2> gpurun_out/r02b/err_s3_$s.txt
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), {k:round(v,4) for k,v in d['kernel_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
P
