# A/B of kernel 3 (errorVersion 1): split-f16 matrix-core kernel vs VALU decision-exact kernel (default) vs value-exact kernel,
# fuzz soak of the matrix-core kernel, kernel trace of the matrix-core run
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
mkdir -p $O
export TMPDIR=/tmp
PUTSLAM_HIP_SCORE=mfma timeout 1500 python tests/fuzz_gpu.py --iters ${FUZZ_ITERS:-6000} --seed 424242 > $O/fuzz_mfma_seed424242.txt 2>&1; tail -2 $O/fuzz_mfma_seed424242.txt
for s in mfma fast exact; do
  PUTSLAM_HIP_SCORE=$s python bench.py --streams 1 --no-cpu-baseline > $O/bench_s1_$s.json 2> $O/err_s1_$s.txt
  PUTSLAM_HIP_SCORE=$s python bench.py --no-cpu-baseline > $O/bench_s3_$s.json 2> $O/err_s3_$s.txt
done
export PUTSLAM_HIP_SCORE=mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o mfma -- python3 bench.py --streams 1 --steps 20 --warmup 20 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace_err.txt
unset PUTSLAM_HIP_SCORE
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_mfma.csv
rm -rf $O/trace
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02i/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), {k:round(v,4) for k,v in d['kernel_ms'].items()}, d.get('score_parked_frac'))
    except Exception as e: print(f, 'ERR', e)
P
head -8 $O/kernel_stats_mfma.csv
