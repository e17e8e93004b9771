# A/B of kernel 1: FP4 MFMA sweep vs integer VALU sweep (single chain and the default three chains)
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
mkdir -p gpurun_out/r02a
for m in mfma valu; do
  PUTSLAM_HIP_MATCHER=$m python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r02a/bench_s1_$m.json 2> gpurun_out/r02a/err_s1_$m.txt
  PUTSLAM_HIP_MATCHER=$m python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_s3_$m.json 2> gpurun_out/r02a/err_s3_$m.txt
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), {k:round(v,4) for k,v in d['kernel_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
P
