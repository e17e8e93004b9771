# kernel 1: query tiles straight into registers (default) vs staged through LDS with a barrier per tile (-DPS_MFMA_DIRECT=0,
# built as putslam_amd/libputslam_hip_lds.so)
# (build the LDS variant first, here or on the box: make putslam_amd/libputslam_hip_lds.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02h
[ -f putslam_amd/libputslam_hip_lds.so ] || make putslam_amd/libputslam_hip_lds.so || { echo "cannot build the LDS variant" >&2; exit 1; }
python -m pytest tests/test_gpu_matcher_variants.py tests/test_gpu_batch.py -x -q 2>&1 | tail -2
for v in direct lds; do
  L=$PWD/putslam_amd/libputslam_hip.so; [ $v = lds ] && L=$PWD/putslam_amd/libputslam_hip_lds.so
  for rep in 1 2; do PUTSLAM_HIP_LIB=$L python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r02h/bench_s1_${v}_$rep.json 2>> gpurun_out/r02h/stderr.txt; done
  PUTSLAM_HIP_LIB=$L python bench.py --no-cpu-baseline > gpurun_out/r02h/bench_s3_$v.json 2>> gpurun_out/r02h/stderr.txt
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02h/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['kernel_ms']['ps_hamming_mfma'],4), round(d['kernel_bounds']['ps_hamming_mfma']['frac'],3))
P
