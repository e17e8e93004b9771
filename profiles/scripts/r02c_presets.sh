# bench lines of the other BASELINE configs: configs[1] demoMatching (one pair per step, single chain) in both error
# modes and with the reference's own schedule, configs[4] stress (5000 kpts, H = 100000) in both error modes
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline"
$B --preset demoMatching --streams 1 --steps 200 --warmup 20 > gpurun_out/r02c/demoMatching_e1_H4096.json 2> gpurun_out/r02c/err1.txt
$B --preset demoMatching --streams 1 --steps 200 --warmup 20 --error-version 0 > gpurun_out/r02c/demoMatching_e0_H4096.json 2>> gpurun_out/r02c/err1.txt
$B --preset demoMatching --streams 1 --steps 200 --warmup 20 --estimator ransac --hyp 487 > gpurun_out/r02c/demoMatching_e1_ransac487.json 2>> gpurun_out/r02c/err1.txt
$B --preset demoMatching --streams 1 --steps 200 --warmup 20 --estimator ransac --hyp 487 --error-version 0 > gpurun_out/r02c/demoMatching_e0_ransac487.json 2>> gpurun_out/r02c/err1.txt
$B --preset stress --streams 1 --steps 10 --warmup 2 > gpurun_out/r02c/stress_e1.json 2>> gpurun_out/r02c/err1.txt
$B --preset stress --streams 1 --steps 10 --warmup 2 --error-version 0 > gpurun_out/r02c/stress_e0.json 2>> gpurun_out/r02c/err1.txt
python profiles/stream_latency.py > gpurun_out/r02c/stream_latency.txt 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, d.get('score_parked_frac'))
    except Exception as e: print(f, 'ERR', e)
P
tail -5 gpurun_out/r02c/stream_latency.txt; tail -3 gpurun_out/r02c/err1.txt
