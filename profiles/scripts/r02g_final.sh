# final evidence of round 2: whole GPU suite, the three rocprofv3 passes + SQ counters of the single-chain command,
# the default bench line, a fuzz soak
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -5
bash profiles/run_profiles.sh r02g > gpurun_out/prof_r02g.log 2>&1; tail -3 gpurun_out/prof_r02g.log
timeout 1500 python tests/fuzz_gpu.py --iters ${FUZZ_ITERS:-60000} --seed ${FUZZ_SEED:-7777} > gpurun_out/fuzz_seed${FUZZ_SEED:-7777}.txt 2>&1; tail -2 gpurun_out/fuzz_seed${FUZZ_SEED:-7777}.txt
