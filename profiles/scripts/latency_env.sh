#!/bin/bash
# Single-pair and pushed-frame latency under runtime settings that change how launches reach the GPU (run through gpurun):
#   bash profiles/scripts/latency_env.sh > gpurun_out/latency_env.txt
# and a kernel trace of the single-pair loop (start / end stamps: the real gaps between the four dependent launches).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
one() {
  echo "== $*"
  env "$@" python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
seq = synth.make_sequence(40, 2000, config=3, index=0)
ctx = api.Context(0)
fs = FrameSetDevice(seq["desc"][:2], seq["pts"][:2], seq["nkpts"][:2]); pb = PairBatchDevice(seq["pairs"][:1], fs.max_kpts)
prm = default_ransac_params(0); cfg, _ = make_config(EST_RANSAC, 487, seed=3)
for _ in range(200): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
torch.cuda.synchronize()
med = []
for turn in range(7):
    t0 = time.perf_counter()
    for _ in range(300): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize(); med.append((time.perf_counter() - t0) / 300 * 1e6)
st = api.VoStream(ctx, 2000); ts = []
for rep in range(6):
    for f in range(40):
        c, _ = make_config(EST_RANSAC, 487, seed=f + 1)
        t0 = time.perf_counter(); st.push(prm, c, TUM_FR1_K, seq["desc"][f], seq["pts"][f]); ts.append(time.perf_counter() - t0)
ts = np.array(ts[40:]) * 1e6
print(f"single pair {np.median(med):.1f} us (min {min(med):.1f})   pushed frame median {np.median(ts):.1f} us p10 {np.percentile(ts,10):.1f} p90 {np.percentile(ts,90):.1f}")
PY
}
one A=0
one HIP_FORCE_DEV_KERNARG=1
one HIP_FORCE_DEV_KERNARG=0
one DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
one DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
one GPU_MAX_HW_QUEUES=1
one ROC_SIGNAL_POOL_SIZE=128
one HSA_ENABLE_INTERRUPT=0
one HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1
one A=0
