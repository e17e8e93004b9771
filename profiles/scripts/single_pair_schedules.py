"""Single-pair latency under other schedules (long RANSAC caps, USAC up to the reference's 850 000): profiles/r04g/single_pair_schedules.txt.
PSLIB=<path> selects another build of the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
seq = synth.make_sequence(2, 2000, config=3, index=0)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"]); pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
ctx = api.Context(0, lib=os.environ.get("PSLIB"))
for name, ev, est, H in (("E0/ransac/487", 0, EST_RANSAC, 487), ("E0/ransac/1157", 0, EST_RANSAC, 1157), ("E0/usac/3000", 0, EST_USAC, 3000), ("E0/usac/100000", 0, EST_USAC, 100000), ("E0/usac/850000", 0, EST_USAC, 850000), ("E1/usac/850000", 1, EST_USAC, 850000)):
    prm = default_ransac_params(ev); cfg, _ = make_config(est, H, seed=3)
    for _ in range(50): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    w = (time.perf_counter() - t0) / 300 * 1e6
    enq = (t1 - t0) / 300 * 1e6
    ctx.enable_timing(True)
    for _ in range(50): run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    kern = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in ctx.kernel_time_totals().items()}
    ctx.enable_timing(False)
    st = pb.download()["stats"][0]
    print(name, f"{w:.1f} us (host enqueue {enq:.1f})", kern, {k: v[1] for k, v in ctx.kernel_time_totals().items()} if False else "", "iterations", int(st["iterationsRun"]), "staged", ctx.get_option("last_staged_pairs"), flush=True)
