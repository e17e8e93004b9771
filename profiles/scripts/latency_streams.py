import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from putslam_amd import api, synth
from putslam_amd._abi import EST_RANSAC, EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
seq = synth.make_sequence(2, 2000, config=3, index=0)
for name, ev, est, H in (("E0/ransac/487", 0, EST_RANSAC, 487), ("E1/fixed/4096", 1, EST_FIXED, 4096)):
    ctx = api.Context(0)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"]); pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    prm = default_ransac_params(ev); cfg, _ = make_config(est, H, seed=3)
    def loop(fn, sync):
        for _ in range(200): fn()
        sync(); med = []
        for turn in range(7):
            t0 = time.perf_counter()
            for _ in range(300): fn()
            sync(); med.append((time.perf_counter() - t0) / 300 * 1e6)
        return np.median(med), min(med)
    a = loop(lambda: run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb), torch.cuda.synchronize)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = loop(lambda: run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb), torch.cuda.synchronize)
    ctx2 = api.Context(0)
    b = loop(lambda: run_pairs(ctx2, prm, cfg, TUM_FR1_K, fs, pb, use_torch_stream=False), torch.cuda.synchronize)
    print(name, "joined with the null stream %.1f (min %.1f)  on a torch side stream %.1f (%.1f)  on the context's own stream %.1f (%.1f)" % (a + c + b))
