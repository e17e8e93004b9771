import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from putslam_amd import api, synth
from putslam_amd._abi import *
ctx = api.Context(0)
seq = synth.make_sequence(41, 2000, config=3, index=0)
prm = default_ransac_params(1)
for est, H, name in ((EST_FIXED, 4096, "fixed4096"), (EST_RANSAC, 487, "ransac487")):
    cfg, _ = make_config(est, H, seed=1)
    st = api.VoStream(ctx, 2000)
    st.push(prm, cfg, TUM_FR1_K, seq["desc"][0], seq["pts"][0])
    for f in range(1, 6):
        st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f])
    t0 = time.perf_counter()
    for f in range(6, 41):
        st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f])
    t1 = time.perf_counter()
    print(name, "streaming host-pointer path (PCIe + sync inclusive):", 35 / (t1 - t0), "pairs/s", (t1 - t0) / 35 * 1e3, "ms/pair")
    t0 = time.perf_counter()
    for f in range(6, 41):
        m = ctx.match_hamming256(seq["desc"][f - 1], seq["desc"][f])
        ctx.ransac_rigid3d(prm, cfg, TUM_FR1_K, seq["pts"][f - 1], seq["pts"][f], m)
    t1 = time.perf_counter()
    print(name, "two-call host-pointer path:", 35 / (t1 - t0), "pairs/s")
