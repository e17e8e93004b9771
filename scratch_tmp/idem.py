import numpy as np, sys
sys.path.insert(0, '.')
from putslam_amd import api, synth
from putslam_amd._abi import *
from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
ctx = api.Context(0)
print("score option", ctx.get_option("score"))
seq = synth.make_sequence(33, 2000, config=3, index=1)
prm = default_ransac_params(REPROJECTION_ERROR)
cfg, _ = make_config(EST_FIXED, 4096, seed=42)
fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
outs = []
for rep in range(4):
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    outs.append(pb.download())
ctx.set_option("score", 1)
pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
ref = pb.download()
for rep, g in enumerate(outs):
    bad = [p for p in range(len(seq["pairs"])) if g["pose"][p].tobytes() != ref["pose"][p].tobytes() or g["stats"][p]["bestInlierCount"] != ref["stats"][p]["bestInlierCount"] or g["stats"][p]["bestHypothesis"] != ref["stats"][p]["bestHypothesis"]]
    print("rep", rep, "pairs differing from the fast kernel:", bad, [(int(g["stats"][p]["bestHypothesis"]), int(ref["stats"][p]["bestHypothesis"]), int(g["stats"][p]["bestInlierCount"]), int(ref["stats"][p]["bestInlierCount"])) for p in bad[:6]])
