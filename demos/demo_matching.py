#!/usr/bin/env python3
"""Counterpart of demos/demoMatching.cpp (= PUTSLAM::startProcessing in matching-VO mode, PUTSLAM.cpp:677-930)
for the hot path only (BASELINE configs[2]): stream a synthetic TUM-style sequence, per frame pair
performMatching -> RANSAC -> refit on the GPU, compose VO_k = VO_{k-1} * inc_k with the 0.1 m gate
(PUTSLAM.cpp:735-740) and write the trajectory in Freiburg format `timestamp tx ty tz qx qy qz qw`
(PUTSLAM.cpp:1006-1016).  Prints the raw position error against the generator's ground truth and, with --groundtruth FILE,
writes the ground truth in the same format and scores the trajectory file the way the reference's scripts/evaluate_ate.py
and evaluate_rpe.py do (putslam_amd/tum_eval.py).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--error-version", type=int, default=0)
    ap.add_argument("--out", default="VO_trajectory.res")
    ap.add_argument("--groundtruth", default=None, help="write the generator's ground truth here and print the TUM ATE / RPE")
    a = ap.parse_args(argv)
    from putslam_amd import api, sharding, synth
    from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs

    seq = synth.make_sequence(a.frames, a.kpts, config=3, index=0)
    ctx = api.Context(0)
    prm = default_ransac_params(a.error_version)
    cfg, _ = make_config(EST_RANSAC, 487, seed=1)         # the reference's own schedule
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, pb)
    res = pb.download()
    inc = res["pose"].reshape(-1, 4, 4).transpose(0, 2, 1)
    traj = sharding.compose_trajectory(inc)
    gt = sharding.compose_trajectory(seq["gt"].astype(np.float32))
    stamps = 1305031102.175304 + np.arange(a.frames) / 30.0
    sharding.write_tum_trajectory(a.out, stamps, traj)
    ate = np.sqrt(np.mean(np.sum((traj[:, :3, 3] - gt[:, :3, 3]) ** 2, axis=1)))
    ratio = res["stats"]["pointInlierRatio"]
    print(f"{a.frames} frames, {len(inc)} pairs: mean inlier ratio {np.nanmean(ratio):.3f}, "
          f"accepted {int(res['stats']['accepted'].sum())}, ATE rmse {ate * 100:.2f} cm -> {a.out}")
    if a.groundtruth:
        from putslam_amd import tum_eval
        sharding.write_tum_trajectory(a.groundtruth, stamps, gt)
        ev = tum_eval.evaluate_files(a.groundtruth, a.out)
        print(f"TUM protocol: ATE rmse {ev['ate']['rmse'] * 100:.2f} cm over {ev['ate']['pairs']} poses, "
              f"RPE {ev['rpe_per_second']['translation']['rmse'] * 100:.2f} cm/s "
              f"{np.degrees(ev['rpe_per_second']['rotation']['rmse']):.3f} deg/s, "
              f"{ev['rpe_per_frame']['translation']['rmse'] * 1000:.2f} mm/frame")
        return traj, gt, ate, ev
    return traj, gt, ate


if __name__ == "__main__":
    main()
