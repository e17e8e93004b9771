#!/usr/bin/env python3
"""Counterpart of the Kabsch part of demos/demoKabsch.cpp:968-1043 (BASELINE configs[0]).

N points uniform in (-1.5, 1.5)^3 (generateSetpoint, demoKabsch.cpp:118-126), B = R(0,0,0) A + (0.1, 0.2, -0.3) +
N(0, sigma = (0.01, 0.02, 0.03)) (:23,25,1003-1019), then TransformEst::computeTransformation (KabschEst) and the
translation / quaternion print-out of :1022-1029.  The fit runs on the GPU (ps_kabsch_f64).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args(argv)
    from putslam_amd import api, sharding
    rng = np.random.Generator(np.random.PCG64(a.seed))
    A = rng.uniform(-1.5, 1.5, (a.points, 3))
    t_true, sigma = np.array([0.1, 0.2, -0.3]), np.array([0.01, 0.02, 0.03])
    B = A + t_true + rng.normal(0, 1, A.shape) * sigma
    T = api.Context(0).kabsch_f64(A, B)
    q = sharding.rotation_to_quaternion_f32(T[:3, :3])
    err = np.abs(T[:3, 3] - t_true)
    print("Kabsch Estimator: translation", T[:3, 3], "quaternion (x y z w)", [float(v) for v in q])
    for i, ax in enumerate("xyz"):
        if err[i] > 3 * sigma[i]:
            print(f"alert! {ax} error {err[i]:.4f} > 3 sigma")   # demoKabsch.cpp:50-67
    return T, err, sigma


if __name__ == "__main__":
    main()
