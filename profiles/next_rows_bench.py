#!/usr/bin/env python3
"""Measurement of the SURVEY section 8(f) rows beside the main path (same bar: GPU time, CPU-oracle time on the box's
host cores, achieved bytes/s against the 8 TB/s HBM peak):
  N2  ps_match_xyz                 guided map matching (matcher.cpp:694-746), nmap map features x ncur keypoints
  N4  ps_remove_image_distortion   cv::undistortPoints, 5 Brown-model iterations per point (RGBD.cpp:254-314)
  A3  ps_keypoints2Dto3D           back-projection from a 640x480 depth image (RGBD.cpp:30-65)
  A10 ps_kabsch_f64                double N-point Kabsch (kabschEst.cpp:24-68): one wavefront up to 16384 points,
                                   G wavefronts in two passes above
The entry points take host pointers, so the call time includes both PCIe directions and one synchronisation; run the
script under `rocprofv3 --kernel-trace --stats` for the kernel-only durations (profiles/run_next_rows.sh).
Prints one JSON line per row."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc  # noqa: E402  (CPU leg only)
from putslam_amd import api  # noqa: E402
from putslam_amd._abi import TUM_FR1_K  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def main():
    ctx = api.Context(0)
    rng = np.random.default_rng(2026)
    rows = []
    # ---- N2: map of 5000 features against a 2000-keypoint frame (O(N_map x N)) ----
    nmap, ncur = 5000, 2000
    cur_pos = (rng.uniform(-1.5, 1.5, (ncur, 3)) + [0, 0, 2.5]).astype(np.float32)
    cur_desc = rng.integers(0, 256, (ncur, 32), dtype=np.uint8)
    src = rng.integers(0, ncur, nmap)
    map_pos = (cur_pos[src] + rng.normal(0, 0.05, (nmap, 3))).astype(np.float32)
    map_desc = cur_desc[src] ^ np.packbits(rng.random((nmap, 256)) < 0.05, axis=1)
    cur_level = rng.integers(0, 8, ncur).astype(np.int32)
    map_level = np.clip(cur_level[src] + rng.integers(-1, 2, nmap), 0, 7).astype(np.int32)
    tg, g = timed(lambda: ctx.match_xyz(map_pos, map_desc, map_level, cur_pos, cur_desc, cur_level, 0.12, 0.55), 20)
    tc, c = timed(lambda: orc.match_xyz(map_pos, map_desc, map_level, cur_pos, cur_desc, cur_level, 0.12, 0.55), 3)
    assert g.tobytes() == c.tobytes()
    alg = nmap * (12 + 32 + 4) + ncur * (12 + 32 + 4) + len(g) * 16
    rows.append(dict(row="N2 ps_match_xyz", nmap=nmap, ncur=ncur, matches=int(len(g)), gpu_call_ms=tg * 1e3,
                     cpu_oracle_ms=tc * 1e3, cpu_threads=1, algorithmic_bytes=alg,
                     pairs_tested_per_s_gpu_call=nmap * ncur / tg))
    # ---- N4: undistortion of 2000 / 200000 keypoints ----
    dist5 = np.array([0.2624, -0.9531, -0.0054, 0.0026, 1.1633])
    for n in (2000, 200000):
        xy = np.stack([rng.uniform(0, 639, n), rng.uniform(0, 479, n)], axis=1).astype(np.float32)
        tg, g = timed(lambda: ctx.remove_image_distortion(xy, TUM_FR1_K, dist5), 20)
        tc, c = timed(lambda: orc.remove_image_distortion(xy, TUM_FR1_K, dist5), 3)
        assert g.tobytes() == c.tobytes()
        rows.append(dict(row="N4 ps_remove_image_distortion", n=n, gpu_call_ms=tg * 1e3, cpu_oracle_ms=tc * 1e3,
                         cpu_threads=1, algorithmic_bytes=n * 16))
    # ---- A3: back-projection of 2000 / 200000 keypoints from a 640x480 depth image ----
    depth = rng.integers(500, 30000, (480, 640)).astype(np.uint16)
    for n in (2000, 200000):
        xy = np.stack([rng.uniform(0, 638.4, n), rng.uniform(0, 478.4, n)], axis=1).astype(np.float32)
        tg, g = timed(lambda: ctx.keypoints2Dto3D(xy, depth, TUM_FR1_K, 5000.0), 20)
        tc, c = timed(lambda: orc.keypoints2Dto3D(xy, depth, TUM_FR1_K, 5000.0), 3)
        assert g.tobytes() == c.tobytes()
        rows.append(dict(row="A3 ps_keypoints2Dto3D", n=n, gpu_call_ms=tg * 1e3, cpu_oracle_ms=tc * 1e3, cpu_threads=1,
                         algorithmic_bytes=n * (8 + 2 + 12), depth_image_bytes=int(depth.nbytes)))
    # ---- A10: double Kabsch on n correspondences (config 1: n = 500) ----
    for n in (500, 100000):
        A = rng.uniform(-1.5, 1.5, (n, 3))
        B = A + [0.1, 0.2, -0.3] + rng.normal(0, 1, (n, 3)) * [0.01, 0.02, 0.03]
        tg, g = timed(lambda: ctx.kabsch_f64(A, B), 20)
        tc, c = timed(lambda: orc.kabsch_f64(A, B), 5)
        assert np.abs(g - np.asarray(c, np.float64)).max() < 1e-12      # summation tree differs (tests: same bound)
        rows.append(dict(row="A10 ps_kabsch_f64", n=n, gpu_call_ms=tg * 1e3, cpu_oracle_ms=tc * 1e3, cpu_threads=1,
                         algorithmic_bytes=n * 48 + 128))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
