"""Harness semantics of the reference's demos + the pose accuracy claim."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_demo_kabsch(ctx):
    from demos import demo_kabsch
    T, err, sigma = demo_kabsch.main(["--points", "500"])
    assert np.all(err < 3 * sigma) and abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12


def test_demo_matching_trajectory(ctx, tmp_path):
    from demos import demo_matching
    out = tmp_path / "VO_trajectory.res"
    traj, gt, ate = demo_matching.main(["--frames", "40", "--kpts", "1000", "--out", str(out)])
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 40 and all(len(l.split()) == 8 for l in lines)
    assert lines[0].split()[1:] == ["0", "0", "0", "0", "0", "0", "1"]
    assert ate < 0.02                                       # 39 increments of <= 5 cm each, 4 mm point noise


def test_demo_matching_scored_with_the_tum_protocol(ctx, tmp_path):
    """The trajectory file the path writes, scored as the reference scores its runs (scripts/evaluate_ate.py / evaluate_rpe.py;
    putslam_amd/tum_eval.py restates them, tests/test_tum_eval.py pins it to the reference's scripts)."""
    from demos import demo_matching
    out, gtf = tmp_path / "VO_trajectory.res", tmp_path / "groundtruth.txt"
    traj, gt, ate, ev = demo_matching.main(["--frames", "60", "--kpts", "1000", "--out", str(out), "--groundtruth", str(gtf)])
    assert ev["ate"]["pairs"] == 60
    assert ev["ate"]["rmse"] <= ate + 1e-4 and ev["ate"]["rmse"] < 0.02      # aligned error never exceeds the raw one
    assert ev["rpe_per_frame"]["pairs"] == 58 and ev["rpe_per_frame"]["translation"]["rmse"] < 0.005
    assert ev["rpe_per_second"]["pairs"] >= 28 and ev["rpe_per_second"]["translation"]["rmse"] < 0.02
    assert ev["rpe_per_second"]["rotation"]["rmse"] < np.radians(0.5)


def test_cpp_demo_latency_runs_through_the_c_abi(ctx):
    """demos/cpp/demo_latency: single-pair call + synchronize, back-to-back calls, ps_vo_stream_push and the pipelined form at one
    frame per chunk timed from C++ (the
    figures bench.py's `other_modes["latency"]` reports); here: it runs, every increment is accepted, the figures parse."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demos", "cpp", "demo_latency")
    if not os.path.exists(exe):
        pytest.fail("demos/cpp/demo_latency is not built (__graft_entry__.build())")
    p = subprocess.run([exe, "1000", "0", "200"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    us = [float(re.search(tag + r"[^\n]*?(?:median|300:) ([0-9.]+) us", p.stdout).group(1)) for tag in (r"\(a\)", r"\(b\)", r"\(c\)")]
    assert all(5.0 < v < 2000.0 for v in us), p.stdout
    assert us[1] <= us[0]                                    # the chain alone is never slower than call + synchronize
    # (d): the pipelined form at one frame per chunk -- every pair came back (exit code 0 checks that) and it beats the synchronous push
    d = [float(x) for x in re.findall(r"\(d[01]\)[^\n]*: ([0-9.]+) frames/s", p.stdout)]
    assert len(d) == 2 and all(v > 1.2e6 / us[2] for v in d), p.stdout


def test_demo_usac(ctx):
    from demos import demo_usac
    out, gt = demo_usac.main([])
    for name in ("RANSAC", "USAC"):
        assert out[name]["stats"]["accepted"] == 1
        assert np.abs(out[name]["pose"] - gt).max() < (5e-3 if name == "RANSAC" else 3e-2)


def test_noise_free_pose_within_1e5(ctx):
    """north_star: pose within 1e-5.  Exact correspondences under a known rigid motion: the refit pose must
    agree with the ground truth to 1e-5 (float32 Umeyama over ~1000 points), for every error mode."""
    from putslam_amd import synth
    from putslam_amd._abi import DMATCH_DTYPE, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config
    rng = np.random.default_rng(7)
    n = 1200
    cur = (rng.uniform(-1.0, 1.0, (n, 3)) * [1.2, 0.9, 1.0] + [0, 0, 2.5]).astype(np.float32)
    R, t = synth.random_motion(rng, 2.0, 0.05)
    prev = (cur.astype(np.float64) @ R.T + t).astype(np.float32)
    m = np.zeros(n, DMATCH_DTYPE)
    m["queryIdx"] = np.arange(n)
    m["trainIdx"] = np.arange(n)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    for mode in (0, 1, 2, 4):
        cfg, _ = make_config(EST_RANSAC, 487, seed=mode)
        r = ctx.ransac_rigid3d(default_ransac_params(mode), cfg, TUM_FR1_K, prev, cur, m)
        assert r["stats"]["numInliers"] == n
        assert np.abs(r["pose"] - T).max() < 1e-5, (mode, np.abs(r["pose"] - T).max())


def test_file_grabber_frames_backproject_bit_exact(ctx, oracle, tmp_path):
    """N4: frames staged in the FileGrabber directory format (fileGrabber.cpp:25-160) -> ps_keypoints2Dto3D and
    ps_remove_image_distortion on the GPU == oracle, bit for bit."""
    from putslam_amd import tum_io
    from putslam_amd._abi import TUM_FR1_K
    r = np.random.default_rng(7)
    frames = []
    for i in range(3):
        yy, xx = np.mgrid[0:480, 0:640]
        depth = (4000 + 11 * xx + 7 * yy + r.integers(0, 50, (480, 640))).astype(np.uint16)
        depth[r.random((480, 640)) < 0.03] = 0
        rgb = r.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        frames.append((100.0 + i / 30, 100.01 + i / 30, rgb, depth))
    d = str(tmp_path / "seq")
    tum_io.write_sequence(d, frames)
    dist5 = np.array([0.2624, -0.9531, -0.0054, 0.0026, 1.1633])       # TUM fr1 rgb distortion
    n = 0
    for fr in tum_io.FileGrabber(d):
        xy = np.stack([r.uniform(0, 638.4, 2000), r.uniform(0, 478.4, 2000)], axis=1).astype(np.float32)
        und_g = ctx.remove_image_distortion(xy, TUM_FR1_K, dist5)
        und_o = oracle.remove_image_distortion(xy, TUM_FR1_K, dist5)
        assert np.array_equal(und_g.view(np.uint32), und_o.view(np.uint32))
        got = ctx.keypoints2Dto3D(xy, fr.depthImage, TUM_FR1_K, fr.depthImageScale)
        exp = oracle.keypoints2Dto3D(xy, fr.depthImage, TUM_FR1_K, fr.depthImageScale)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        assert np.array_equal(fr.depthImage, frames[n][3])
        n += 1
    assert n == 3
