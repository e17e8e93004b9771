"""C++ drop-in classes (reference Matcher / RANSAC / RANSAC_USAC / TransformEst surface over the C ABI):
the oracle writes a case file, tests/cpp/test_dropin replays it through the classes on the GPU."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from putslam_amd import synth
from putslam_amd._abi import EST_RANSAC, EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_dropin")


@pytest.mark.parametrize("mode", [EUCLIDEAN_ERROR, REPROJECTION_ERROR])
def test_cpp_dropin(oracle, tmp_path, mode):
    if not os.path.exists(EXE):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_dropin()
    n, seed = 600, 0x1234_5678_9ABC
    a, b = synth.make_pair(n, config=2, index=40 + mode)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(mode)
    cfg, _ = make_config(EST_RANSAC, 487, seed=seed)
    r = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
    rng = np.random.default_rng(1)
    A = rng.uniform(-1.5, 1.5, (500, 3))
    B = A + [0.1, 0.2, -0.3] + rng.normal(0, 1, (500, 3)) * [0.01, 0.02, 0.03]
    T = oracle.kabsch_f64(A, B)
    path = tmp_path / "case.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", n, len(m), 487, mode, seed & 0xFFFFFFFF, seed >> 32, int(r["stats"]["numInliers"]), 0))
        f.write(a["desc"].tobytes())
        f.write(b["desc"].tobytes())
        f.write(a["pts"].tobytes())
        f.write(b["pts"].tobytes())
        f.write(m.tobytes())
        f.write(np.ascontiguousarray(r["pose"].T).tobytes())   # column-major
        f.write(r["mask"].tobytes())
        f.write(struct.pack("<i", 500))
        f.write(np.asfortranarray(A).tobytes(order="F"))
        f.write(np.asfortranarray(B).tobytes(order="F"))
        f.write(np.ascontiguousarray(T.T).tobytes())
    p = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all checks passed" in p.stdout


def test_cpp_demo_matching_sequence(tmp_path):
    """demos/cpp/demo_matching: the reference's demoMatching loop over the drop-in classes (Matcher factory ->
    detectInitFeatures / runVO -> VOTrajectory), synthetic frames with known motion; exit code 0 = every increment
    within 5e-3 of the ground truth; the trajectory file has one TUM line per frame."""
    exe = os.path.join(ROOT, "demos", "cpp", "demo_matching")
    if not os.path.exists(exe):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_dropin()
    traj = tmp_path / "traj.txt"
    p = subprocess.run([exe, "40", "1200", str(traj)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = traj.read_text().strip().split("\n")
    assert len(lines) == 40 and all(len(l.split()) == 8 for l in lines)
    assert "39 increments accepted, 0 rejected" in p.stdout


@pytest.mark.parametrize("chunk", [1, 7, 32])
def test_cpp_demo_matching_pipelined_writes_the_same_trajectory(tmp_path, chunk):
    """FrameMatcher::enqueueFrame / dequeueResult (ps_vo_stream_push_async / ps_vo_stream_pop: results with a lag) against one
    synchronous runVO per frame: the TUM trajectory files are identical, whatever the chunk size."""
    exe = os.path.join(ROOT, "demos", "cpp", "demo_matching")
    if not os.path.exists(exe):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_dropin()
    a, b = tmp_path / "sync.txt", tmp_path / "pipe.txt"
    p = subprocess.run([exe, "45", "1200", str(a)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    q = subprocess.run([exe, "45", "1200", str(b), "--pipelined", str(chunk)], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout + q.stderr
    assert "44 increments accepted, 0 rejected" in q.stdout
    assert a.read_text() == b.read_text() and len(a.read_text().strip().split("\n")) == 45


@pytest.mark.parametrize("chunk", [1, 8])
def test_cpp_pipelined_matcher_grows_with_the_frames(tmp_path, chunk):
    """the pipelined drop-in fixed its frame capacity at the first frame and refused larger frames for ever.
    A sequence whose first third carries 2200 keypoints and the rest 5500 (capacity 2750 at first): enqueueFrame answers
    "busy" until the pending results are dequeued, rebuilds the pipeline with more room and continues -- same pairs, same
    hypothesis seeds: the trajectory file equals the synchronous runVO's, which regrows its stream in place."""
    exe = os.path.join(ROOT, "demos", "cpp", "demo_matching")
    a, b = tmp_path / "sync.txt", tmp_path / "pipe.txt"
    p = subprocess.run([exe, "24", "5500", str(a), "--ragged"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    q = subprocess.run([exe, "24", "5500", str(b), "--ragged", "--pipelined", str(chunk)], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout + q.stderr
    assert "23 increments accepted, 0 rejected" in q.stdout
    assert a.read_text() == b.read_text() and len(a.read_text().strip().split("\n")) == 24


def test_reference_shaped_plugin_links_and_matches_oracle(oracle, tmp_path):
    """tests/cpp/test_reference_shaped.cpp: a translation unit with classes named putslam::Matcher / ::MatcherOpenCV and
    the reference's factories (bodies as INTEGRATION.md section 2 prescribes) linked against the drop-in.  N3: the
    loop-closure matcher runs matchFeatureLoopClosure(std::vector<MapFeature>[2], int[2], pairedFeatures, T) on a second
    thread while the VO matcher processes a sequence; pairs, pose and ratio must equal the oracle's (match_hamming256 +
    ransac_rigid3d with the LC parameter set and errorVersionMap), and so must every VO pose."""
    from putslam_amd._abi import DMATCH_DTYPE
    exe = os.path.join(ROOT, "tests", "cpp", "test_reference_shaped")
    if not os.path.exists(exe):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_dropin()
    mode = REPROJECTION_ERROR
    seed_lc, seed_vo = 0x0123_4567_89AB_CDEF, 0x0F0F_1234_5678
    a, b = synth.make_pair(500, config=2, index=91, inlier_frac=0.6)
    m = oracle.match_hamming256(a["desc"], b["desc"])
    prm = default_ransac_params(mode, lc=True)                       # putslammatcherOpenCVParametersLC.xml:30
    prm.errorVersionMap = mode
    cfg, _ = make_config(EST_RANSAC, 1157, seed=(seed_lc + 0x9E3779B97F4A7C15) & (2 ** 64 - 1))
    r = oracle.ransac_rigid3d(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
    inl = m[r["mask"].astype(bool)]
    assert len(inl) > 20
    ratio = oracle.point_inlier_ratio(inl, m)
    F, nV = 7, 800
    seq = synth.make_sequence(F, nV, config=3, index=17)
    vprm = default_ransac_params(EUCLIDEAN_ERROR)                     # errorVersionVO = 0
    poses = []
    for k in range(1, F):
        mk = oracle.match_hamming256(seq["desc"][k - 1], seq["desc"][k])
        ck, _ = make_config(EST_RANSAC, 487, seed=seed_vo + (k - 1))
        poses.append(oracle.ransac_rigid3d(vprm, ck, TUM_FR1_K, seq["pts"][k - 1], seq["pts"][k], mk)["pose"])
    path = tmp_path / "lc_case.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4i2I2i", 500, 500, len(m), mode, seed_lc & 0xFFFFFFFF, seed_lc >> 32, len(inl), 0))
        f.write(a["desc"].tobytes()); f.write(b["desc"].tobytes())
        f.write(a["pts"].tobytes()); f.write(b["pts"].tobytes())
        f.write(np.ascontiguousarray(m, DMATCH_DTYPE).tobytes())
        f.write(np.stack([inl["queryIdx"], inl["trainIdx"]], axis=1).astype(np.int32).tobytes())
        f.write(np.ascontiguousarray(r["pose"].T, np.float32).tobytes())
        f.write(struct.pack("<d", ratio))
        f.write(struct.pack("<2i2I", F, nV, seed_vo & 0xFFFFFFFF, seed_vo >> 32))
        for k in range(F):
            f.write(seq["desc"][k].tobytes()); f.write(seq["pts"][k].tobytes())
        for T in poses:
            f.write(np.ascontiguousarray(T.T, np.float32).tobytes())
    p = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all checks passed" in p.stdout
