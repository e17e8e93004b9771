"""The defaults this repository hard-codes are the ones the reference ships in its resource files:
resources/putslammatcherOpenCVParameters.xml:29-37 (RANSAC) and :64-78 (matchXYZ), ...ParametersLC.xml:30,
resources/datasetConfig/freiburg1_desk.xml:5-8,20.  tests/golden/reference_xml_defaults.json holds the parsed values (data
only; tests/golden/make_reference_xml_defaults.py wrote it where /root/reference exists).  Where the reference IS present the
XML is parsed again, so the fixture cannot drift from it; everywhere the product's defaults are compared with the fixture:
putslam_amd._abi (default_ransac_params, TUM_FR1_K, TUM_FR1_DIST, TUM_DEPTH_SCALE), the C++ drop-in
(FrameMatcher::MatcherParameters) and the demos' parameter blocks.  No GPU."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_xml_defaults.json")))
RANSAC_FIELDS = ("verbose", "errorVersionVO", "errorVersionMap", "inlierThresholdEuclidean", "inlierThresholdReprojection",
                 "inlierThresholdMahalanobis", "minimalInlierRatioThreshold", "minimalNumberOfMatches", "usedPairs")


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="no reference checkout on this box (the fixture was parsed where there is one)")
def test_fixture_is_what_the_reference_ships():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_reference_xml_defaults as m
    assert m.parse("/root/reference") == FIX


@pytest.mark.parametrize("lc", [False, True])
def test_python_defaults_equal_the_shipped_xml(lc):
    from putslam_amd._abi import default_ransac_params
    p = default_ransac_params(0, lc=lc)
    want = FIX["lc" if lc else "vo"]["RANSAC"]
    for f in RANSAC_FIELDS:
        assert getattr(p, f) == want[f], (f, getattr(p, f), want[f])
    assert set(want) == set(RANSAC_FIELDS)                  # nothing in the element that the struct does not carry


def test_camera_constants_equal_the_dataset_config():
    from putslam_amd._abi import TUM_DEPTH_SCALE, TUM_FR1_DIST, TUM_FR1_K
    c = FIX["camera"]
    K = np.array([c["fu"], 0, c["Cu"], 0, c["fv"], c["Cv"], 0, 0, 1], np.float32)
    assert np.array_equal(TUM_FR1_K, K)
    assert list(TUM_FR1_DIST) == c["rgbDistortion"] and TUM_DEPTH_SCALE == c["depthImageScale"]
    assert (c["sizeU"], c["sizeV"]) == (640, 480)           # BASELINE.json's "640x480"


def test_dropin_defaults_equal_the_shipped_xml(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_hip()
    g.build_dropin()
    exe = tmp_path / "print_defaults"
    d = os.path.join(ROOT, "putslam_amd", "csrc", "dropin")
    lib = os.path.join(ROOT, "putslam_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", d, os.path.join(ROOT, "tests", "cpp", "print_defaults.cpp"),
                           "-o", str(exe), "-L", lib, "-lputslam_dropin", "-lputslam_hip", "-Wl,-rpath," + lib])
    got = json.loads(subprocess.check_output([str(exe)], text=True))
    want = FIX["vo"]
    for f in RANSAC_FIELDS:
        assert got[f] == want["RANSAC"][f], (f, got[f], want["RANSAC"][f])
    for f in ("matchingXYZSphereRadius", "matchingXYZacceptRatioOfBestMatch"):
        assert got[f] == want["MatcherOpenCV"][f]
    c = FIX["camera"]
    assert [np.float32(v) for v in got["K"]] == [np.float32(v) for v in (c["fu"], 0, c["Cu"], 0, c["fv"], c["Cv"], 0, 0, 1)]   # CV_32FC1


def test_demo_parameter_blocks_equal_the_shipped_xml():
    """The C++ demos fill PsRansacParams by hand (they link the C ABI only): the literals are the XML's."""
    want = FIX["vo"]["RANSAC"]
    c = FIX["camera"]
    for name in ("demo_latency.cpp", "demo_batch_queue.cpp", "demo_sequences_multi_gpu.cpp"):
        src = open(os.path.join(ROOT, "demos", "cpp", name)).read()
        for f in ("inlierThresholdEuclidean", "inlierThresholdReprojection", "inlierThresholdMahalanobis", "minimalInlierRatioThreshold",
                  "minimalNumberOfMatches", "usedPairs"):
            m = re.search(r"\b%s = ([0-9.]+);" % f, src)
            assert m and float(m.group(1)) == want[f], (name, f)
        m = re.search(r"K\[9\] = \{([^}]*)\}", src)
        vals = [float(x.strip().rstrip("f")) for x in m.group(1).split(",")]
        assert vals == [c["fu"], 0.0, c["Cu"], 0.0, c["fv"], c["Cv"], 0.0, 0.0, 1.0], name
