#!/usr/bin/env python3
"""Writes the raw little-endian input files of the reference harnesses into oracle/_ref/inputs/ (same seeded
generators as the tests: putslam_amd/synth.py).  File = int32 header fields, then arrays as documented per harness."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from putslam_amd import synth  # noqa: E402

# PUTSLAM_REF_DIR / PUTSLAM_REF_N: used by tests/test_ref_recipe_plumbing.py (a temporary directory, fewer samples)
OUT = os.path.join(os.environ.get("PUTSLAM_REF_DIR") or os.path.join(ROOT, "oracle", "_ref"), "inputs")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    # eigen core: N 3-point samples (src, dst), NK k-point sets, N 4x4 rigid matrices to invert, N 3x3 matrices
    n = int(os.environ.get("PUTSLAM_REF_N", "20000"))
    src = (rng.uniform(-2.5, 2.5, (n, 3, 3)) + [0, 0, 3]).astype(np.float32)
    dst = (src + rng.normal(0, 0.01, src.shape) + rng.uniform(-0.2, 0.2, (n, 1, 3))).astype(np.float32)
    dst[::7] = rng.uniform(-3, 3, dst[::7].shape).astype(np.float32)
    src[::97, 1] = src[::97, 0]
    ks = [4, 5, 17, 63, 64, 65, 128, 500, 1500]
    sets = [((rng.uniform(-2, 2, (k, 3)) + [0, 0, 3]).astype(np.float32)) for k in ks]
    sets_d = [(s + rng.normal(0, 0.004, s.shape) + [0.03, -0.02, 0.01]).astype(np.float32) for s in sets]
    mats = (rng.standard_normal((n, 3, 3)) * 10.0 ** rng.uniform(-4, 4, (n, 1, 1))).astype(np.float32)
    with open(os.path.join(OUT, "eigen_core.bin"), "wb") as f:
        f.write(struct.pack("<3i", n, len(ks), 0))
        f.write(src.tobytes()); f.write(dst.tobytes()); f.write(mats.tobytes())
        for k, s, d in zip(ks, sets, sets_d):
            f.write(struct.pack("<i", k)); f.write(s.tobytes()); f.write(d.tobytes())
    # kabsch: sizes of demoKabsch (demos/demoKabsch.cpp:983) and of BASELINE config 1
    with open(os.path.join(OUT, "kabsch.bin"), "wb") as f:
        sizes = [3, 4, 100, 500, 5000]
        f.write(struct.pack("<i", len(sizes)))
        for m in sizes:
            A = rng.uniform(-1.5, 1.5, (m, 3))
            B = A + [0.1, 0.2, -0.3] + rng.normal(0, 1, (m, 3)) * [0.01, 0.02, 0.03]
            f.write(struct.pack("<i", m)); f.write(np.asfortranarray(A).tobytes(order="F")); f.write(np.asfortranarray(B).tobytes(order="F"))
    # matcher + whole RANSAC: frame pairs of the synthetic generator
    with open(os.path.join(OUT, "pairs.bin"), "wb") as f:
        cases = [(64, 1), (500, 2), (2000, 3), (777, 4)]
        f.write(struct.pack("<i", len(cases)))
        for nk, idx in cases:
            a, b = synth.make_pair(nk, config=2, index=idx)
            f.write(struct.pack("<i", nk))
            f.write(a["desc"].tobytes()); f.write(b["desc"].tobytes()); f.write(a["pts"].tobytes()); f.write(b["pts"].tobytes())
    print("inputs written to", OUT)


if __name__ == "__main__":
    main()
