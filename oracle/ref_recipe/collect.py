#!/usr/bin/env python3
"""Turns the harness outputs under oracle/_ref/ (+ the inputs they were computed from) into tests/golden/ref_*.npz:
data files (inputs and the reference's outputs), nothing else."""
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# PUTSLAM_REF_DIR / PUTSLAM_GOLD_DIR: used by tests/test_ref_recipe_plumbing.py (temporary directories)
REF = os.environ.get("PUTSLAM_REF_DIR") or os.path.join(ROOT, "oracle", "_ref")
GOLD = os.environ.get("PUTSLAM_GOLD_DIR") or os.path.join(ROOT, "tests", "golden")


def _r(f, fmt):
    return struct.unpack(fmt, f.read(struct.calcsize(fmt)))


def eigen_core():
    fi, fo = os.path.join(REF, "inputs", "eigen_core.bin"), os.path.join(REF, "eigen_core.out")
    if not os.path.exists(fo):
        return
    with open(fi, "rb") as f:
        n, nk, _ = _r(f, "<3i")
        src = np.frombuffer(f.read(n * 36), np.float32).reshape(n, 3, 3)
        dst = np.frombuffer(f.read(n * 36), np.float32).reshape(n, 3, 3)
        mats = np.frombuffer(f.read(n * 36), np.float32).reshape(n, 3, 3)
        sets = []
        for _ in range(nk):
            (k,) = _r(f, "<i")
            sets.append((np.frombuffer(f.read(k * 12), np.float32).reshape(k, 3), np.frombuffer(f.read(k * 12), np.float32).reshape(k, 3)))
    with open(fo, "rb") as f:
        _r(f, "<3i")
        rec = np.frombuffer(f.read(n * 140), np.float32).reshape(n, 35)
        svd = np.frombuffer(f.read(n * 84), np.float32).reshape(n, 21)
        refits = []
        for _ in range(nk):
            _r(f, "<i")
            refits.append(np.frombuffer(f.read(64), np.float32).reshape(4, 4).T)
    out = dict(src=src, dst=dst, mats=mats, T=rec[:, :16].reshape(n, 4, 4).transpose(0, 2, 1),
               Tinv=rec[:, 16:32].reshape(n, 4, 4).transpose(0, 2, 1), xform=rec[:, 32:35],
               U=svd[:, :9].reshape(n, 3, 3), S=svd[:, 9:12], V=svd[:, 12:21].reshape(n, 3, 3))
    for i, ((s, d), T) in enumerate(zip(sets, refits)):
        out[f"set{i}_src"], out[f"set{i}_dst"], out[f"set{i}_T"] = s, d, T
    np.savez_compressed(os.path.join(GOLD, "ref_eigen_core.npz"), **out)


def kabsch():
    fi, fo = os.path.join(REF, "inputs", "kabsch.bin"), os.path.join(REF, "kabsch.out")
    if not os.path.exists(fo):
        return
    out = {}
    with open(fi, "rb") as f, open(fo, "rb") as g:
        (cases,) = _r(f, "<i")
        _r(g, "<i")
        for c in range(cases):
            (m,) = _r(f, "<i")
            out[f"A{c}"] = np.frombuffer(f.read(m * 24), np.float64).reshape(3, m).T
            out[f"B{c}"] = np.frombuffer(f.read(m * 24), np.float64).reshape(3, m).T
            _r(g, "<i")
            out[f"T{c}"] = np.frombuffer(g.read(128), np.float64).reshape(4, 4).T
    np.savez_compressed(os.path.join(GOLD, "ref_kabsch.npz"), **out)


def _pairs():
    with open(os.path.join(REF, "inputs", "pairs.bin"), "rb") as f:
        (cases,) = _r(f, "<i")
        for _ in range(cases):
            (n,) = _r(f, "<i")
            yield (n, np.frombuffer(f.read(n * 32), np.uint8).reshape(n, 32), np.frombuffer(f.read(n * 32), np.uint8).reshape(n, 32),
                   np.frombuffer(f.read(n * 12), np.float32).reshape(n, 3), np.frombuffer(f.read(n * 12), np.float32).reshape(n, 3))


def bfmatcher():
    fo = os.path.join(REF, "bfmatcher.out")
    if not os.path.exists(fo):
        return
    out = {}
    with open(fo, "rb") as g:
        _r(g, "<i")
        for c, (n, d0, d1, p0, p1) in enumerate(_pairs()):
            (m,) = _r(g, "<i")
            out[f"desc0_{c}"], out[f"desc1_{c}"] = d0, d1
            out[f"matches_{c}"] = np.frombuffer(g.read(m * 16), np.uint8).reshape(m, 16)   # cv::DMatch records
    np.savez_compressed(os.path.join(GOLD, "ref_bfmatcher.npz"), **out)


def ransac():
    fo = os.path.join(REF, "ransac.out")
    if not os.path.exists(fo):
        return
    out = {}
    with open(fo, "rb") as g:
        _r(g, "<i")
        for c, (n, d0, d1, p0, p1) in enumerate(_pairs()):
            out[f"desc0_{c}"], out[f"desc1_{c}"], out[f"pts0_{c}"], out[f"pts1_{c}"] = d0, d1, p0, p1
            for mode in range(2):
                _, _, nm, ninl, nit = _r(g, "<5i")
                out[f"pose_{c}_{mode}"] = np.frombuffer(g.read(64), np.float32).reshape(4, 4).T
                out[f"inliers_{c}_{mode}"] = np.frombuffer(g.read(ninl * 8), np.int32).reshape(ninl, 2)
                out[f"samples_{c}_{mode}"] = np.frombuffer(g.read(nit * 12), np.int32).reshape(nit, 3)
    np.savez_compressed(os.path.join(GOLD, "ref_ransac.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    eigen_core(); kabsch(); bfmatcher(); ransac()
    print("collected:", [f for f in os.listdir(GOLD) if f.startswith("ref_")])
