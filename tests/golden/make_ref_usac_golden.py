#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own USAC code (include/putslam/USAC/USAC.h, compiled where it lies under /root/reference
by oracle/ref_usac/build.sh into oracle/_ref/usac_harness; run in the build container):

    python tests/golden/make_ref_usac_golden.py          # writes tests/golden/ref_usac.npz

  stop_*    USAC<T>::updateStandardStopping(numInliers, totPoints, 3)            (USAC.h:944-971)
  solve_*   USAC<T>::solve() under RANSAC_USAC's configuration over replayed outcomes (USAC.h:296-520, USAC_wrapper.cpp:62-100)
  sample_*  USAC<T>::generateUniformRandomSample fed the build's draw stream      (USAC.h:562-579)
Only data is committed: the queries made here and the numbers the reference's code answered."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def run(text):
    exe = os.path.join(ROOT, "oracle", "_ref", "usac_harness")
    p = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stderr[-500:]
    return p.stdout.split("\n")


def main():
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "ref_usac", "build.sh")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(20261004)
    out = {}
    # ---- stop: every count of every M up to 400, then larger M with every count up to M
    q = [(c, M) for M in range(3, 401) for c in range(0, M + 1)]
    for M in (487, 1000, 1700, 1776, 1777, 1778, 2000, 5000, 12000, 40000):
        q += [(c, M) for c in range(0, M + 1)]
    q = np.array(q, np.int64)
    lines = run("stop %d\n" % len(q) + "".join("%d %d\n" % (a, b) for a, b in q))
    out["stop_query"] = q.astype(np.uint32)
    out["stop_answer"] = np.array([int(x) for x in lines[:len(q)]], np.int64)
    # ---- solve: replayed outcomes
    cases, answers = [], []
    for k in range(600):
        M = int(rng.choice([8, 9, 15, 40, 133, 400, 1200, 3000]))
        n = int(rng.integers(1, 4000))
        frac = rng.choice([0.02, 0.05, 0.1, 0.3, 0.6, 0.9])
        good = rng.random(n) < frac ** 3                                   # hypotheses whose sample is all inliers
        counts = np.where(good, rng.integers(max(3, int(frac * M * 0.7)), max(4, int(frac * M) + 2), n), rng.integers(0, max(4, M // 20 + 4), n))
        counts = np.minimum(counts, M).astype(np.int32)
        valid = (rng.random(n) > 0.03).astype(np.int32)
        if k % 7 == 0:
            counts[:] = rng.integers(0, 4, n)                              # nothing ever good: the loop runs long
            n_keep = min(n, 300)
            counts, valid, n = counts[:n_keep], valid[:n_keep], n_keep
        lines = run("solve %d %d\n" % (M, n) + "".join("%d %d\n" % (v, c) for v, c in zip(valid, counts)))
        ok, hyp, best, stored = (int(x) for x in lines[0].split())
        cases.append((M, n, valid, counts))
        answers.append((ok, hyp, best, stored))
    out["solve_M"] = np.array([c[0] for c in cases], np.int32)
    out["solve_n"] = np.array([c[1] for c in cases], np.int32)
    out["solve_valid"] = np.concatenate([c[2] for c in cases]).astype(np.int8)
    out["solve_counts"] = np.concatenate([c[3] for c in cases]).astype(np.int32)
    out["solve_answer"] = np.array(answers, np.int64)
    # ---- sample: the reference's sampler on the build's draw stream
    sq, sa = [], []
    for seed in (0, 1, 42, 0xB0B0, 2 ** 40 + 12345):
        for M in (3, 4, 5, 8, 50, 333, 2000, 40000):
            H = 400
            lines = run("sample %d %d %d\n" % (seed, M, H))
            sq.append((seed, M, H))
            sa.append(np.array([[int(x) for x in l.split()] for l in lines[:H]], np.int32))
    out["sample_query"] = np.array(sq, np.uint64)
    out["sample_answer"] = np.stack(sa)
    # ---- end to end: the oracle's own per-hypothesis counts of synthetic frame pairs, replayed through the reference's solve()
    from oracle import oracle_py as po
    from putslam_amd import synth
    from putslam_amd._abi import EST_USAC, TUM_FR1_K, default_ransac_params, make_config
    e2e_q, e2e_a = [], []
    for n, index, frac, mode in ((300, 1, 0.7, 0), (600, 2, 0.4, 0), (600, 3, 0.15, 1), (1000, 4, 0.08, 0), (1500, 5, 0.04, 1),
                                 (2000, 6, 0.6, 2), (2000, 7, 0.03, 0), (800, 8, 0.25, 4), (64, 9, 0.5, 0), (2000, 10, 0.0, 0)):
        a, b = synth.make_pair(n, config=2, index=index, inlier_frac=frac)
        m = po.match_hamming256(a["desc"], b["desc"])
        prm = default_ransac_params(mode)
        H = 6000
        cfg, _ = make_config(EST_USAC, H, seed=1000 + index)
        counts, M = po.hypothesis_counts(prm, cfg, TUM_FR1_K, a["pts"], b["pts"], m)
        if M < 8:
            continue
        lines = run("solve %d %d\n" % (M, H) + "".join("1 %d\n" % c for c in counts))
        ok, hyp, best, stored = (int(x) for x in lines[0].split())
        e2e_q.append((n, index, int(frac * 1000), mode, H, M, int(np.asarray(counts, np.int64).sum())))
        e2e_a.append((ok, hyp, best, stored))
    out["e2e_query"] = np.array(e2e_q, np.int64)
    out["e2e_answer"] = np.array(e2e_a, np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_usac.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_usac.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
