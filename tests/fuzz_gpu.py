#!/usr/bin/env python3
"""Randomised soak test: HIP path vs oracle over many random configurations (sizes, inlier ratios, noise,
error modes, estimators, thresholds, seeds).  tests/test_gpu_fuzz_slice.py collects a 300-configuration slice under
`-m gpu`; longer soaks are run by hand on the GPU box:

    python tests/fuzz_gpu.py --iters 3000 --seed 1
    python tests/fuzz_gpu.py --iters 100000 --procs 8 --modes 0,2,4 --counts     (round 3: the new scoring kernels)
    python tests/fuzz_gpu.py --batch --iters 400 --procs 4                       (staged scoring: random batches)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as po  # noqa: E402
from putslam_amd import api, synth  # noqa: E402
from putslam_amd._abi import EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config  # noqa: E402


def run(iters, seed, max_kpts=1500, ctx=None, verbose=True, modes=(0, 1, 2, 4, 3), counts=False):
    """Returns the number of configurations in which the HIP path and the oracle disagree.  counts=True also compares
    the inlier count of EVERY hypothesis (ps_debug_ransac_counts vs po_hypothesis_counts), not only the call's outputs."""
    rng = np.random.default_rng(seed)
    ctx = ctx or api.Context(0)
    t0 = time.time()
    bad = 0
    for it in range(iters):
        n = int(rng.integers(4, max_kpts))
        frac = float(rng.uniform(0.05, 0.95))
        noise = float(10 ** rng.uniform(-4, -1.3))
        pa, pb = synth.make_pair(n, config=7, index=int(rng.integers(0, 2 ** 31)), inlier_frac=frac, noise=noise)
        mode = int(rng.choice(list(modes)))
        # (RANSAC: the reference's 1157 -- the schedule never needs more than 574 with the default ratios -- or a cap of at most
        # 520: kernel 4 replays schedules of up to 512 iterations per wavefront out of registers, longer ones as a work-group)
        est, H = [(EST_RANSAC, [1157, 487, int(rng.integers(1, 520))][int(rng.integers(0, 3))]),
                  (EST_USAC, int(rng.integers(50, 3000))), (EST_FIXED, int(rng.integers(1, 3000)))][int(rng.integers(0, 3))]
        if est == EST_RANSAC and n <= 150 and rng.random() < 0.3:
            # long RANSAC caps: with minimalInlierRatioThreshold = 0.05 the schedule may run for 31 294 iterations (a stop table
            # of that many floats, the work-group replay, stage 1 walked by looping work-groups)
            H = int(rng.integers(2000, 60000))
        if est == EST_USAC and n <= 150 and rng.random() < 0.3:
            # long schedules (the reference's cap is 850 000): few matches keep the oracle's loop affordable; the stop table has
            # one entry per hypothesis and the work-group replay walks all of its search windows
            H = int(rng.integers(100000, 850001))
        prm = default_ransac_params(mode, lc=bool(rng.integers(0, 2)))
        prm.inlierThresholdEuclidean = float(10 ** rng.uniform(-3, -0.5))
        prm.inlierThresholdReprojection = float(10 ** rng.uniform(-1, 1.5))
        prm.minimalInlierRatioThreshold = float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.3, 0.5]))
        prm.minimalNumberOfMatches = int(rng.choice([3, 8, 10, 15, 40]))
        K = TUM_FR1_K if rng.random() < 0.9 else TUM_FR1_K * np.float32(10 ** rng.uniform(-3, 3))
        cfg, _ = make_config(est, H, seed=int(rng.integers(0, 2 ** 62)))
        mg = ctx.match_hamming256(pa["desc"], pb["desc"])
        mc = po.match_hamming256(pa["desc"], pb["desc"])
        ok = mg.tobytes() == mc.tobytes()
        g = ctx.ransac_rigid3d(prm, cfg, K, pa["pts"], pb["pts"], mc)
        c = po.ransac_rigid3d(prm, cfg, K, pa["pts"], pb["pts"], mc)
        ok &= np.array_equal(g["mask"], c["mask"]) and g["pose"].tobytes() == c["pose"].tobytes()
        for f in c["stats"].dtype.names:
            x, y = g["stats"][f], c["stats"][f]
            ok &= bool(x == y or (np.isnan(x) and np.isnan(y)))
        if counts:
            # (the scoring kernels do not run below the schedule's minimum number of matches -- RANSAC.cpp:77-80,
            # USAC_wrapper.cpp:120 -- and the diagnostic then returns zeros; the oracle's count function has no such gate)
            min_run = max(3, 8 if est == EST_USAC else int(prm.minimalNumberOfMatches))
            if int(c["stats"]["numMatchesValid"]) >= min_run:
                cg = ctx.debug_ransac_counts(prm, cfg, K, pa["pts"], pb["pts"], mc)
                cc, _ = po.hypothesis_counts(prm, cfg, K, pa["pts"], pb["pts"], mc)
                ok &= np.array_equal(cg, cc[: len(cg)])
        if not ok:
            bad += 1
            print("MISMATCH", dict(it=it, n=n, frac=frac, noise=noise, mode=mode, est=est, H=H), g["stats"], c["stats"], flush=True)
        if verbose and (it + 1) % 200 == 0:
            print(f"{it + 1} iterations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    return bad


STAT_FIELDS = ("numMatchesIn", "numMatchesValid", "bestHypothesis", "bestInlierCount", "iterationsRun", "numInliers",
               "accepted", "bestInlierRatio", "pointInlierRatio")


def run_batch(iters, seed, verbose=True, modes=(0, 1, 2, 4), oracle_pairs=3):
    """Random BATCHES large enough for the staged scoring (ps_score_fast.h): the staged run with the reordered match record
    (options prune = 1, reorder = 1) against the staged run in the original order (reorder = 0) and the complete one
    (prune = 0: the launch form the single-pair soak above checks against the oracle) on every pair and every output
    field, and against the oracle on a few pairs of each batch.  Returns the number of batches with a difference."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    ctxs = {}
    for pr in (1, 2, 0):   # 1: staged, reordered match record; 2: staged, original order; 0: complete
        ctxs[pr] = api.Context(0)
        ctxs[pr].set_option("prune", 2 if pr else 0)   # (2: staged whatever the batch size; the default asks the cost model)
        ctxs[pr].set_option("reorder", 1 if pr == 1 else 0)
    for it in range(iters):
        mode = int(rng.choice(list(modes)))
        est, H = [(EST_RANSAC, int(rng.choice([487, 1157, 2000]))), (EST_USAC, int(rng.integers(300, 5000))),
                  (EST_FIXED, int(rng.integers(257, 6000)))][int(rng.integers(0, 3))]
        long_usac = est == EST_USAC and rng.random() < 0.08
        if long_usac:  # caps up to the reference's 850 000: one stop-table entry per hypothesis, survivor lists of that length
            H = int(rng.integers(100000, 850001))
        if est == EST_FIXED and rng.random() < 0.08:  # many hypotheses, few pairs: long survivor lists, list stages in several passes
            H = int(rng.integers(20000, 120001))
            long_usac = True  # (the same shape of batch: few small frames)
        hb = (H + 255) // 256
        kpts = int(rng.integers(40, 300 if long_usac else 900))
        # (staged from P (hb - 1) >= 256 (Euclidean kernels) / 768 (reprojection kernels) on: most batches are above)
        frames = int(max(6, min(400, 800 // max(hb - 1, 1) + int(rng.integers(2, 60)))))
        if long_usac:
            frames = int(rng.integers(3, 9))  # (the oracle walks the schedule hypothesis by hypothesis)
        frac = float(rng.uniform(0.05, 0.95))
        noise = float(10 ** rng.uniform(-4, -1.5))
        seq = synth.make_sequence(frames, kpts, config=3, index=int(rng.integers(0, 2 ** 31)), inlier_frac=frac, noise=noise)
        if rng.random() < 0.3:  # ragged keypoint counts, a few nearly empty frames
            for f in rng.integers(0, frames, max(1, frames // 8)):
                seq["nkpts"][f] = int(rng.integers(0, kpts))
        if rng.random() < 0.2:  # quantised points: groups of bit-identical models and counts (ties)
            seq["pts"] = (np.round(seq["pts"] * 64) / 64).astype(np.float32)
        prm = default_ransac_params(mode, lc=bool(rng.integers(0, 2)))
        prm.inlierThresholdEuclidean = float(10 ** rng.uniform(-2.5, -0.7))
        prm.inlierThresholdReprojection = float(10 ** rng.uniform(-0.5, 1.2))
        prm.minimalInlierRatioThreshold = float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.3]))
        base = int(rng.integers(0, 2 ** 40))
        cfg, _ = make_config(est, H, seed=base)
        outs = {}
        for pr in (1, 2, 0):
            fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
            pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
            run_pairs(ctxs[pr], prm, cfg, TUM_FR1_K, fs, pb)
            outs[pr] = pb.download()
            if hasattr(ctxs[pr], "debug_keys_clean") and it % 8 == 0:
                # the matcher's atomicMin merge starts from an all-ones keys block that kernel 2 restores (no clearing launch)
                assert ctxs[pr].debug_keys_clean() == 0, ("keys block not all-ones at rest", it, pr)
        a = outs[1]
        P = len(seq["pairs"])
        why = []   # what differs, for the log: (against what, pair, field)

        def note(cond, *what):
            if not cond and len(why) < 12:
                why.append(what)
            return bool(cond)

        ok = True
        for tag, b in (("staged/original order", outs[2]), ("complete", outs[0])):
            ok &= note(np.array_equal(a["numMatches"], b["numMatches"]), tag, "numMatches")
            for p in range(P):
                n = int(a["numMatches"][p])
                ok &= note(a["pose"][p].tobytes() == b["pose"][p].tobytes(), tag, p, "pose")
                ok &= note(np.array_equal(a["inlierMask"][p, :n], b["inlierMask"][p, :n]), tag, p, "mask")
                for f in STAT_FIELDS:
                    x, y = a["stats"][p][f], b["stats"][p][f]
                    ok &= note(bool(x == y or (np.isnan(x) and np.isnan(y))), tag, p, f, x, y)
        for p in rng.integers(0, P, oracle_pairs):
            cfgp, _ = make_config(est, H, seed=base + int(p))
            c = po.vo_pairs(prm, cfgp, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"][p:p + 1], threads=1)
            n = int(c["numMatches"][0])
            ok &= note(n == int(a["numMatches"][p]) and np.array_equal(a["inlierMask"][p, :n], c["inlierMask"][0, :n]), "oracle", int(p), "mask")
            ok &= note(a["pose"][p].tobytes() == c["pose"][0].tobytes(), "oracle", int(p), "pose")
            for f in STAT_FIELDS:
                x, y = a["stats"][p][f], c["stats"][0][f]
                ok &= note(bool(x == y or (np.isnan(x) and np.isnan(y))), "oracle", int(p), f, x, y)
        if not ok:
            bad += 1
            print("MISMATCH", dict(it=it, mode=mode, est=est, H=H, frames=frames, kpts=kpts, frac=frac, noise=noise, seed=base),
                  "differences:", why, flush=True)
        if verbose and (it + 1) % 20 == 0:
            print(f"{it + 1} batches, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    for c in ctxs.values():
        c.close()
    return bad


def main():
    import faulthandler
    faulthandler.enable()   # a worker that dies on a signal leaves its Python stack in its log
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-kpts", type=int, default=1500)
    ap.add_argument("--modes", default="0,1,2,4,3", help="RANSAC::ERROR_VERSION values to draw from")
    ap.add_argument("--counts", action="store_true", help="also compare every hypothesis's inlier count")
    ap.add_argument("--procs", type=int, default=1, help="worker processes (seeds seed, seed+1, ...), iterations split evenly")
    ap.add_argument("--batch", action="store_true", help="random batches: staged scoring vs complete scoring vs oracle")
    ap.add_argument("--log-dir", default=None,
                    help="where every worker's FULL output is kept (default: gpurun_out/fuzz_logs under the repository)")
    ap.add_argument("--tag", default="", help="label for the log files of this run")
    a = ap.parse_args()
    modes = tuple(int(x) for x in a.modes.split(","))
    if a.procs > 1:
        import subprocess
        po.lib()  # (build the oracle once, before the workers start)
        per = (a.iters + a.procs - 1) // a.procs
        log_dir = a.log_dir or os.path.join(ROOT, "gpurun_out", "fuzz_logs")
        os.makedirs(log_dir, exist_ok=True)
        stamp = f"{a.tag or 'run'}_{'batch' if a.batch else 'single'}_{int(time.time())}_{os.getpid()}"
        ps, logs = [], []
        for i in range(a.procs):
            # every worker's complete output goes to a file of its own (round 3 lost the only failing worker's output to a
            # `tail -1`): nothing a worker printed before it died can be lost, whatever the caller does with this
            # process's stdout
            path = os.path.join(log_dir, f"{stamp}_worker{i}_seed{a.seed + i}.log")
            f = open(path, "w")
            logs.append((path, f))
            ps.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--iters", str(per), "--seed", str(a.seed + i),
                                        "--max-kpts", str(a.max_kpts), "--modes", a.modes] + (["--counts"] if a.counts else [])
                                       + (["--batch"] if a.batch else []), stdout=f, stderr=subprocess.STDOUT, text=True))
        failed = 0
        for i, p in enumerate(ps):
            p.wait()
            path, f = logs[i]
            f.close()
            out = open(path).read()
            lines = out.splitlines()
            tail = [l for l in lines if l.startswith("fuzz done") or l.startswith("MISMATCH")]
            print(f"[worker {i}, seed {a.seed + i}, rc {p.returncode}] " + " | ".join(tail[-4:]), flush=True)
            finished = any(l.startswith("fuzz done") for l in lines)
            if p.returncode != 0 or not finished:
                failed += 1
                # ANY non-zero exit or missing summary line fails the run and is shown in full (last 60 lines), whether it was a
                # mismatch, a Python exception, a HIP error or a signal (negative return code)
                kind = "MISMATCH" if any(l.startswith("MISMATCH") for l in lines) else \
                    (f"killed by signal {-p.returncode}" if p.returncode < 0 else "died")
                print(f"[worker {i}] {kind}: exit code {p.returncode}, full log kept at {path}; last output:\n  "
                      + "\n  ".join(lines[-60:]), flush=True)
        print(f"fuzz done: {per * a.procs} iterations over {a.procs} workers, modes {a.modes}, "
              f"{'no mismatches' if failed == 0 else f'{failed} WORKER(S) FAILED (MISMATCHES or crashes, see above)'}")
        return 1 if failed else 0
    if a.batch:
        bad = run_batch(a.iters, a.seed, modes=tuple(m for m in modes if m != 3))
    else:
        bad = run(a.iters, a.seed, a.max_kpts, modes=modes, counts=a.counts)
    print(f"fuzz done: {a.iters} iterations, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
