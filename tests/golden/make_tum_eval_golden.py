#!/usr/bin/env python3
"""Golden vectors for putslam_amd/tum_eval.py from the reference's OWN evaluation scripts (scripts/associate.py,
evaluate_ate.py, evaluate_rpe.py under /root/reference), run in the build container:

    python tests/golden/make_tum_eval_golden.py          # writes tests/golden/tum_eval.npz

The scripts are Python 2; they are converted with lib2to3 into a temporary directory (nothing of them is kept) and their
functions are called on synthetic trajectories made here.  One interpreter-compatibility alias is set (`numpy.linalg.linalg`,
a module path numpy 2 no longer has).  The committed .npz holds only data: the input trajectories and the numbers the
reference's functions returned for them."""
import importlib
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    tmp = tempfile.mkdtemp(prefix="tum_ref_")
    for f in ("associate.py", "evaluate_ate.py", "evaluate_rpe.py"):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):                     # (tabs and spaces are mixed in one of the scripts: Python 3 refuses that)
        if f.endswith(".py"):
            p = os.path.join(tmp, f)
            text = open(p).read().expandtabs(8)
            open(p, "w").write(text)
    sys.path.insert(0, tmp)
    np.linalg.linalg = np.linalg
    mods = [importlib.import_module(m) for m in ("associate", "evaluate_ate", "evaluate_rpe")]
    return tmp, mods


def make_trajectories(seed, n=180, est_every=1, drop=0.0, jitter=0.004, noise=0.01, drift=0.0005):
    """Ground truth on a smooth path sampled at 100 Hz-ish stamps, estimate at 30 Hz with timestamp jitter, pose noise, drift
    and dropped frames."""
    rng = np.random.default_rng(seed)

    def quat(axis, ang):
        axis = axis / np.linalg.norm(axis)
        return np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])

    def pose(t):
        p = np.array([0.8 * np.sin(0.7 * t), 0.3 * np.cos(0.9 * t), 0.5 * np.sin(0.4 * t) + 0.1 * t])
        q = quat(np.array([0.2, 1.0, 0.1]), 0.5 * np.sin(0.6 * t))
        return np.concatenate([p, q])

    t0 = 1305031102.175304
    gs = t0 + np.arange(3 * n) / 100.0 + rng.uniform(-0.001, 0.001, 3 * n)
    gp = np.array([pose(t - t0) for t in gs])
    es, ep = [], []
    for k in range(n):
        if rng.random() < drop:
            continue
        t = t0 + k / 30.0 + rng.uniform(-jitter, jitter)
        p = pose(t - t0)
        p[:3] += rng.normal(0, noise, 3) + drift * k * np.array([1.0, -0.5, 0.2])
        dq = quat(rng.normal(size=3), rng.normal(0, 0.01))
        # q * dq
        x1, y1, z1, w1 = p[3:]
        x2, y2, z2, w2 = dq
        p[3:] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                 w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
        es.append(t)
        ep.append(p)
    return np.array(gs), gp, np.array(es), np.array(ep)


def main():
    tmp, (associate, evaluate_ate, evaluate_rpe) = load_reference()
    out = {}
    cases = [dict(seed=1), dict(seed=2, drop=0.2, jitter=0.012), dict(seed=3, n=60, noise=0.05, drift=0.004), dict(seed=4, n=300, jitter=0.03)]
    for ci, kw in enumerate(cases):
        gs, gp, es, ep = make_trajectories(**kw)
        out[f"c{ci}_gs"], out[f"c{ci}_gp"], out[f"c{ci}_es"], out[f"c{ci}_ep"] = gs, gp, es, ep
        first = {float(s): [repr(float(v)) for v in p] for s, p in zip(gs, gp)}
        second = {float(s): [repr(float(v)) for v in p] for s, p in zip(es, ep)}
        for oi, (offset, maxd) in enumerate(((0.0, 0.02), (0.013, 0.01))):
            m = associate.associate(dict(first), dict(second), offset, maxd)
            out[f"c{ci}_assoc{oi}"] = np.array(m, np.float64).reshape(-1, 2)
            out[f"c{ci}_assoc{oi}_args"] = np.array([offset, maxd])
            if len(m) >= 2:
                fx = np.matrix([[float(v) for v in first[a][0:3]] for a, b in m]).transpose()
                sx = np.matrix([[float(v) * 1.0 for v in second[b][0:3]] for a, b in m]).transpose()
                rot, trans, err = evaluate_ate.align(sx, fx)
                out[f"c{ci}_ate{oi}_rot"], out[f"c{ci}_ate{oi}_trans"], out[f"c{ci}_ate{oi}_err"] = np.asarray(rot), np.asarray(trans), np.asarray(err)
        G = {float(s): evaluate_rpe.transform44([float(s)] + [float(v) for v in p]) for s, p in zip(gs, gp)}
        E = {float(s): evaluate_rpe.transform44([float(s)] + [float(v) for v in p]) for s, p in zip(es, ep)}
        out[f"c{ci}_T0"] = G[float(gs[0])]
        for ri, rk in enumerate((dict(param_fixed_delta=True, param_delta=1.0, param_delta_unit="s"),
                                 dict(param_fixed_delta=True, param_delta=1, param_delta_unit="f"),
                                 dict(param_fixed_delta=True, param_delta=0.3, param_delta_unit="m", param_scale=1.02),
                                 dict(param_fixed_delta=True, param_delta=5.0, param_delta_unit="deg", param_offset=0.004),
                                 dict(param_fixed_delta=True, param_delta=0.1, param_delta_unit="rad", param_max_pairs=40),
                                 dict(param_fixed_delta=False, param_max_pairs=500),
                                 dict(param_fixed_delta=False, param_max_pairs=0) if len(es) <= 70 else dict(param_fixed_delta=False, param_max_pairs=200))):
            random.seed(0)
            res = evaluate_rpe.evaluate_trajectory(dict(G), dict(E), **rk)
            out[f"c{ci}_rpe{ri}"] = np.array(res, np.float64)
            out[f"c{ci}_rpe{ri}_args"] = np.array([rk.get("param_max_pairs", 10000), 1.0 if rk.get("param_fixed_delta") else 0.0,
                                                    float(rk.get("param_delta", 1.0)), "s m rad deg f".split().index(rk.get("param_delta_unit", "s")),
                                                    rk.get("param_offset", 0.0), rk.get("param_scale", 1.0)])
    np.savez_compressed(os.path.join(HERE, "tum_eval.npz"), **out)
    shutil.rmtree(tmp)
    print("wrote", os.path.join(HERE, "tum_eval.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
