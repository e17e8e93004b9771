"""Scoring a TUM-format trajectory the way the reference does ("so results can be scored with
ATE/RPE"): the reference's verification is its demos + the TUM RGB-D benchmark's evaluation protocol, shipped as
`scripts/associate.py`, `scripts/evaluate_ate.py`, `scripts/evaluate_rpe.py` (reference tree).  This module restates that
protocol -- host side, numpy -- so that the trajectories `saveTrajectoryFreiburgFormat` (`src/PUTSLAM/PUTSLAM.cpp:1006-1016`;
here `sharding.write_tum_trajectory`, `putslam_hip::VOTrajectory::freiburgLine`) produces can be scored without Python 2:

  associate   time-stamp association: every pair closer than `max_difference` is a candidate, candidates are taken in order
              of their difference, each stamp is used once (scripts/associate.py:71-100)
  ate         absolute trajectory error: closed-form (Horn) alignment of the associated positions, then the translational
              error per pose -- rmse / mean / median / std / min / max (scripts/evaluate_ate.py:48-83,128-161)
  rpe         relative pose error over pose pairs a fixed delta apart (seconds, frames, metres, radians or degrees) or over
              all / sampled pairs: translational and rotational error of (est_i^-1 est_j)^-1 (gt_i^-1 gt_j)
              (scripts/evaluate_rpe.py:47-76,117-291)

`tests/test_tum_eval.py` checks every function against golden vectors produced by the reference's own scripts run in the
build container (`tests/golden/make_tum_eval_golden.py`, which needs /root/reference; the vectors are committed).
Nothing here touches the GPU; it is the evaluation side of the path's output format.
"""
import math
import random

import numpy as np

_EPS4 = np.finfo(float).eps * 4.0


def read_trajectory(path):
    """TUM trajectory file: `stamp tx ty tz qx qy qz qw` per line, `#` comments, commas / tabs as separators
    (evaluate_rpe.py:78-108: lines with an all-zero quaternion or a NaN are skipped).  Returns (stamps (N,), poses (N, 7)),
    ascending in time; a stamp that occurs twice keeps its last line, like the reference's dict."""
    rows = {}
    with open(path) as f:
        for line in f.read().replace(",", " ").replace("\t", " ").split("\n"):
            if not line or line[0] == "#":
                continue
            v = [float(x) for x in line.split(" ") if x.strip() != ""]
            if len(v) < 8:
                continue
            if v[4:8] == [0, 0, 0, 0] or any(math.isnan(x) for x in v):
                continue
            rows[v[0]] = v[1:8]
    stamps = np.array(sorted(rows), np.float64)
    poses = np.array([rows[s] for s in stamps], np.float64).reshape(-1, 7)
    return stamps, poses


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """Index pairs (i, j) with |first[i] - (second[j] + offset)| < max_difference, each index used once, the closest
    candidates first (ties: the earlier first stamp, then the earlier second stamp), returned ascending in first stamp."""
    a = np.asarray(first_stamps, np.float64)
    b = np.asarray(second_stamps, np.float64)
    cand = []
    order_b = np.argsort(b, kind="stable")
    bs = b[order_b] + offset
    for i, t in enumerate(a):
        lo = np.searchsorted(bs, t - max_difference, side="left")
        hi = np.searchsorted(bs, t + max_difference, side="right")
        for k in range(max(lo - 1, 0), min(hi + 1, len(bs))):
            d = abs(t - (b[order_b[k]] + offset))
            if d < max_difference:
                cand.append((d, t, b[order_b[k]], i, int(order_b[k])))
    cand.sort(key=lambda c: (c[0], c[1], c[2]))
    used_a, used_b, out = set(), set(), []
    for d, ta, tb, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i)
            used_b.add(j)
            out.append((ta, tb, i, j))
    out.sort(key=lambda c: (c[0], c[1]))
    return [(i, j) for _, _, i, j in out]


def horn_align(model, data):
    """Rigid alignment of `model` onto `data` (both (3, n)) in closed form: R, t minimising sum |R model + t - data|^2,
    and the residual length per point."""
    model = np.asarray(model, np.float64)
    data = np.asarray(data, np.float64)
    mc = model - model.mean(axis=1, keepdims=True)
    dc = data - data.mean(axis=1, keepdims=True)
    W = np.zeros((3, 3))
    for k in range(model.shape[1]):
        W += np.outer(mc[:, k], dc[:, k])
    U, _, Vh = np.linalg.svd(W.T)
    S = np.identity(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = data.mean(axis=1, keepdims=True) - R @ model.mean(axis=1, keepdims=True)
    err = R @ model + t - data
    return R, t, np.sqrt(np.sum(err * err, axis=0))


def _stats(e):
    e = np.asarray(e, np.float64)
    return dict(rmse=float(np.sqrt(np.dot(e, e) / len(e))), mean=float(np.mean(e)), median=float(np.median(e)),
                std=float(np.std(e)), min=float(np.min(e)), max=float(np.max(e)))


def ate(gt_stamps, gt_xyz, est_stamps, est_xyz, offset=0.0, scale=1.0, max_difference=0.02):
    """Absolute trajectory error of `est` against `gt` (positions (N, 3)).  Returns dict(pairs, rmse, mean, median, std, min,
    max, rotation, translation, errors): the estimated positions are scaled, associated by time stamp and aligned onto the
    ground truth before the error is taken."""
    pairs = associate(gt_stamps, est_stamps, offset, max_difference)
    if len(pairs) < 2:
        raise ValueError("fewer than two associated poses: are these the right trajectories?")
    first = np.array([gt_xyz[i] for i, _ in pairs], np.float64).T
    second = np.array([np.asarray(est_xyz[j], np.float64) * scale for _, j in pairs]).T
    R, t, err = horn_align(second, first)
    out = _stats(err)
    out.update(pairs=len(pairs), rotation=R, translation=t, errors=err, associations=pairs)
    return out


def pose_matrix(pose7):
    """4 x 4 matrix of (tx, ty, tz, qx, qy, qz, qw); a quaternion of (numerically) zero length gives the identity rotation."""
    t = pose7[0:3]
    q = np.array(pose7[3:7], np.float64)
    nq = float(np.dot(q, q))
    M = np.identity(4)
    M[0:3, 3] = t
    if nq < _EPS4:
        return M
    q = q * math.sqrt(2.0 / nq)
    o = np.outer(q, q)
    M[0:3, 0:3] = [[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                   [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                   [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]]
    return M


def _closest(sorted_list, t):
    """The reference's bisection for "the entry closest to t" (evaluate_rpe.py:110-136): it remembers the closest entry among
    the ones it visits, which for an exact hit or a monotone list is the closest one overall."""
    lo, hi = 0, len(sorted_list)
    best, best_d = 0, abs(sorted_list[0] - t)
    while lo < hi:
        mid = int((hi + lo) / 2)
        d = abs(sorted_list[mid] - t)
        if d < best_d:
            best_d, best = d, mid
        if t == sorted_list[mid]:
            return mid
        if sorted_list[mid] > t:
            hi = mid
        else:
            lo = mid + 1
    return best


def _relative(a, b):
    return np.linalg.inv(a) @ b


def _angle(T):
    return math.acos(min(1.0, max(-1.0, (np.trace(T[0:3, 0:3]) - 1.0) / 2.0)))


def _scaled(T, s):
    out = np.array(T, np.float64)
    out[0:3, 3] *= s
    return out


def rpe(gt_stamps, gt_poses, est_stamps, est_poses, max_pairs=10000, fixed_delta=False, delta=1.0, delta_unit="s", offset=0.0,
        scale=1.0, seed=0):
    """Relative pose error (poses (N, 7)).  Returns dict(pairs, translation=stats [m], rotation=stats [rad], rows) with rows =
    (stamp_est_0, stamp_est_1, stamp_gt_0, stamp_gt_1, translational error, rotational error) per evaluated pair.  Pair
    sampling uses Python's `random` seeded with `seed` exactly as the reference's command line does (`random.seed(0)`)."""
    rng = random.Random(seed)
    gs = [float(s) for s in gt_stamps]
    es = [float(s) for s in est_stamps]
    G = {s: pose_matrix(p) for s, p in zip(gs, gt_poses)}
    E = {s: pose_matrix(p) for s, p in zip(es, est_poses)}
    gs, es = sorted(G), sorted(E)
    overlap = []
    for t_est in es:
        t_gt = gs[_closest(gs, t_est + offset)]
        back = es[_closest(es, t_gt - offset)]
        if back not in overlap:
            overlap.append(back)
    if len(overlap) < 2:
        raise ValueError("the time stamps of the two trajectories hardly overlap")
    n = len(es)
    if delta_unit == "s":
        index = es
    elif delta_unit in ("m", "rad", "deg"):
        steps = [_relative(E[es[i + 1]], E[es[i]]) for i in range(n - 1)]
        k = {"m": None, "rad": 1.0, "deg": 180.0 / math.pi}[delta_unit]
        index, acc = [0], 0.0
        for T in steps:
            acc += float(np.linalg.norm(T[0:3, 3])) if k is None else _angle(T) * k
            index.append(acc)
    elif delta_unit == "f":
        index = list(range(n))
    else:
        raise ValueError("unknown unit for delta: %r" % delta_unit)
    if not fixed_delta:
        if max_pairs == 0 or n < math.sqrt(max_pairs):
            pairs = [(i, j) for i in range(n) for j in range(n)]
        else:
            pairs = [(rng.randint(0, n - 1), rng.randint(0, n - 1)) for _ in range(max_pairs)]
    else:
        pairs = []
        for i in range(n):
            j = _closest(index, index[i] + delta)
            if j != n - 1:
                pairs.append((i, j))
        if max_pairs != 0 and len(pairs) > max_pairs:
            pairs = rng.sample(pairs, max_pairs)
    gt_interval = float(np.median([s - t for s, t in zip(gs[1:], gs[:-1])]))
    limit = 2 * gt_interval
    rows = []
    for i, j in pairs:
        e0, e1 = es[i], es[j]
        g0, g1 = gs[_closest(gs, e0 + offset)], gs[_closest(gs, e1 + offset)]
        if abs(g0 - (e0 + offset)) > limit or abs(g1 - (e1 + offset)) > limit:
            continue
        err = _relative(_scaled(_relative(E[e1], E[e0]), scale), _relative(G[g1], G[g0]))
        rows.append((e0, e1, g0, g1, float(np.linalg.norm(err[0:3, 3])), _angle(err)))
    if len(rows) < 2:
        raise ValueError("no matching time stamps between ground truth and estimate")
    rows = np.array(rows, np.float64)
    return dict(pairs=len(rows), translation=_stats(rows[:, 4]), rotation=_stats(rows[:, 5]), rows=rows)


def evaluate_files(gt_path, est_path, **kw):
    """ATE + RPE (one second apart, and frame to frame) of an estimated trajectory file against a ground-truth file."""
    gs, gp = read_trajectory(gt_path)
    es, ep = read_trajectory(est_path)
    a = ate(gs, gp[:, :3], es, ep[:, :3], **{k: v for k, v in kw.items() if k in ("offset", "scale", "max_difference")})
    r1 = rpe(gs, gp, es, ep, fixed_delta=True, delta=1.0, delta_unit="s", **{k: v for k, v in kw.items() if k in ("offset", "scale")})
    rf = rpe(gs, gp, es, ep, fixed_delta=True, delta=1, delta_unit="f", **{k: v for k, v in kw.items() if k in ("offset", "scale")})
    return dict(ate=a, rpe_per_second=r1, rpe_per_frame=rf)


def main(argv=None):
    """`python -m putslam_amd.tum_eval groundtruth.txt estimate.txt [--offset s] [--scale k] [--max_difference s]`: the summary
    lines of evaluate_ate.py --verbose and evaluate_rpe.py --fixed_delta --verbose."""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("groundtruth")
    ap.add_argument("estimate")
    ap.add_argument("--offset", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--max_difference", type=float, default=0.02)
    a = ap.parse_args(argv)
    ev = evaluate_files(a.groundtruth, a.estimate, offset=a.offset, scale=a.scale, max_difference=a.max_difference)
    print("compared_pose_pairs %d pairs" % ev["ate"]["pairs"])
    for k in ("rmse", "mean", "median", "std", "min", "max"):
        print("absolute_translational_error.%s %f m" % (k, ev["ate"][k]))
    for name, r in (("per_second", ev["rpe_per_second"]), ("per_frame", ev["rpe_per_frame"])):
        print("relative_pose_error.%s compared_pose_pairs %d pairs" % (name, r["pairs"]))
        for k in ("rmse", "mean", "median", "std", "min", "max"):
            print("relative_pose_error.%s translational_error.%s %f m" % (name, k, r["translation"][k]))
        for k in ("rmse", "mean", "median", "std", "min", "max"):
            print("relative_pose_error.%s rotational_error.%s %f deg" % (name, k, math.degrees(r["rotation"][k])))
    return ev


if __name__ == "__main__":
    main()
