"""putslam_amd/tum_eval.py (scoring a trajectory with the TUM protocol) against golden vectors from the
reference's own scripts/associate.py, evaluate_ate.py and evaluate_rpe.py (tests/golden/make_tum_eval_golden.py made them in
the build container; only the data is committed)."""
import os

import numpy as np
import pytest

from putslam_amd import tum_eval

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tum_eval.npz"))
CASES = sorted({k.split("_")[0] for k in G.files})
UNITS = "s m rad deg f".split()


@pytest.mark.parametrize("c", CASES)
def test_associate_matches_reference(c):
    gs, es = G[c + "_gs"], G[c + "_es"]
    for oi in range(2):
        offset, maxd = G[f"{c}_assoc{oi}_args"]
        want = G[f"{c}_assoc{oi}"]
        got = tum_eval.associate(gs, es, offset, maxd)
        assert len(got) == len(want) and len(got) > 10
        assert np.array_equal(gs[[i for i, _ in got]], want[:, 0])
        assert np.array_equal(es[[j for _, j in got]], want[:, 1])
        assert len({i for i, _ in got}) == len(got) == len({j for _, j in got})


@pytest.mark.parametrize("c", CASES)
def test_ate_matches_reference(c):
    gs, gp, es, ep = (G[f"{c}_{k}"] for k in ("gs", "gp", "es", "ep"))
    for oi in range(2):
        offset, maxd = G[f"{c}_assoc{oi}_args"]
        out = tum_eval.ate(gs, gp[:, :3], es, ep[:, :3], offset=offset, max_difference=maxd)
        np.testing.assert_allclose(out["rotation"], G[f"{c}_ate{oi}_rot"], atol=1e-12)
        np.testing.assert_allclose(out["translation"], G[f"{c}_ate{oi}_trans"], atol=1e-12)
        err = G[f"{c}_ate{oi}_err"]
        np.testing.assert_allclose(out["errors"], err, atol=1e-12)
        assert out["rmse"] == pytest.approx(np.sqrt(np.dot(err, err) / len(err)), abs=1e-12)
        assert out["median"] == pytest.approx(np.median(err), abs=1e-12) and out["pairs"] == len(err)


@pytest.mark.parametrize("c", CASES)
def test_pose_matrix_matches_reference(c):
    np.testing.assert_allclose(tum_eval.pose_matrix(G[c + "_gp"][0]), G[c + "_T0"], atol=1e-15)


@pytest.mark.parametrize("c", CASES)
def test_rpe_matches_reference(c):
    gs, gp, es, ep = (G[f"{c}_{k}"] for k in ("gs", "gp", "es", "ep"))
    ri = 0
    while f"{c}_rpe{ri}" in G.files:
        max_pairs, fixed, delta, unit, offset, scale = G[f"{c}_rpe{ri}_args"]
        want = G[f"{c}_rpe{ri}"]
        out = tum_eval.rpe(gs, gp, es, ep, max_pairs=int(max_pairs), fixed_delta=bool(fixed), delta=delta if UNITS[int(unit)] != "f" else int(delta),
                           delta_unit=UNITS[int(unit)], offset=offset, scale=scale, seed=0)
        assert out["rows"].shape == want.shape and len(want) >= 2, (c, ri)
        assert np.array_equal(out["rows"][:, :4], want[:, :4]), (c, ri)          # the same pose pairs, in the same order
        np.testing.assert_allclose(out["rows"][:, 4], want[:, 4], atol=1e-12)
        # acos near 1 amplifies the last bits of the trace: 1e-7 rad is far below the protocol's resolution
        np.testing.assert_allclose(out["rows"][:, 5], want[:, 5], atol=1e-7)
        assert out["translation"]["rmse"] == pytest.approx(np.sqrt(np.dot(want[:, 4], want[:, 4]) / len(want)), abs=1e-12)
        ri += 1
    assert ri == 7


def test_read_trajectory_and_evaluate_files(tmp_path):
    c = CASES[1]
    gs, gp, es, ep = (G[f"{c}_{k}"] for k in ("gs", "gp", "es", "ep"))
    gt, est = tmp_path / "gt.txt", tmp_path / "est.txt"
    with open(gt, "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for s, p in zip(gs, gp):
            f.write("%.6f %s\n" % (s, " ".join(repr(float(v)) for v in p)))
        f.write("%.6f 0 0 0 0 0 0 0\n" % (gs[-1] + 1))                            # no orientation: skipped
        f.write("%.6f nan 0 0 0 0 0 1\n" % (gs[-1] + 2))                          # NaN: skipped
    with open(est, "w") as f:
        for s, p in zip(es, ep):
            f.write("%.6f,%s\n" % (s, ",".join(repr(float(v)) for v in p)))       # commas are separators too
    s2, p2 = tum_eval.read_trajectory(str(gt))
    assert len(s2) == len(gs) and np.allclose(s2, gs, atol=1e-6) and np.array_equal(p2, gp)
    out = tum_eval.evaluate_files(str(gt), str(est))
    assert 0.0 < out["ate"]["rmse"] < 0.05 and out["ate"]["pairs"] > 100
    assert out["rpe_per_second"]["pairs"] > 50 and out["rpe_per_frame"]["pairs"] > 100
    # a trajectory scored against itself is exact
    zero = tum_eval.evaluate_files(str(gt), str(gt))
    assert zero["ate"]["max"] < 1e-12 and zero["rpe_per_frame"]["translation"]["max"] < 1e-12


def test_ate_is_invariant_to_a_rigid_motion_of_the_estimate():
    c = CASES[0]
    gs, gp, es, ep = (G[f"{c}_{k}"] for k in ("gs", "gp", "es", "ep"))
    base = tum_eval.ate(gs, gp[:, :3], es, ep[:, :3])
    a = 0.7
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    moved = ep[:, :3] @ R.T + np.array([3.0, -2.0, 0.5])
    assert tum_eval.ate(gs, gp[:, :3], es, moved)["rmse"] == pytest.approx(base["rmse"], abs=1e-10)
    with pytest.raises(ValueError):
        tum_eval.ate(gs, gp[:, :3], es + 1000.0, ep[:, :3])
