"""Common-mode check of the oracle (CPU only): tests/eigen_emulation_np.py is a third, independently written and
differently structured (batched numpy) restatement of Eigen 3.3's JacobiSVD / umeyama; it must agree with the oracle
(oracle/po_svd.inc, the device code's deliberate twin) BIT FOR BIT on 10^5 random 3x3 matrices of the kinds the path
produces.  This does not pin the oracle to a real Eigen build (nothing in this image can: DESIGN.md section 2) -- it
shows that the restated algorithm has one reading, not two authors' shared slip."""
import numpy as np
import pytest

import eigen_emulation_np as emu


def _matrices(rng, n, dtype):
    """Dense, badly scaled, rank-deficient (3-point covariances), near-diagonal and special matrices."""
    k = n // 5
    dense = rng.standard_normal((k, 3, 3)) * (10.0 ** rng.uniform(-6, 6, (k, 1, 1)))
    pts_s = rng.uniform(-3, 3, (k, 3, 3))
    pts_d = rng.uniform(-3, 3, (k, 3, 3))
    pts_s -= pts_s.mean(axis=1, keepdims=True)
    pts_d -= pts_d.mean(axis=1, keepdims=True)
    rank2 = np.einsum("npr,npc->nrc", pts_d, pts_s) / 3.0                    # what Umeyama hands to the SVD
    neardiag = np.zeros((k, 3, 3))
    neardiag[:, [0, 1, 2], [0, 1, 2]] = rng.standard_normal((k, 3))
    neardiag += rng.standard_normal((k, 3, 3)) * (10.0 ** rng.uniform(-12, -1, (k, 1, 1)))
    rot = np.linalg.qr(rng.standard_normal((k, 3, 3)))[0] * rng.choice([0.5, 1.0, 1.0, 2.0], (k, 1, 1))  # equal sigmas
    special = rng.integers(-2, 3, (n - 4 * k, 3, 3)).astype(np.float64)      # zeros, ties, singular, negative diagonals
    return np.concatenate([dense, rank2, neardiag, rot, special]).astype(dtype)


@pytest.mark.parametrize("dtype,n", [(np.float32, 100000), (np.float64, 20000)])
def test_jacobi_svd3_three_way_bitwise(oracle, dtype, n):
    A = _matrices(np.random.default_rng(20261003), n, dtype)
    U, S, V = emu.jacobi_svd3(A)
    bad = 0
    for i in range(n):
        u, s, v = oracle.jacobi_svd3(A[i], dtype)
        if u.tobytes() != U[i].tobytes() or s.tobytes() != S[i].tobytes() or v.tobytes() != V[i].tobytes():
            bad += 1
            if bad <= 3:
                print("mismatch", i, A[i], "\noracle", u, s, v, "\nemulation", U[i], S[i], V[i])
    assert bad == 0, f"{bad} of {n} matrices differ"
    # and the decomposition is one: U S V^T reproduces A, U and V orthogonal (loose tolerance, scaled by |A|)
    rec = np.einsum("nik,nk,njk->nij", U.astype(np.float64), S.astype(np.float64), V.astype(np.float64))
    tol = (300 if dtype == np.float32 else 3000) * np.finfo(dtype).eps
    scale = np.abs(A).reshape(n, 9).max(axis=1).astype(np.float64) + 1e-300
    assert (np.abs(rec - A).reshape(n, 9).max(axis=1) / scale).max() < tol
    assert np.abs(np.einsum("nki,nkj->nij", U, U) - np.eye(3)).max() < tol
    assert np.all(np.diff(S, axis=1) <= 0)


def test_umeyama3_three_way_bitwise(oracle):
    rng = np.random.default_rng(7)
    n = 30000
    src = (rng.uniform(-2.5, 2.5, (n, 3, 3)) + [0, 0, 3]).astype(np.float32)
    ang = rng.uniform(0, 0.3, n)
    axis = rng.standard_normal((n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    Kx = np.zeros((n, 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0] = -axis[:, 2], axis[:, 1], axis[:, 2]
    Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -axis[:, 0], -axis[:, 1], axis[:, 0]
    R = np.eye(3) + np.sin(ang)[:, None, None] * Kx + (1 - np.cos(ang))[:, None, None] * (Kx @ Kx)
    dst = (np.einsum("nij,npj->npi", R, src) + rng.uniform(-0.2, 0.2, (n, 1, 3)) +
           rng.normal(0, 0.004, (n, 3, 3))).astype(np.float32)
    # a share of degenerate samples: collinear, coincident, outlier correspondences (reflection candidates)
    src[::11, 2] = (2 * src[::11, 1] - src[::11, 0])
    src[::97, 1] = src[::97, 0]
    dst[::7] = rng.uniform(-3, 3, dst[::7].shape).astype(np.float32)
    T = emu.umeyama3(src, dst)
    bad = nan = 0
    for i in range(n):
        To, ok = oracle.umeyama_f32(src[i], dst[i])
        if not ok:                                    # isnan(T(0,0)) in the reference (RANSAC.cpp:239-242)
            nan += 1
            assert np.isnan(T[i, 0, 0]), i
            continue
        if To.tobytes() != T[i].tobytes():
            bad += 1
            if bad <= 3:
                print("mismatch", i, "\noracle", To, "\nemulation", T[i])
    assert bad == 0, f"{bad} of {n} samples differ"
    assert nan < n // 50
    det = np.linalg.det(T[~np.isnan(T[:, 0, 0]), :3, :3].astype(np.float64))
    assert np.abs(det - 1).max() < 1e-4              # always a proper rotation, reflection cases included
