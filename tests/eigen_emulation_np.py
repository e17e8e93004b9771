"""An independent restatement of Eigen 3.3's JacobiSVD (3x3, real) and umeyama(src, dst, false) in numpy,
vectorised over a batch, written from the algorithm as Eigen documents / structures it (JacobiSVD::compute,
real_2x2_jacobi_svd, JacobiRotation::makeJacobi, apply_rotation_in_the_plane, Umeyama.h) -- NOT from
oracle/po_svd.inc or putslam_amd/csrc/ps_device_math.h.  tests/test_oracle_independent.py compares it bit for bit
with the oracle, so that a misreading of Eigen shared by the oracle and the device code (they are deliberate twins)
would have to be shared by a third, differently structured implementation as well.

Every arithmetic operation below is a single IEEE operation in the array's dtype (float32 or float64): numpy does not
fuse or reassociate elementwise operations, division and sqrt are correctly rounded.

Conventions where Eigen leaves the order to its product / reduction kernels (stated, not pinned -- oracle header):
  * means and the 3x3 covariance of a k = 3 sample: sequential sums ((a0 + a1) + a2), then * (1/n);
  * R = U * S * V^T for dynamic-size matrices: coefficient-based lazy product, sequential inner sum;
  * t = dst_mean - R * src_mean: column by column.
"""
import numpy as np


def _rot_rows(M, p, q, c, s, act):
    """apply_rotation_in_the_plane on rows p, q of every matrix of the batch: x' = c x + s y, y' = -s x + c y
    (skipped where the rotation is the identity, like Eigen's early return, and where `act` is False)."""
    x, y = M[:, p, :].copy(), M[:, q, :].copy()
    do = act & ~((c == 1) & (s == 0))
    cc, ss = c[:, None], s[:, None]
    nx = cc * x + ss * y
    ny = (-ss) * x + cc * y
    M[:, p, :] = np.where(do[:, None], nx, x)
    M[:, q, :] = np.where(do[:, None], ny, y)


def _rot_cols(M, p, q, c, s, act):
    x, y = M[:, :, p].copy(), M[:, :, q].copy()
    do = act & ~((c == 1) & (s == 0))
    cc, ss = c[:, None], s[:, None]
    nx = cc * x + ss * y
    ny = (-ss) * x + cc * y
    M[:, :, p] = np.where(do[:, None], nx, x)
    M[:, :, q] = np.where(do[:, None], ny, y)


def jacobi_svd3(A, max_sweeps=64):
    """A: (N,3,3).  Returns U (N,3,3), S (N,3) descending, V (N,3,3) with A = U diag(S) V^T, as JacobiSVD computes
    them for a square matrix (no QR preconditioner), ComputeFullU | ComputeFullV."""
    A = np.asarray(A)
    dt = A.dtype.type
    N = A.shape[0]
    tiny = np.finfo(A.dtype).tiny
    eps = np.finfo(A.dtype).eps
    one, zero, two = dt(1), dt(0), dt(2)
    with np.errstate(all="ignore"):
        scale = np.abs(A).reshape(N, 9).max(axis=1)
        scale = np.where(scale == 0, one, scale)
        W = (A / scale[:, None, None]).astype(A.dtype)
        U = np.broadcast_to(np.eye(3, dtype=A.dtype), (N, 3, 3)).copy()
        V = U.copy()
        max_diag = np.abs(np.stack([W[:, 0, 0], W[:, 1, 1], W[:, 2, 2]], 1)).max(axis=1)
        precision = two * eps
        running = np.ones(N, bool)
        for _ in range(max_sweeps):
            if not running.any():
                break
            finished = np.ones(N, bool)
            for p, q in ((1, 0), (2, 0), (2, 1)):
                thr = np.maximum(dt(tiny), precision * max_diag)
                act = running & ((np.abs(W[:, p, q]) > thr) | (np.abs(W[:, q, p]) > thr))
                finished &= ~act
                # --- real_2x2_jacobi_svd on the (p,q) block ---
                m00, m01, m10, m11 = W[:, p, p].copy(), W[:, p, q].copy(), W[:, q, p].copy(), W[:, q, q].copy()
                t = m00 + m11
                d = m10 - m01
                small = np.abs(d) < tiny
                u = t / d
                tmp = np.sqrt(one + u * u)
                s1 = np.where(small, zero, one / tmp)
                c1 = np.where(small, one, u / tmp)
                ident1 = (c1 == 1) & (s1 == 0)
                a00 = np.where(ident1, m00, c1 * m00 + s1 * m10)      # rot1 applied on the left of the 2x2 block
                a10 = np.where(ident1, m10, (-s1) * m00 + c1 * m10)
                a01 = np.where(ident1, m01, c1 * m01 + s1 * m11)
                a11 = np.where(ident1, m11, (-s1) * m01 + c1 * m11)
                del a10
                # --- makeJacobi(x = a00, y = a01, z = a11) ---
                deno = two * np.abs(a01)
                flat = deno < tiny
                tau = (a00 - a11) / deno
                w = np.sqrt(tau * tau + one)
                tt = np.where(tau > 0, one / (tau + w), one / (tau - w))
                sign_t = np.where(tt > 0, one, -one)
                n = one / np.sqrt(tt * tt + one)
                sr = np.where(flat, zero, (((-sign_t) * (a01 / np.abs(a01))) * np.abs(tt)) * n)
                cr = np.where(flat, one, n)
                # --- j_left = rot1 * j_right^T ---
                nsr = -sr
                cl = c1 * cr - s1 * nsr
                sl = c1 * nsr + s1 * cr
                _rot_rows(W, p, q, cl, sl, act)          # W.applyOnTheLeft(p, q, j_left)
                _rot_cols(U, p, q, cl, sl, act)          # U.applyOnTheRight(p, q, j_left^T)
                _rot_cols(W, p, q, cr, nsr, act)         # W.applyOnTheRight(p, q, j_right)
                _rot_cols(V, p, q, cr, nsr, act)         # V.applyOnTheRight(p, q, j_right)
                md = np.maximum(np.abs(W[:, p, p]), np.abs(W[:, q, q]))
                max_diag = np.where(act, np.maximum(max_diag, md), max_diag)
            running &= ~finished
        diag = np.stack([W[:, 0, 0], W[:, 1, 1], W[:, 2, 2]], 1)
        S = np.abs(diag)
        neg = diag < 0
        U = np.where(neg[:, None, :], -U, U)
        S = (S * scale[:, None]).astype(A.dtype)
        # descending selection sort, first maximum wins, stop at an all-zero tail
        stopped = np.zeros(N, bool)
        idx = np.arange(N)
        for i in range(3):
            tail = S[:, i:]
            pos = tail.argmax(axis=1) + i
            best = S[idx, pos]
            stopped |= best == 0
            swap = ~stopped & (pos != i)
            if swap.any():
                k, j = idx[swap], pos[swap]
                S[k, i], S[k, j] = S[k, j].copy(), S[k, i].copy()
                ui, uj = U[k, :, i].copy(), U[k, :, j].copy()
                U[k, :, i], U[k, :, j] = uj, ui
                vi, vj = V[k, :, i].copy(), V[k, :, j].copy()
                V[k, :, i], V[k, :, j] = vj, vi
    return U, S, V


def umeyama3(src, dst):
    """Eigen::umeyama(src, dst, with_scaling=false) for batches of 3-point samples.  src, dst: (N,3,3) float32,
    [sample][point][coordinate].  Returns T (N,4,4) row-major (T[:, :3, :3] = R, T[:, :3, 3] = t)."""
    src = np.asarray(src, np.float32)
    dst = np.asarray(dst, np.float32)
    N = src.shape[0]
    f = np.float32
    with np.errstate(all="ignore"):
        one_over_n = f(1) / f(3)
        sm = ((src[:, 0] + src[:, 1]) + src[:, 2]) * one_over_n
        dm = ((dst[:, 0] + dst[:, 1]) + dst[:, 2]) * one_over_n
        sd = src - sm[:, None, :]
        dd = dst - dm[:, None, :]
        sigma = np.empty((N, 3, 3), np.float32)
        for r in range(3):
            for c in range(3):
                acc = dd[:, 0, r] * sd[:, 0, c]
                acc = acc + dd[:, 1, r] * sd[:, 1, c]
                acc = acc + dd[:, 2, r] * sd[:, 2, c]
                sigma[:, r, c] = one_over_n * acc
        U, S, V = jacobi_svd3(sigma)
        # reflection fix: S(2) = -1 where det(U) det(V) < 0 (only the sign of the determinants matters: |det| = 1)
        sgn = np.linalg.det(U.astype(np.float64)) * np.linalg.det(V.astype(np.float64))
        s2 = np.where(sgn < 0, f(-1), f(1)).astype(np.float32)
        R = np.empty((N, 3, 3), np.float32)
        for i in range(3):
            for j in range(3):
                acc = U[:, i, 0] * V[:, j, 0]
                acc = acc + U[:, i, 1] * V[:, j, 1]
                acc = acc + (U[:, i, 2] * s2) * V[:, j, 2]
                R[:, i, j] = acc
        t = np.empty((N, 3), np.float32)
        for i in range(3):
            acc = dm[:, i] - R[:, i, 0] * sm[:, 0]
            acc = acc - R[:, i, 1] * sm[:, 1]
            acc = acc - R[:, i, 2] * sm[:, 2]
            t[:, i] = acc
    T = np.zeros((N, 4, 4), np.float32)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    T[:, 3, 3] = 1
    return T
