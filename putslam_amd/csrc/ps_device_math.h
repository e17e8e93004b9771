// ps_device_math.h -- device-side arithmetic of the Matcher -> RANSAC -> Kabsch path (gfx950).
//
// Every function here reproduces, operation for operation, the arithmetic the reference
// delegates to Eigen 3.3 / OpenCV 3.x at the call sites cited below, with each product and
// sum rounded separately (the translation unit is compiled with -ffp-contract=off; float
// divide and sqrt are the correctly rounded HIP defaults).  The reference is an x86-64 SSE2
// scalar build without FMA (reference CMakeLists.txt:23,147), so inlier decisions taken on
// the device are bit-for-bit the ones a sequential IEEE evaluation takes.
//
// No code in this directory includes or links anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PS_HD __host__ __device__ __forceinline__
#define PS_D __device__ __forceinline__

#define PS_SVD_MAX_SWEEPS 64

namespace psdev {

template <typename T> struct Lim;
template <> struct Lim<float> {
    static PS_HD float min_normal() { return 1.17549435e-38f; }
    static PS_HD float eps() { return 1.1920929e-07f; }
};
template <> struct Lim<double> {
    static PS_HD double min_normal() { return 2.2250738585072014e-308; }
    static PS_HD double eps() { return 2.220446049250313e-16; }
};

PS_HD float ps_sqrt(float x) { return sqrtf(x); }
PS_HD double ps_sqrt(double x) { return sqrt(x); }
PS_HD float ps_abs(float x) { return fabsf(x); }
PS_HD double ps_abs(double x) { return fabs(x); }


// ---- exact float sqrt / reciprocal / quotients with the range fix-ups removed (device, float only) ----------------
// hipcc expands a correctly rounded sqrtf(x) into 16 vector instructions: three that rescale a tiny argument
// (x < 2^-96), v_sqrt_f32, the one-ulp correction (two FMAs on the neighbours, two compares, two selects), two that scale
// the result back and a v_cmp_class + select that returns the argument for +-0 and +inf.  For an argument that is known to be
// >= 1 (or +inf, or NaN) the rescaling is a no-op and the class select changes nothing (v_sqrt_f32(+inf) = +inf and the
// correction's compares are false for it; NaN stays NaN), so the nine instructions below ARE the compiler's sequence and
// return the same bits.  Every square root of the Jacobi SVD has the form sqrt(v * v + 1).  Likewise a correctly rounded
// a / b is div_scale x 2, rcp, 3 + 3 FMA / MUL, div_fmas, div_fixup: with numerator and denominator inside the window
// where neither div_scale rescales and div_fixup passes its first operand through (ps_kernels.h, div2_shared) the plain
// FMA chain below is that sequence.  ps_debug_mathcheck compares each of these with the operator bit for bit
// (tests/test_gpu_parity.py: every float >= 1 for the square root, every float of [1, 2] for the reciprocal, 10^9 random
// operands for the quotients).
#if defined(__HIP_DEVICE_COMPILE__)
#define PS_FAST_EXACT 1 // (the PS_HD functions below take the short forms in the device pass only)
#endif
constexpr float kExactDivLo = 9.094947017729282e-13f; // 2^-40
constexpr float kExactDivHi = 1.099511627776e12f;     // 2^+40
PS_D bool wave_every(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }

// sqrtf(x) for x >= 1, x = +inf or x = NaN
PS_D float sqrt_ge1(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    float r = (0.0f >= rl) ? lo : s;
    r = (0.0f < rh) ? hi : r;
    return r;
}
// the reciprocal chain shared by the quotients below: rcp + one Newton step (the compiler's first three instructions)
PS_D float rcp_refined(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
// a / b given r1 = rcp_refined(b), operands inside the window
PS_D float div_with(float a, float b, float r1)
{
    const float m = a * r1;
    const float f = __builtin_fmaf(-b, m, a);
    const float g = __builtin_fmaf(f, r1, m);
    const float h = __builtin_fmaf(-b, g, a);
    return __builtin_fmaf(h, r1, g);
}
// 1 / b for 1 <= b <= 2^40 (no window check needed by the callers that know the range)
PS_D float rcp_exact_in_window(float b)
{
    const float r1 = rcp_refined(b);
    const float f = __builtin_fmaf(-b, r1, 1.0f);
    const float g = __builtin_fmaf(f, r1, r1);
    const float h = __builtin_fmaf(-b, g, 1.0f);
    return __builtin_fmaf(h, r1, g);
}
// y / |y| for |y| >= FLT_MIN (the caller's guard), +-inf or NaN: +-1, NaN for the non-finite ones -- three instructions
// instead of a division
PS_D float unit_sign(float y) { return __builtin_copysignf(1.0f, y) + (y - y); }

// 1 / d and u / d for d = sqrt(1 + u * u) >= 1 (JacobiSVD's real_2x2_jacobi_svd): one reciprocal for both inside the
// window, the wavefront's ordinary divisions otherwise
PS_HD void inv_and_quot(float u, float d, float &inv, float &quo)
{
#ifdef PS_FAST_EXACT
    // d >= 1 and |u| < d by construction; a zero, tiny or NaN numerator and a huge denominator leave the window
    if (wave_every(ps_abs(u) >= kExactDivLo && d <= kExactDivHi)) {
        const float r1 = rcp_refined(d);
        inv = div_with(1.0f, d, r1);
        quo = div_with(u, d, r1);
        return;
    }
#endif
    inv = 1.0f / d;
    quo = u / d;
}
PS_HD void inv_and_quot(double u, double d, double &inv, double &quo)
{
    inv = 1.0 / d;
    quo = u / d;
}
// W = A / scale, entry by entry (JacobiSVD's m_workMatrix = matrix / scale): nine quotients with one denominator
PS_HD void scale_down3(const float (&A)[3][3], float scale, float (&W)[3][3])
{
#ifdef PS_FAST_EXACT
    float lo = fminf(fminf(ps_abs(A[0][0]), ps_abs(A[0][1])), ps_abs(A[0][2]));
    lo = fminf(fminf(lo, ps_abs(A[1][0])), ps_abs(A[1][1]));
    lo = fminf(fminf(lo, ps_abs(A[1][2])), ps_abs(A[2][0]));
    lo = fminf(fminf(lo, ps_abs(A[2][1])), ps_abs(A[2][2]));
    // (scale = the largest |entry|: with the smallest one and the scale itself inside the window every operand is; a NaN
    // entry is skipped by fminf and yields NaN on either path; a NaN scale fails the comparison)
    if (wave_every(lo >= kExactDivLo && scale <= kExactDivHi)) {
        const float r1 = rcp_refined(scale);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) W[i][j] = div_with(A[i][j], scale, r1);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) W[i][j] = A[i][j] / scale;
}
PS_HD void scale_down3(const double (&A)[3][3], double scale, double (&W)[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) W[i][j] = A[i][j] / scale;
}
PS_HD float sqrt_1p(float v2) // sqrt(v2 + 1) for v2 = a square (>= 0, +inf or NaN)
{
#ifdef PS_FAST_EXACT
    return sqrt_ge1(v2 + 1.0f);
#else
    return sqrtf(v2 + 1.0f);
#endif
}
PS_HD double sqrt_1p(double v2) { return sqrt(v2 + 1.0); }
PS_HD float rcp_1to2(float x) // 1 / x for x in [1, 2] (or NaN)
{
#ifdef PS_FAST_EXACT
    return rcp_exact_in_window(x);
#else
    return 1.0f / x;
#endif
}
PS_HD double rcp_1to2(double x) { return 1.0 / x; }
PS_HD float unit_of(float y) // y / |y|
{
#ifdef PS_FAST_EXACT
    return unit_sign(y);
#else
    return y / fabsf(y);
#endif
}
PS_HD double unit_of(double y) { return y / fabs(y); }

// Plane rotation of a scalar pair: x' = c*x + s*y, y' = -s*x + c*y.
template <typename T> PS_HD void rot_pair(T &x, T &y, T c, T s)
{
    T xi = x, yi = y;
    x = c * xi + s * yi;
    y = (-s) * xi + c * yi;
}

// Symmetric 2x2 Jacobi rotation for (x, y, z) = (m00, m01, m11).
template <typename T> PS_HD void make_jacobi(T x, T y, T z, T &c, T &s)
{
    T deno = T(2) * ps_abs(y);
    if (deno < Lim<T>::min_normal()) {
        c = T(1);
        s = T(0);
        return;
    }
    T tau = (x - z) / deno;
    T w = sqrt_1p(tau * tau); // (tau * tau + 1 rounds the product first, then the sum: the same two operations)
    T t;
    if (tau > T(0))
        t = T(1) / (tau + w);
    else
        t = T(1) / (tau - w);
    T sign_t = t > T(0) ? T(1) : T(-1);
    // |t| <= 1 (|tau +- w| >= w >= 1), so t * t + 1 lies in [1, 2] and its root in [1, 1.4143]
    T n = rcp_1to2(sqrt_1p(t * t));
    s = (((-sign_t) * unit_of(y)) * ps_abs(t)) * n;
    c = n;
}

// Two-sided Jacobi SVD of a square real 3x3 matrix as Eigen 3.3's JacobiSVD performs it when
// rows == cols (no QR preconditioner step): used by Eigen::umeyama (reference
// src/TransformEst/RANSAC.cpp:225) and by src/TransformEst/kabschEst.cpp:47.
// A, U, V are row-major [3][3]; S descending.  Sweeps are capped at PS_SVD_MAX_SWEEPS.
template <typename T> PS_HD void jacobi_svd3(const T (&A)[3][3], T (&U)[3][3], T (&S)[3], T (&V)[3][3])
{
    T W[3][3];
    // cwiseAbs().maxCoeff(): the visitor starts from (0,0) and walks column by column keeping
    // `value > current`, so a NaN at (0,0) makes the scale (and then the whole work matrix) NaN.
    T scale = ps_abs(A[0][0]);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T a = ps_abs(A[i][j]);
            if (a > scale) scale = a;
        }
    if (scale == T(0)) scale = T(1);
    scale_down3(A, scale, W);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) U[i][j] = V[i][j] = (i == j) ? T(1) : T(0);
    const T precision = T(2) * Lim<T>::eps();
    const T considerAsZero = Lim<T>::min_normal();
    T maxDiag = ps_abs(W[0][0]);
    if (ps_abs(W[1][1]) > maxDiag) maxDiag = ps_abs(W[1][1]);
    if (ps_abs(W[2][2]) > maxDiag) maxDiag = ps_abs(W[2][2]);

    bool finished = false;
    for (int sweep = 0; sweep < PS_SVD_MAX_SWEEPS && !finished; ++sweep) {
        finished = true;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            // (p,q) visits (1,0), (2,0), (2,1) with compile-time indices (no scratch arrays)
            const int p = (pq == 0) ? 1 : 2;
            const int q = (pq == 2) ? 1 : 0;
            T pm = precision * maxDiag;
            T threshold = (considerAsZero < pm) ? pm : considerAsZero;
            if (ps_abs(W[p][q]) > threshold || ps_abs(W[q][p]) > threshold) {
                finished = false;
                T m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                T t = m00 + m11;
                T d = m10 - m01;
                T c1, s1;
                if (ps_abs(d) < Lim<T>::min_normal()) {
                    s1 = T(0);
                    c1 = T(1);
                } else {
                    T u = t / d;
                    T tmp = sqrt_1p(u * u); // (1 + u * u: the sum is commutative)
                    inv_and_quot(u, tmp, s1, c1); // s1 = 1 / tmp, c1 = u / tmp
                }
                if (!(c1 == T(1) && s1 == T(0))) {
                    rot_pair(m00, m10, c1, s1);
                    rot_pair(m01, m11, c1, s1);
                }
                T cr, sr;
                make_jacobi(m00, m01, m11, cr, sr);
                T nsr = -sr;
                T cl = c1 * cr - s1 * nsr;
                T sl = c1 * nsr + s1 * cr;
                if (!(cl == T(1) && sl == T(0))) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) rot_pair(W[p][i], W[q][i], cl, sl);
#pragma unroll
                    for (int i = 0; i < 3; ++i) rot_pair(U[i][p], U[i][q], cl, sl);
                }
                if (!(cr == T(1) && nsr == T(0))) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) rot_pair(W[i][p], W[i][q], cr, nsr);
#pragma unroll
                    for (int i = 0; i < 3; ++i) rot_pair(V[i][p], V[i][q], cr, nsr);
                }
                T a = ps_abs(W[p][p]), b = ps_abs(W[q][q]);
                T mx = (a < b) ? b : a;
                if (maxDiag < mx) maxDiag = mx;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T a = W[i][i];
        S[i] = ps_abs(a);
        if (a < T(0)) {
#pragma unroll
            for (int r = 0; r < 3; ++r) U[r][i] = -U[r][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) S[i] = S[i] * scale;
    // selection sort, descending, first maximum wins; stops at the first all-zero tail
    bool stop = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (stop) continue;
        int pos = i;
        T best = S[i];
#pragma unroll
        for (int j = i + 1; j < 3; ++j)
            if (S[j] > best) {
                best = S[j];
                pos = j;
            }
        if (best == T(0)) {
            stop = true;
            continue;
        }
        if (pos != i) {
            // pos > i, both compile-time bounded: spell out the swaps to keep registers
#pragma unroll
            for (int j = i + 1; j < 3; ++j)
                if (j == pos) {
                    T tmp = S[i];
                    S[i] = S[j];
                    S[j] = tmp;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        tmp = U[r][i]; U[r][i] = U[r][j]; U[r][j] = tmp;
                        tmp = V[r][i]; V[r][i] = V[r][j]; V[r][j] = tmp;
                    }
                }
        }
    }
}

PS_HD float det3(const float (&M)[3][3])
{
    return (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0])) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// Rigid model: R row-major, t.  valid == false <=> isnan(T(0,0)) in the reference (RANSAC.cpp:239-242).
struct Rigid {
    float R[3][3];
    float t[3];
};

// Tail of Eigen::umeyama(src, dst, false) once means and sigma are known:
// SVD, reflection fix by sign(det U * det V), R = U*S*V^T, t = dst_mean - R*src_mean (column by column).
PS_HD bool umeyama_finish(const float (&sigma)[3][3], const float (&sm)[3], const float (&dm)[3], Rigid &M)
{
    float U[3][3], V[3][3], S[3];
    jacobi_svd3<float>(sigma, U, S, V);
    float s2 = (det3(U) * det3(V) < 0.0f) ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            M.R[i][j] = (U[i][0] * V[j][0] + U[i][1] * V[j][1]) + (U[i][2] * s2) * V[j][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) M.t[i] = ((dm[i] - M.R[i][0] * sm[0]) - M.R[i][1] * sm[1]) - M.R[i][2] * sm[2];
    return !(M.R[0][0] != M.R[0][0]);
}

PS_HD void set_identity(Rigid &M)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) M.R[i][j] = (i == j) ? 1.0f : 0.0f;
        M.t[i] = 0.0f;
    }
}

// 3-point minimal fit (RANSAC.cpp:100 -> :207-244).  Sums follow the canonical order used for any k
// (64 strided partials + stride-1,2,4.. tree), which for three points is ((a+b)+c)+0.
PS_HD bool umeyama3(const float (&src)[3][3], const float (&dst)[3][3], Rigid &M)
{
    const float one_over_n = 1.0f / 3.0f;
    float sm[3], dm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sm[c] = (((src[0][c] + src[1][c]) + src[2][c]) + 0.0f) * one_over_n;
        dm[c] = (((dst[0][c] + dst[1][c]) + dst[2][c]) + 0.0f) * one_over_n;
    }
    float sigma[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float p0 = (dst[0][r] - dm[r]) * (src[0][c] - sm[c]);
            float p1 = (dst[1][r] - dm[r]) * (src[1][c] - sm[c]);
            float p2 = (dst[2][r] - dm[r]) * (src[2][c] - sm[c]);
            sigma[r][c] = one_over_n * (((p0 + p1) + p2) + 0.0f);
        }
    bool ok = umeyama_finish(sigma, sm, dm, M);
    if (!ok) set_identity(M);
    return ok;
}

// Eigen Matrix4f::inverse(), generic cofactor path, for T = [R t; 0 0 0 1]
// (RANSAC.cpp:337-338,386-387).  m is the full 4x4 accessed as m(r,c).
// (Elements through a compile-time accessor and the cofactors as twelve named scalars: with float[4][4] temporaries the
// compiler turned rows into 4-vectors and, to take them apart again, wrote one to scratch memory -- 20 bytes per lane that
// made every launch of the scoring kernels a launch with a scratch allocation.)
template <int R, int C> PS_HD float mat4_at(const Rigid &M)
{
    if (R < 3 && C < 3) return M.R[R < 3 ? R : 0][C < 3 ? C : 0];
    if (R < 3) return M.t[R < 3 ? R : 0];
    return C == 3 ? 1.0f : 0.0f;
}
template <int I1, int I2, int I3, int J1, int J2, int J3> PS_HD float det3_helper(const Rigid &m)
{
    return mat4_at<I1, J1>(m) * (mat4_at<I2, J2>(m) * mat4_at<I3, J3>(m) - mat4_at<I2, J3>(m) * mat4_at<I3, J2>(m));
}
template <int I, int J> PS_HD float cofactor4(const Rigid &m)
{
    constexpr int i1 = (I + 1) % 4, i2 = (I + 2) % 4, i3 = (I + 3) % 4;
    constexpr int j1 = (J + 1) % 4, j2 = (J + 2) % 4, j3 = (J + 3) % 4;
    return (det3_helper<i1, i2, i3, j1, j2, j3>(m) + det3_helper<i2, i3, i1, j1, j2, j3>(m)) +
           det3_helper<i3, i1, i2, j1, j2, j3>(m);
}
// result(j,i) = (-1)^(i+j) * cofactor(i,j)
template <int I, int J> PS_HD float adj4(const Rigid &m) { return ((I + J) & 1) ? -cofactor4<I, J>(m) : cofactor4<I, J>(m); }
PS_HD void inverse_rigid_general(const Rigid &M, Rigid &Inv)
{
    // rows 0 .. 2 of the adjugate (row 3 of the inverse is never used), r_ji = adj4<i, j>
    const float r00 = adj4<0, 0>(M), r01 = adj4<1, 0>(M), r02 = adj4<2, 0>(M), r03 = adj4<3, 0>(M);
    const float r10 = adj4<0, 1>(M), r11 = adj4<1, 1>(M), r12 = adj4<2, 1>(M), r13 = adj4<3, 1>(M);
    const float r20 = adj4<0, 2>(M), r21 = adj4<1, 2>(M), r22 = adj4<2, 2>(M), r23 = adj4<3, 2>(M);
    const float p0 = mat4_at<0, 0>(M) * r00, p1 = mat4_at<1, 0>(M) * r01, p2 = mat4_at<2, 0>(M) * r02, p3 = 0.0f * r03;
    const float det = (p0 + p1) + (p2 + p3);
#ifdef PS_FAST_EXACT
    {
        // twelve quotients with one denominator: one reciprocal inside the window (a rigid model has det ~ 1 and rotation /
        // translation cofactors far above 2^-40 unless an entry is exactly zero: then the ordinary divisions run)
        float lo = fminf(fminf(ps_abs(r00), ps_abs(r01)), ps_abs(r02));
        lo = fminf(fminf(lo, ps_abs(r03)), ps_abs(r10));
        lo = fminf(fminf(lo, ps_abs(r11)), ps_abs(r12));
        lo = fminf(fminf(lo, ps_abs(r13)), ps_abs(r20));
        lo = fminf(fminf(lo, ps_abs(r21)), ps_abs(r22));
        lo = fminf(fminf(lo, ps_abs(r23)), ps_abs(det));
        float hi = fmaxf(fmaxf(ps_abs(r00), ps_abs(r01)), ps_abs(r02));
        hi = fmaxf(fmaxf(hi, ps_abs(r03)), ps_abs(r10));
        hi = fmaxf(fmaxf(hi, ps_abs(r11)), ps_abs(r12));
        hi = fmaxf(fmaxf(hi, ps_abs(r13)), ps_abs(r20));
        hi = fmaxf(fmaxf(hi, ps_abs(r21)), ps_abs(r22));
        hi = fmaxf(fmaxf(hi, ps_abs(r23)), ps_abs(det));
        if (wave_every(lo >= kExactDivLo && hi <= kExactDivHi)) {
            const float q = rcp_refined(det);
            Inv.R[0][0] = div_with(r00, det, q); Inv.R[0][1] = div_with(r01, det, q); Inv.R[0][2] = div_with(r02, det, q);
            Inv.t[0] = div_with(r03, det, q);
            Inv.R[1][0] = div_with(r10, det, q); Inv.R[1][1] = div_with(r11, det, q); Inv.R[1][2] = div_with(r12, det, q);
            Inv.t[1] = div_with(r13, det, q);
            Inv.R[2][0] = div_with(r20, det, q); Inv.R[2][1] = div_with(r21, det, q); Inv.R[2][2] = div_with(r22, det, q);
            Inv.t[2] = div_with(r23, det, q);
            return;
        }
    }
#endif
    Inv.R[0][0] = r00 / det; Inv.R[0][1] = r01 / det; Inv.R[0][2] = r02 / det; Inv.t[0] = r03 / det;
    Inv.R[1][0] = r10 / det; Inv.R[1][1] = r11 / det; Inv.R[1][2] = r12 / det; Inv.t[1] = r13 / det;
    Inv.R[2][0] = r20 / det; Inv.R[2][1] = r21 / det; Inv.R[2][2] = r22 / det; Inv.t[2] = r23 / det;
}

// R*p + t with Eigen 3.3 fixed-size evaluation order a0 + (a1 + a2) (RANSAC.cpp:266,346-348).
PS_HD void xform(const Rigid &M, float x, float y, float z, float &ox, float &oy, float &oz)
{
    ox = (M.R[0][0] * x + (M.R[0][1] * y + M.R[0][2] * z)) + M.t[0];
    oy = (M.R[1][0] * x + (M.R[1][1] * y + M.R[1][2] * z)) + M.t[1];
    oz = (M.R[2][0] * x + (M.R[2][1] * y + M.R[2][2] * z)) + M.t[2];
}

// RGBD::point3Dto2D, reference src/RGBD/RGBD.cpp:92-98: multiply, divide, add.
PS_HD void project(float x, float y, float z, float fx, float fy, float cx, float cy, float &u, float &v)
{
    u = x * fx / z + cx;
    v = y * fy / z + cy;
}

// ---- sample stream (replaces srand(time(0)) + rand() % M, RANSAC.cpp:13,180-205; USAC.h:562-579) ----
PS_HD uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
PS_HD uint32_t draw31(uint64_t seed, uint32_t h, uint32_t j)
{
    return (uint32_t)(mix64(seed ^ mix64(((uint64_t)h << 8) | (uint64_t)j)) >> 33);
}
// Three distinct indices in [0, M): draw % M, redraw on a repeat (seeded stream), or
// raw % M with a repeat moved to the next free index (explicit stream).
PS_HD void sample_triplet(uint64_t seed, const uint32_t *raw, uint32_t h, uint32_t M, uint32_t (&idx)[3])
{
    if (raw) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint32_t v = raw[3 * (size_t)h + j] % M;
            for (;;) {
                bool rep = false;
                for (int i = 0; i < j; ++i) rep |= (idx[i] == v);
                if (!rep) break;
                v = (v + 1) % M;
            }
            idx[j] = v;
        }
        return;
    }
    int count = 0;
    idx[0] = idx[1] = idx[2] = 0xFFFFFFFFu;
    for (uint32_t j = 0; count < 3 && j < 256; ++j) {
        uint32_t v = draw31(seed, h, j) % M;
        bool rep = (count > 0 && idx[0] == v) || (count > 1 && idx[1] == v);
        if (!rep) {
            if (count == 0) idx[0] = v;
            else if (count == 1) idx[1] = v;
            else idx[2] = v;
            ++count;
        }
    }
    for (uint32_t v = 0; count < 3; ++v) {
        bool rep = (count > 0 && idx[0] == v) || (count > 1 && idx[1] == v);
        if (!rep) {
            if (count == 0) idx[0] = v;
            else if (count == 1) idx[1] = v;
            else idx[2] = v;
            ++count;
        }
    }
}

// ---- exact square-domain thresholds -------------------------------------------------------
// The reference compares  (double) fl32(sqrt(s)) < thr  (Eigen norm(), RANSAC.cpp:272) and
// sqrt((double)dx*dx + (double)dy*dy) < thr (cv::norm, RANSAC.cpp:361-366).  sqrt is correctly
// rounded and monotone, so each test equals  s < B  for the smallest B whose root reaches thr.
PS_HD float next_up_pos(float x)
{
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    ++b;
    __builtin_memcpy(&x, &b, 4);
    return x;
}
PS_HD float next_down_pos(float x)
{
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    --b;
    __builtin_memcpy(&x, &b, 4);
    return x;
}
PS_HD float sq_bound_f32(double thr)
{
    if (!(thr > 0.0)) return 0.0f;                    // also NaN: nothing passes
    if (thr > 1.8446742974197924e19) return __builtin_inff(); // > sqrtf(FLT_MAX): every finite s passes
    float x = (float)(thr * thr);
    if (!(x > 0.0f)) x = 1.401298464324817e-45f;
    while (x > 1.401298464324817e-45f && (double)sqrtf(next_down_pos(x)) >= thr) x = next_down_pos(x);
    while ((double)sqrtf(x) < thr) x = next_up_pos(x);
    return x;
}

// depth filter of RANSAC.cpp:65-74 (float promoted to double against 0.1 and 6)
PS_HD bool depth_ok(float x, float y, float z)
{
    if (x != x || y != y || z != z) return false;
    if ((double)z < 0.1 || (double)z > 6.0) return false;
    return true;
}

} // namespace psdev
