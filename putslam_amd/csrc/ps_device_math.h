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
    T w = ps_sqrt(tau * tau + T(1));
    T t;
    if (tau > T(0))
        t = T(1) / (tau + w);
    else
        t = T(1) / (tau - w);
    T sign_t = t > T(0) ? T(1) : T(-1);
    T n = T(1) / ps_sqrt(t * t + T(1));
    s = (((-sign_t) * (y / ps_abs(y))) * ps_abs(t)) * n;
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
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            W[i][j] = A[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? T(1) : T(0);
        }
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
                    T tmp = ps_sqrt(T(1) + u * u);
                    s1 = T(1) / tmp;
                    c1 = u / tmp;
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
