/* putslam_oracle.c -- CPU restatement of PUTSLAM's Matcher -> RANSAC/USAC -> Kabsch path.
 *
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline); see putslam_oracle.h.
 * PARITY UNPINNED for everything that goes through OpenCV 3.x / Eigen 3.3 (the matcher, Umeyama / JacobiSVD, the inverse,
 * the metrics: restated from the published algorithms -- neither library is available to compile the reference here --
 * and checked against analytic KATs, float64 numpy and an independent emulation).
 * PINNED to the reference's own code: row A11's stopping rule, main loop and sampler (po_usac_stopping, UsacLoop /
 * po_usac_replay, po_sample_triplet's seeded rule) -- include/putslam/USAC/USAC.h compiles from its own sources and
 * oracle/ref_usac drives it; its answers are tests/golden/ref_usac.npz (tests/test_ref_usac.py).
 *
 * Citations are file:line under the reference tree (LRMPUT/PUTSLAM).
 * Build: gcc -O3 -march=native -ffp-contract=off -fno-fast-math (no FMA contraction: the
 * reference is an x86-64 SSE2 scalar build, CMakeLists.txt:23,147).
 */
#include "putslam_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * A1  cv::BFMatcher(NORM_HAMMING, crossCheck = true).match(query = prev, train = cur)
 *     call site: src/Matcher/matcherOpenCV.cpp:198-206 (matcher built at :100-105)
 * OpenCV 3.x semantics (BFMatcher::knnMatchImpl + cv::batchDistance with crosscheck, K = 1):
 *   1. for every train row t: nn(t) = argmin_q ham(t,q), q ascending, strict '<'
 *   2. dist[q] = INT_MAX, idx[q] = -1; for t ascending: q = nn(t); if d(t) < dist[q]: take it
 *   3. emit DMatch(q, idx[q], 0, (float)dist[q]) for q ascending with idx[q] >= 0
 * ------------------------------------------------------------------------------------------ */
int po_hamming256(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/* SIMD form of the same sweep for the TIMED baseline: OpenCV's hal::normHamming is vectorised (universal intrinsics:
 * per-nibble lookup + byte sums on AVX2, VPOPCNT where the CPU has it), so a scalar popcnt loop would not be the
 * "most generous to the CPU" figure BASELINE.md promises.  Distances are integers: the result is identical to the
 * scalar path by construction (asserted in tests/test_oracle_kat.py::test_simd_matcher_equals_scalar). */
#if defined(__AVX2__)
#include <immintrin.h>
#define PO_HAVE_SIMD 1
static inline int ham256_simd(__m256i t, const uint8_t *q)
{
    __m256i x = _mm256_xor_si256(t, _mm256_loadu_si256((const __m256i *)q));
#if defined(__AVX512VPOPCNTDQ__) && defined(__AVX512VL__)
    __m256i c = _mm256_popcnt_epi64(x);
#else
    const __m256i lut = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4, 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3,
                                         2, 3, 3, 4);
    const __m256i mask = _mm256_set1_epi8(0x0F);
    __m256i lo = _mm256_and_si256(x, mask), hi = _mm256_and_si256(_mm256_srli_epi16(x, 4), mask);
    __m256i c = _mm256_sad_epu8(_mm256_add_epi8(_mm256_shuffle_epi8(lut, lo), _mm256_shuffle_epi8(lut, hi)),
                                _mm256_setzero_si256());
#endif
    __m128i s2 = _mm_add_epi64(_mm256_castsi256_si128(c), _mm256_extracti128_si256(c, 1));
    return (int)(_mm_cvtsi128_si64(s2) + _mm_extract_epi64(s2, 1));
}
int po_matcher_simd_kind(void)
{
#if defined(__AVX512VPOPCNTDQ__) && defined(__AVX512VL__)
    return 2; /* VPOPCNTQ on 256-bit vectors */
#else
    return 1; /* AVX2 nibble lookup + psadbw */
#endif
}
#else
#define PO_HAVE_SIMD 0
int po_matcher_simd_kind(void) { return 0; }
#endif

static int g_matcher_simd = PO_HAVE_SIMD; /* which sweep po_match_hamming256 runs (po_set_matcher_simd) */
void po_set_matcher_simd(int on) { g_matcher_simd = (on && PO_HAVE_SIMD) ? 1 : 0; }
int po_get_matcher_simd(void) { return g_matcher_simd; }

static void nearest_query(const uint8_t *tr, const uint8_t *query, int nq, size_t qstep, int simd, int *bestOut,
                          int *bqOut)
{
    int best = INT_MAX, bq = -1;
#if PO_HAVE_SIMD
    if (simd) {
        const __m256i t = _mm256_loadu_si256((const __m256i *)tr);
        for (int q = 0; q < nq; ++q) {
            int d = ham256_simd(t, query + (size_t)q * qstep);
            if (d < best) {
                best = d;
                bq = q;
            }
        }
        *bestOut = best;
        *bqOut = bq;
        return;
    }
#else
    (void)simd;
#endif
    for (int q = 0; q < nq; ++q) {
        int d = po_hamming256(tr, query + (size_t)q * qstep);
        if (d < best) {
            best = d;
            bq = q;
        }
    }
    *bestOut = best;
    *bqOut = bq;
}

int po_match_hamming256(const uint8_t *query, int nq, size_t qstep, const uint8_t *train, int nt,
                        size_t tstep, PsDMatch *out, int *nout)
{
    *nout = 0;
    if (nq <= 0 || nt <= 0) return 0;
    int *dist = (int *)malloc(sizeof(int) * (size_t)nq);
    int *idx = (int *)malloc(sizeof(int) * (size_t)nq);
    if (!dist || !idx) {
        free(dist);
        free(idx);
        return -1;
    }
    for (int q = 0; q < nq; ++q) {
        dist[q] = INT_MAX;
        idx[q] = -1;
    }
    const int simd = g_matcher_simd;
    for (int t = 0; t < nt; ++t) {
        int best, bq;
        nearest_query(train + (size_t)t * tstep, query, nq, qstep, simd, &best, &bq);
        if (bq >= 0 && best < dist[bq]) {
            dist[bq] = best;
            idx[bq] = t;
        }
    }
    int n = 0;
    for (int q = 0; q < nq; ++q)
        if (idx[q] >= 0) {
            out[n].queryIdx = q;
            out[n].trainIdx = idx[q];
            out[n].imgIdx = 0;
            out[n].distance = (float)dist[q];
            ++n;
        }
    *nout = n;
    free(dist);
    free(idx);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A3  RGBD helpers, src/RGBD/RGBD.cpp
 * ------------------------------------------------------------------------------------------ */
int po_round_size(double x, int size) /* RGBD.cpp:10-16, including the size (not size-1) clamp */
{
    if (x < 0)
        x = 0;
    else if (x > size - 1)
        x = size;
    return (int)round(x);
}

void po_keypoints2Dto3D(const float *xy, int n, const uint16_t *depth, int rows, int cols,
                        size_t depthStep, const float *K, double depthImageScale, float *out)
{
    for (int i = 0; i < n; ++i) { /* RGBD.cpp:47-65 */
        float fx = xy[2 * i], fy = xy[2 * i + 1];
        int uR = po_round_size(fx, cols);
        int vR = po_round_size(fy, rows);
        /* cv::Mat::at(v,u) is plain pointer arithmetic: u == cols walks into the next row.
         * Reads past the last row are undefined in the reference; they yield depth 0 here. */
        size_t off = (size_t)vR * depthStep + (size_t)uR * 2;
        uint16_t dv = 0;
        if (off + 2 <= (size_t)rows * depthStep) memcpy(&dv, (const uint8_t *)depth + off, 2);
        float Z = (float)(((double)dv) / depthImageScale);
        float u = (fx - K[2]) / K[0];
        float v = (fy - K[5]) / K[4];
        out[3 * i] = u * Z;
        out[3 * i + 1] = v * Z;
        out[3 * i + 2] = Z;
    }
}

static void project_pt(const float *p, const float *K, float *u, float *v) /* RGBD.cpp:92-98 */
{
    *u = p[0] * K[0] / p[2] + K[2];
    *v = p[1] * K[4] / p[2] + K[5];
}

void po_points3Dto2D(const float *xyz, int n, const float *K, float *uv)
{
    for (int i = 0; i < n; ++i) project_pt(xyz + 3 * i, K, &uv[2 * i], &uv[2 * i + 1]);
}

/* ------------------------------------------------------------------------------------------
 * 3x3 Jacobi SVD, float and double instances
 * ------------------------------------------------------------------------------------------ */
#define REAL float
#define REAL_MIN FLT_MIN
#define REAL_EPS FLT_EPSILON
#define REAL_SQRT sqrtf
#define REAL_ABS fabsf
#define FN(n) n##_f32
#include "po_svd.inc"
#undef REAL
#undef REAL_MIN
#undef REAL_EPS
#undef REAL_SQRT
#undef REAL_ABS
#undef FN

#define REAL double
#define REAL_MIN DBL_MIN
#define REAL_EPS DBL_EPSILON
#define REAL_SQRT sqrt
#define REAL_ABS fabs
#define FN(n) n##_f64
#include "po_svd.inc"
#undef REAL
#undef REAL_MIN
#undef REAL_EPS
#undef REAL_SQRT
#undef REAL_ABS
#undef FN

void po_jacobi_svd3_f32(const float *A, float *U, float *S, float *V) { jacobi_svd3_f32(A, U, S, V); }
void po_jacobi_svd3_f64(const double *A, double *U, double *S, double *V) { jacobi_svd3_f64(A, U, S, V); }

/* ------------------------------------------------------------------------------------------
 * A7  RANSAC::computeTransformationModel (RANSAC.cpp:207-244) = Eigen::umeyama(src = current,
 *     dst = previous, with_scaling = false) on dynamic-size float matrices, then isnan(T(0,0)).
 * Eigen 3.3 Umeyama: mean = rowwise sum * (1/n); demean; sigma = (1/n) * dst_dem * src_dem^T;
 * JacobiSVD(sigma, FullU|FullV); S = diag(1,1,-1) iff det(U)*det(V) < 0; R = U*S*V^T (dynamic
 * 3x3 lazy products: sequential inner sums); t = dst_mean; t -= R*src_mean (gemv, column by column).
 *
 * Summation order over the k points (the one place where it is a free choice, because Eigen's
 * own order for k >= ~14 is its blocked GEMM): 64 strided partial sums, partial[i % 64] taken
 * in ascending i, then a binary tree with strides 1,2,4,...,32.  For k = 3 this IS Eigen's
 * sequential ((a+b)+c).  The device kernel uses the identical order (one wavefront).
 * ------------------------------------------------------------------------------------------ */
static float canon_reduce64(float *p)
{
    for (int o = 1; o < 64; o <<= 1)
        for (int l = 0; l < 64; l += 2 * o) p[l] = p[l] + p[l + o];
    return p[0];
}

static float det3_f32(const float *M) /* row-major; only its sign is used */
{
    return (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void set_identity4(float *T)
{
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

int po_umeyama_f32(const float *src, const float *dst, int k, float *T)
{
    float part[64];
    float sm[3], dm[3];
    const float one_over_n = 1.0f / (float)k;
    for (int c = 0; c < 3; ++c) {
        for (int l = 0; l < 64; ++l) part[l] = 0.0f;
        for (int i = 0; i < k; ++i) part[i & 63] = part[i & 63] + src[3 * i + c];
        sm[c] = canon_reduce64(part) * one_over_n;
        for (int l = 0; l < 64; ++l) part[l] = 0.0f;
        for (int i = 0; i < k; ++i) part[i & 63] = part[i & 63] + dst[3 * i + c];
        dm[c] = canon_reduce64(part) * one_over_n;
    }
    float sigma[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            for (int l = 0; l < 64; ++l) part[l] = 0.0f;
            for (int i = 0; i < k; ++i) {
                float dd = dst[3 * i + r] - dm[r];
                float ss = src[3 * i + c] - sm[c];
                part[i & 63] = part[i & 63] + dd * ss;
            }
            sigma[3 * r + c] = one_over_n * canon_reduce64(part);
        }
    float U[9], S[3], V[9];
    jacobi_svd3_f32(sigma, U, S, V);
    float s2 = (det3_f32(U) * det3_f32(V) < 0.0f) ? -1.0f : 1.0f;
    float R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = (U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1]) + (U[3 * i + 2] * s2) * V[3 * j + 2];
    float t[3];
    for (int i = 0; i < 3; ++i)
        t[i] = ((dm[i] - R[3 * i] * sm[0]) - R[3 * i + 1] * sm[1]) - R[3 * i + 2] * sm[2];
    if (isnan(R[0])) { /* RANSAC.cpp:239-242 */
        set_identity4(T);
        return 0;
    }
    /* column-major 4x4 */
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];
        T[12 + i] = t[i];
        T[4 * i + 3] = 0.0f;
    }
    T[15] = 1.0f;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Eigen Matrix4f::inverse(), generic cofactor path (RANSAC.cpp:337-338,386-387).
 * ------------------------------------------------------------------------------------------ */
#define M4(m, r, c) ((m)[4 * (c) + (r)])
static float det3_helper(const float *m, int i1, int i2, int i3, int j1, int j2, int j3)
{
    return M4(m, i1, j1) * (M4(m, i2, j2) * M4(m, i3, j3) - M4(m, i2, j3) * M4(m, i3, j2));
}
static float cofactor4(const float *m, int i, int j)
{
    int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4;
    int j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
    return (det3_helper(m, i1, i2, i3, j1, j2, j3) + det3_helper(m, i2, i3, i1, j1, j2, j3)) +
           det3_helper(m, i3, i1, i2, j1, j2, j3);
}
void po_inverse4_f32(const float *T, float *R)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float c = cofactor4(T, i, j);
            M4(R, j, i) = ((i + j) & 1) ? -c : c;
        }
    float p0 = M4(T, 0, 0) * M4(R, 0, 0), p1 = M4(T, 1, 0) * M4(R, 0, 1);
    float p2 = M4(T, 2, 0) * M4(R, 0, 2), p3 = M4(T, 3, 0) * M4(R, 0, 3);
    float det = (p0 + p1) + (p2 + p3);
    for (int i = 0; i < 16; ++i) R[i] = R[i] / det;
}

/* ------------------------------------------------------------------------------------------
 * A8  inlier metrics, RANSAC.cpp:251-281 (Euclid / adaptive), :325-375 (reprojection),
 *     :377-436 (both).  Fixed-size Eigen 3.3 products: a 3-term sum is a0 + (a1 + a2).
 * ------------------------------------------------------------------------------------------ */
static void xform(const float *T, const float *p, float *o) /* R*p + t, T column-major 4x4 */
{
    for (int i = 0; i < 3; ++i)
        o[i] = (M4(T, i, 0) * p[0] + (M4(T, i, 1) * p[1] + M4(T, i, 2) * p[2])) + M4(T, i, 3);
}
static float norm3(const float *a, const float *b)
{
    float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return sqrtf(d0 * d0 + (d1 * d1 + d2 * d2));
}
static double cvnorm2(float ax, float ay, float bx, float by) /* cv::norm(Point2f a - b) */
{
    float dx = ax - bx, dy = ay - by;
    return sqrt((double)dx * dx + (double)dy * dy);
}

int po_is_inlier(int mode, const float *T, const float *Tinv, const float *K, const float *pp,
                 const float *cp, double thrE, double thrR)
{
    float estOld[3];
    xform(T, cp, estOld);
    if (mode == PS_EUCLIDEAN_ERROR || mode == PS_ADAPTIVE_ERROR) {
        double thr = thrE;
        if (mode == PS_ADAPTIVE_ERROR) thr *= pp[2];
        return norm3(estOld, pp) < thr;
    }
    if (mode == PS_REPROJECTION_ERROR || mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
        float estNew[3];
        xform(Tinv, pp, estNew);
        float pnu, pnv, rnu, rnv, pou, pov, rou, rov;
        project_pt(estNew, K, &pnu, &pnv);
        project_pt(cp, K, &rnu, &rnv);
        project_pt(estOld, K, &pou, &pov);
        project_pt(pp, K, &rou, &rov);
        double e0 = cvnorm2(pnu, pnv, rnu, rnv);
        double e1 = cvnorm2(pou, pov, rou, rov);
        if (mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
            double e3 = norm3(estOld, pp);
            return e3 < thrE && e0 < thrR && e1 < thrR;
        }
        return e0 < thrR && e1 < thrR;
    }
    return 0; /* Mahalanobis (dead, RANSAC.cpp:301-303) and unknown modes (RANSAC.cpp:134-135) score 0 */
}

/* The error VALUES behind po_is_inlier, as the reference computes them (RANSAC.cpp:266-272,346-366): err[0] = the float
 * Euclidean norm (as double), err[1] = cv::norm(predictedNew - realNew), err[2] = cv::norm(predictedOld - realOld).
 * For the directed band-edge tests: a threshold set to one of these values (or its neighbours) puts that evaluation
 * exactly on the decision boundary. */
void po_eval_errors(const float *T, const float *Tinv, const float *K, const float *pp, const float *cp, double *err)
{
    float estOld[3], estNew[3];
    xform(T, cp, estOld);
    err[0] = (double)norm3(estOld, pp);
    xform(Tinv, pp, estNew);
    float pnu, pnv, rnu, rnv, pou, pov, rou, rov;
    project_pt(estNew, K, &pnu, &pnv);
    project_pt(cp, K, &rnu, &rnv);
    project_pt(estOld, K, &pou, &pov);
    project_pt(pp, K, &rou, &rov);
    err[1] = cvnorm2(pnu, pnv, rnu, rnv);
    err[2] = cvnorm2(pou, pov, rou, rov);
}

/* ------------------------------------------------------------------------------------------
 * A6  iteration schedule
 * ------------------------------------------------------------------------------------------ */
int po_ransac_iterations(double inlierRatio, double successProbability, int numberOfPairs)
{
    double v = log(1 - successProbability) / log(1 - pow(inlierRatio, numberOfPairs)); /* RANSAC.cpp:459-460 */
    if (!(v < 2147483647.0)) return INT_MAX; /* int(v) is UB there; also catches NaN / +inf */
    if (v < 0) return 0;
    return (int)v;
}

#define USAC_CONF 0.99
#define USAC_MAX_HYP 850000u
unsigned po_usac_stopping(unsigned numInliers, unsigned totPoints, unsigned sampleSize)
{
    double n_inliers = 1.0, n_pts = 1.0; /* USAC.h:944-971 */
    for (unsigned i = 0; i < sampleSize; ++i) {
        n_inliers *= numInliers - i;
        n_pts *= totPoints - i;
    }
    double prob_good_model = n_inliers / n_pts;
    if (prob_good_model < DBL_EPSILON) return USAC_MAX_HYP;
    if (1 - prob_good_model < DBL_EPSILON) return 1;
    double nusample_s = log(1 - USAC_CONF) / log(1 - prob_good_model);
    /* The reference returns (unsigned int) ceil(nusample_s): undefined from 2^32 on (good-model probabilities below 1.07e-9:
     * three inliers among 1777 matches and more).  What the reference's code does there -- an SSE2 x86-64 build keeps the low
     * 32 bits of the 64-bit conversion, a pseudo-random number -- is on file (tests/golden/ref_usac.npz, from the reference's
     * USAC.h itself); the build returns the cap instead, here and on the device, whatever the compiler makes of the cast. */
    const double c = ceil(nusample_s);
    if (!(c < 4294967296.0)) return USAC_MAX_HYP;
    return (unsigned)c;
}

/* The bookkeeping of USAC<T>::solve's main loop (USAC.h:299,326-329,409-414,498-509) under RANSAC_USAC's configuration
 * (USAC_wrapper.cpp:62-100): hypotheses are counted when they start; a model that could not be generated is skipped (0
 * solutions, USAC.h:376-379); a count STRICTLY above the best so far is stored and re-derives the stopping count. */
typedef struct UsacLoop {
    unsigned adaptive, hyp;
    int bestCount, best;
} UsacLoop;
static void usac_loop_init(UsacLoop *l)
{
    l->adaptive = USAC_MAX_HYP;
    l->hyp = 0;
    l->bestCount = 0;
    l->best = -1;
}
static int usac_loop_continues(const UsacLoop *l, int H) { return l->hyp < l->adaptive && l->hyp < USAC_MAX_HYP && (int)l->hyp < H; }
static int usac_loop_next(UsacLoop *l) { return (int)l->hyp++; }
static int usac_loop_result(UsacLoop *l, int i, int cnt, int M) /* 1: hypothesis i is the best so far */
{
    if (cnt <= l->bestCount) return 0;
    l->bestCount = cnt;
    l->best = i;
    l->adaptive = po_usac_stopping((unsigned)cnt, (unsigned)M, 3);
    return 1;
}
/* The loop over REPLAYED outcomes (valid[i] != 0: hypothesis i has a model, counts[i] its inlier count; hypotheses from n on have
 * a model and count 0), at most H hypotheses: what the reference's solve() does with the same outcomes. */
void po_usac_replay(const int32_t *valid, const int32_t *counts, int n, int H, int M, int32_t *iterations, int32_t *bestCount,
                    int32_t *best)
{
    UsacLoop l;
    usac_loop_init(&l);
    while (usac_loop_continues(&l, H)) {
        const int i = usac_loop_next(&l);
        if (i < n && !valid[i]) continue;
        usac_loop_result(&l, i, i < n ? counts[i] : 0, M);
    }
    *iterations = (int32_t)l.hyp;
    *bestCount = l.bestCount;
    *best = l.best;
}

/* ------------------------------------------------------------------------------------------
 * A5  sample stream.  The reference draws rand() % M after srand(time(0)) (RANSAC.cpp:13,191;
 * USAC.h:562-579) and redraws on a repeat; the build makes the stream an explicit input.
 * ------------------------------------------------------------------------------------------ */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint32_t po_draw31(uint64_t seed, uint32_t h, uint32_t j)
{
    return (uint32_t)(mix64(seed ^ mix64(((uint64_t)h << 8) | (uint64_t)j)) >> 33);
}
void po_sample_triplet(const PsRansacConfig *cfg, uint64_t seed, int h, int M, int idx[3])
{
    if (cfg->sampleIdx) {
        for (int j = 0; j < 3; ++j) {
            int v = (int)(cfg->sampleIdx[3 * (size_t)h + j] % (uint32_t)M);
            for (;;) {
                int rep = 0;
                for (int i = 0; i < j; ++i) rep |= (idx[i] == v);
                if (!rep) break;
                v = (v + 1) % M;
            }
            idx[j] = v;
        }
        return;
    }
    int count = 0;
    for (uint32_t j = 0; count < 3 && j < 256; ++j) {
        int v = (int)(po_draw31(seed, (uint32_t)h, j) % (uint32_t)M);
        int rep = 0;
        for (int i = 0; i < count; ++i) rep |= (idx[i] == v);
        if (!rep) idx[count++] = v;
    }
    for (int v = 0; count < 3; ++v) { /* unreachable for M >= 3 in practice: fill deterministically */
        int rep = 0;
        for (int i = 0; i < count; ++i) rep |= (idx[i] == v);
        if (!rep) idx[count++] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * A4 / A9 / A11  estimateTransformation
 * ------------------------------------------------------------------------------------------ */
static int depth_ok(const float *p) /* RANSAC.cpp:65-74, USAC_wrapper.cpp:41-60 */
{
    if (isnan(p[0]) || isnan(p[1]) || isnan(p[2])) return 0;
    if (p[2] < 0.1 || p[2] > 6) return 0; /* float promoted to double against 0.1 / 6 */
    return 1;
}

typedef struct {
    int n;
    int *src; /* index into the caller's match list */
} MatchList;

static int fit_sample(const float *prev, const float *cur, const PsDMatch *matches, const int *valid,
                      const int idx[3], float *T)
{
    float s[9], d[9];
    for (int j = 0; j < 3; ++j) {
        const PsDMatch *mm = &matches[valid[idx[j]]];
        memcpy(&d[3 * j], &prev[3 * (size_t)mm->queryIdx], 12);
        memcpy(&s[3 * j], &cur[3 * (size_t)mm->trainIdx], 12);
    }
    return po_umeyama_f32(s, d, 3, T);
}

static int score_all(int mode, const float *T, const float *K, const float *prev, const float *cur,
                     const PsDMatch *matches, const int *valid, int M, double thrE, double thrR,
                     uint8_t *flags)
{
    float Tinv[16];
    int need_inv = (mode == PS_REPROJECTION_ERROR || mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR);
    if (need_inv) po_inverse4_f32(T, Tinv);
    int cnt = 0;
    for (int i = 0; i < M; ++i) {
        const PsDMatch *mm = &matches[valid[i]];
        int in = po_is_inlier(mode, T, need_inv ? Tinv : NULL, K, &prev[3 * (size_t)mm->queryIdx],
                              &cur[3 * (size_t)mm->trainIdx], thrE, thrR);
        if (flags) flags[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

static const float K_ZERO[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

int po_hypothesis_counts(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                         const float *prev, int nprev, const float *cur, int ncur,
                         const PsDMatch *matches, int m, int32_t *counts, int *Mvalid)
{
    (void)nprev;
    (void)ncur;
    if (!K) K = K_ZERO;
    int *valid = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int M = 0;
    for (int i = 0; i < m; ++i)
        if (depth_ok(&prev[3 * (size_t)matches[i].queryIdx]) && depth_ok(&cur[3 * (size_t)matches[i].trainIdx]))
            valid[M++] = i;
    *Mvalid = M;
    for (int h = 0; h < cfg->numHypotheses; ++h) {
        counts[h] = 0;
        if (M < 3) continue;
        int idx[3];
        float T[16];
        po_sample_triplet(cfg, cfg->seed, h, M, idx);
        if (!fit_sample(prev, cur, matches, valid, idx, T)) continue;
        counts[h] = score_all(params->errorVersion, T, K, prev, cur, matches, valid, M,
                              params->inlierThresholdEuclidean, params->inlierThresholdReprojection, NULL);
    }
    free(valid);
    return 0;
}

/* The model of hypothesis h exactly as po_hypothesis_counts builds it (sample -> 3-point Umeyama); returns 0 when the
 * model is invalid (iteration skipped, RANSAC.cpp:107), M through *Mvalid, the sampled valid-match indices in idx3. */
int po_hypothesis_model(const PsRansacConfig *cfg, const float *prev, const float *cur, const PsDMatch *matches, int m,
                        int h, float *T, int *validOut, int *Mvalid, int *idx3)
{
    int M = 0;
    for (int i = 0; i < m; ++i)
        if (depth_ok(&prev[3 * (size_t)matches[i].queryIdx]) && depth_ok(&cur[3 * (size_t)matches[i].trainIdx]))
            validOut[M++] = i;
    *Mvalid = M;
    if (M < 3) return 0;
    po_sample_triplet(cfg, cfg->seed, h, M, idx3);
    return fit_sample(prev, cur, matches, validOut, idx3, T);
}

double po_point_inlier_ratio(const PsDMatch *inl, int ninl, const PsDMatch *all, int nall)
{
    /* RANSAC.h:56-66: |unique trainIdx of inliers| / |unique trainIdx of all matches| */
    int maxIdx = -1;
    for (int i = 0; i < nall; ++i)
        if (all[i].trainIdx > maxIdx) maxIdx = all[i].trainIdx;
    for (int i = 0; i < ninl; ++i)
        if (inl[i].trainIdx > maxIdx) maxIdx = inl[i].trainIdx;
    uint8_t *seen = (uint8_t *)calloc((size_t)maxIdx + 2, 1);
    int ua = 0, ui = 0;
    for (int i = 0; i < nall; ++i)
        if (all[i].trainIdx >= 0 && !(seen[all[i].trainIdx] & 1)) {
            seen[all[i].trainIdx] |= 1;
            ++ua;
        }
    for (int i = 0; i < ninl; ++i)
        if (inl[i].trainIdx >= 0 && !(seen[inl[i].trainIdx] & 2)) {
            seen[inl[i].trainIdx] |= 2;
            ++ui;
        }
    free(seen);
    return (double)ui / (double)ua;
}

int po_ransac_rigid3d(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const float *prev, int nprev, const float *cur, int ncur,
                      const PsDMatch *matches, int m, float *pose, PsDMatch *inliers, int *ninl,
                      uint8_t *mask, PsRansacStats *stats, int32_t *hypCounts)
{
    (void)nprev;
    (void)ncur;
    if (!K) K = K_ZERO;
    const int H = cfg->numHypotheses;
    const int mode = params->errorVersion;
    const double thrE = params->inlierThresholdEuclidean, thrR = params->inlierThresholdReprojection;
    PsRansacStats st;
    memset(&st, 0, sizeof st);
    st.numMatchesIn = m;
    st.bestHypothesis = -1;
    st.pointInlierRatio = NAN;
    set_identity4(pose);
    *ninl = 0;
    if (mask && m > 0) memset(mask, 0, (size_t)m);
    if (hypCounts)
        for (int h = 0; h < H; ++h) hypCounts[h] = -1;

    int *valid = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    uint8_t *flags = (uint8_t *)malloc((size_t)(m > 0 ? m : 1));
    uint8_t *bestFlags = (uint8_t *)calloc((size_t)(m > 0 ? m : 1), 1);
    int M = 0;
    for (int i = 0; i < m; ++i)
        if (depth_ok(&prev[3 * (size_t)matches[i].queryIdx]) && depth_ok(&cur[3 * (size_t)matches[i].trainIdx]))
            valid[M++] = i;
    st.numMatchesValid = M;

    const int usac = (cfg->estimator == PS_EST_USAC);
    const int minMatches = usac ? 8 : params->minimalNumberOfMatches; /* USAC_wrapper.cpp:120-122 / RANSAC.cpp:77-80 */
    /* M < 3: the reference's sampler would never return */
    const int run = !(M < minMatches || M < 3);

    float bestT[16];
    set_identity4(bestT);
    int bestCount = 0;
    float bestRatioF = 0.0f;
    double bestRatio = 0.0;
    int iterationsRun = 0;

    if (!run) {
        /* too few matches: identity, inliers cleared (RANSAC.cpp:77-80) */
    } else if (!usac) {
        /* RANSAC.cpp:87-150.  iterationCount starts at computeRANSACIteration(0.20) (ctor, :30). */
        int iterationCount = (cfg->estimator == PS_EST_FIXED) ? H : po_ransac_iterations(0.20, 0.98, 3);
        for (int i = 0; i < iterationCount && i < H; ++i) {
            ++iterationsRun;
            int idx[3];
            float T[16];
            po_sample_triplet(cfg, cfg->seed, i, M, idx);
            if (!fit_sample(prev, cur, matches, valid, idx, T)) {
                if (hypCounts) hypCounts[i] = 0;
                continue; /* :107 model not computed -> iteration skipped */
            }
            int cnt = score_all(mode, T, K, prev, cur, matches, valid, M, thrE, thrR, flags);
            if (hypCounts) hypCounts[i] = cnt;
            float ratio = (float)cnt / (float)M; /* :280 */
            if ((double)ratio > bestRatio) {     /* :443 strict > */
                memcpy(bestT, T, sizeof bestT);
                bestRatio = (double)ratio;
                bestRatioF = ratio;
                bestCount = cnt;
                st.bestHypothesis = i;
                memcpy(bestFlags, flags, (size_t)M);
                if (cfg->estimator != PS_EST_FIXED) {
                    int a = po_ransac_iterations(params->minimalInlierRatioThreshold, 0.98, 3);
                    int b = po_ransac_iterations(bestRatio, 0.98, 3);
                    iterationCount = a < b ? a : b; /* :450-453 */
                }
            }
        }
        /* :152-158 refit on the best inliers, then the (always Euclidean / adaptive) re-selection */
        int kk = 0;
        for (int i = 0; i < M; ++i) kk += bestFlags[i];
        float *s = (float *)malloc(sizeof(float) * 3 * (size_t)(kk > 0 ? kk : 1));
        float *d = (float *)malloc(sizeof(float) * 3 * (size_t)(kk > 0 ? kk : 1));
        int j = 0;
        for (int i = 0; i < M; ++i)
            if (bestFlags[i]) {
                const PsDMatch *mm = &matches[valid[i]];
                memcpy(&d[3 * j], &prev[3 * (size_t)mm->queryIdx], 12);
                memcpy(&s[3 * j], &cur[3 * (size_t)mm->trainIdx], 12);
                ++j;
            }
        float refT[16];
        po_umeyama_f32(s, d, kk, refT); /* NaN (e.g. kk == 0) -> identity, return value ignored (:153) */
        free(s);
        free(d);
        int refMode = (mode == PS_ADAPTIVE_ERROR) ? PS_ADAPTIVE_ERROR : PS_EUCLIDEAN_ERROR; /* :268-271 */
        int nfinal = 0;
        for (int i = 0; i < M; ++i) {
            if (!bestFlags[i]) continue;
            const PsDMatch *mm = &matches[valid[i]];
            if (po_is_inlier(refMode, refT, NULL, K, &prev[3 * (size_t)mm->queryIdx],
                             &cur[3 * (size_t)mm->trainIdx], thrE, thrR)) {
                inliers[nfinal++] = *mm;
                if (mask) mask[valid[i]] = 1;
            }
        }
        memcpy(pose, refT, sizeof refT);
        *ninl = nfinal;
        st.accepted = 1;
        if (bestRatio < params->minimalInlierRatioThreshold) { /* :161-164 */
            set_identity4(pose);
            *ninl = 0;
            if (mask) memset(mask, 0, (size_t)m);
            st.accepted = 0;
        }
    } else {
        /* USAC<T>::solve, USAC.h:326,409-414,498-509 with SAMP_UNIFORM / VERIF_STANDARD / LO_NONE: the loop's bookkeeping is
         * UsacLoop (below), the same code po_usac_replay runs against the reference's own solve() (tests/test_ref_usac.py) */
        UsacLoop lp;
        usac_loop_init(&lp);
        while (usac_loop_continues(&lp, H)) {
            int i = usac_loop_next(&lp);
            ++iterationsRun;
            int idx[3];
            float T[16];
            po_sample_triplet(cfg, cfg->seed, i, M, idx);
            if (!fit_sample(prev, cur, matches, valid, idx, T)) {
                if (hypCounts) hypCounts[i] = 0;
                continue;
            }
            int cnt = score_all(mode, T, K, prev, cur, matches, valid, M, thrE, thrR, flags);
            if (hypCounts) hypCounts[i] = cnt;
            if (usac_loop_result(&lp, i, cnt, M)) {
                bestCount = cnt;
                memcpy(bestT, T, sizeof bestT);
                memcpy(bestFlags, flags, (size_t)M);
                st.bestHypothesis = i;
                bestRatioF = (float)cnt / (float)M;
            }
        }
        int nfinal = 0;
        for (int i = 0; i < M; ++i)
            if (bestFlags[i]) {
                inliers[nfinal++] = matches[valid[i]];
                if (mask) mask[valid[i]] = 1;
            }
        *ninl = nfinal;
        memcpy(pose, bestT, sizeof bestT);
        st.accepted = 1;
        if ((double)bestRatioF < params->minimalInlierRatioThreshold) { /* USAC_wrapper.cpp:139-141: inliers kept */
            set_identity4(pose);
            st.accepted = 0;
        }
    }
    st.bestInlierCount = bestCount;
    st.bestInlierRatio = bestRatioF;
    st.iterationsRun = iterationsRun;

    st.numInliers = *ninl;
    if (m > 0) st.pointInlierRatio = po_point_inlier_ratio(inliers, *ninl, matches, m);
    if (stats) *stats = st;
    free(valid);
    free(flags);
    free(bestFlags);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A10  KabschEst::computeTransformation, src/TransformEst/kabschEst.cpp:24-68 (double)
 * ------------------------------------------------------------------------------------------ */
static double det3_lu_f64(const double *Ain) /* Eigen dynamic-size determinant = PartialPivLU */
{
    double a[9];
    memcpy(a, Ain, sizeof a);
    double det = 1.0;
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        double big = fabs(a[3 * k + k]);
        for (int r = k + 1; r < 3; ++r)
            if (fabs(a[3 * r + k]) > big) {
                big = fabs(a[3 * r + k]);
                piv = r;
            }
        if (big == 0.0) return 0.0;
        if (piv != k) {
            for (int c = 0; c < 3; ++c) {
                double t = a[3 * k + c];
                a[3 * k + c] = a[3 * piv + c];
                a[3 * piv + c] = t;
            }
            det = -det;
        }
        det *= a[3 * k + k];
        for (int r = k + 1; r < 3; ++r) {
            double f = a[3 * r + k] / a[3 * k + k];
            for (int c = k + 1; c < 3; ++c) a[3 * r + c] -= f * a[3 * k + c];
        }
    }
    return det;
}

void po_kabsch_f64(const double *A, const double *B, int n, int ld, double *T)
{
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (n == 0) return; /* :28 */
    double cA[3], cB[3];
    for (int c = 0; c < 3; ++c) { /* :31-34 col(i).mean() = sequential sum / n */
        double sa = 0, sb = 0;
        for (int i = 0; i < n; ++i) {
            sa += A[(size_t)c * ld + i];
            sb += B[(size_t)c * ld + i];
        }
        cA[c] = sa / (double)n;
        cB[c] = sb / (double)n;
    }
    double Am[9]; /* :44 A = setAnew^T * setBnew */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int i = 0; i < n; ++i) s += (A[(size_t)r * ld + i] - cA[r]) * (B[(size_t)c * ld + i] - cB[c]);
            Am[3 * r + c] = s;
        }
    double V[9], S[3], W[9]; /* :47-49  V = svd.matrixU(), W = svd.matrixV() */
    jacobi_svd3_f64(Am, V, S, W);
    double det = det3_lu_f64(Am);
    double dsg = (det != 0) ? det : 1; /* :53 */
    double d = (double)((dsg > 0) - (dsg < 0));
    double R[9]; /* :56 U = W * diag(1,1,d) * V^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = (W[3 * i] * V[3 * j] + W[3 * i + 1] * V[3 * j + 1]) + (W[3 * i + 2] * d) * V[3 * j + 2];
    for (int i = 0; i < 3; ++i) { /* :59-62 T = U*(-cA) + cB */
        double ti = (R[3 * i] * (-cA[0]) + (R[3 * i + 1] * (-cA[1]) + R[3 * i + 2] * (-cA[2]))) + cB[i];
        for (int j = 0; j < 3; ++j) T[4 * j + i] = R[3 * i + j];
        T[12 + i] = ti;
    }
}

/* ------------------------------------------------------------------------------------------
 * N2  guided map matching, Matcher::matchXYZ (src/Matcher/matcher.cpp:606-746)
 * ------------------------------------------------------------------------------------------ */
int po_predicted_level(int octave, double detDist, double curDist)
{
    const double scaleFactor = 1.2; /* matcher.h:26-28 */
    const int nLevels = 8;
    const double logScaleFactor = log(scaleFactor);
    double detLevelScaleFactor = pow(scaleFactor, octave);
    double curLevelScaleFactor = detLevelScaleFactor * detDist / curDist;
    int curLevel = (int)ceil(log(curLevelScaleFactor) / logScaleFactor); /* :648,690 */
    if (curLevel < 0) curLevel = 0;
    if (curLevel > nLevels - 1) curLevel = nLevels - 1;
    return curLevel;
}

int po_satdiff_hamming256(const uint8_t *a, const uint8_t *b)
{
    int v = 0;
    for (int k = 0; k < 32; ++k) {
        int d = (int)a[k] - (int)b[k];
        if (d < 0) d = 0; /* cv::Mat subtraction of CV_8U saturates */
        v += __builtin_popcount((unsigned)d);
    }
    return v;
}

int po_match_xyz(const float *mapPos, const uint8_t *mapDesc, size_t mapStep, const int32_t *mapLevel, int nmap,
                 const float *curPos, const uint8_t *curDesc, size_t curStep, const int32_t *curLevel, int ncur,
                 double sphereRadius, double acceptRatio, PsDMatch *out, int cap, int *nout)
{
    int n = 0;
    int *cand = (int *)malloc(sizeof(int) * (size_t)(ncur > 0 ? ncur : 1));
    for (int j = 0; j < nmap; ++j) {
        const float *tmp = &mapPos[3 * (size_t)j];
        const int curLevelJ = mapLevel[j];
        int nc = 0;
        for (int i = 0; i < ncur; ++i) { /* :699-711 */
            float nrm = norm3(tmp, &curPos[3 * (size_t)i]);
            int scaleCheck = (curLevel[i] - 1 <= curLevelJ) && (curLevelJ <= curLevel[i] + 1);
            int posCheck = nrm < sphereRadius;
            if (posCheck && scaleCheck) cand[nc++] = i;
        }
        int bestId = -1;
        float bestVal = 99999;
        for (int c = 0; c < nc; ++c) { /* :714-727 */
            float value = (float)po_satdiff_hamming256(mapDesc + (size_t)j * mapStep, curDesc + (size_t)cand[c] * curStep);
            if (value < bestVal || bestId == -1) {
                bestVal = value;
                bestId = cand[c];
            }
        }
        for (int c = 0; c < nc; ++c) { /* :734-746 */
            float value = (float)po_satdiff_hamming256(mapDesc + (size_t)j * mapStep, curDesc + (size_t)cand[c] * curStep);
            if (acceptRatio * value <= bestVal) {
                if (n < cap) {
                    out[n].queryIdx = j;
                    out[n].trainIdx = cand[c];
                    out[n].imgIdx = -1; /* default-constructed cv::DMatch */
                    out[n].distance = value;
                }
                ++n;
            }
        }
    }
    free(cand);
    *nout = n;
    return n <= cap ? 0 : 1;
}

/* ------------------------------------------------------------------------------------------
 * N4  RGBD::removeImageDistortion, src/RGBD/RGBD.cpp:254-314
 * ------------------------------------------------------------------------------------------ */
void po_remove_image_distortion(const float *xy, int n, const float *K, const double *dist5, float *out)
{
    double k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) k[i] = dist5[i];
    const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; ++i) {
        double x = (double)xy[2 * i], y = (double)xy[2 * i + 1];
        double x0 = x = (x - cx) * ifx;
        double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; ++j) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        float xn = (float)x, yn = (float)y;             /* dst is CV_32FC2 */
        out[2 * i] = xn * K[0] + K[2];                  /* RGBD.cpp:276-282 */
        out[2 * i + 1] = yn * K[4] + K[5];
    }
}

/* ------------------------------------------------------------------------------------------
 * A2  Matcher::match data flow over a batch of pairs (matcher.cpp:470-515), host memory.
 * ------------------------------------------------------------------------------------------ */
int po_vo_pairs(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                const PsFrameSet *fs, const int32_t *pairs, int P, const PsPairResults *out, int threads)
{
    int rc = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int p = 0; p < P; ++p) {
        int fa = pairs[2 * p], fb = pairs[2 * p + 1];
        size_t cap = (size_t)fs->maxKpts;
        const uint8_t *qd = fs->desc + (size_t)fa * cap * 32;
        const uint8_t *td = fs->desc + (size_t)fb * cap * 32;
        const float *pp = fs->pts + (size_t)fa * cap * 3;
        const float *cp = fs->pts + (size_t)fb * cap * 3;
        PsDMatch *mm = out->matches + (size_t)p * cap;
        int nm = 0;
        if (po_match_hamming256(qd, fs->nkpts[fa], 32, td, fs->nkpts[fb], 32, mm, &nm)) rc = -1;
        out->numMatches[p] = nm;
        PsRansacConfig c = *cfg;
        c.seed = cfg->seed + (uint64_t)p;
        PsDMatch *inl = (PsDMatch *)malloc(sizeof(PsDMatch) * (cap ? cap : 1));
        int ninl = 0;
        po_ransac_rigid3d(params, &c, K, pp, fs->nkpts[fa], cp, fs->nkpts[fb], mm, nm,
                          out->pose + (size_t)p * 16, inl, &ninl, out->inlierMask + (size_t)p * cap,
                          &out->stats[p], NULL);
        free(inl);
    }
    return rc;
}
