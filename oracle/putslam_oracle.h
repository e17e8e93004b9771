/* putslam_oracle.h -- CPU restatement of PUTSLAM's Matcher -> RANSAC/USAC -> Kabsch path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under putslam_amd/ or include/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in OpenCV
 * (cv::BFMatcher, unpinned, CMakeLists.txt:118) and Eigen (umeyama / JacobiSVD,
 * unpinned, CMakeLists.txt:145; needs >= 3.3 because transformEst.h:45 uses
 * Eigen::Index).  Neither library is vendored under the reference tree nor
 * installed in the build image, the reference has no tests and its only recorded
 * vectors (resources/USAC/ .features/.matches/.ransac files, demoUSAC.cpp:252-286) are
 * absent.  This file therefore restates the PUBLISHED algorithms of OpenCV 3.x
 * BFMatcher cross-check and Eigen 3.3 Umeyama/JacobiSVD/inverse, anchored on the
 * reference's call sites, and is checked against analytic known answers and a
 * float64 numpy SVD (tests/test_oracle_*.py).  Agreement with a real
 * OpenCV/Eigen build is a tolerance claim, not a bit claim.
 * PINNED (the exception): the USAC template library include/putslam/USAC/USAC.h compiles from its own
 * sources, so row A11's stopping rule, main loop and sampler are checked against the reference's code itself
 * (oracle/ref_usac -> oracle/_ref/usac_harness -> tests/golden/ref_usac.npz, tests/test_ref_usac.py).
 *
 * POD types are shared with the public C ABI (include/putslam_hip.h).
 */
#ifndef PUTSLAM_ORACLE_H_
#define PUTSLAM_ORACLE_H_

#include "../include/putslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* popcount(a XOR b) over 32 bytes: cv::NORM_HAMMING on CV_8U rows (matcherOpenCV.cpp:105). */
int po_hamming256(const uint8_t *a, const uint8_t *b);

/* cv::BFMatcher(NORM_HAMMING, crossCheck=true).match(query, train)
 * as called from MatcherOpenCV::performMatching (matcherOpenCV.cpp:198-206). */
int po_match_hamming256(const uint8_t *query, int nq, size_t qstep,
                        const uint8_t *train, int nt, size_t tstep,
                        PsDMatch *out, int *nout);

/* RGBD::roundSize (RGBD.cpp:10-16), point2Dto3D (:47-65), keypoints2Dto3D (:30-45), point3Dto2D (:92-98). */
/* timed-baseline switch: SIMD popcount sweep (OpenCV's normHamming is vectorised) vs the scalar popcnt loop; identical results */
void po_set_matcher_simd(int on);
int po_get_matcher_simd(void);
int po_matcher_simd_kind(void); /* 0 none (scalar only), 1 AVX2 nibble lookup, 2 AVX-512 VPOPCNTQ */

int po_round_size(double x, int size);
void po_keypoints2Dto3D(const float *xy, int n, const uint16_t *depth, int rows, int cols,
                        size_t depthStep, const float *K, double depthImageScale, float *out);
void po_points3Dto2D(const float *xyz, int n, const float *K, float *uv);

/* Eigen::umeyama(src, dst, false) in float + isnan(T(0,0)) (RANSAC.cpp:207-244).
 * src/dst: k x 3 floats. T: 16 floats column-major. Returns 1 if valid, 0 => T = identity. */
int po_umeyama_f32(const float *src, const float *dst, int k, float *T);

/* 3x3 Jacobi SVD restatements (exposed for tests). Row-major 3x3 in, U,V row-major, s[3] descending. */
void po_jacobi_svd3_f32(const float *A, float *U, float *S, float *V);
void po_jacobi_svd3_f64(const double *A, double *U, double *S, double *V);

/* Eigen general 4x4 inverse, generic (non-SSE) cofactor path, column-major in/out (RANSAC.cpp:337-338). */
void po_inverse4_f32(const float *T, float *Tinv);

/* RANSAC::computeRANSACIteration (RANSAC.cpp:457-461), int conversion saturated instead of UB. */
int po_ransac_iterations(double inlierRatio, double successProbability, int numberOfPairs);
/* USAC<T>::updateStandardStopping (USAC.h:944-971) with conf 0.99 / maxHypotheses 850000 (USAC_wrapper.cpp:66,70). */
unsigned po_usac_stopping(unsigned numInliers, unsigned totPoints, unsigned sampleSize);

/* The sample stream shared by oracle and device (replaces srand(time(0)) + rand()%M, RANSAC.cpp:13,180-205). */
void po_usac_replay(const int32_t *valid, const int32_t *counts, int n, int H, int M, int32_t *iterations, int32_t *bestCount,
                    int32_t *best);
uint32_t po_draw31(uint64_t seed, uint32_t h, uint32_t j);
void po_sample_triplet(const PsRansacConfig *cfg, uint64_t seed, int h, int M, int idx[3]);

/* Inlier test of ONE match under a 4x4 column-major model (RANSAC.cpp:251-281,325-436).
 * Tinv may be NULL for modes 0/4. Returns 0/1. */
int po_is_inlier(int mode, const float *T, const float *Tinv, const float *K,
                 const float *prevPt, const float *curPt, double thrEuclid, double thrReproj);

/* RANSAC::estimateTransformation (RANSAC.cpp:50-174) / RANSAC_USAC::estimateTransformation
 * (USAC_wrapper.cpp:104-151) / fixed-H variant, selected by cfg->estimator.
 * hypCounts (may be NULL): receives the inlier count of every hypothesis that the
 * sequential loop evaluated (others -1); length cfg->numHypotheses. */
/* error values behind po_is_inlier (Euclid norm, reprojection new, reprojection old) and the model of hypothesis h:
 * used by the directed band-edge tests */
void po_eval_errors(const float *T, const float *Tinv, const float *K, const float *pp, const float *cp, double *err);
int po_hypothesis_model(const PsRansacConfig *cfg, const float *prev, const float *cur, const PsDMatch *matches, int m,
                        int h, float *T, int *validOut, int *Mvalid, int *idx3);

int po_ransac_rigid3d(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const float *prev, int nprev, const float *cur, int ncur,
                      const PsDMatch *matches, int m,
                      float *pose, PsDMatch *inliers, int *ninl, uint8_t *mask,
                      PsRansacStats *stats, int32_t *hypCounts);

/* Score EVERY hypothesis 0..H-1 (no early stop): counts[h] = inliers, 0 for an invalid model. */
int po_hypothesis_counts(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                         const float *prev, int nprev, const float *cur, int ncur,
                         const PsDMatch *matches, int m, int32_t *counts, int *Mvalid);

/* RANSAC::pointInlierRatio (RANSAC.h:56-66). */
double po_point_inlier_ratio(const PsDMatch *inliers, int ninl, const PsDMatch *all, int nall);

/* KabschEst::computeTransformation (kabschEst.cpp:24-68). A,B n x 3 column-major (ld), T 4x4 column-major. */
void po_kabsch_f64(const double *A, const double *B, int n, int ld, double *T);

/* N2: guided map matching core of Matcher::matchXYZ, src/Matcher/matcher.cpp:606-746.
 * Predicted pyramid level (matcher.cpp:639-652 for keypoints, :681-692 for map features). */
int po_predicted_level(int octave, double detDist, double curDist);
/* value = (float)cv::norm(a - b, NORM_HAMMING) on CV_8U rows (matcher.cpp:719-721): the subtraction
 * SATURATES per byte, so this is popcount(max(a_k - b_k, 0)) summed over 32 bytes, not XOR Hamming. */
int po_satdiff_hamming256(const uint8_t *a, const uint8_t *b);
int po_match_xyz(const float *mapPos, const uint8_t *mapDesc, size_t mapStep, const int32_t *mapLevel, int nmap,
                 const float *curPos, const uint8_t *curDesc, size_t curStep, const int32_t *curLevel, int ncur,
                 double sphereRadius, double acceptRatio, PsDMatch *out, int cap, int *nout);

/* N4: RGBD::removeImageDistortion (src/RGBD/RGBD.cpp:254-314) = cv::undistortPoints(pts, K, dist)
 * (5 fixed-point iterations of the Brown model in double, OpenCV 3.x cvUndistortPoints with R = P = I)
 * followed by u = x_n * fx + cx in float.  dist = (k1, k2, p1, p2, k3). */
void po_remove_image_distortion(const float *xy, int n, const float *K, const double *dist5, float *out);

/* Matcher::match data flow (matcher.cpp:470-515) over P independent pairs of a frame set
 * held in HOST memory (same layout as PsFrameSet/PsPairResults but host pointers);
 * threads > 1 runs pairs in parallel with OpenMP (the reference itself is single-threaded). */
int po_vo_pairs(const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                const PsFrameSet *frames, const int32_t *pairs, int P,
                const PsPairResults *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
