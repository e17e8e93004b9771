/* putslam_hip.h -- C ABI of the MI355X-native PUTSLAM visual-odometry front end.
 *
 * One data-parallel path of LRMPUT/PUTSLAM, hand-written for gfx950 (CDNA4):
 *   brute-force 256-bit Hamming matching with OpenCV cross-check semantics
 *   -> depth filter -> 3-point RANSAC / USAC with a float Umeyama fit
 *   -> inlier scoring (Euclidean / reprojection) -> refit -> acceptance gate,
 * plus the double-precision N-point Kabsch fit behind TransformEst and the
 * pinhole back-projection helper.
 *
 * Every entry point names the reference interface (file:line under the
 * PUTSLAM tree) it replaces.  Signatures are plain C: pointers, sizes, PODs.
 * All functions return PS_OK (0) or a negative PsStatus; on failure the
 * reference's own fallback outputs (identity pose, zero inliers) are still
 * written, because the reference path has no exceptions and no error codes
 * (src/TransformEst/RANSAC.cpp:77-80,161-164,239-242).
 *
 * There is NO CPU fallback behind this ABI.  If no HIP device is usable the
 * context constructor fails with PS_ERR_NO_DEVICE and every call fails loudly.
 */
#ifndef PUTSLAM_HIP_H_
#define PUTSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_ABI_VERSION 2          /* 2: PsFrameSet carries frame strides (packed frames) */
#define PS_DESC_BYTES 32          /* ORB (matcherOpenCV.cpp:90) and LDB (ldb.cpp:61,657) rows: 256 bit */
#define PS_MAX_KPTS 16384         /* keypoints per frame handled by one launch (LDS-resident cross-check) */
#define PS_MAX_HYPOTHESES (1 << 20)

typedef enum PsStatus {
    PS_OK = 0,
    PS_ERR_BAD_ARG = -1,
    PS_ERR_NO_DEVICE = -2,
    PS_ERR_HIP = -3,
    PS_ERR_ALLOC = -4,
    PS_ERR_UNSUPPORTED = -5,
    PS_ERR_BUSY = -6            /* pipelined streaming: every lane holds results the caller has not popped yet */
} PsStatus;

/* cv::DMatch: same field order and size (16 B) as OpenCV's struct, so a
 * std::vector<cv::DMatch>::data() can be passed straight through. */
typedef struct PsDMatch {
    int32_t queryIdx;   /* row of the PREVIOUS frame's descriptors (matcher.cpp:470-471) */
    int32_t trainIdx;   /* row of the CURRENT frame's descriptors */
    int32_t imgIdx;     /* always 0 */
    float distance;     /* integer Hamming distance 0..256 as float */
} PsDMatch;

/* RANSAC::ERROR_VERSION, include/putslam/TransformEst/RANSAC.h:22 */
typedef enum PsErrorVersion {
    PS_EUCLIDEAN_ERROR = 0,
    PS_REPROJECTION_ERROR = 1,
    PS_EUCLIDEAN_AND_REPROJECTION_ERROR = 2,
    PS_MAHALANOBIS_ERROR = 3,  /* dead in the reference (cov never set, RANSAC.cpp:301-303): scores 0 */
    PS_ADAPTIVE_ERROR = 4
} PsErrorVersion;

/* RANSAC::parameters, include/putslam/TransformEst/RANSAC.h:23-31 (same fields, same order). */
typedef struct PsRansacParams {
    int32_t verbose;
    int32_t errorVersion, errorVersionVO, errorVersionMap;
    double inlierThresholdEuclidean, inlierThresholdReprojection, inlierThresholdMahalanobis;
    double minimalInlierRatioThreshold;
    int32_t minimalNumberOfMatches;
    int32_t usedPairs;        /* only 3 is supported (the shipped value, putslammatcherOpenCVParameters.xml:37) */
    int32_t iterationCount;   /* ignored on input, like the reference ctor (RANSAC.cpp:30) */
} PsRansacParams;

/* Which sequential selection rule is replayed over the per-hypothesis inlier counts. */
typedef enum PsEstimator {
    PS_EST_RANSAC = 0,  /* RANSAC.cpp:87-164: strict-> best ratio, adaptive iterationCount, refit, ratio gate */
    PS_EST_USAC = 1,    /* USAC.h:326,409-414,944-971 + USAC_wrapper.cpp:104-151: best count, std stopping, no refit */
    PS_EST_FIXED = 2    /* all H hypotheses, first best wins, then RANSAC's refit + gate (adaptive stop disabled) */
} PsEstimator;

/* Controls that have no counterpart in the reference because its sampling is
 * srand(time(0)) + rand() (RANSAC.cpp:13,191): the sample stream is an input. */
typedef struct PsRansacConfig {
    int32_t estimator;        /* PsEstimator */
    int32_t numHypotheses;    /* H: number of 3-point samples that may be consumed (>= the schedule's maximum) */
    uint64_t seed;            /* counter-based stream: draw(h,j) = mix(seed,h,j) >> 33, index = draw % M, redraw on repeat */
    const uint32_t *sampleIdx;/* optional HOST pointer, H x 3 raw draws replacing the seeded stream:
                                 index_j = raw % M, a repeat is moved to the next free index (+1 mod M) */
} PsRansacConfig;

typedef struct PsRansacStats {
    int32_t numMatchesIn;     /* matches handed in (cross-check survivors) */
    int32_t numMatchesValid;  /* M after the depth filter RANSAC.cpp:65-74 */
    int32_t bestHypothesis;   /* index of the selected sample, -1 if none */
    int32_t bestInlierCount;  /* its inlier count inside the loop */
    int32_t iterationsRun;    /* loop trips the sequential reference would have made */
    int32_t numInliers;       /* final inliers (after refit re-selection and the ratio gate) */
    int32_t accepted;         /* 0 => identity returned (too few matches or ratio gate) */
    float bestInlierRatio;    /* float(count)/float(M) as in RANSAC.cpp:280 */
    double pointInlierRatio;  /* RANSAC::pointInlierRatio, RANSAC.h:56-66 (NaN if no input matches) */
} PsRansacStats;

typedef struct PsContext PsContext;

/* ---- context: one HIP stream + scratch arena per instance (reference threading
 * contract: main-thread matcher and loop-closure matcher are separate instances,
 * PUTSLAM.cpp:566,570; featuresMap.cpp:650-652,794). ------------------------- */
int ps_context_create(int device, PsContext **out);
void ps_context_destroy(PsContext *ctx);
/* Use an externally owned hipStream_t (e.g. the caller's current stream); NULL restores the private stream, which
 * is created hipStreamNonBlocking: it is NOT ordered with the legacy default stream, so a caller working on the
 * default stream hands over an explicit stream (forked from / joined to the default one) or synchronises.
 * The scratch arena belongs to the context: when the stream changes, the new stream is made to wait for the event
 * recorded at the end of the last asynchronous call (ps_vo_pairs_device), so consecutive calls on different streams
 * never overlap on it.  The previous stream itself is not touched by ps_context_set_stream: its owner may destroy it once
 * the calls submitted to it have been synchronised or a later stream has been selected.
 * One context still serves one caller thread at a time; concurrent chains use one context each. */
int ps_context_set_stream(PsContext *ctx, void *hipStream);
int ps_context_synchronize(PsContext *ctx);
/* The hipStream_t the context's calls are queued on (its private stream unless ps_context_set_stream chose another) and the
 * device it was created for: what a host needs to order its own kernels, copies or collectives with the context's work
 * (include/putslam_shard.h queues its RCCL calls there). */
void *ps_context_stream(PsContext *ctx);
int ps_context_device(const PsContext *ctx);
/* Kernel variants kept side by side for A/B measurements and as tested twins (results are identical):
 *   "matcher": 1 = FP4 matrix-core sweep ps_hamming_mfma, 0 = integer VALU sweep ps_hamming_nn, 2 = by batch size
 *              (default: the matrix-core form, one launch more, from about five 2000-keypoint pairs per call on)
 *              (environment: PUTSLAM_HIP_MATCHER=mfma|valu|auto, read at context creation); "matcher_used" (read only)
 *              tells which of the two the last matching call ran.
 *   "score":   1 = decision-exact fast scoring kernels (cheap evaluation with a proven error band, in-band evaluations
 *              re-done by the value-exact code; default): ps_ransac_score_euclid for errorVersion 0 / 4 (the metric every
 *              shipped reference config runs), ps_ransac_score_fast for errorVersion 1 / 2;
 *              0 = value-exact ps_ransac_score<M> for every evaluation (PUTSLAM_HIP_SCORE=fast|exact).
 *              Per-hypothesis counts are identical between 0 and 1 by proof + tests.  (Round 2's matrix-core experiment,
 *              value 2, left the library in round 4: profiles/variants/ps_score_mfma.h.txt.)
 *   "prune":   1 (default) = staged scoring of large batches: the first 256 hypotheses of every pair are scored
 *              completely, the later ones in three stages over growing match ranges; between the stages every
 *              hypothesis that cannot become a record of the sequential selection any more (count so far + matches left
 *              <= best count of the earlier ones, RANSAC.cpp:438-455) is abandoned, and hypotheses beyond the adaptive
 *              trip limit (RANSAC.cpp:450-453) are never started; all outputs are unchanged, only the scratch counts of
 *              abandoned hypotheses are lower bounds.  "Large" is decided by a cost model: pairs x (ceil(H / 256) - 1) x
 *              maxKpts against base + perRow x maxKpts, per metric family and schedule (the staged form costs six or seven
 *              dependent launches and pays them back with the evaluations it abandons).  2 = the staged form whenever the
 *              kernels have it (H > 256), whatever the batch size; 0 = every hypothesis is scored completely
 *              (PUTSLAM_HIP_PRUNE=0|1|2).  The diagnostic ps_debug_ransac_counts always scores completely.
 *   "reorder": the stages after the first sweep a copy of the pair's match record in which the matches the best hypotheses
 *              so far reject come first (written by a launch of its own between the stages): whatever is not better than
 *              those hypotheses rejects nearly all of them too and is abandoned a few matches later.  Counts are sums over
 *              all matches, so the order changes no output.  1 = always, 0 = original match order, 2 (default) = with the
 *              fixed schedule only (under the adaptive ones the trip limit usually ends the scoring before the stages
 *              start and the launch would buy nothing)  (PUTSLAM_HIP_REORDER=0|1|2).
 *   "bail":    1 (default) = "nothing to gain" handling of the staged scoring for the Euclidean metrics: a pair whose prefix
 *              leaves a miss budget so large that the first stage sweeps every match anyway skips the reorder vote, and -- fixed
 *              schedule only -- while most pairs of the last observed batched call OF THE SAME KIND (metric, schedule, H,
 *              batch-size class, frame capacity, frame set; eight kinds are tracked) were such pairs, the next calls of that
 *              kind are scored completely, probing with the staged form every 16th call (identical outputs either way;
 *              "hopeless", read only, is the state of the last call's kind).  Adaptive schedules always keep the staged form.
 *              0 = always the staged form.  Setting the option (to either value) forgets what was observed.
 *   "last_staged_pairs" / "last_reordered_pairs" (read only): pairs of the last scoring step if it was staged / reordered,
 *              else 0.
 *   "model_room_mib": room for the staged scoring's parked models (48 bytes per pair and leading hypothesis); 0 (default) =
 *              256 MiB under the adaptive schedules, 2 GiB under the fixed one.  Hypotheses without a slot are swept in one
 *              piece by stage 1 and rebuilt by kernel 4 if one of them wins: identical outputs, tests force it small.
 *   "score_stats": 1 = count the evaluations the fast kernel hands to the value-exact code (ps_debug_score_stats).
 *   "stamps":  1 = kernels 2 and 4 record the shader clock at their phase boundaries (ps_debug_stamps); 0 (default) = they
 *              are passed a null pointer and record nothing.
 *   "side_by_side": 0 (default) = the context's launches have the chip to themselves; n >= 2 = it is one of n launch chains
 *              that run side by side (a PsBatchQueue sets it on its chains, the pipelined stream on its lanes from chunks of 48
 *              frames on; a host that drives several contexts on streams of its own sets it itself).  The other chains fill the
 *              gaps between the staged scoring's dependent launches, so "prune" = 1 takes the staged form from far smaller
 *              batches on (E1 / fixed / H = 4096 / 2000 keypoints: from 16 pairs instead of 77; two chains: from 35).  Identical
 *              outputs whatever the value  (PUTSLAM_HIP_SIDE_BY_SIDE).
 * Twelve options in all are the surface: "matcher", "matcher_fused", "score", "prune", "side_by_side", "reorder", "bail", "model_room_mib",
 * "stream_copy_kernels", "stream_ahead" (places of the pipelined stream beyond one per lane, see below), "score_stats", "stamps",
 * and the read-only ones.
 * NOT part of it -- launch-shape and tuning knobs of the sweeps and of the staged scoring, every value of which gives the same
 * results; they exist for the parity tests (tests/test_gpu_prune.py runs every one next to the default) and for A/B
 * measurements, answer only to the name "debug.<knob>" (and PUTSLAM_HIP_<KNOB> at context creation) and may change between
 * versions: qsplit / msplit (work-groups the query range of kernel 1 / the match range of kernel 3 is split over, 0 =
 * automatic), gensplit (stage 0 as two launches, models then sweep), singlerest (one stage after the prefix under the adaptive
 * schedules), pretest (stage 1's one-direction pre-test), prefix (64 / 128 / 192 / 256 hypotheses of stage 0), list_g2
 * (work-groups per pair of stage 2), reorder_top (voters), reorder_margin.  (list_g3, list_r3, reorder_c2div and reorder_gran
 * left in round 6: every A/B had their other values within 1 % of the defaults; they are constants of the library now.)
 * ps_context_get_option returns the value or a negative PsStatus. */
int ps_context_set_option(PsContext *ctx, const char *name, int value);
int ps_context_get_option(const PsContext *ctx, const char *name);
const char *ps_last_error(const PsContext *ctx);
int ps_abi_version(void);
/* Name of the device the context runs on, e.g. "gfx950". */
const char *ps_device_arch(const PsContext *ctx);

/* ---- A1: MatcherOpenCV::performMatching, src/Matcher/matcherOpenCV.cpp:198-206
 * = cv::BFMatcher(NORM_HAMMING, crossCheck=true).match(query=prev, train=cur)
 * (matcher object built at matcherOpenCV.cpp:100-105).  Host pointers; rows are
 * 32 bytes wide with a row pitch of qstep/tstep bytes (cv::Mat::step).
 * out must hold nq entries; *nout receives the number written (ascending queryIdx). */
int ps_match_hamming256(PsContext *ctx,
                        const uint8_t *query, int nq, size_t qstep,
                        const uint8_t *train, int nt, size_t tstep,
                        PsDMatch *out, int *nout);

/* ---- A4-A9 / A11: RANSAC::estimateTransformation, src/TransformEst/RANSAC.cpp:50-174
 * and RANSAC_USAC::estimateTransformation, src/USAC/USAC_wrapper.cpp:104-151.
 * prev/cur: N x 3 floats, 12-byte stride (std::vector<Eigen::Vector3f> storage).
 * K: row-major 3x3 float camera matrix (cv::Mat CV_32FC1, RGBD.cpp:93-96); may be NULL for
 *    the Euclidean/adaptive modes.
 * pose: column-major 4x4 float (Eigen::Matrix4f storage), maps current-frame points into
 *    the previous frame (umeyama(src=cur,dst=prev), RANSAC.cpp:225-226).
 * inliers (capacity m) / ninl: final inlier matches in input order; mask (m bytes, may be
 *    NULL): 1 where matches[i] is a final inlier. stats may be NULL. */
int ps_ransac_rigid3d(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg,
                      const float *K,
                      const float *prev, int nprev, const float *cur, int ncur,
                      const PsDMatch *matches, int m,
                      float *pose, PsDMatch *inliers, int *ninl, uint8_t *mask,
                      PsRansacStats *stats);

/* ---- A7: RANSAC::computeTransformationModel, RANSAC.cpp:207-244
 * (Eigen::umeyama(src, dst, false) in float + the isnan(T(0,0)) check).
 * nsets independent fits of k points each: src/dst are nsets x k x 3 floats.
 * T: nsets x 16 column-major; valid: nsets flags (0 => identity written). */
int ps_umeyama_f32(PsContext *ctx, const float *src, const float *dst, int k, int nsets,
                   float *T, int32_t *valid);

/* ---- A10: KabschEst::computeTransformation, src/TransformEst/kabschEst.cpp:24-68
 * (interface include/putslam/TransformEst/transformEst.h:23).
 * A, B: n x 3 doubles, COLUMN-major with leading dimension ld (Eigen::MatrixXd storage).
 * T: column-major 4x4 double (Mat34 = Eigen::Transform<double,3,Affine>), maps A onto B. */
int ps_kabsch_f64(PsContext *ctx, const double *A, const double *B, int n, int ld, double *T);

/* ---- A3: RGBD::keypoints2Dto3D / point2Dto3D / roundSize, src/RGBD/RGBD.cpp:10-16,30-65.
 * xy: n x 2 floats (cv::Point2f), depth: rows x cols uint16 with a pitch of depthStep BYTES,
 * K row-major 3x3 float, out: n x 3 floats. */
int ps_keypoints2Dto3D(PsContext *ctx, const float *xy, int n,
                       const uint16_t *depth, int rows, int cols, size_t depthStep,
                       const float *K, double depthImageScale, float *out);

/* ---- N4: RGBD::removeImageDistortion, src/RGBD/RGBD.cpp:254-314 = cv::undistortPoints(pts, K, dist) with
 * R = P = I (5 fixed-point iterations of the Brown model, double) followed by u = x_n*fx + cx in float.
 * xy, out: n x 2 floats (cv::Point2f); dist5 = (k1, k2, p1, p2, k3) (datasetConfig rgbDistortion). */
int ps_remove_image_distortion(PsContext *ctx, const float *xy, int n, const float *K, const double *dist5,
                               float *out);

/* ---- A3: RGBD::point3Dto2D, src/RGBD/RGBD.cpp:92-98 (n points). */
int ps_points3Dto2D(PsContext *ctx, const float *xyz, int n, const float *K, float *uv);

/* ---- N2: guided map matching, core of Matcher::matchXYZ, src/Matcher/matcher.cpp:606-746.
 * For every map feature j: candidates i among the current frame's keypoints with
 * |mapPos[j] - curPos[i]| < sphereRadius and |curLevel[i] - mapLevel[j]| <= 1 (:699-711); their value is
 * cv::norm(mapDesc[j] - curDesc[i], NORM_HAMMING) on CV_8U rows = popcount of the per-byte SATURATING
 * difference (:719-721); every candidate with acceptRatio * value <= best value is emitted (:734-746) as
 * DMatch(queryIdx = j, trainIdx = i, imgIdx = -1, distance = value), ordered by (j, i).
 * mapPos: (float) casts of MapFeature::position (:701-702); levels from ps_predicted_level.
 * Returns PS_ERR_BAD_ARG with *nout = required capacity if cap is too small.  Host pointers. */
int ps_match_xyz(PsContext *ctx, const float *mapPos, const uint8_t *mapDesc, size_t mapDescStep,
                 const int32_t *mapLevel, int nmap, const float *curPos, const uint8_t *curDesc,
                 size_t curDescStep, const int32_t *curLevel, int ncur, double sphereRadius,
                 double acceptRatio, PsDMatch *out, int cap, int *nout);
/* Predicted ORB pyramid level, matcher.cpp:639-652 (keypoints) and :681-692 (map features):
 * clamp(ceil(log(1.2^octave * detDist / curDist) / log 1.2), 0, 7).  Pure host arithmetic (libm). */
int ps_predicted_level(int octave, double detDist, double curDist);

/* ---- A2 + A12: the data flow of Matcher::match (src/Matcher/matcher.cpp:470-515) for a
 * whole batch of independent frame pairs, everything resident in HBM.  All pointers in
 * PsFrameSet / PsPairResults are DEVICE pointers; nothing is copied to or from the host. */
typedef struct PsFrameSet {
    const uint8_t *desc;      /* numFrames x maxKpts x 32 B descriptors */
    const float *pts;         /* numFrames x maxKpts x 3 floats, back-projected 3-D points */
    const int32_t *nkpts;     /* numFrames keypoint counts (<= maxKpts) */
    int32_t numFrames;
    int32_t maxKpts;          /* row capacity per frame; also the capacity of per-pair outputs */
    /* ABI 2: bytes from one frame's block to the next; 0 = dense (maxKpts x 32 / maxKpts x 12).  A frame set whose frames keep
     * their descriptors and points together -- [maxKpts x 32 B][maxKpts x 12 B] per frame, what one transfer per frame or per
     * chunk of frames delivers (the prevDescriptors / prevFeatures3D pair of matcher.h:379-384 as one block) -- has
     * pts = desc + maxKpts x 32 and both strides = the frame's size.  descFrameStride: a multiple of 16, >= maxKpts x 32;
     * ptsFrameStride: a multiple of 4, >= maxKpts x 12. */
    size_t descFrameStride;
    size_t ptsFrameStride;
} PsFrameSet;

typedef struct PsPairResults {
    PsDMatch *matches;        /* P x maxKpts: cross-check matches per pair, ascending queryIdx */
    int32_t *numMatches;      /* P */
    uint8_t *inlierMask;      /* P x maxKpts: 1 where matches[p][i] is a final inlier */
    float *pose;              /* P x 16 column-major */
    PsRansacStats *stats;     /* P */
} PsPairResults;

/* pairs: DEVICE array of P (prevFrame, curFrame) index pairs, int32 x 2 each.
 * Hypothesis h of pair p draws from the seeded stream with seed cfg->seed + p
 * (cfg->sampleIdx must be NULL). Asynchronous on the context's stream.
 * Scratch: counts take 4 bytes per pair and hypothesis of cfg->numHypotheses (1.7 GB for 499 pairs under USAC's cap of
 * 850 000, USAC_wrapper.cpp:70; touched only up to each pair's trip limit); the staged scoring's parked models take 48 bytes
 * per pair and LEADING hypothesis, 256 MB at most under the adaptive schedules (2 GiB under the fixed one): a hypothesis
 * beyond the slots is swept in one piece and, should it win, rebuilt (options "arena_mib", "last_model_slots", read only).
 * Throughput: a host that loops over batches gets 18 % more (499 pairs) to 2.9 x (64 pairs) from a PsBatchQueue (below), which
 * hands the batches to four contexts on four streams in turn -- launch chains that are never joined. */
int ps_vo_pairs_device(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg,
                       const float *K, const PsFrameSet *frames,
                       const int32_t *pairs, int P, const PsPairResults *out);

/* ---- A2, for a host that loops over batches (the loop of src/PUTSLAM/PUTSLAM.cpp:677-740 around Matcher::match,
 * src/Matcher/matcher.cpp:470-515): ps_vo_pairs_device through launch chains that are never joined.
 * One context is one launch chain: a batch's matrix-core Hamming sweep, then its vector scoring stages, dependent launches with
 * the chip partly idle between them.  A queue owns `chains` contexts + streams (0 = the default, 4; 1 .. PS_BATCH_QUEUE_MAX_CHAINS)
 * on ctx's device, with ctx's options, and hands batch n to chain n mod chains, WHOLE: consecutive batches run side by side, one
 * in its Hamming sweep while the others score -- 610 k instead of 517 k frame-pairs/s on batches of 499 pairs, 488 k instead of
 * 274 k at 125 pairs, 408 k instead of 142 k at 64 (round 6's measurements: splitting every batch over two chains, rounds 3 - 5's
 * recipe, gave 559 k at 499 pairs, profiles/r06h/queue_split_vs_turns.txt; FOUR chains are the best count at every batch size
 * from 16 to 1000 pairs, two read 601 k / 406 k / 183 k, six are worse than four: profiles/r06u/small_batch_chains.txt).  The chains'
 * contexts carry option "side_by_side" = chains: with other chains filling the gaps between its dependent launches the staged
 * scoring pays from 2 - 5 times smaller batches on than it does for a lone context (profiles/r06u/bench_data_crossover.txt).
 * The chains are ordered only within themselves; nothing ever makes one wait for another.
 *   submit: arguments of ps_vo_pairs_device (device pointers); pair p draws from cfg->seed + p: the outputs are byte for byte
 *           those of ONE ps_vo_pairs_device call.  Returns at once; *ticket (may be NULL) names the batch.  Inputs and outputs
 *           must stay valid until the batch is complete.  Batches in flight run CONCURRENTLY: give consecutive batches output
 *           blocks of their own (`chains` blocks used in turn are enough: batch n + chains runs on batch n's chain, behind it).
 *           A batch that is handed the block (out->pose) of a batch still in flight on another chain is queued behind that one on
 *           its chain instead -- correct, and as fast as one context.  Reading a batch's results needs its ticket waited for.  At most 64 batches are in flight: the 65th submit waits for
 *           the first.  If the chain's call fails the error is returned (text: ps_last_error of ctx); the ticket still stands
 *           for whatever part of the batch was queued.
 *   wait / query: the host blocks until / asks whether that batch is complete (query: 1 complete, 0 not yet).
 *   wait_on_stream: the given hipStream_t waits for the batch instead (device-side; the host does not block): what a host
 *           that post-processes on a stream of its own queues behind a batch.  A device-side wait for a batch that is still
 *           running is not free: while it is pending the chains themselves lose 5 % (499 pairs per batch) to 19 % (125) of
 *           their rate (profiles/r06v/pending_waits.txt) -- a host that can, queues its dependent work once query / wait says
 *           the batch is complete (the sharding layer and bench.py issue their gathers that way).
 *   context(q, i): chain i's context -- for ps_context_stream (work to be queued behind that chain's batch),
 *           ps_context_enable_timing, options; not for calls of its own while batches are in flight.
 *   last_split: bounds[0 .. chains] of the last submitted batch: pairs [bounds[i], bounds[i+1]) ran on chain i (all of them on
 *           one chain, unless PUTSLAM_HIP_QUEUE_SPLIT_FROM=<pairs> asks for rounds 3 - 5's split of batches that large: A/B runs).
 * Hardware queues: every chain wants one of its own.  The HIP runtime gives a process GPU_MAX_HW_QUEUES of them (default 4,
 * shared with the host's other streams) and serialises streams that share one.  The library sets GPU_MAX_HW_QUEUES=16 when it
 * is loaded if the variable is unset (a constructor, before the process' first HIP call for a program that links the library;
 * a host that set the variable keeps its value).  Read-only option "hw_queues_seen" = the value found; ps_batch_queue_create
 * leaves a warning in ps_last_error(ctx) when it is too small.  A process that initialises HIP before loading the library sets
 * the variable itself (putslam_amd/_lib.py does). */
#define PS_BATCH_QUEUE_MAX_CHAINS 8
typedef struct PsBatchQueue PsBatchQueue;
int ps_batch_queue_create(PsContext *ctx, int chains, PsBatchQueue **out);
void ps_batch_queue_destroy(PsBatchQueue *q);
int ps_batch_queue_submit(PsBatchQueue *q, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                          const PsFrameSet *frames, const int32_t *pairs, int P, const PsPairResults *out, int64_t *ticket);
int ps_batch_queue_wait(PsBatchQueue *q, int64_t ticket);
int ps_batch_queue_query(PsBatchQueue *q, int64_t ticket);
int ps_batch_queue_wait_on_stream(PsBatchQueue *q, int64_t ticket, void *hipStream);
int ps_batch_queue_synchronize(PsBatchQueue *q);
int ps_batch_queue_chains(const PsBatchQueue *q);
PsContext *ps_batch_queue_context(PsBatchQueue *q, int chain);
int ps_batch_queue_last_split(const PsBatchQueue *q, int32_t *bounds);

/* ---- The path's only exchange between GPUs: what travels to the rank that composes the trajectories -- the one
 * sequential step of the reference, VO pose composition with the 0.1 m gate, src/PUTSLAM/PUTSLAM.cpp:735-740 -- is a 72-byte record
 * per pair: pose[16] (column-major) + numInliers + numMatchesIn, as PS_RECORD_FLOATS floats.  ONE launch on hipStream (NULL: the
 * context's stream) packs `pairs` records from a batch's device-resident results: rows [0, valid) from pose / stats, rows
 * [valid, pairs) zero-filled (ranks of a gather send blocks of one size).  Queued behind a batch on the chain it ran on
 * (ps_batch_queue_context(q, chain)), it reads the block before that chain's next batch can overwrite it.  include/putslam_shard.h
 * and bench.py's multi-rank steps pack with it (torch's slice assignments were five launches on the chain). */
#define PS_RECORD_FLOATS 18
int ps_pack_records_device(PsContext *ctx, void *hipStream, const float *pose, const PsRansacStats *stats, int valid, int pairs,
                           float *records);

/* ---- A2, streaming form: Matcher::match (src/Matcher/matcher.cpp:452-516) with the previous frame's
 * descriptors and 3-D points resident in HBM (the prevDescriptors / prevFeatures3D state, matcher.h:379-384).
 * The first push only stores the frame (detectInitFeatures, matcher.cpp:17-64) and returns *nmatches = -1;
 * every later push matches previous (query) against the new frame (train), runs the estimator selected by
 * cfg on the surviving matches and makes the new frame the previous one (:506-513).
 * Hypothesis stream of every push = cfg->seed (the caller varies it per frame if wanted).
 * matches / inlierMask: capacity maxKpts; pose column-major 4x4; host pointers.
 * A push is launch-bound, so from the third push with unchanged params / estimator / numHypotheses / K on, the
 * copy-in -> four kernels -> copy-out sequence is replayed from a captured hipGraph (one per frame slot; frame,
 * row count and seed travel as data).  Results are identical; PUTSLAM_HIP_NO_GRAPH=1 keeps ordinary launches. */
typedef struct PsVoStream PsVoStream;
int ps_vo_stream_create(PsContext *ctx, int maxKpts, PsVoStream **out);
void ps_vo_stream_destroy(PsVoStream *s);
/* Forget the resident frame: the next push is a first frame again (Matcher::detectInitFeatures, matcher.cpp:17-64). */
int ps_vo_stream_reset(PsVoStream *s);
int ps_vo_stream_push(PsVoStream *s, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const uint8_t *desc, size_t descStep, const float *pts, int n,
                      PsDMatch *matches, int *nmatches, uint8_t *inlierMask, float *pose, PsRansacStats *stats);

/* ---- A2, PIPELINED streaming form (BASELINE configs[2]: "500 frames streamed through Matcher -> USAC -> Kabsch"):
 * the same call shape -- frames arrive on the host one after the other, each is matched against its predecessor
 * (src/Matcher/matcher.cpp:452-516, loop src/PUTSLAM/PUTSLAM.cpp:677-740) -- with the results returned with a LAG instead
 * of inside the push, so that uploads, kernels and downloads of consecutive frames overlap.  Frames are collected into
 * chunks of `chunkFrames`; a full chunk is uploaded on a copy stream into a ring of frames resident in HBM (the last
 * frame of the previous chunk is still there: no halo is sent twice), runs as ONE batched call (ps_vo_pairs_device's
 * launches) on one of `lanes` private contexts in turn (stream + scratch arena each, so that one chunk's Hamming sweep runs
 * beside another's scoring sweep; a lane's stream may hold further chunks queued behind the running one), and its results come
 * back in one download into pinned host memory.  chunkFrames = 1 is the
 * lowest-latency setting, 64..256 the throughput setting; ps_vo_stream_push stays the synchronous per-frame form.
 *
 * Hardware queues: every lane, the upload stream and the download stream want a hardware queue of their own; the HIP runtime
 * gives a process GPU_MAX_HW_QUEUES of them (default 4) and lets streams share beyond that, which serialises what shares.  The
 * library sets GPU_MAX_HW_QUEUES = 16 itself when it is loaded and the variable is unset (see PsBatchQueue above); with fewer than lanes + 6 the downloads
 * are queued on the lanes' own streams instead of a stream of their own (results identical, 10 - 20 % slower).
 *
 * Small chunks (chunkFrames <= 4; 1 is the reference's own call shape): a chunk of one to four frames is launch-bound, so every
 * place keeps a private copy of its frames (the previous frame is read again from its pinned staging slot: no ring, no copy
 * stream, no event pair) and its whole chunk is seven launches on the place's lane -- frames in by a copy kernel, kernels 1 - 4,
 * results out by a kernel --, with row counts, seed and frame addresses travelling as data: 25 - 28 k frames/s at 0.2 ms of lag from
 * one host thread, twice the synchronous push.  (PUTSLAM_HIP_STREAM_GRAPH=1 replays each place's chunk from ONE captured hipGraph
 * instead: measured slower on this runtime, profiles/r06h/mini_chunks.txt.)  Frames go through the library's pinned staging
 * areas in this form (88 KB per 2000-keypoint frame) unless they are pinned packed blocks handed to
 * ps_vo_stream_push_many_packed, which are read in place (untouched until the pair that has a frame as its PREVIOUS frame has been
 * popped); pop_many's pointers are a copy.  PUTSLAM_HIP_STREAM_MINI=0 keeps the ring form for such chunks (results identical).
 *
 * Pair k of the stream (frames k, k+1 counted from the last reset; frame k is the query = previous frame) draws its
 * hypotheses from the seeded stream cfg->seed + k: the results are byte for byte those of ONE ps_vo_pairs_device call
 * over the whole sequence with the same cfg, whatever the chunking.
 *
 * ps_vo_stream_configure_async: parameters of the pipelined form, fixed until the next configure (which drains).  chunkFrames
 *   1..1024 (0 = 128), lanes 2..8 (0 = by chunk size: 2 from 48 frames per chunk on, 3 below, 6 for chunks of one to four frames).
 *   cfg->sampleIdx must be NULL.  The lanes inherit the options of the stream's context.  Throughput grows with the chunk: 417 /
 *   504 / 550 k frame-pairs/s at 125 / 250 / 500 frames of 2000 keypoints per chunk (1.2 / 2.0 / 2.7 ms from push to results).
 * ps_vo_stream_push_async: ONE frame (host pointers, rows of descStep bytes) is copied into the pinned staging area of
 *   the chunk being collected; the chunk is submitted when it is full.  Returns at once.
 * ps_vo_stream_push_many: numFrames frames laid out like a PsFrameSet on the HOST (desc numFrames x maxKpts x 32 B,
 *   pts numFrames x maxKpts x 3 floats, nkpts numFrames) are submitted as chunks of at most chunkFrames.  If desc and pts
 *   are pinned host memory (ps_host_alloc, hipHostMalloc, hipHostRegister) the upload reads them in place -- they must
 *   stay untouched until the results of their frames have been popped; pageable memory is staged through pinned buffers.
 *   Frames staged by push_async before are submitted first, as a chunk of their own.
 * ps_vo_stream_flush: submits a partly filled chunk.
 * Flow control: a chunk needs a PLACE (a meta block, device and pinned result blocks, events); there are lanes + `ahead` of
 *   them (context option "stream_ahead", 0 .. 8, or -1 = by chunk size, the default: 1 from 96 frames per chunk on, 2 from 48,
 *   else six places in all; read by configure_async).  An accepted chunk is uploaded and
 *   launched at once, on the next lane in turn -- behind that lane's running chunk if it has one --, so a lane never waits for
 *   the host between chunks.  A place is busy from the launch of its chunk until pop_many has returned its results (the pinned
 *   block they lie in changes hands: the place goes on with a spare one while the caller reads).  push_async (at the first frame of a chunk) / push_many return
 *   PS_ERR_BUSY -- and take nothing -- when there is no place: pop first.  (Pinned frames handed to push_many are read in
 *   place: untouched until their results have been popped, as before.)
 * Errors other than PS_ERR_BUSY from push_async / push_many / flush / reset (an allocation or a HIP call failed while a chunk
 *   was being queued): whatever part of that chunk was queued is drained, the chunk's frames are DROPPED AS A UNIT -- for
 *   push_async: every frame collected in the partly filled chunk, the one just pushed included; for push_many: the failing chunk
 *   and the frames after it in the call (chunks before it are in flight and keep their numbering) -- and the stream goes on as
 *   after ps_vo_stream_reset: the next frame has no predecessor, its pair is number 0 of a new epoch (cfg->seed + 0).  No
 *   place is lost, nothing is in flight that pop would not return.  The reference's convention (no exceptions, fallback
 *   outputs, src/TransformEst/RANSAC.cpp:77-80,161-164) has no counterpart for a lost frame: the status code is the signal.
 * ps_vo_stream_pop_many: results of the oldest chunk in flight, as HOST pointers into that lane's pinned result block
 *   (valid until the next pop_many / pop / configure / destroy of this stream).  wait = 0: out->count = 0 if that chunk
 *   has not finished (or nothing is in flight); wait = 1: blocks until it has.
 * ps_vo_stream_pop: the same, one pair at a time, copied out (matches / inlierMask: capacity maxKpts; *nmatches = -1 and
 *   PS_OK when nothing is ready / in flight).  PS_RESULTS_INLIERS: *nmatches = number of inliers, matches = the inlier
 *   matches, inlierMask all ones; PS_RESULTS_POSES: *nmatches = 0, pose and stats only.
 * ps_vo_stream_reset on a pipelined stream: the next frame has no predecessor (Matcher::detectInitFeatures) and pair
 *   numbering restarts at 0; a partly filled chunk is submitted first (its place was reserved by its first frame); chunks in
 *   flight or waiting are unaffected and keep their numbering (`epoch` tells them apart). */
/* What a chunk's download carries (ps_vo_stream_set_result_mode, before ps_vo_stream_configure_async):
 *   PS_RESULTS_FULL     every cross-check match + the inlier mask + pose + stats (34 KB per 2000-keypoint pair);
 *   PS_RESULTS_INLIERS  what Matcher::match hands back (matcher.cpp:452-516: estimatedTransformation, inlierMatches): the
 *                       inlier matches of every pair in input order -- the first stats[i].numInliers entries of
 *                       matches[i * maxKpts ...] -- + pose + stats; inlierMask is NULL (12 KB per pair);
 *   PS_RESULTS_POSES    pose + stats only; matches and inlierMask are NULL (108 bytes per pair: a host that only composes
 *                       the trajectory, PUTSLAM.cpp:735-740).
 * numMatches is the number of cross-check matches in every mode.  Modes 1 and 2 are written by a kernel straight into the
 * lane's pinned block (no copy engine, no blit kernel beside the other lanes' launches). */
typedef enum PsStreamResults { PS_RESULTS_FULL = 0, PS_RESULTS_INLIERS = 1, PS_RESULTS_POSES = 2 } PsStreamResults;
/* How frames lie on the host and in the ring (ps_vo_stream_set_frame_layout, before ps_vo_stream_configure_async):
 *   PS_FRAMES_TWO_ARRAYS  descriptors and points in two arrays (push_many's form): two uploads per chunk;
 *   PS_FRAMES_PACKED      every frame is ONE block [maxKpts x 32 B descriptors][maxKpts x 12 B points] padded to
 *                         ps_vo_stream_packed_stride() bytes (maxKpts x 44 rounded up to 16; 88 000 at 2000 keypoints) -- what
 *                         a front end that detects, describes and back-projects a frame naturally fills, the reference's
 *                         prevDescriptors / prevFeatures3D state (include/putslam/Matcher/matcher.h:379-384) as one block --,
 *                         the ring in HBM has the same layout (PsFrameSet strides) and a chunk is ONE transfer: the link does
 *                         not idle between a descriptor and a point upload (+ 10 ... 15 % at chunks of 125).
 *                         ps_vo_stream_push_many_packed reads pinned frames in place (pageable ones are staged);
 *                         push_async / push_many still work on such a stream: they fill packed staging blocks on the host. */
typedef enum PsStreamFrames { PS_FRAMES_TWO_ARRAYS = 0, PS_FRAMES_PACKED = 1 } PsStreamFrames;

typedef struct PsHostPairResults {
    const PsDMatch *matches;      /* count x maxKpts (PS_RESULTS_INLIERS: the inlier matches first; PS_RESULTS_POSES: NULL) */
    const int32_t *numMatches;    /* count */
    const uint8_t *inlierMask;    /* count x maxKpts (NULL unless PS_RESULTS_FULL) */
    const float *pose;            /* count x 16, column-major */
    const PsRansacStats *stats;   /* count */
    int64_t firstPair;            /* number of the first pair of the block since the reset it belongs to */
    int32_t count;                /* pairs in this block; 0 = nothing ready */
    int32_t maxKpts;
    int32_t epoch;                /* resets of the stream before this block's frames */
    int32_t resultMode;           /* PsStreamResults the block was written with */
} PsHostPairResults;
int ps_vo_stream_set_result_mode(PsVoStream *s, int mode /* PsStreamResults; takes effect at the next configure_async */);
int ps_vo_stream_configure_async(PsVoStream *s, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                                 int chunkFrames, int lanes);
int ps_vo_stream_push_async(PsVoStream *s, const uint8_t *desc, size_t descStep, const float *pts, int n);
int ps_vo_stream_push_many(PsVoStream *s, const uint8_t *desc, const float *pts, const int32_t *nkpts, int numFrames);
int ps_vo_stream_set_frame_layout(PsVoStream *s, int layout /* PsStreamFrames; takes effect at the next configure_async */);
size_t ps_vo_stream_packed_stride(const PsVoStream *s);
int ps_vo_stream_push_many_packed(PsVoStream *s, const uint8_t *frames, size_t frameStride, const int32_t *nkpts, int numFrames);
int ps_vo_stream_flush(PsVoStream *s);
int ps_vo_stream_pop_many(PsVoStream *s, int wait, PsHostPairResults *out);
int ps_vo_stream_pop(PsVoStream *s, int wait, PsDMatch *matches, int *nmatches, uint8_t *inlierMask, float *pose,
                     PsRansacStats *stats);
/* Pairs submitted or staged whose results have not been popped yet (negative PsStatus on error). */
int ps_vo_stream_pending(const PsVoStream *s);
/* Pushes (synchronous form) / chunks (pipelined form, chunkFrames <= 4) replayed from a captured hipGraph so far. */
long long ps_vo_stream_graph_launches(const PsVoStream *s);
/* Pinned (page-locked) host memory for frames handed to ps_vo_stream_push_many in place; NULL on failure. */
void *ps_host_alloc(size_t bytes);
void ps_host_free(void *p);

/* Algorithmic bytes one call of ps_vo_pairs_device moves per SURVEY.md section 8(d):
 * computed from the per-pair stats already on the host (numMatchesIn, numMatchesValid). */
uint64_t ps_algorithmic_bytes(int nkpts, int matchesIn, int matchesValid, int numHypotheses);

/* Names of the kernels launched by ps_vo_pairs_device, in launch order, NUL separated,
 * double-NUL terminated (used by the bench to pick rows out of rocprofv3 output). */
const char *ps_kernel_names(void);

/* Kernel timing: when enabled, every ps_vo_pairs_device call brackets each of its kernel
 * launches with HIP events recorded on the context's stream (the stream the kernels run on).
 * Enabling (again) resets the record; the last 128 calls are kept.
 * ps_last_kernel_times_ms: the most recent call (ms holds 8 floats; returns the kernel count).
 * ps_kernel_time_totals: sums and launch counts over all kept calls (arrays of 8). */
int ps_context_enable_timing(PsContext *ctx, int enable);
int ps_last_kernel_times_ms(PsContext *ctx, float *ms);
int ps_kernel_time_totals(PsContext *ctx, double *sum_ms, int *launches);

/* ---- diagnostics (used by the parity tests; no reference counterpart) -------------------- */
/* ps_ransac_rigid3d's scoring stage only: counts[h] = inlier count kernel 3 produced for
 * hypothesis h (0 for an invalid model); *numScored = hypotheses scored (<= cfg->numHypotheses:
 * the RANSAC estimator never needs more than max(iterations(0.2), iterations(minRatio))). */
int ps_debug_ransac_counts(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg,
                           const float *K, const float *prev, int nprev, const float *cur, int ncur,
                           const PsDMatch *matches, int m, int32_t *counts, int *numScored);
/* Words of the context's keys block that are not all-ones once its queued work has drained; must be 0 after any sequence of
 * calls (the matcher's query splits merge with atomicMin on a block kernel 2 leaves all-ones: no clearing launch per call). */
int ps_debug_keys_clean(PsContext *ctx, uint64_t *bad);
/* Device-side trip limit after a best model with c inliers out of M, for c = 1..M:
 * min(H, iterations(minRatio), iterations(float(c)/float(M))) for PS_EST_RANSAC (RANSAC.cpp:450-461),
 * min(H, updateStandardStopping(c, M, 3)) for PS_EST_USAC (USAC.h:944-971). */
int ps_debug_limits(PsContext *ctx, int estimator, double minRatio, int H, int M, int32_t *out);
/* Runs blocks*256*perThread random (a0, a1, b) triples through the scoring kernel's shared-reciprocal
 * division and through the '/' operator; *mismatches must come back 0 (bitwise comparison). */
int ps_debug_fastdiv(PsContext *ctx, uint64_t seed, int blocks, int perThread, uint64_t *mismatches, uint64_t *tested);
/* The exact short forms of sqrt / reciprocal / shared-denominator quotients inside the hypothesis prologue (the float
 * Umeyama + Jacobi SVD of RANSAC.cpp:207-244 and the 4x4 inverse of :337-338) against sqrtf and '/', bit for bit; *mismatches
 * must come back 0.  mode 0: every float from 1.0f to +inf and beyond (elements = 0x40001000), 1: every float of [1, 2]
 * (0x00800001), 2: y / |y|, 3: 1 / d and u / d with d = sqrt(1 + u u), 4: nine numerators over one denominator (random). */
int ps_debug_mathcheck(PsContext *ctx, int mode, uint64_t seed, uint64_t elements, uint64_t *mismatches, uint64_t *tested);
/* After a scoring launch with option "score_stats" = 1: evaluations the fast kernel parked for the value-exact
 * code, and (hypothesis, match) evaluations it made in all (lanes of partially filled wavefronts included). */
int ps_debug_score_stats(PsContext *ctx, uint64_t *parked, uint64_t *evaluations);
/* All eight counters of the last scoring step: [0] parked, [1] evaluations made (with the staged scoring: what is left of
 * the complete sweep of H hypotheses x M matches); [2] / [3] trips (wavefront x two matches) of stage 1's pre-test that left
 * no lane / some lane to the value-exact test; [4..7] reserved (0). */
int ps_debug_score_stats_ex(PsContext *ctx, uint64_t *out8);
/* Staged scoring: hypotheses of every pair that survived stage 1 (out[0..P)) and stage 2 (out[P..2P)) of the last call
 * that was scored in stages: zeros if the LAST scoring step of the context was not staged, PS_ERR_BAD_ARG if P is not that
 * step's number of pairs (the counters are laid out with it; option "last_staged_pairs" says how many pairs that was). */
int ps_debug_stage_survivors(PsContext *ctx, int P, int32_t *out);
/* Staged scoring with the reordered match record: perm[P][cap] = for every pair, the match (index into its depth-valid
 * matches) at each position of the order stages 1+ swept; front[P] = leading positions whose matches every voting hypothesis
 * rejected and found far off (stage 1's pre-test range).  Entries beyond a pair's number of valid matches are unspecified.
 * PS_ERR_BAD_ARG unless the context's last scoring step was staged AND reordered with exactly this P and cap. */
int ps_debug_stage_order(PsContext *ctx, int P, int cap, int32_t *perm, int32_t *front);
/* Latency study (option "stamps" = 1): the shader-clock stamps (s_memtime) work-group 0 of kernels 2 and 4 of the last call
 * wrote into a private buffer: out16[0..3] = ps_crosscheck_prep (start, best[q] built, matches compacted + records written,
 * end), out16[4..9] = ps_select_refit (start, selection replayed, winner's inlier pass, refit, re-selection, end). */
int ps_debug_stamps(PsContext *ctx, uint64_t *out16);
/* sizeof() of the PODs as compiled into the library (layout check for foreign-language bindings). */
size_t ps_abi_sizeof_dmatch(void);
size_t ps_abi_sizeof_params(void);
size_t ps_abi_sizeof_config(void);
size_t ps_abi_sizeof_stats(void);
size_t ps_abi_sizeof_frameset(void);
size_t ps_abi_sizeof_results(void);
size_t ps_abi_sizeof_host_results(void);

#ifdef __cplusplus
}
#endif
#endif /* PUTSLAM_HIP_H_ */
