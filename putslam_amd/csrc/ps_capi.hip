// ps_capi.hip -- C ABI of the MI355X-native PUTSLAM front end (include/putslam_hip.h).
//
// Host side: one PsContext = one HIP stream + one grow-only scratch arena (the reference's
// Matcher/RANSAC objects are per-thread instances with no shared state, PUTSLAM.cpp:566,570).
// There is no CPU fallback anywhere in this file: every entry point launches the HIP kernels
// of ps_kernels.h or fails with a negative PsStatus.
#include "ps_kernels.h"
#include "ps_matcher_mfma.h"
#include "ps_score_fast.h"
#include "ps_score_euclid.h"
#include "ps_internal.h"

#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace psdev;

namespace {

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

constexpr int kWidePairs = 16; // batches of at most this many pairs run kernel 2 with 1024-thread work-groups
// where the staged scoring takes over from complete scoring (prepare_score): batch size in work units = pairs x
// (ceil(H / 256) - 1) x frame capacity, threshold = base + perRow x frame capacity (profiles/r05c/staged_crossover.txt)
struct StagedFrom {
    double base, perRow;
};
constexpr StagedFrom kStagedFromEuclidFixed = {7.8e5, 250.0}, kStagedFromReprojFixed = {1.3e6, 500.0},
                     kStagedFromEuclidAdaptive = {6.0e4, 0.0}, kStagedFromReprojAdaptive = {4.5e4, 0.0};
constexpr int kMaxTimed = 8;   // kernels timed per call
constexpr int kTimingRing = 128; // calls kept (HIP events on the launch stream around every kernel)

} // namespace

struct PsContext {
    int device = 0;
    hipStream_t own = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t handoff = nullptr; // recorded at every exit of the asynchronous call (ps_vo_pairs_device) once it has queued work:
                                  // a newly selected stream waits for it
    bool handoffPending = false;
    std::string err;
    char arch[64] = {0};
    // scratch arena (device)
    Buf keys, recA, recB, recC, recD, recE, recF, counts, mvalid, cmax, idxList, raw;
    Buf models; // [P][H][12] hypothesis models parked by kernel 3 for kernel 4 (small batches) and for the later stages
                // of the staged scoring (large batches)
    Buf survA, survB, survN; // staged scoring: survivor lists [P][H] of stages 1 / 2 and their counters [2][P]
    Buf validMask;           // staged scoring, stage 0 in two launches: which prefix hypotheses have a model, [P][prefix / 64]
    Buf frontRec;            // staged scoring: the pre-test operands of the all-reject front, [P][cap / 2][10] floats
    Buf prefInfo;            // staged scoring: per pair (best count, trip limit) of the prefix, written by ps_stage_reorder
    Buf recF2, permBuf;      // staged scoring: the reordered hot record of stages 1+ and position -> original match [P][cap]
    Buf dbgCnt; // {parked evaluations, evaluations} of the fast scoring kernel (option "score_stats")
    Buf xq; // FP4 image of every pair's query frame (ps_matcher_mfma.h)
    Buf recShadow; // (-DPS_STREAM_DIAG builds with PUTSLAM_HIP_DIAG_SHADOW_RECORDS=1: kernel 2's records once more; never read)
    Buf tabR, tabU;
    // staging for the host-pointer entry points (device)
    Buf sDesc, sNk, sMatches, sNumM, sMask, sPose, sStats, sMisc0, sMisc1, sMisc2;
    // cached stop tables
    int tabEstimator = -1, tabH = -1, tabRN = 0, tabUN = 0, tabIter0 = 0;
    double tabMinRatio = -1.0;
    float tabTiny = 0.0f;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev; // [kTimingRing][kMaxTimed][2], created when timing is first enabled
    long long timedCalls = 0;   // calls recorded since timing was (re)enabled
    int curCall = 0;            // ring slot of the call being recorded
    int nTimed = 0;             // highest timed slot + 1
    unsigned slotMask[kTimingRing] = {0}; // per kept call: which slots were recorded
    // tuning overrides (options "qsplit" / "msplit"; 0 = automatic)
    int forceQsplit = 0, forceMsplit = 0;
    // kernel variants (option "matcher"): 1 = FP4 MFMA matcher, 0 = integer VALU matcher,
    // 2 = by batch size (default): the MFMA form costs one more launch (the FP4 expansion), which a handful of pairs does
    // not earn back
    int matcher = 2;
    int matcherUsed = 1; // what the last matching call ran (1 MFMA, 0 VALU)
    // matrix-core matcher: 1 = the work-group expands its query tiles itself through LDS (default), 0 = round 2's form
    // with the FP4 image of the query frames written to HBM by a launch of its own (option "matcher_fused")
    int matcherFused = 1;
    // 1 = the decision-exact kernels (ps_score_fast.h / ps_score_euclid.h, default), 0 = the value-exact ps_ransac_score<MODE>
    // (the matrix-core scoring experiment of round 2 -- split-f16 transforms on v_mfma_f32_32x32x16_f16, correct, no gain on
    // the headline -- left the tree in round 4: profiles/variants/ps_score_mfma.h.txt)
    int scoreFast = 1;
    int scoreStats = 0;
    // pruned scoring (ps_score_euclid.h): 1 = large batches score the first 256 hypotheses of every pair completely and
    // abandon later hypotheses that cannot become records (default), 0 = every hypothesis is scored completely
    int prune = 1;
    // staged scoring: stages 1+ sweep the matches in the order ps_stage_reorder writes (those the prefix's best hypotheses
    // reject first: hypotheses end sooner, ps_score_fast.h).  1 = always, 0 = never (original order), 2 (default) = for the
    // fixed schedule only: under the adaptive schedules the trip limit usually ends the scoring inside the prefix, and the
    // extra launch (6 us per call) buys nothing (option "reorder")
    int reorder = 2;
    int reorderGran = 64; // stage cuts of the reprojection kernels: multiples of this (option "reorder_gran": 2 .. 64; finer cuts
                          // shorten stage 1 and lengthen stages 2 / 3 by as much, profiles/r03n)
    int genSplit = 1;    // staged scoring: stage 0 as two launches -- the prefix's models once, then the sweep with the match range
                         // split over twice as many work-groups (option "gensplit" = 0: one launch, every part repeats the
                         // sample -> SVD chain)
    int singleRest = 1;  // adaptive schedules, no reordering: one stage after the prefix instead of three (option "singlerest")
    int pretest = 1;     // stage 1: one-direction pre-test on the all-reject front (option "pretest")
    int listRsplit3 = 4; // option "list_r3": work-groups the last stage's match range is split over
    int listGroups2 = 0, listGroups3 = 0; // work-groups per pair of stages 2 / 3 (options list_g2 / list_g3; 0 = automatic)
    int forcePrefix = 0; // option "prefix": hypotheses stage 0 scores completely under the fixed schedule (64 .. 256; 0 = default)
    int bail = 1;        // option "bail": a pair whose prefix leaves nothing to abandon is swept in ONE stage (ps_stage_reorder)
    int streamCopyKernels = 1; // option "stream_copy_kernels": ps_vo_stream_push moves its frame in / results out with a copy
                               // kernel over mapped pinned memory (1) or with hipMemcpyAsync (0: rounds 1 - 4)
    int streamAhead = -1; // option "stream_ahead": places of the pipelined stream beyond one per lane (chunks queued behind the running
                          // ones); -1 = six places in all
    int modelRoomMiB = 0; // option "model_room_mib": room for the staged scoring's parked models (0 = 256 MiB adaptive / 2 GiB fixed)
    int reorderTop = 8, reorderMargin = kReorderMargin, reorderC2div = 16; // (options "reorder_top" / "reorder_margin" / "reorder_c2div")
    int stampsOn = 0; // option "stamps": kernels 2 and 4 record their phase boundaries (ps_debug_stamps)
    Buf stamps;
    // Every (re)allocation of an arena block bumps this: captured graphs (ps_vo_stream_push) carry the pointers of the
    // blocks they were captured with and are dropped when the generation they saw is no longer the current one.
    unsigned long long arenaGen = 0;
    // what the LAST scoring step left in the staged-scoring buffers (ps_debug_stage_survivors / ps_debug_stage_order):
    // pairs and capacity the survivor counters / the order were laid out with, 0 = that step was not staged / not reordered
    int stagedP = 0, stagedCap = 0, reorderedP = 0;
    int lastModelH = 0; // hypotheses per pair with a parked-model slot in the last scoring step (0: nothing parked)
    // "Nothing to gain" policy of the staged scoring (option "bail", Euclidean metrics, fixed schedule, batched calls):
    // ps_stage_reorder counts the pairs it replayed and those whose prefix leaves nothing to abandon (stage 1 sweeps every match:
    // hopeless data, no pair accepted); kernel 4 forwards the two counters to mapped host memory.  While the last observation
    // says "most pairs", the next calls OF THE SAME KIND score completely -- one launch, what the staged form costs on such data
    // is its extra launches, 14 - 19 % (profiles/r03p/data_sweep.txt) -- and every 16th call probes with the staged form again.
    // Results are bit-identical either way; the observation arrives asynchronously, so the switch lags the data by a call or two.
    // The state is kept per KIND of call -- (errorVersion, estimator, H, batch-size class, frame capacity, frame set) --, eight
    // kinds at a time: a context that alternates a mostly failing batch (loop-closure candidates) with good VO batches keeps
    // the staged form for the good ones (round 4 kept ONE flag per context: ADVICE round 4).  Adaptive schedules never drop the
    // staged form: under a long cap complete scoring is the minutes-long path.
    struct BailKind {
        int mode = -1, estimator = 0, H = 0, pclass = 0, cap = 0;
        const void *frames = nullptr;
        unsigned seen[2] = {0, 0};
        int hopeless = 0, calls = 0;
        unsigned long long used = 0; // (least recently used slot is recycled)
    };
    static constexpr int kBailKinds = 8;
    BailKind bailKinds[kBailKinds];
    unsigned long long bailClock = 0;
    int bailSlot = -1;            // slot of the last call the policy looked at
    Buf bailCnt;                  // device: [kBailKinds]{pairs replayed, pairs with nothing to gain}, monotonic per slot
    unsigned *bailHost = nullptr; // mapped host mirror [kBailKinds][2]
    unsigned *bailHostDev = nullptr;
    int hopeless = 0;             // state of the last call's kind (option "hopeless", read only)
    // The keys block is all-ones at rest: kernel 2 puts kNoKey back into every entry it reads, so the matcher forms that merge
    // their query splits with atomicMin need no clearing launch in front of them (a single pair paid a memset launch and its
    // gap for that on every call: 6 of 96 us).  keysCleanPtr / keysCleanBytes = the block and the leading bytes the invariant
    // holds for once the stream's queued work has drained; a fresh or regrown block is cleared once, completely.
    void *keysCleanPtr = nullptr;
    size_t keysCleanBytes = 0;
};

namespace {

int fail(PsContext *c, int code, const char *what, hipError_t e = hipSuccess)
{
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define PS_HIP(call)                                                  \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) return fail(ctx, PS_ERR_HIP, #call, e_); \
    } while (0)

int ensure(PsContext *ctx, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return PS_OK;
    // (a quarter more than asked for, 16 MiB at most: the large blocks -- counts and parked models under a long cap -- grow in
    // steps of the batch size, not by doubling)
    size_t want = bytes + (bytes / 4 < ((size_t)16 << 20) ? bytes / 4 : ((size_t)16 << 20)) + 256;
    if (b.p) {
        // The old block may still be referenced by work queued on the stream.
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, PS_ERR_HIP, "hipStreamSynchronize", e);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return fail(ctx, PS_ERR_ALLOC, "hipMalloc", e);
    b.cap = want;
    ctx->arenaGen++;
    return PS_OK;
}
#define PS_ENSURE(buf, bytes)                        \
    do {                                             \
        int rc_ = ensure(ctx, (buf), (bytes));       \
        if (rc_ != PS_OK) return rc_;                \
    } while (0)

// The matcher forms that merge with atomicMin start from an all-ones keys block (see PsContext::keysCleanPtr).
int keys_clean(PsContext *ctx, size_t bytes)
{
    if (ctx->keysCleanPtr == ctx->keys.p && ctx->keysCleanBytes >= bytes) return PS_OK;
    PS_HIP(hipMemsetAsync(ctx->keys.p, 0xFF, ctx->keys.cap, ctx->stream));
    ctx->keysCleanPtr = ctx->keys.p;
    ctx->keysCleanBytes = ctx->keys.cap;
    return PS_OK;
}

void release(Buf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// RANSAC::computeRANSACIteration (reference src/TransformEst/RANSAC.cpp:457-461) evaluated with the
// host's libm exactly as the reference evaluates it; the int conversion (UB there for huge
// quotients) saturates.
int ransac_iterations_host(double inlierRatio, double successProbability = 0.98, int numberOfPairs = 3)
{
    double v = std::log(1 - successProbability) / std::log(1 - std::pow(inlierRatio, numberOfPairs));
    if (!(v < 2147483647.0)) return INT_MAX;
    if (v < 0) return 0;
    return (int)v;
}

// USAC<T>::updateStandardStopping (reference include/putslam/USAC/USAC.h:944-971) as a function of the
// good-model probability, with confThreshold 0.99 and maxHypotheses 850000 (USAC_wrapper.cpp:66,70).
constexpr unsigned kUsacMaxHyp = 850000u;
unsigned usac_stopping_host(double prob_good_model)
{
    if (prob_good_model < DBL_EPSILON) return kUsacMaxHyp;
    if (1 - prob_good_model < DBL_EPSILON) return 1;
    double n = std::log(1 - 0.99) / std::log(1 - prob_good_model);
    return (unsigned)std::ceil(n);
}

// Threshold tables: the device never evaluates log/pow, it binary-searches these host-built
// (hence libm-identical) step positions.
void build_ransac_table(double minRatio, int H, std::vector<float> &tab, int &iter0, float &tiny)
{
    // For r below ~6e-6, 1 - r^3 rounds to 1, log(1) = +0 and the quotient is -inf: the reference's
    // int(-inf) is UB (INT_MIN on x86: the loop ends); ransac_iterations_host returns 0 there.  The
    // step positions below are searched above that range, and the range itself is passed to the
    // device as `tiny` (limit 0), so device and host agree for every float ratio.
    uint32_t tlo = 0, thi;
    {
        float probe = 1e-4f; // iterations(1e-4) saturates at INT_MAX
        memcpy(&thi, &probe, 4);
        while (thi - tlo > 1) {
            uint32_t mid = tlo + (thi - tlo) / 2;
            float mf;
            memcpy(&mf, &mid, 4);
            if (ransac_iterations_host((double)mf) == 0) tlo = mid; else thi = mid;
        }
        memcpy(&tiny, &tlo, 4); // largest float whose quotient is -inf
    }
    int itersMin = ransac_iterations_host(minRatio);
    int kcap = itersMin < H ? itersMin : H;
    if (kcap < 0) kcap = 0;
    tab.resize((size_t)kcap);
    uint32_t one;
    float onef = 1.0f;
    memcpy(&one, &onef, 4);
    uint32_t hi = one; // iterations(1.0) = 0 <= k for every k
    for (int k = 0; k < kcap; ++k) {
        // smallest float r in (0,1] with iterations(r) <= k.  The table is non-increasing in k, so the
        // previous entry (iterations <= k-1 <= k) is a valid upper end of the bracket.
        uint32_t lo = thi; // just above the -inf range: iterations saturate at INT_MAX > k
        while (hi - lo > 1) {
            uint32_t mid = lo + (hi - lo) / 2;
            float mf;
            memcpy(&mf, &mid, 4);
            if (ransac_iterations_host((double)mf) <= k) hi = mid; else lo = mid;
        }
        memcpy(&tab[(size_t)k], &hi, 4);
    }
    int i0 = ransac_iterations_host(0.20); // RANSAC ctor, RANSAC.cpp:30
    iter0 = i0 < H ? i0 : H;
}

void build_usac_table(int H, std::vector<double> &tab)
{
    int n = H < (int)kUsacMaxHyp ? H : (int)kUsacMaxHyp;
    tab.resize((size_t)(n > 0 ? n : 0));
    uint64_t oneb;
    double oned = 1.0;
    memcpy(&oneb, &oned, 8);
    uint64_t hiPrev = oneb;
    // The bisection runs where the rule is monotone.  Below p = 1.07e-9 the quotient exceeds 2^32 and the reference's
    // (unsigned) cast is undefined (x86 keeps the low 32 bits: pseudo-random in p); a probe in that region that happens to
    // land below the target sent the bisection to the region's edge and every later entry with it -- rounds 1 to 4 built
    // tables that were right up to entry 78 774 only, so that schedules that should run longer (fewer than 4 % inliers)
    // stopped there.  From 2^-29 = 1.86e-9 up the quotient is below 2.5e9: every target (< 850 000) lies above it.
    // Good-model probabilities below 1.07e-9 (three or four inliers among more than 1777 / 2820 matches) get the cap,
    // where the reference's cast is undefined and the oracle returns what x86 makes of it.
    const double pFloor = 1.862645149230957e-09; // 2^-29
    uint64_t floorb;
    memcpy(&floorb, &pFloor, 8);
    for (int k = 0; k < n; ++k) {
        unsigned target = (unsigned)k + 1u; // smallest p with stopping(p) <= k+1
        uint64_t lo = floorb, hi = hiPrev;  // stopping(2^-29) = 2.47e9 > target
        if (target >= kUsacMaxHyp) {
            tab[(size_t)k] = 0.0;
            continue;
        }
        {
            // The root of log(0.01) / log(1 - p) = target in closed form brackets the entry to a few thousand neighbouring
            // doubles; either end is taken only if the rule itself confirms it, so the bisection's invariant -- and its
            // result -- are those of the wide bracket (850 000 entries: 0.41 s instead of 0.75).
            const double ps = -std::expm1(std::log(1 - 0.99) / (double)target);
            const double a = ps * (1.0 - 1e-11), b = ps * (1.0 + 1e-11);
            uint64_t ab, bb;
            memcpy(&ab, &a, 8);
            memcpy(&bb, &b, 8);
            if (ab > lo && ab < hi && usac_stopping_host(a) > target) lo = ab;
            if (bb > lo && bb < hi && usac_stopping_host(b) <= target) hi = bb;
        }
        while (hi - lo > 1) {
            uint64_t mid = lo + (hi - lo) / 2;
            double md;
            memcpy(&md, &mid, 8);
            if (usac_stopping_host(md) <= target) hi = mid; else lo = mid;
        }
        memcpy(&tab[(size_t)k], &hi, 8);
        hiPrev = hi;
    }
}

double sq_bound_f64(double thr)
{
    // smallest double x >= 0 with sqrt(x) >= thr  (cv::norm(Point2f) < thr  <=>  dx^2+dy^2 < x)
    if (!(thr > 0.0)) return 0.0;
    if (thr > 1.3407807929942596e154) return INFINITY;
    double x = thr * thr;
    const double tiny = 4.9406564584124654e-324;
    if (!(x > 0.0)) x = tiny;
    while (x > tiny && std::sqrt(std::nextafter(x, 0.0)) >= thr) x = std::nextafter(x, 0.0);
    while (std::sqrt(x) < thr) x = std::nextafter(x, INFINITY);
    return x;
}

int prepare_tables(PsContext *ctx, int estimator, double minRatio, int H, SelectArgs &sa)
{
    sa.ransacTab = nullptr;
    sa.ransacTabN = 0;
    sa.usacTab = nullptr;
    sa.usacTabN = 0;
    if (estimator == PS_EST_FIXED) {
        sa.iter0 = H;
        return PS_OK;
    }
    if (!(ctx->tabEstimator == estimator && ctx->tabH == H && ctx->tabMinRatio == minRatio)) {
        if (estimator == PS_EST_RANSAC) {
            std::vector<float> tab;
            int iter0 = 0;
            build_ransac_table(minRatio, H, tab, iter0, ctx->tabTiny);
            PS_ENSURE(ctx->tabR, tab.size() * sizeof(float) + 4);
            if (!tab.empty()) {
                PS_HIP(hipMemcpyAsync(ctx->tabR.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice,
                                      ctx->stream));
                PS_HIP(hipStreamSynchronize(ctx->stream)); // tab goes out of scope
            }
            ctx->tabRN = (int)tab.size();
            ctx->tabIter0 = iter0;
        } else {
            std::vector<double> tab;
            build_usac_table(H, tab);
            PS_ENSURE(ctx->tabU, tab.size() * sizeof(double) + 8);
            if (!tab.empty()) {
                PS_HIP(hipMemcpyAsync(ctx->tabU.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice,
                                      ctx->stream));
                PS_HIP(hipStreamSynchronize(ctx->stream));
            }
            ctx->tabUN = (int)tab.size();
            ctx->tabIter0 = H < (int)kUsacMaxHyp ? H : (int)kUsacMaxHyp;
        }
        ctx->tabEstimator = estimator;
        ctx->tabH = H;
        ctx->tabMinRatio = minRatio;
    }
    sa.iter0 = ctx->tabIter0;
    if (estimator == PS_EST_RANSAC) {
        sa.ransacTab = (const float *)ctx->tabR.p;
        sa.ransacTabN = ctx->tabRN;
        sa.ransacTiny = ctx->tabTiny;
    } else {
        sa.usacTab = (const double *)ctx->tabU.p;
        sa.usacTabN = ctx->tabUN;
    }
    return PS_OK;
}

int effective_mode(int errorVersion)
{
    switch (errorVersion) {
    case PS_EUCLIDEAN_ERROR:
    case PS_REPROJECTION_ERROR:
    case PS_EUCLIDEAN_AND_REPROJECTION_ERROR:
    case PS_ADAPTIVE_ERROR:
        return errorVersion;
    default:
        // MAHALANOBIS is dead in the reference (RANSAC.cpp:301-303) and unknown values print
        // "incorrect error version" and score 0 (RANSAC.cpp:134-135): both score 0 here.
        return PS_MAHALANOBIS_ERROR;
    }
}

struct Plan {
    int mode = 0;
    int H = 0;        // hypotheses scored
    int minRun = 3;
    ScoreConsts sc{};
    FastConsts fc{};
    EuclidConsts ec{};
    PrepArgs pa{};
    SelectArgs sa{};
    ModelArgs ma{};
    int msplit = 1;   // work-groups the match range of kernel 3 is split over (prepare_score)
    bool genSplit = false; // staged scoring: stage 0 as two launches (models, then the sweep)
    bool genPlain = false; // complete scoring with a split match range, two pairs or more: the same two launches over [0, H)
    bool reorder = false; // staged scoring: stages 1+ sweep the reordered hot record (ps_stage_reorder)
    bool prune = false; // staged scoring: hypotheses [0, prefix) completely (msplit applies to it), the rest in pruned stages
    bool bailWatch = false; // this staged call feeds the "nothing to gain" policy (PsContext::bailHost)
    int bailSlot = 0;       // ... for this kind of call (PsContext::bailKinds)
    int prefix = 0;     // 256 (fixed schedule) or 64 (adaptive schedules)
    int lastStage = 0;  // staged scoring: 1 = ONE stage after the prefix (adaptive schedules without reordering), else kStages
};

int make_plan(PsContext *ctx, const PsRansacParams *prm, const PsRansacConfig *cfg, const float *K, int cap,
              int trainRange, Plan &pl)
{
    if (!prm || !cfg) return fail(ctx, PS_ERR_BAD_ARG, "null params/config");
    if (prm->usedPairs != 3) return fail(ctx, PS_ERR_UNSUPPORTED, "usedPairs must be 3");
    if (cfg->numHypotheses < 1 || cfg->numHypotheses > PS_MAX_HYPOTHESES)
        return fail(ctx, PS_ERR_BAD_ARG, "numHypotheses out of range");
    if (cfg->estimator < PS_EST_RANSAC || cfg->estimator > PS_EST_FIXED)
        return fail(ctx, PS_ERR_BAD_ARG, "unknown estimator");
    pl.mode = effective_mode(prm->errorVersion);
    float k[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (K) memcpy(k, K, sizeof k);
    pl.sc.fx = k[0]; pl.sc.fy = k[4]; pl.sc.cx = k[2]; pl.sc.cy = k[5];
    pl.sc.boundR = sq_bound_f64(prm->inlierThresholdReprojection);
    // float pre-filter band (see inlier_test): only when boundR is comfortably inside the float range
    pl.sc.bLo = -1.0f;      // never "surely inside"
    pl.sc.bHi = INFINITY;   // never "surely outside"
    if (pl.sc.boundR >= 1e-20 && pl.sc.boundR <= 1e30) {
        const double w = 4.76837158203125e-07; // 2^-21
        double lo = pl.sc.boundR * (1.0 - w), hi = pl.sc.boundR * (1.0 + w);
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = std::nextafterf(flo, 0.0f);
        if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
        pl.sc.bLo = flo;
        pl.sc.bHi = fhi;
    }
    // constants of the decision-exact fast path (ps_score_fast.h); directed roundings keep every bound on its safe side
    {
        FastConsts &fc = pl.fc;
        const double fm = std::fmax(std::fmax(std::fabs((double)k[0]), std::fabs((double)k[4])), 1.0);
        const double cm = std::fmax(std::fabs((double)k[2]), std::fabs((double)k[5]));
        const double bR = pl.sc.boundR;
        fc.enabled = (bR >= 1e-12 && bR <= 1e12 && fm <= 1e6 && cm <= 1e6) ? 1 : 0; // false for NaN
        if (fc.enabled) {
            const double T = std::sqrt(bR);
            auto up = [](double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; };
            auto down = [](double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; };
            fc.fmaxK = up(fm);
            fc.cmaxK = up(cm);
            fc.thrUp = up(T * (1.0 + 1e-5));
            fc.bIn0 = down(bR * (1.0 - 20.0 * 5.9604644775390625e-08));
            fc.cIn = up(2.0 * std::sqrt(2.0) * T * (1.0 + 1e-5));
            const double c = std::sqrt(2.0) * (1.0 + 1e-5);
            fc.thr2Up = up((double)fc.thrUp * (double)fc.thrUp);
            fc.cHi = up((2.0 * c + c * c) * (double)fc.thrUp);
        }
    }
    // constants of the decision-exact Euclidean path (ps_score_euclid.h): the threshold in the root domain, pushed
    // 6u (7u) to the safe side; errorVersion 0 works against the exact float bound B the value-exact code compares with,
    // errorVersion 4 against thr itself (the record is normalised by the match's depth)
    {
        EuclidConsts &ec = pl.ec;
        const double thr = prm->inlierThresholdEuclidean;
        ec.enabled = (thr >= 1e-10 && thr <= 1e10) ? 1 : 0; // false for NaN
        ec.tbLo = ec.tbHi = 0.0f;
        if (ec.enabled) {
            const double u = 5.9604644775390625e-08;
            auto up = [](double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; };
            auto down = [](double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; };
            if (pl.mode == PS_ADAPTIVE_ERROR) {
                ec.tbLo = down(thr * (1.0 - 7.0 * u));
                ec.tbHi = up(thr * (1.0 + 7.0 * u));
            } else {
                const double Tb = std::sqrt((double)sq_bound_f32(thr));
                ec.tbLo = down(Tb * (1.0 - 6.0 * u));
                ec.tbHi = up(Tb * (1.0 + 6.0 * u));
            }
        }
    }
    pl.pa.fx = k[0]; pl.pa.fy = k[4]; pl.pa.cx = k[2]; pl.pa.cy = k[5];
    pl.pa.thrE = prm->inlierThresholdEuclidean;
    pl.pa.mode = pl.mode;
    pl.pa.cap = cap;
    pl.sa.estimator = cfg->estimator;
    pl.sa.mode = pl.mode;
    pl.sa.cap = cap;
    pl.sa.minMatches = (cfg->estimator == PS_EST_USAC) ? 8 : prm->minimalNumberOfMatches;
    pl.sa.minRatio = prm->minimalInlierRatioThreshold;
    pl.sa.trainRange = trainRange;
    pl.minRun = pl.sa.minMatches > 3 ? pl.sa.minMatches : 3;
    // hypotheses that can ever be consumed by the sequential schedule
    int H = cfg->numHypotheses;
    if (cfg->estimator == PS_EST_RANSAC) {
        int a = ransac_iterations_host(0.20), b = ransac_iterations_host(prm->minimalInlierRatioThreshold);
        int most = a > b ? a : b;
        if (most < H) H = most;
        if (H < 1) H = 1;
    } else if (cfg->estimator == PS_EST_USAC) {
        if (H > (int)kUsacMaxHyp) H = (int)kUsacMaxHyp;
    }
    pl.H = H;
    pl.sa.H = H;
    int rc = prepare_tables(ctx, cfg->estimator, prm->minimalInlierRatioThreshold, H, pl.sa);
    if (rc != PS_OK) return rc;
    if (pl.sa.iter0 > H) pl.sa.iter0 = H;
    pl.ma.seed = cfg->seed;
    pl.ma.raw = nullptr;
    pl.ma.seedDev = nullptr;
    return PS_OK;
}

void tick(PsContext *ctx, int slot, bool stop)
{
    if (!ctx->timing || slot >= kMaxTimed || ctx->ev.empty()) return;
    (void)hipEventRecord(ctx->ev[((size_t)ctx->curCall * kMaxTimed + slot) * 2 + (stop ? 1 : 0)], ctx->stream);
    if (stop) {
        ctx->slotMask[ctx->curCall] |= 1u << slot;
        if (slot + 1 > ctx->nTimed) ctx->nTimed = slot + 1;
    }
}

// The Euclidean fast scoring kernel reads its own pair-interleaved record, kept in the block the reprojection kernels use
// for theirs (recF): kernel 2 writes one or the other.
bool with_euclid_fast(const PsContext *ctx, int mode)
{
    return ctx->scoreFast != 0 && (mode == PS_EUCLIDEAN_ERROR || mode == PS_ADAPTIVE_ERROR);
}

RecPtrs rec_ptrs(PsContext *ctx, int cap, int mode)
{
    RecPtrs r;
    r.G = with_euclid_fast(ctx, mode) ? (float *)ctx->recF.p : nullptr;
    r.A = (float4 *)ctx->recA.p; r.B = (float4 *)ctx->recB.p; r.C = (float4 *)ctx->recC.p; r.D = (int4 *)ctx->recD.p;
    r.E = (float4 *)ctx->recE.p;
    r.F = (float2 *)ctx->recF.p;
#ifdef PS_STREAM_DIAG
    r.S = (float4 *)ctx->recShadow.p; // (null unless PUTSLAM_HIP_DIAG_SHADOW_RECORDS=1: ensure_records)
#endif
    return r;
}

int pick_split(long long blocksWithout, int maxSplit, int minChunkOf, int total)
{
    // Few pairs and many CUs: split the inner range so that ~8 workgroups per CU are in flight.
    const long long want = 2048;
    if (blocksWithout >= want) return 1;
    long long s = (want + blocksWithout - 1) / (blocksWithout > 0 ? blocksWithout : 1);
    if (s > maxSplit) s = maxSplit;
    while (s > 1 && total / s < minChunkOf) --s;
    return (int)(s < 1 ? 1 : s);
}

template <int MODE>
void launch_score(PsContext *ctx, dim3 grid, const Plan &pl, int cap, int msplit)
{
    hipLaunchKernelGGL(ps_ransac_score<MODE>, grid, dim3(kBlock), 0, ctx->stream, (const float4 *)ctx->recA.p,
                       (const float4 *)ctx->recB.p, (const float4 *)ctx->recC.p, (const int32_t *)ctx->mvalid.p,
                       (const float2 *)ctx->cmax.p, pl.ma, pl.sc, pl.H, cap, pl.minRun, msplit, (int32_t *)ctx->counts.p);
}

// Decisions of the scoring stage that the preceding kernels need to know: how the match range is split (kernel 2 then
// clears the counts, instead of a memset launch) and whether kernel 3 parks its models for kernel 4.
// launches of the reprojection kernels with more work-groups than this use the packed match record (ps_score_fast.h, BIG)
unsigned big_limit(int mode) { return mode == PS_REPROJECTION_ERROR ? 1536u : 1280u; }

// complete = true: every hypothesis is scored completely whatever the batch size (ps_debug_ransac_counts returns the counts
// themselves: the staged scoring leaves lower bounds for abandoned hypotheses)
int prepare_score(PsContext *ctx, Plan &pl, int P, int cap, bool complete = false, bool adaptive = false, const void *dataKey = nullptr)
{
    const int H = pl.H;
    const int hb = (H + kBlock - 1) / kBlock;
    // pruned scoring: worth its second launch when the hypotheses beyond the prefix fill the chip by themselves
    const bool prunable = with_euclid_fast(ctx, pl.mode) ||
                          (pl.mode == PS_REPROJECTION_ERROR && ctx->scoreFast == 1) ||
                          (pl.mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR && ctx->scoreFast != 0);
    // (the later stages read the models back from HBM, 48 B per hypothesis: ps_score_fast.h)
    const size_t mbytes = (size_t)P * H * 12 * sizeof(float);
    pl.prefix = pl.sa.estimator == PS_EST_FIXED ? kPrefixFixed : kPrefixAdaptive;
    if (ctx->forcePrefix > 0 && pl.sa.estimator == PS_EST_FIXED) pl.prefix = ctx->forcePrefix; // (tuning knob)
    // (the staged form is six or seven dependent launches, 0.23 ms at the least with the reprojection kernels and 0.08 ms with
    // the Euclidean ones: it pays from about 48 / 16 pairs of H = 4096 on, profiles/r03p/small_batches.txt)
    // (adaptive schedules: the trip limit the prefix leaves cuts most of the work whatever the batch size)
    // Cost model (round 5): the staged form trades six or seven dependent launches for the evaluations it abandons, and an
    // evaluation's worth of work is a (hypothesis, match) pair -- so the batch is sized in work-groups of 256 hypotheses beyond the
    // first one x rows of the frame capacity (matches <= keypoints), and the crossover is where (1 - share of evaluations left)
    // of that pays for the launches: base + perRow x capacity, fitted on profiles/r05c/staged_crossover.txt over 500 ... 4000
    // keypoints x H = 1024 ... 16384 (rounds 3 - 4 counted work-groups only, with constants from 2000 keypoints that DESIGN
    // section 8 knew to be 2 - 14 % off on either side; adaptive schedules gain from far smaller batches than they were given).
    // Option "prune" = 2 takes the staged form whenever the kernels have it (tests; A/B).
    const double units = (double)P * (double)(hb - 1) * (double)cap;
    const StagedFrom sf = with_euclid_fast(ctx, pl.mode) ? (pl.sa.estimator == PS_EST_FIXED ? kStagedFromEuclidFixed : kStagedFromEuclidAdaptive)
                                                         : (pl.sa.estimator == PS_EST_FIXED ? kStagedFromReprojFixed : kStagedFromReprojAdaptive);
    const double stagedFrom = sf.base + sf.perRow * (double)cap;
    pl.prune = !complete && ctx->prune != 0 && prunable && H > kPrefixFixed && (ctx->prune == 2 || units >= stagedFrom);
    pl.bailWatch = false;
    pl.bailSlot = 0;
    const bool willReorder = ctx->reorder == 1 || (ctx->reorder == 2 && pl.sa.estimator == PS_EST_FIXED);
    if (adaptive && pl.prune && willReorder && ctx->bail != 0 && with_euclid_fast(ctx, pl.mode) && pl.sa.estimator == PS_EST_FIXED) {
        if (!ctx->bailHost) {
            const size_t nb = (size_t)PsContext::kBailKinds * 2 * sizeof(unsigned);
            if (hipHostMalloc((void **)&ctx->bailHost, nb, hipHostMallocMapped) == hipSuccess &&
                hipHostGetDevicePointer((void **)&ctx->bailHostDev, ctx->bailHost, 0) == hipSuccess) {
                memset(ctx->bailHost, 0, nb);
                if (ensure(ctx, ctx->bailCnt, nb) == PS_OK) (void)hipMemsetAsync(ctx->bailCnt.p, 0, nb, ctx->stream);
            } else {
                (void)hipGetLastError();
                ctx->bailHostDev = nullptr; // (no mapped memory here: the policy stays off)
            }
        }
        if (ctx->bailHostDev && ctx->bailCnt.p) {
            // the kind of this call: its own slot (a new kind takes the least recently used one and starts from "staged")
            int pclass = 0;
            for (int q = P; q > 1; q >>= 1) ++pclass;
            int slot = -1, lru = 0;
            for (int i = 0; i < PsContext::kBailKinds; ++i) {
                const PsContext::BailKind &b = ctx->bailKinds[i];
                if (b.mode == pl.mode && b.estimator == pl.sa.estimator && b.H == H && b.pclass == pclass && b.cap == cap && b.frames == dataKey) slot = i;
                if (b.used < ctx->bailKinds[lru].used) lru = i;
            }
            if (slot < 0) {
                slot = lru;
                PsContext::BailKind &b = ctx->bailKinds[slot];
                b.mode = pl.mode; b.estimator = pl.sa.estimator; b.H = H; b.pclass = pclass; b.cap = cap; b.frames = dataKey;
                b.hopeless = 0;
                b.calls = 0;
                // (the slot's counters are monotonic: what its previous kind left there is simply "seen" -- once it HAS landed:
                // a call of the previous kind may still be in flight, and counts arriving after the snapshot would read as an
                // observation of the new kind.  Recycling is rare -- more than eight kinds alternating on one context -- and
                // drains the stream first.)
                (void)hipStreamSynchronize(ctx->stream);
                b.seen[0] = ((volatile unsigned *)ctx->bailHost)[2 * slot];
                b.seen[1] = ((volatile unsigned *)ctx->bailHost)[2 * slot + 1];
            }
            PsContext::BailKind &b = ctx->bailKinds[slot];
            b.used = ++ctx->bailClock;
            const unsigned s0 = ((volatile unsigned *)ctx->bailHost)[2 * slot], s1 = ((volatile unsigned *)ctx->bailHost)[2 * slot + 1];
            if (s0 != b.seen[0]) { // a new observation of this kind has landed
                const unsigned d0 = s0 - b.seen[0], d1 = s1 - b.seen[1];
                const int was = b.hopeless;
                b.hopeless = (2ull * d1 > d0) ? 1 : 0;
                if (b.hopeless != was) b.calls = 0;
                b.seen[0] = s0;
                b.seen[1] = s1;
            }
            if (b.hopeless && (++b.calls & 15) != 0)
                pl.prune = false; // complete scoring while nothing can be abandoned; every 16th call looks again
            pl.bailWatch = pl.prune;
            pl.bailSlot = slot;
            ctx->bailSlot = slot;
            ctx->hopeless = b.hopeless;
        }
    }
    // Complete scoring of a batch under a long cap is the reference's own worst case times P (850 000 iterations over every
    // match, USAC_wrapper.cpp:70): minutes of GPU time behind an asynchronous call.  It is refused -- here, before the counts
    // block (4 bytes per pair and hypothesis: hundreds of GB for such a batch) is asked for; the staged scoring (option "prune",
    // the default) takes the same batch in milliseconds whenever the schedules end early.
    if (adaptive && !pl.prune && (double)P * (double)H * (double)cap > 2.0e14)
        return fail(ctx, PS_ERR_UNSUPPORTED, "complete scoring of this batch (pairs x hypotheses x matches > 2e14) would run for minutes: "
                                             "leave the staged scoring on (option \"prune\") or pass fewer pairs per call");
    PS_ENSURE(ctx->counts, (size_t)P * H * sizeof(int32_t));
    pl.msplit = pick_split((long long)P * (pl.prune ? 1 : hb), 32, 64, cap);
    pl.genSplit = pl.prune && ctx->genSplit != 0 && pl.msplit > 1;
    if (pl.genSplit) pl.msplit = pl.msplit * 2 < 32 ? pl.msplit * 2 : 32; // (the parts no longer repeat the prologue)
    if (ctx->forceMsplit > 0) pl.msplit = ctx->forceMsplit;
    if (pl.msplit <= 1) pl.genSplit = false;
    pl.pa.zeroCounts = pl.msplit > 1 ? (int32_t *)ctx->counts.p : nullptr;
    pl.pa.zeroH = pl.prune ? pl.prefix : H;
    pl.pa.zeroStride = H;
    pl.ma.models = nullptr;
    pl.ma.modelH = H;
    // adaptive schedules without reordering: ONE stage after the prefix (all matches, hypotheses below the trip limit only)
    pl.lastStage = (pl.sa.estimator != PS_EST_FIXED && !willReorder && ctx->singleRest != 0) ? 1 : kStages;
    if (pl.prune) {
        // Parked models in proportion to the work, not to the cap: under a long cap (USAC's 850 000, USAC_wrapper.cpp:70) the
        // schedules end after a handful of iterations (USAC.h:944-971), and 48 bytes per pair and cap entry would be 20 GB for
        // 499 pairs.  Only the leading hypotheses get a slot (256 MB under the adaptive schedules, 2 GiB under the fixed one,
        // whose stages do read the models back); a hypothesis beyond is swept in one piece by stage 1 and, if it wins, rebuilt
        // by kernel 4.
        const size_t room = ctx->modelRoomMiB > 0 ? ((size_t)ctx->modelRoomMiB << 20)
                                                  : (pl.sa.estimator == PS_EST_FIXED ? ((size_t)2 << 30) : ((size_t)256 << 20));
        if (mbytes > room) {
            long long fit = (long long)(room / ((size_t)P * 12 * sizeof(float))) & ~(long long)(kBlock - 1);
            pl.ma.modelH = (int)(fit < kPrefixFixed ? kPrefixFixed : fit);
        }
    }
    if ((P <= kWidePairs && mbytes <= ((size_t)64 << 20)) || pl.prune) {
        PS_ENSURE(ctx->models, (size_t)P * pl.ma.modelH * 12 * sizeof(float));
        pl.ma.models = (float *)ctx->models.p;
    }
    // Complete scoring with the match range split over msplit work-groups repeats the sample -> SVD chain in every part (16
    // pairs, H = 4096, errorVersion 1: 23 M of the launch's 45 M instructions).  From two pairs on the models are generated
    // once by a launch of their own, as stage 0 of the staged scoring does, and the sweep reads them back: 4 to 15 % of the
    // call for 2 ... 48 pairs (profiles/r04h/ab_gen_plain.txt).  A single pair keeps one launch: the chain's latency is all
    // there is, and a second launch costs 3 - 5 us more than it saves.
    pl.genPlain = false;
    {
        const bool fastKernels = with_euclid_fast(ctx, pl.mode) || (pl.mode == PS_REPROJECTION_ERROR && ctx->scoreFast == 1) ||
                                 (pl.mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR && ctx->scoreFast != 0);
#ifndef PS_GENPLAIN_FROM
#define PS_GENPLAIN_FROM 2
#endif
        if (!pl.prune && fastKernels && ctx->genSplit != 0 && pl.msplit > 1 && P >= PS_GENPLAIN_FROM && mbytes <= ((size_t)1 << 30)) {
            PS_ENSURE(ctx->models, mbytes);
            pl.ma.models = (float *)ctx->models.p;
            PS_ENSURE(ctx->validMask, (size_t)P * ((H + 63) / 64) * sizeof(unsigned long long));
            pl.genPlain = true;
        }
    }
    pl.pa.zeroSurvA = pl.pa.zeroSurvB = nullptr;
    {
        // which record form of the reprojection kernels the launches of this call read (kernel 2 writes only those:
        // 16 + 40 bytes per match otherwise): the packed form F for launches that fill the chip and every stage after
        // the prefix, the three-record form (A, B, E) for small ones -- the same decisions as in run_ransac_stage
        const bool fastRep = (pl.mode == PS_REPROJECTION_ERROR && ctx->scoreFast == 1) ||
                             (pl.mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR && ctx->scoreFast != 0);
        const unsigned bigLimit = big_limit(pl.mode);
        const bool firstBig = (unsigned)pl.msplit * (unsigned)P * (pl.prune ? 1u : (unsigned)hb) > bigLimit;
        pl.pa.skipF = !(fastRep && (pl.prune || firstBig));
        pl.pa.skipE = !(fastRep && !firstBig);
    }
    if (pl.prune) {
        if (pl.lastStage > 1) { // (only hypotheses with a model slot are ever listed)
            PS_ENSURE(ctx->survA, (size_t)P * pl.ma.modelH * sizeof(int32_t));
            PS_ENSURE(ctx->survB, (size_t)P * pl.ma.modelH * sizeof(int32_t));
        }
        PS_ENSURE(ctx->survN, (size_t)2 * P * sizeof(int32_t));
        if (pl.genSplit) PS_ENSURE(ctx->validMask, (size_t)P * ((pl.prefix + 63) / 64) * sizeof(unsigned long long));
        pl.reorder = ctx->reorder == 1 || (ctx->reorder == 2 && pl.sa.estimator == PS_EST_FIXED);
        if (pl.reorder) {
            const size_t n = (size_t)P * cap;
            PS_ENSURE(ctx->recF2, n * 40 > (size_t)P * ((cap + 1) / 2) * 64 ? n * 40 : (size_t)P * ((cap + 1) / 2) * 64);
            PS_ENSURE(ctx->permBuf, n * sizeof(int32_t));
            PS_ENSURE(ctx->prefInfo, (size_t)4 * P * sizeof(int32_t));
            PS_ENSURE(ctx->frontRec, (size_t)P * ((cap + 1) / 2) * 10 * sizeof(float));
        }
        pl.pa.zeroSurvA = (int32_t *)ctx->survN.p;       // cleared by kernel 2, one counter per pair and stage
        pl.pa.zeroSurvB = (int32_t *)ctx->survN.p + P;
    }
    ctx->lastModelH = pl.ma.models ? pl.ma.modelH : 0;
    ctx->stagedP = pl.prune ? P : 0;
    ctx->stagedCap = pl.prune ? cap : 0;
    ctx->reorderedP = (pl.prune && pl.reorder) ? P : 0;
    return PS_OK;
}

// Kernels 3 + 4 over records already in the arena.
int run_ransac_stage(PsContext *ctx, const Plan &pl, int P, int cap, const PsDMatch *dMatches,
                     const int32_t *dNumMatches, int matchStride, float *dPose, uint8_t *dMask, PsRansacStats *dStats,
                     int slot0)
{
    const int H = pl.H;
    PS_ENSURE(ctx->counts, (size_t)P * H * sizeof(int32_t));
    PS_ENSURE(ctx->idxList, (size_t)P * cap * sizeof(int32_t));
    const int hb = (H + kBlock - 1) / kBlock;
    const int msplit = pl.msplit; // counts were cleared by kernel 2 when the range is split (prepare_score)
    dim3 grid((unsigned)hb * (unsigned)msplit * (unsigned)P);
    tick(ctx, slot0, false);
    unsigned long long *dbgE = nullptr;
    if (ctx->scoreStats && with_euclid_fast(ctx, pl.mode)) {
        PS_ENSURE(ctx->dbgCnt, 8 * sizeof(unsigned long long));
        PS_HIP(hipMemsetAsync(ctx->dbgCnt.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        dbgE = (unsigned long long *)ctx->dbgCnt.p;
    }
    // Staged scoring (ps_score_fast.h): stage 0 = the prefix completely, stages 1 .. 3 = the rest with hypotheses abandoned
    // between the launches.  stage_args(i) describes launch i.
    // stages 2+: work-groups per pair (they loop over longer lists; after the reordered stage 1 few hypotheses are left)
    auto list_groups = [&](int stage) {
        const int listed = (pl.H < pl.ma.modelH ? pl.H : pl.ma.modelH) - pl.prefix; // hypotheses that can be on a survivor list
        const int all = ((listed > 0 ? listed : 1) + kBlock - 1) / kBlock;
        // (stage 3 by default: one looping group per eight possible ones -- 1 for H = 4096, where 21 of 3840 hypotheses per
        // pair are left, 48 for the stress configuration's H = 100 000, which one group swept in 4.3 ms instead of 1.0)
        const int auto3 = all / 8 > 1 ? all / 8 : 1;
        // (stage 2 by default: every possible group with the reprojection kernels -- a fifth of the hypotheses survives stage 1
        // there --, two looping groups with the Euclidean ones, whose counts saturate: hardly anything survives and the
        // launch is mostly work-groups that find nothing, 2 - 3 % of the step at every inlier share tried,
        // profiles/r04b/ab_euclid_list_groups.txt)
        const int auto2 = with_euclid_fast(ctx, pl.mode) ? (all / 8 > 2 ? all / 8 : 2) : 64; // (many hypotheses: lists can be long)
        const int want = pl.reorder ? (stage == 2 ? (ctx->listGroups2 > 0 ? ctx->listGroups2 : auto2)
                                                  : (ctx->listGroups3 > 0 ? ctx->listGroups3 : auto3)) : all;
        return want < all ? want : all;
    };
    // the last stage: work-groups its match range is split over (their counts add up in counts[]; a short survivor list
    // swept by one wavefront per SIMD pays the full latency of every record load, 0.25 us per match)
    auto list_rsplit = [&](int stage) { return (pl.reorder && stage == kStages) ? ctx->listRsplit3 : 1; };
    // stage 1's one-direction pre-test on the far-off front of the reordered record: the reprojection metrics (ps_score_fast.h)
    const bool usePretest = pl.reorder && ctx->pretest != 0 &&
                            (pl.mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR || pl.mode == PS_REPROJECTION_ERROR);
    const int lastStage = pl.lastStage;
    // Stage 1 of an adaptive schedule with a long cap (USAC's 850 000 = 3320 blocks of 256 hypotheses per pair, of which the
    // trip limit leaves a handful): one work-group per block is hundreds of thousands of work-groups that look at the limit and
    // leave -- 0.9 ms per 210 pairs.  From 64 blocks per pair on a fixed number of work-groups per pair walks the blocks and
    // stops at the first one beyond the limit (enough of them to fill the chip when the limit does stay at the cap).
    const int blocks1 = (pl.H - pl.prefix + kBlock - 1) / kBlock;
    int loopGroups = 0;
    if (pl.prune && pl.sa.estimator != PS_EST_FIXED && blocks1 > 64) {
        // (64 per pair at the least until round 5: 32 000 work-groups per 499 pairs that read the limit and leave, a third of the
        // scoring step under USAC's cap; 4096 in all still fill the chip when the limits do stay at the cap)
        loopGroups = (4096 + P - 1) / P;
        loopGroups = loopGroups < 4 ? 4 : loopGroups;
        loopGroups = loopGroups > blocks1 ? blocks1 : loopGroups;
    }
    auto stage_args = [&](int stage) {
        StageArgs st{};
        st.stage = stage;
        st.hBase = stage == 0 ? 0 : pl.prefix;
        st.hCount = stage == 0 ? pl.prefix : pl.H - pl.prefix;
        if (stage >= 2) st.hCount = list_groups(stage) * kBlock; // work-groups per pair that sweep the survivor list
        int32_t *nA = (int32_t *)ctx->survN.p, *nB = nA + P;
        if (stage == 1) { st.listOut = (int32_t *)ctx->survA.p; st.countOut = nA; }
        if (stage == 2) { st.listIn = (const int32_t *)ctx->survA.p; st.countIn = nA; st.listOut = (int32_t *)ctx->survB.p; st.countOut = nB; }
        if (stage == 3) { st.listIn = (const int32_t *)ctx->survB.p; st.countIn = nB; }
        if (stage >= 1 && pl.reorder) st.perm = (const int32_t *)ctx->permBuf.p;
        if (stage >= 1 && pl.reorder) st.prefInfo = (const int32_t *)ctx->prefInfo.p;
        if (stage == 1 && usePretest) st.frontRec = (const float2 *)ctx->frontRec.p;
        st.listStride = pl.ma.modelH;
        st.loopGroups = stage == 1 ? loopGroups : 0;
        st.single = lastStage == 1 ? 1 : 0;
        st.margin = ctx->reorderMargin;
        st.gran = with_euclid_fast(ctx, pl.mode) ? 64 : ctx->reorderGran;
        st.c2div = ctx->reorderC2div;
        return st;
    };
    // the hot record of stages 1+: reordered between stage 0 and stage 1 (ps_stage_reorder) unless the option is off
    const float2 *hotF = (const float2 *)(pl.reorder ? ctx->recF2.p : ctx->recF.p);
#define PS_LAUNCH_REORDER(MODE)                                                                                        \
    do {                                                                                                               \
        if (pl.reorder)                                                                                                \
            hipLaunchKernelGGL(ps_stage_reorder<MODE>, dim3((unsigned)P), dim3(kBlock), 0, ctx->stream,                \
                               (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p, (const float4 *)ctx->recC.p,  \
                               (const float2 *)ctx->recF.p, (const int32_t *)ctx->mvalid.p, pl.ma, pl.sc, pl.sa,       \
                               pl.prefix, ctx->reorderTop,                                                              \
                               (ctx->bail != 0 && with_euclid_fast(ctx, pl.mode)) ? 64 : 0, ctx->reorderMargin, pl.H, cap,  \
                               pl.minRun, (const int32_t *)ctx->counts.p, (float2 *)ctx->recF2.p,                       \
                               (int32_t *)ctx->permBuf.p, (int32_t *)ctx->prefInfo.p,                            \
                               usePretest ? (float2 *)ctx->frontRec.p : (float2 *)nullptr,                             \
                               pl.bailWatch ? (unsigned *)ctx->bailCnt.p + 2 * pl.bailSlot : (unsigned *)nullptr);     \
    } while (0)
    // stage 0 as two launches: models + validity, then the sweep reading them back
    auto stage0_args = [&](bool gen) {
        StageArgs st = stage_args(0);
        if (pl.genSplit) {
            st.validMask = (unsigned long long *)ctx->validMask.p;
            st.genOnly = gen ? 1 : 0;
        }
        return st;
    };
    StageArgs stAll{}; // the plain launch: every hypothesis of [0, H) completely
    stAll.hCount = pl.H;
    StageArgs stAllGen = stAll, stAllSweep = stAll; // the same as two launches (Plan::genPlain): models, then the sweep
    if (pl.genPlain) {
        stAllGen.validMask = stAllSweep.validMask = (unsigned long long *)ctx->validMask.p;
        stAllGen.genOnly = 1;
    }
#define PS_LAUNCH_EUCLID_ONE(MODE, KIND, ST, HCOUNT, MSPLIT)                                                           \
    hipLaunchKernelGGL((ps_ransac_score_euclid<MODE, KIND>),                                                           \
                       dim3((unsigned)(((HCOUNT) + kBlock - 1) / kBlock) * (unsigned)(MSPLIT) * (unsigned)P),          \
                       dim3(kBlock), 0, ctx->stream, (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p,         \
                       (KIND) >= 1 ? hotF : (const float2 *)ctx->recF.p, (const int32_t *)ctx->mvalid.p,               \
                       (const float2 *)ctx->cmax.p,                                                                    \
                       pl.ma, pl.sc, pl.ec, pl.sa, (ST), pl.H, cap, pl.minRun, (MSPLIT), (int32_t *)ctx->counts.p, dbgE)
#define PS_LAUNCH_EUCLID(MODE)                                                                                         \
    do {                                                                                                               \
        if (pl.prune) {                                                                                                \
            if (pl.genSplit) PS_LAUNCH_EUCLID_ONE(MODE, 0, stage0_args(true), pl.prefix, 1);                           \
            PS_LAUNCH_EUCLID_ONE(MODE, 0, stage0_args(false), pl.prefix, msplit);                                      \
            PS_LAUNCH_REORDER(MODE);                                                                                   \
            if (loopGroups > 0)                                                                                        \
                hipLaunchKernelGGL((ps_ransac_score_euclid<MODE, 1, true>), dim3((unsigned)loopGroups * (unsigned)P),  \
                                   dim3(kBlock), 0, ctx->stream, (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p, \
                                   hotF, (const int32_t *)ctx->mvalid.p, (const float2 *)ctx->cmax.p, pl.ma, pl.sc, pl.ec, \
                                   pl.sa, stage_args(1), pl.H, cap, pl.minRun, 1, (int32_t *)ctx->counts.p, dbgE);     \
            else                                                                                                       \
                PS_LAUNCH_EUCLID_ONE(MODE, 1, stage_args(1), pl.H - pl.prefix, 1);                                     \
            for (int sg = 2; sg <= lastStage; ++sg)                                                                    \
                PS_LAUNCH_EUCLID_ONE(MODE, 2, stage_args(sg), list_groups(sg) * kBlock, list_rsplit(sg));              \
        } else if (pl.genPlain) {                                                                                      \
            PS_LAUNCH_EUCLID_ONE(MODE, 0, stAllGen, pl.H, 1);                                                          \
            PS_LAUNCH_EUCLID_ONE(MODE, 0, stAllSweep, pl.H, msplit);                                                   \
        } else                                                                                                         \
            PS_LAUNCH_EUCLID_ONE(MODE, 0, stAll, pl.H, msplit);                                                        \
    } while (0)
    switch (pl.mode) {
    case PS_EUCLIDEAN_ERROR:
        if (with_euclid_fast(ctx, pl.mode))
            PS_LAUNCH_EUCLID(PS_EUCLIDEAN_ERROR);
        else
            launch_score<PS_EUCLIDEAN_ERROR>(ctx, grid, pl, cap, msplit);
        break;
    case PS_REPROJECTION_ERROR:
        if (ctx->scoreFast >= 1) {
            unsigned long long *dbg = nullptr;
            if (ctx->scoreStats) {
                PS_ENSURE(ctx->dbgCnt, 8 * sizeof(unsigned long long));
                PS_HIP(hipMemsetAsync(ctx->dbgCnt.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
                dbg = (unsigned long long *)ctx->dbgCnt.p;
            }
            // more work-groups than fit at once (256 CUs x 6): the build for big launches (ps_score_fast.h)
#define PS_LAUNCH_FAST_ONE(MODE, BIG, KIND, ST, HCOUNT, MSPLIT)                                                        \
    hipLaunchKernelGGL((ps_ransac_score_fast<MODE, BIG, KIND>),                                                        \
                       dim3((unsigned)(((HCOUNT) + kBlock - 1) / kBlock) * (unsigned)(MSPLIT) * (unsigned)P),          \
                       dim3(kBlock), 0, ctx->stream, (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p,         \
                       (const float4 *)ctx->recC.p,                                                                    \
                       ((KIND) == 1 && (ST).frontRec != nullptr) ? (const float4 *)(ST).frontRec : (const float4 *)ctx->recE.p, \
                       (KIND) >= 1 ? hotF : (const float2 *)ctx->recF.p,                                               \
                       (const int32_t *)ctx->mvalid.p, (const float2 *)ctx->cmax.p, pl.ma, pl.sc, pl.fc, pl.ec, pl.sa, \
                       (ST), pl.H, cap, pl.minRun, (MSPLIT), (int32_t *)ctx->counts.p, dbg)
    // more work-groups than fit at once: the build for big launches (ps_score_fast.h); staged: prefix, then the stages
#define PS_LAUNCH_FAST(MODE, BIGLIMIT)                                                                                 \
    do {                                                                                                               \
        if (pl.prune) {                                                                                                \
            if ((unsigned)msplit * (unsigned)P > (BIGLIMIT)) {                                                         \
                if (pl.genSplit) PS_LAUNCH_FAST_ONE(MODE, true, 0, stage0_args(true), pl.prefix, 1);                   \
                PS_LAUNCH_FAST_ONE(MODE, true, 0, stage0_args(false), pl.prefix, msplit);                              \
            } else {                                                                                                   \
                if (pl.genSplit) PS_LAUNCH_FAST_ONE(MODE, false, 0, stage0_args(true), pl.prefix, 1);                  \
                PS_LAUNCH_FAST_ONE(MODE, false, 0, stage0_args(false), pl.prefix, msplit);                             \
            }                                                                                                          \
            PS_LAUNCH_REORDER(MODE);                                                                                   \
            if (loopGroups > 0) {                                                                                      \
                const StageArgs st1 = stage_args(1);                                                                   \
                hipLaunchKernelGGL((ps_ransac_score_fast<MODE, true, 1, true>),                                        \
                                   dim3((unsigned)loopGroups * (unsigned)P), dim3(kBlock), 0, ctx->stream,             \
                                   (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p, (const float4 *)ctx->recC.p, \
                                   st1.frontRec != nullptr ? (const float4 *)st1.frontRec : (const float4 *)ctx->recE.p, \
                                   hotF, (const int32_t *)ctx->mvalid.p, (const float2 *)ctx->cmax.p, pl.ma, pl.sc, pl.fc, \
                                   pl.ec, pl.sa, st1, pl.H, cap, pl.minRun, 1, (int32_t *)ctx->counts.p, dbg);         \
            } else                                                                                                     \
                PS_LAUNCH_FAST_ONE(MODE, true, 1, stage_args(1), pl.H - pl.prefix, 1);                                 \
            for (int sg = 2; sg <= lastStage; ++sg)                                                                    \
                PS_LAUNCH_FAST_ONE(MODE, true, 2, stage_args(sg), list_groups(sg) * kBlock, list_rsplit(sg));          \
        } else if (grid.x > (BIGLIMIT)) {                                                                              \
            if (pl.genPlain) {                                                                                         \
                PS_LAUNCH_FAST_ONE(MODE, true, 0, stAllGen, pl.H, 1);                                                  \
                PS_LAUNCH_FAST_ONE(MODE, true, 0, stAllSweep, pl.H, msplit);                                           \
            } else                                                                                                     \
                PS_LAUNCH_FAST_ONE(MODE, true, 0, stAll, pl.H, msplit);                                                \
        } else {                                                                                                       \
            if (pl.genPlain) {                                                                                         \
                PS_LAUNCH_FAST_ONE(MODE, false, 0, stAllGen, pl.H, 1);                                                 \
                PS_LAUNCH_FAST_ONE(MODE, false, 0, stAllSweep, pl.H, msplit);                                          \
            } else                                                                                                     \
                PS_LAUNCH_FAST_ONE(MODE, false, 0, stAll, pl.H, msplit);                                               \
        }                                                                                                              \
    } while (0)
            PS_LAUNCH_FAST(PS_REPROJECTION_ERROR, big_limit(PS_REPROJECTION_ERROR));
        } else
            launch_score<PS_REPROJECTION_ERROR>(ctx, grid, pl, cap, msplit);
        break;
    case PS_EUCLIDEAN_AND_REPROJECTION_ERROR:
        if (ctx->scoreFast != 0) {
            unsigned long long *dbg = nullptr;
            if (ctx->scoreStats) {
                PS_ENSURE(ctx->dbgCnt, 8 * sizeof(unsigned long long));
                PS_HIP(hipMemsetAsync(ctx->dbgCnt.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
                dbg = (unsigned long long *)ctx->dbgCnt.p;
            }
            PS_LAUNCH_FAST(PS_EUCLIDEAN_AND_REPROJECTION_ERROR, big_limit(PS_EUCLIDEAN_AND_REPROJECTION_ERROR));
        } else
            launch_score<PS_EUCLIDEAN_AND_REPROJECTION_ERROR>(ctx, grid, pl, cap, msplit);
        break;
    case PS_ADAPTIVE_ERROR:
        if (with_euclid_fast(ctx, pl.mode))
            PS_LAUNCH_EUCLID(PS_ADAPTIVE_ERROR);
        else
            launch_score<PS_ADAPTIVE_ERROR>(ctx, grid, pl, cap, msplit);
        break;
    default: launch_score<PS_MAHALANOBIS_ERROR>(ctx, grid, pl, cap, msplit); break;
    }
#undef PS_LAUNCH_REORDER
#undef PS_LAUNCH_EUCLID
#undef PS_LAUNCH_EUCLID_ONE
#undef PS_LAUNCH_FAST
#undef PS_LAUNCH_FAST_ONE
    tick(ctx, slot0, true);
    PS_HIP(hipGetLastError());
    SelectArgs sa = pl.sa;
    // LDS of kernel 4: the two bitmaps over train indices, then up to 48 KiB for the refit's and the re-selection's operands
    // (what is left of the 64 KiB a work-group gets without an attribute, with room for the kernel's static 1 KiB)
    const size_t bitmapBytes = (((2 * (size_t)((sa.trainRange + 31) / 32)) + 3) & ~(size_t)3) * sizeof(uint32_t);
    const size_t stageRoom = ((size_t)63 << 10) - bitmapBytes;
    sa.stageCap = cap < 1536 ? cap : 1536;
    if ((size_t)sa.stageCap * 32 > stageRoom) sa.stageCap = (int)(stageRoom / 32);
    size_t lds = bitmapBytes + (size_t)sa.stageCap * 8 * sizeof(float);
    tick(ctx, slot0 + 1, false);
    // (1024-thread work-groups were measured for this kernel too: 48 us instead of 40 for a single pair)
    hipLaunchKernelGGL(ps_select_refit<kBlock>, dim3((unsigned)P), dim3(kBlock), lds, ctx->stream,
                       (const float4 *)ctx->recA.p, (const float4 *)ctx->recB.p, (const float4 *)ctx->recC.p,
                       (const int4 *)ctx->recD.p, (const int32_t *)ctx->mvalid.p, (const int32_t *)ctx->counts.p,
                       dMatches, dNumMatches, matchStride, pl.ma, pl.sc, sa, (int32_t *)ctx->idxList.p, dPose,
                       dMask, dStats, ctx->stampsOn ? (unsigned long long *)ctx->stamps.p : (unsigned long long *)nullptr,
                       (pl.bailWatch && pl.reorder) ? (const unsigned *)ctx->bailCnt.p + 2 * pl.bailSlot : (const unsigned *)nullptr,
                       (pl.bailWatch && pl.reorder) ? ctx->bailHostDev + 2 * pl.bailSlot : (unsigned *)nullptr);
    tick(ctx, slot0 + 1, true);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ensure_records(PsContext *ctx, size_t P, size_t cap)
{
    const size_t n = P * cap;
    PS_ENSURE(ctx->recA, n * 16);
    PS_ENSURE(ctx->recB, n * 16);
    PS_ENSURE(ctx->recC, n * 16);
    PS_ENSURE(ctx->recD, n * 16);
    PS_ENSURE(ctx->recE, n * 16);
    // 40 B per match (reprojection kernels) or up to 64 B per match pair (Euclidean kernel, ps_score_euclid.h)
    PS_ENSURE(ctx->recF, n * 40 > P * ((cap + 1) / 2) * 64 ? n * 40 : P * ((cap + 1) / 2) * 64);
#ifdef PS_STREAM_DIAG
    static const bool shadow = std::getenv("PUTSLAM_HIP_DIAG_SHADOW_RECORDS") != nullptr && std::atoi(std::getenv("PUTSLAM_HIP_DIAG_SHADOW_RECORDS")) != 0;
    if (shadow) PS_ENSURE(ctx->recShadow, n * 7 * 16);
#endif
    return PS_OK;
}

// Kernels 1 + 2 for P pairs of a device-resident frame set.
int run_match_stage(PsContext *ctx, const PsFrameSet &fs, const int32_t *dPairs, int P, bool withRecords,
                    const PrepArgs &paIn, PsDMatch *dMatches, int32_t *dNumMatches, int slot0)
{
    const int cap = fs.maxKpts;
    // frame strides (PsFrameSet, ABI 2): dense unless the frames keep descriptors and points together
    const size_t descStride = fs.descFrameStride ? fs.descFrameStride : (size_t)cap * 32;
    const size_t ptsStride = fs.ptsFrameStride ? fs.ptsFrameStride : (size_t)cap * 12;
    if ((descStride & 15) != 0 || descStride < (size_t)cap * 32 || (ptsStride & 3) != 0 || (fs.pts && ptsStride < (size_t)cap * 12) ||
        descStride / 4 > (size_t)INT_MAX || ptsStride / 4 > (size_t)INT_MAX || ((uintptr_t)fs.desc & 15) != 0)
        return fail(ctx, PS_ERR_BAD_ARG, "frame set: descFrameStride must be a multiple of 16 and >= maxKpts x 32 (desc 16-byte aligned), "
                                         "ptsFrameStride a multiple of 4 and >= maxKpts x 12");
    const int fstrideDw = (int)(descStride / 4);
    PrepArgs pa = paIn;
    pa.ptsStride = (int)(ptsStride / 4);
    PS_ENSURE(ctx->keys, (size_t)P * cap * sizeof(uint32_t));
    PS_ENSURE(ctx->mvalid, (size_t)P * sizeof(int32_t));
    PS_ENSURE(ctx->cmax, (size_t)P * sizeof(float2));
    if (withRecords) {
        int rc = ensure_records(ctx, (size_t)P, (size_t)cap);
        if (rc != PS_OK) return rc;
    }
    // by batch size: VALU sweep 1.85 us, MFMA sweep 0.43 us per 2000 x 2000 pair on a full chip, + ~7 us for the extra launch
    const bool useMfma = ctx->matcher == 1 || (ctx->matcher == 2 && (double)P * cap * cap > 2.0e7);
    ctx->matcherUsed = useMfma ? 1 : 0;
    const size_t keyBytes = (size_t)P * cap * sizeof(uint32_t);
    // from the matcher's launch until kernel 2 has been queued behind it the keys block is in use; kernel 2 leaves what it
    // read all-ones again, so a block that was clean before this call is clean after it
    const size_t cleanBefore = ctx->keysCleanPtr == ctx->keys.p ? ctx->keysCleanBytes : 0;
    struct KeysInUse {
        PsContext *c;
        size_t restore;
        bool done = false;
        ~KeysInUse() { c->keysCleanBytes = done ? restore : 0; }
    } keysInUse{ctx, cleanBefore};
    if (useMfma) {
        // matrix-core form: expand every pair's query frame to FP4 once, then the MFMA sweep
#ifndef PS_MFMA_TT
#define PS_MFMA_TT 4
#endif
        constexpr int TT = PS_MFMA_TT;
        const int tpf = (cap + kTileRows - 1) / kTileRows;
        const int groups = (tpf + kWavesPerWG * TT - 1) / (kWavesPerWG * TT);
        int qsplit = pick_split((long long)P * groups, tpf, 1, tpf);
        if (ctx->forceQsplit > 0) qsplit = ctx->forceQsplit < tpf ? ctx->forceQsplit : tpf;
        if (ctx->matcherFused) {
            // fused expansion: every work-group of a query split expands its own share of the query tiles, so a split only
            // pays when the groups do not fill the chip by themselves; the splits merge with atomicMin on an all-ones keys block
            // (kept so by kernel 2: keys_clean)
            if (ctx->forceQsplit <= 0 && (long long)P * groups >= 1024) qsplit = 1;
            if (qsplit > 1) {
                int rc = keys_clean(ctx, keyBytes);
                if (rc != PS_OK) return rc;
                keysInUse.restore = ctx->keysCleanBytes;
            }
            tick(ctx, 5, false);
            hipLaunchKernelGGL(ps_hamming_mfma_fused<TT>, dim3((unsigned)(groups * qsplit) * (unsigned)P), dim3(kBlock), 0,
                               ctx->stream, (const uint32_t *)fs.desc, fs.nkpts, dPairs, cap, fstrideDw, tpf, groups, qsplit,
                               (uint32_t *)ctx->keys.p);
            tick(ctx, 5, true);
            PS_HIP(hipGetLastError());
        } else {
        PS_ENSURE(ctx->xq, (size_t)P * tpf * kTileU4 * sizeof(uint4));
        int xchunks = tpf < 8 ? tpf : 8;
        if ((long long)P * xchunks < 1024) xchunks = tpf < 64 ? tpf : 64;
        tick(ctx, 4, false);
        hipLaunchKernelGGL(ps_expand_query_fp4, dim3((unsigned)xchunks * (unsigned)P), dim3(kBlock), 0, ctx->stream,
                           (const uint32_t *)fs.desc, fs.nkpts, dPairs, cap, fstrideDw, tpf, xchunks, (uint4 *)ctx->xq.p,
                           qsplit > 1 ? (uint32_t *)ctx->keys.p : (uint32_t *)nullptr); // also clears the keys
        tick(ctx, 4, true);
        tick(ctx, 5, false);
        hipLaunchKernelGGL(ps_hamming_mfma<TT>, dim3((unsigned)(groups * qsplit) * (unsigned)P), dim3(kBlock), 0,
                           ctx->stream, (const uint32_t *)fs.desc, fs.nkpts, dPairs, cap, fstrideDw, tpf, groups, qsplit,
                           (const uint4 *)ctx->xq.p, (uint32_t *)ctx->keys.p);
        tick(ctx, 5, true);
        PS_HIP(hipGetLastError());
        }
    } else {
        constexpr int TPL = 2;
        const int tiles = (cap + kBlock * TPL - 1) / (kBlock * TPL);
        // single pair: 500 work-groups of 16 query rows (84.1 us per pair against 84.6 with 64 splits of 31 rows and 85.6
        // with 250 of 8, profiles/r04g/ab_latency.txt)
        int qsplit = pick_split((long long)P * tiles, 128, 16, cap);
        if (ctx->forceQsplit > 0) qsplit = ctx->forceQsplit;
        if (qsplit > 1) {
            int rc = keys_clean(ctx, keyBytes);
            if (rc != PS_OK) return rc;
            keysInUse.restore = ctx->keysCleanBytes;
        }
        tick(ctx, slot0, false);
        hipLaunchKernelGGL(ps_hamming_nn<TPL>, dim3((unsigned)(tiles * qsplit) * (unsigned)P), dim3(kBlock), 0, ctx->stream,
                           (const uint4 *)fs.desc, fs.nkpts, dPairs, cap, fstrideDw / 4, tiles, qsplit, (uint32_t *)ctx->keys.p);
        tick(ctx, slot0, true);
        PS_HIP(hipGetLastError());
    }
    size_t lds = (size_t)cap * sizeof(uint32_t);
    tick(ctx, slot0 + 1, false);
    const bool wide = P <= kWidePairs; // a handful of pairs: 1024-thread work-groups shorten the per-pair serial walk
#define PS_LAUNCH_PREP(REC, BLK)                                                                                       \
    hipLaunchKernelGGL((ps_crosscheck_prep<REC, BLK>), dim3((unsigned)P), dim3(BLK), lds, ctx->stream, fs.pts, fs.nkpts, \
                       dPairs, (uint32_t *)ctx->keys.p, pa, dMatches, dNumMatches, rp,                                 \
                       (int32_t *)ctx->mvalid.p, (float2 *)ctx->cmax.p,                                                \
                       ctx->stampsOn ? (unsigned long long *)ctx->stamps.p : (unsigned long long *)nullptr)
    RecPtrs rp{};
    if (withRecords) {
        rp = rec_ptrs(ctx, cap, pa.mode);
        if (wide)
            PS_LAUNCH_PREP(true, 1024);
        else
            PS_LAUNCH_PREP(true, kBlock);
    } else {
        if (wide)
            PS_LAUNCH_PREP(false, 1024);
        else
            PS_LAUNCH_PREP(false, kBlock);
    }
#undef PS_LAUNCH_PREP
    tick(ctx, slot0 + 1, true);
    PS_HIP(hipGetLastError());
    keysInUse.done = true;
    return PS_OK;
}

void identity16(float *T)
{
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

// The host-pointer entry points are not part of the timed record.
struct TimingOff {
    PsContext *c;
    bool saved;
    explicit TimingOff(PsContext *ctx) : c(ctx), saved(ctx->timing) { c->timing = false; }
    ~TimingOff() { c->timing = saved; }
};

int bind(PsContext *ctx)
{
    if (!ctx) return PS_ERR_BAD_ARG;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, PS_ERR_HIP, "hipSetDevice", e);
    ctx->err.clear();
    return PS_OK;
}


// ---- options: one table for ps_context_set_option / ps_context_get_option / the PUTSLAM_HIP_* environment ----
// Every kernel variant and every tuning knob of the staged scoring is settable per context (the environment only
// supplies the initial value), so that tests can run each twin next to the default in one process.
struct OptDesc {
    const char *name;      // option name of ps_context_set_option
    const char *env;       // PUTSLAM_HIP_<env>: initial value (nullptr: none)
    int PsContext::*field;
    int lo, hi;            // accepted range
    const char *what;      // error text
    bool shape = false;    // a launch-shape / tuning knob of the staged scoring and the sweeps: every value gives the same results;
                           // kept for the parity tests and A/B measurements, addressed as "debug.<name>" (not part of the surface)
};
const OptDesc kOptions[] = {
    {"matcher", "MATCHER", &PsContext::matcher, 0, 2, "matcher: 0 (VALU), 1 (MFMA) or 2 (by batch size)"},
    {"matcher_fused", "MATCHER_FUSED", &PsContext::matcherFused, 0, 1, "matcher_fused: 0 or 1"},
    {"score", "SCORE", &PsContext::scoreFast, 0, 1, "score: 0 (value-exact kernels) or 1 (decision-exact kernels)"},
    {"score_stats", nullptr, &PsContext::scoreStats, 0, 1, "score_stats: 0 or 1"},
    {"prune", "PRUNE", &PsContext::prune, 0, 2, "prune: 0 (complete scoring), 1 (staged from the cost model's batch size on) or 2 (staged whenever possible)"},
    {"reorder", "REORDER", &PsContext::reorder, 0, 2, "reorder: 0, 1 or 2"},
    {"qsplit", "QSPLIT", &PsContext::forceQsplit, 0, 1024, "qsplit: 0 (automatic) .. 1024", true},
    {"msplit", "MSPLIT", &PsContext::forceMsplit, 0, 1024, "msplit: 0 (automatic) .. 1024", true},
    // staged scoring (ps_score_fast.h): the twins of its launch forms ...
    {"gensplit", "GENSPLIT", &PsContext::genSplit, 0, 1, "gensplit: 0 (stage 0 as one launch) or 1 (models, then the sweep)", true},
    {"singlerest", "SINGLEREST", &PsContext::singleRest, 0, 1, "singlerest: 0 (three stages) or 1 (one stage after the prefix, adaptive schedules)", true},
    {"pretest", "PRETEST", &PsContext::pretest, 0, 1, "pretest: 0 or 1 (stage 1's one-direction pre-test)", true},
    // ... and its tuning knobs
    {"list_r3", "LISTR3", &PsContext::listRsplit3, 1, 32, "list_r3: 1 .. 32 work-groups the last stage's match range is split over", true},
    {"list_g2", "LISTG2", &PsContext::listGroups2, 0, 64, "list_g2: 0 (automatic) .. 64 work-groups per pair of stage 2", true},
    {"list_g3", "LISTG3", &PsContext::listGroups3, 0, 512, "list_g3: 0 (automatic) .. 512 work-groups per pair of stage 3", true},
    {"prefix", "PREFIX", &PsContext::forcePrefix, 0, 256, "prefix: 0 (default) or 64, 128, 192, 256 hypotheses of stage 0 (fixed schedule)", true},
    {"reorder_top", "REORDER_TOP", &PsContext::reorderTop, 1, kReorderTopMax, "reorder_top: 1 .. 16 voters", true},
    {"reorder_margin", "REORDER_MARGIN", &PsContext::reorderMargin, 1, 4096, "reorder_margin: 1 .. 4096 matches", true},
    {"reorder_c2div", "REORDER_C2DIV", &PsContext::reorderC2div, 1, 64, "reorder_c2div: 1 .. 64", true},
    {"reorder_gran", "REORDER_GRAN", &PsContext::reorderGran, 2, 64, "reorder_gran: 2, 4, 8, 16, 32 or 64", true},
    {"bail", "BAIL", &PsContext::bail, 0, 1, "bail: 0 or 1 (pairs whose prefix leaves nothing to abandon are swept in one stage)"},
    {"stream_copy_kernels", "STREAM_COPY_KERNELS", &PsContext::streamCopyKernels, 0, 1, "stream_copy_kernels: 0 (hipMemcpyAsync) or 1 (copy kernels over mapped pinned memory)"},
    {"stream_ahead", "STREAM_AHEAD", &PsContext::streamAhead, -1, 8, "stream_ahead: -1 (automatic: six places in all) or 0 .. 8 chunks the pipelined stream takes beyond one per lane (queued on the lanes' streams)"},
    {"model_room_mib", "MODEL_ROOM_MIB", &PsContext::modelRoomMiB, 0, 65536, "model_room_mib: 0 (default) .. 65536 MiB for the staged scoring's parked models"},
};
const OptDesc *find_option(const char *name)
{
    const bool dbg = strncmp(name, "debug.", 6) == 0;
    if (dbg) name += 6;
    for (const OptDesc &o : kOptions)
        if (o.shape == dbg && strcmp(name, o.name) == 0) return &o;
    return nullptr;
}
// value checks beyond the range
bool option_value_ok(const OptDesc &o, int v)
{
    if (v < o.lo || v > o.hi) return false;
    if (strcmp(o.name, "prefix") == 0) return (v & 63) == 0;
    if (strcmp(o.name, "reorder_gran") == 0) return (v & (v - 1)) == 0;
    return true;
}
int parse_option_text(const OptDesc &o, const char *v)
{
    if (strcmp(o.name, "score") == 0) {
        if (strcmp(v, "exact") == 0) return 0;
        if (strcmp(v, "fast") == 0) return 1;
    }
    if (strcmp(o.name, "matcher") == 0) {
        if (strcmp(v, "valu") == 0) return 0;
        if (strcmp(v, "mfma") == 0) return 1;
        if (strcmp(v, "auto") == 0) return 2;
    }
    // (a number, all of it: "mfma" for an option that takes no such word, or a typo, is not 0 -- it is ignored)
    char *end = nullptr;
    const long x = std::strtol(v, &end, 10);
    if (end == v || *end != '\0' || x < INT_MIN || x > INT_MAX) return INT_MIN;
    return (int)x;
}

} // namespace

extern "C" {

int ps_abi_version(void) { return PS_ABI_VERSION; }

size_t ps_abi_sizeof_dmatch(void) { return sizeof(PsDMatch); }
size_t ps_abi_sizeof_params(void) { return sizeof(PsRansacParams); }
size_t ps_abi_sizeof_config(void) { return sizeof(PsRansacConfig); }
size_t ps_abi_sizeof_stats(void) { return sizeof(PsRansacStats); }
size_t ps_abi_sizeof_frameset(void) { return sizeof(PsFrameSet); }
size_t ps_abi_sizeof_results(void) { return sizeof(PsPairResults); }
size_t ps_abi_sizeof_host_results(void) { return sizeof(PsHostPairResults); }

int ps_context_create(int device, PsContext **out)
{
    if (!out) return PS_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return PS_ERR_NO_DEVICE; // no CPU fallback: fail loudly
    if (device < 0 || device >= n) return PS_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return PS_ERR_HIP;
    PsContext *ctx = new PsContext();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        strncpy(ctx->arch, prop.gcnArchName, sizeof(ctx->arch) - 1);
    }
    if (hipStreamCreateWithFlags(&ctx->own, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PS_ERR_HIP;
    }
    ctx->stream = ctx->own;
    for (const OptDesc &o : kOptions) { // initial values from the environment (out-of-range values are ignored)
        if (!o.env) continue;
        const std::string name = std::string("PUTSLAM_HIP_") + o.env;
        if (const char *v = std::getenv(name.c_str())) {
            const int x = parse_option_text(o, v);
            if (option_value_ok(o, x)) ctx->*(o.field) = x;
        }
    }
    // the cross-check kernel keeps best[q] for up to PS_MAX_KPTS queries in LDS (64 KiB of the CU's 160 KiB)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<true, 1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<false, 1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    *out = ctx;
    return PS_OK;
}

void ps_context_destroy(PsContext *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    Buf *all[] = {&ctx->keys, &ctx->recA, &ctx->recB, &ctx->recC, &ctx->recD, &ctx->recE, &ctx->recF, &ctx->recShadow, &ctx->models, &ctx->survA, &ctx->survB, &ctx->survN, &ctx->recF2, &ctx->permBuf, &ctx->prefInfo, &ctx->frontRec, &ctx->validMask, &ctx->stamps, &ctx->dbgCnt, &ctx->bailCnt, &ctx->counts, &ctx->mvalid,
                  &ctx->cmax, &ctx->idxList, &ctx->raw, &ctx->xq, &ctx->tabR, &ctx->tabU, &ctx->sDesc, &ctx->sNk,
                  &ctx->sMatches, &ctx->sNumM, &ctx->sMask, &ctx->sPose, &ctx->sStats,
                  &ctx->sMisc0, &ctx->sMisc1, &ctx->sMisc2};
    for (Buf *b : all) release(*b);
    for (hipEvent_t e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->bailHost) (void)hipHostFree(ctx->bailHost);
    if (ctx->handoff) (void)hipEventDestroy(ctx->handoff);
    if (ctx->own) (void)hipStreamDestroy(ctx->own);
    delete ctx;
}

int ps_context_set_stream(PsContext *ctx, void *s)
{
    if (!ctx) return PS_ERR_BAD_ARG;
    hipStream_t next = s ? (hipStream_t)s : ctx->own;
    if (next == ctx->stream) return PS_OK;
    int rc = bind(ctx);
    if (rc) return rc;
    // The scratch arena and the stop tables belong to the context, not to a stream: work queued on the new stream
    // must not start before the work already queued on the old one has finished with them.  The event was recorded
    // at the end of the last asynchronous call, on the stream that call ran on: the previous stream is not touched
    // here, so it may already have been destroyed by its owner.
    if (ctx->handoff && ctx->handoffPending) PS_HIP(hipStreamWaitEvent(next, ctx->handoff, 0));
    ctx->stream = next;
    return PS_OK;
}

void *ps_context_stream(PsContext *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int ps_context_device(const PsContext *ctx) { return ctx ? ctx->device : (int)PS_ERR_BAD_ARG; }

int ps_context_set_option(PsContext *ctx, const char *name, int value)
{
    if (!ctx || !name) return PS_ERR_BAD_ARG;
    if (strcmp(name, "stamps") == 0) {
        if (value != 0 && value != 1) return fail(ctx, PS_ERR_BAD_ARG, "stamps: 0 or 1");
        if (value) {
            int rc = bind(ctx);
            if (rc) return rc;
            PS_ENSURE(ctx->stamps, 16 * sizeof(unsigned long long));
            PS_HIP(hipMemsetAsync(ctx->stamps.p, 0, 16 * sizeof(unsigned long long), ctx->stream));
        }
        ctx->stampsOn = value;
        return PS_OK;
    }
    const OptDesc *o = find_option(name);
    if (!o) return fail(ctx, PS_ERR_BAD_ARG, "unknown option");
    if (!option_value_ok(*o, value)) return fail(ctx, PS_ERR_BAD_ARG, o->what);
    ctx->*(o->field) = value;
    if (strcmp(name, "bail") == 0) { // (setting the option also forgets what the policy has observed: every kind starts staged)
        for (PsContext::BailKind &b : ctx->bailKinds) b = PsContext::BailKind();
        ctx->hopeless = 0;
        ctx->bailSlot = -1;
    }
    return PS_OK;
}

int ps_context_get_option(const PsContext *ctx, const char *name)
{
    if (!ctx || !name) return PS_ERR_BAD_ARG;
    if (strcmp(name, "matcher_used") == 0) return ctx->matcherUsed;
    if (strcmp(name, "stamps") == 0) return ctx->stampsOn;
    if (strcmp(name, "last_staged_pairs") == 0) return ctx->stagedP;       // pairs of the last scoring step if it was staged, else 0
    if (strcmp(name, "hopeless") == 0) return ctx->hopeless;               // the "nothing to gain" policy's current state
    if (strcmp(name, "arena_mib") == 0) {                                  // device memory the context's scratch arena holds, MiB
        const Buf *all[] = {&ctx->keys, &ctx->recA, &ctx->recB, &ctx->recC, &ctx->recD, &ctx->recE, &ctx->recF, &ctx->models,
                            &ctx->survA, &ctx->survB, &ctx->survN, &ctx->recF2, &ctx->permBuf, &ctx->prefInfo, &ctx->frontRec,
                            &ctx->validMask, &ctx->stamps, &ctx->dbgCnt, &ctx->bailCnt, &ctx->counts, &ctx->mvalid, &ctx->cmax,
                            &ctx->idxList, &ctx->raw, &ctx->xq, &ctx->tabR, &ctx->tabU, &ctx->sDesc, &ctx->sNk, &ctx->sMatches,
                            &ctx->sNumM, &ctx->sMask, &ctx->sPose, &ctx->sStats, &ctx->sMisc0, &ctx->sMisc1, &ctx->sMisc2};
        size_t sum = 0;
        for (const Buf *b : all) sum += b->cap;
        return (int)((sum + (((size_t)1 << 20) - 1)) >> 20);
    }
    if (strcmp(name, "hw_queues_seen") == 0) return psi_hw_queues_seen();   // GPU_MAX_HW_QUEUES when the library was loaded (ps_env.cpp)
    if (strcmp(name, "last_model_slots") == 0) return ctx->lastModelH;     // hypotheses per pair with a parked-model slot, last scoring step
    if (strcmp(name, "last_reordered_pairs") == 0) return ctx->reorderedP; // ... and reordered (ps_stage_reorder ran)
    const OptDesc *o = find_option(name);
    return o ? ctx->*(o->field) : (int)PS_ERR_BAD_ARG;
}

int ps_context_synchronize(PsContext *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

const char *ps_last_error(const PsContext *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// (ps_internal.h: for the library's other translation units)
void psi_set_error(PsContext *ctx, const char *what)
{
    if (ctx) ctx->err = what ? what : "";
}

void psi_copy_options(PsContext *dst, const PsContext *src)
{
    if (!dst || !src) return;
    for (const OptDesc &o : kOptions) dst->*(o.field) = src->*(o.field);
}
const char *ps_device_arch(const PsContext *ctx) { return ctx ? ctx->arch : ""; }

int ps_context_enable_timing(PsContext *ctx, int enable)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (enable && ctx->ev.empty()) {
        ctx->ev.assign((size_t)kTimingRing * kMaxTimed * 2, nullptr);
        for (hipEvent_t &e : ctx->ev) PS_HIP(hipEventCreate(&e));
    }
    ctx->timing = enable != 0;
    ctx->timedCalls = 0;
    ctx->curCall = 0;
    ctx->nTimed = 0;
    memset(ctx->slotMask, 0, sizeof ctx->slotMask);
    return PS_OK;
}

int ps_last_kernel_times_ms(PsContext *ctx, float *ms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->timing || ctx->timedCalls == 0) return 0;
    PS_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ctx->nTimed; ++i) {
        float t = 0.f;
        size_t b = ((size_t)ctx->curCall * kMaxTimed + i) * 2;
        if (ctx->slotMask[ctx->curCall] & (1u << i)) PS_HIP(hipEventElapsedTime(&t, ctx->ev[b], ctx->ev[b + 1]));
        ms[i] = t;
    }
    return ctx->nTimed;
}

int ps_kernel_time_totals(PsContext *ctx, double *sum_ms, int *launches)
{
    int rc = bind(ctx);
    if (rc) return rc;
    for (int i = 0; i < kMaxTimed; ++i) {
        sum_ms[i] = 0.0;
        launches[i] = 0;
    }
    if (!ctx->timing || ctx->timedCalls == 0) return 0;
    PS_HIP(hipStreamSynchronize(ctx->stream));
    long long n = ctx->timedCalls < kTimingRing ? ctx->timedCalls : kTimingRing;
    for (long long c = 0; c < n; ++c)
        for (int i = 0; i < ctx->nTimed; ++i) {
            if (!(ctx->slotMask[c] & (1u << i))) continue;
            float t = 0.f;
            size_t b = ((size_t)c * kMaxTimed + i) * 2;
            PS_HIP(hipEventElapsedTime(&t, ctx->ev[b], ctx->ev[b + 1]));
            sum_ms[i] += t;
            launches[i] += 1;
        }
    return ctx->nTimed;
}

const char *ps_kernel_names(void)
{
    return "ps_hamming_nn\0ps_crosscheck_prep\0ps_ransac_score\0ps_select_refit\0ps_expand_query_fp4\0ps_hamming_mfma\0";
}

uint64_t ps_algorithmic_bytes(int nkpts, int matchesIn, int matchesValid, int H)
{
    // descriptors read + matches written + 3-D points read + match index
    // pairs read by RANSAC + sample triplets + inlier counts + final mask + pose.
    return 2ull * nkpts * 32 + 16ull * matchesIn + 2ull * nkpts * 12 + 8ull * matchesValid + 12ull * H + 4ull * H +
           (uint64_t)matchesValid + 64ull;
}

// ---------------------------------------------------------------------------------------------
int ps_match_hamming256(PsContext *ctx, const uint8_t *query, int nq, size_t qstep, const uint8_t *train, int nt,
                        size_t tstep, PsDMatch *out, int *nout)
{
    int rc = bind(ctx);
    if (rc) return rc;
    TimingOff toff(ctx);
    if (nout) *nout = 0;
    if (!out || !nout || nq < 0 || nt < 0 || (nq > 0 && !query) || (nt > 0 && !train))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_match_hamming256: bad argument");
    if ((nq > 0 && qstep < PS_DESC_BYTES) || (nt > 0 && tstep < PS_DESC_BYTES))
        return fail(ctx, PS_ERR_UNSUPPORTED, "descriptor rows must be 32 bytes (ORB/LDB); float descriptors are out of scope");
    if (nq > PS_MAX_KPTS || nt > PS_MAX_KPTS) return fail(ctx, PS_ERR_UNSUPPORTED, "more than PS_MAX_KPTS rows");
    if (nq == 0 || nt == 0) return PS_OK; // BFMatcher on an empty side: no matches
    int cap = nq > nt ? nq : nt;
    PS_ENSURE(ctx->sDesc, (size_t)2 * cap * 32);
    PS_ENSURE(ctx->sNk, 2 * sizeof(int32_t) + 2 * sizeof(int32_t));
    PS_ENSURE(ctx->sMatches, (size_t)cap * sizeof(PsDMatch));
    PS_ENSURE(ctx->sNumM, sizeof(int32_t));
    uint8_t *dDesc = (uint8_t *)ctx->sDesc.p;
    PS_HIP(hipMemcpy2DAsync(dDesc, 32, query, qstep, 32, (size_t)nq, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpy2DAsync(dDesc + (size_t)cap * 32, 32, train, tstep, 32, (size_t)nt, hipMemcpyHostToDevice,
                            ctx->stream));
    int32_t hostMeta[4] = {nq, nt, 0, 1}; // nkpts[2], pair (0,1)
    PS_HIP(hipMemcpyAsync(ctx->sNk.p, hostMeta, sizeof hostMeta, hipMemcpyHostToDevice, ctx->stream));
    PsFrameSet fs;
    fs.desc = dDesc;
    fs.pts = nullptr;
    fs.nkpts = (const int32_t *)ctx->sNk.p;
    fs.numFrames = 2;
    fs.maxKpts = cap;
    fs.descFrameStride = fs.ptsFrameStride = 0;
    PrepArgs pa{};
    pa.cap = cap;
    rc = run_match_stage(ctx, fs, (const int32_t *)ctx->sNk.p + 2, 1, false, pa, (PsDMatch *)ctx->sMatches.p,
                         (int32_t *)ctx->sNumM.p, 0);
    if (rc) return rc;
    int32_t n = 0;
    PS_HIP(hipMemcpyAsync(&n, ctx->sNumM.p, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        PS_HIP(hipMemcpyAsync(out, ctx->sMatches.p, (size_t)n * sizeof(PsDMatch), hipMemcpyDeviceToHost, ctx->stream));
        PS_HIP(hipStreamSynchronize(ctx->stream));
    }
    *nout = n;
    return PS_OK;
}

// ---------------------------------------------------------------------------------------------
static int ransac_host_entry(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                             const float *prev, int nprev, const float *cur, int ncur, const PsDMatch *matches, int m,
                             float *pose, PsDMatch *inliers, int *ninl, uint8_t *mask, PsRansacStats *stats,
                             int32_t *countsOut)
{
    int rc = bind(ctx);
    if (rc) return rc;
    TimingOff toff(ctx);
    PsRansacStats st;
    memset(&st, 0, sizeof st);
    st.bestHypothesis = -1;
    st.numMatchesIn = m > 0 ? m : 0;
    st.pointInlierRatio = NAN;
    if (pose) identity16(pose);
    if (ninl) *ninl = 0;
    if (stats) *stats = st;
    if (!pose || !ninl || m < 0 || nprev < 0 || ncur < 0 || (m > 0 && (!matches || !prev || !cur || !inliers)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_ransac_rigid3d: bad argument");
    if (mask && m > 0) memset(mask, 0, (size_t)m);
    if (m > (1 << 22)) return fail(ctx, PS_ERR_UNSUPPORTED, "too many matches");
    for (int i = 0; i < m; ++i)
        if (matches[i].queryIdx < 0 || matches[i].queryIdx >= nprev || matches[i].trainIdx < 0 ||
            matches[i].trainIdx >= ncur)
            return fail(ctx, PS_ERR_BAD_ARG, "match index out of range");
    const int cap = m > 0 ? m : 1;
    const int trainRange = ncur <= 65536 ? ncur : 0;
    Plan pl;
    rc = make_plan(ctx, params, cfg, K, cap, trainRange, pl);
    if (rc) return rc;
    if (cfg->sampleIdx) {
        PS_ENSURE(ctx->raw, (size_t)pl.H * 3 * sizeof(uint32_t));
        PS_HIP(hipMemcpyAsync(ctx->raw.p, cfg->sampleIdx, (size_t)pl.H * 3 * sizeof(uint32_t), hipMemcpyHostToDevice,
                              ctx->stream));
        pl.ma.raw = (const uint32_t *)ctx->raw.p;
    }
    rc = prepare_score(ctx, pl, 1, cap, countsOut != nullptr);
    if (rc) return rc;
    PS_ENSURE(ctx->sMisc0, (size_t)(nprev > 0 ? nprev : 1) * 12);
    PS_ENSURE(ctx->sMisc1, (size_t)(ncur > 0 ? ncur : 1) * 12);
    PS_ENSURE(ctx->sMatches, (size_t)cap * sizeof(PsDMatch));
    PS_ENSURE(ctx->sNumM, sizeof(int32_t));
    PS_ENSURE(ctx->sMask, (size_t)cap);
    PS_ENSURE(ctx->sPose, 16 * sizeof(float));
    PS_ENSURE(ctx->sStats, sizeof(PsRansacStats));
    PS_ENSURE(ctx->mvalid, sizeof(int32_t));
    PS_ENSURE(ctx->cmax, sizeof(float2));
    rc = ensure_records(ctx, 1, (size_t)cap);
    if (rc) return rc;
    if (nprev > 0) PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, prev, (size_t)nprev * 12, hipMemcpyHostToDevice, ctx->stream));
    if (ncur > 0) PS_HIP(hipMemcpyAsync(ctx->sMisc1.p, cur, (size_t)ncur * 12, hipMemcpyHostToDevice, ctx->stream));
    if (m > 0)
        PS_HIP(hipMemcpyAsync(ctx->sMatches.p, matches, (size_t)m * sizeof(PsDMatch), hipMemcpyHostToDevice,
                              ctx->stream));
    int32_t mm = m;
    PS_HIP(hipMemcpyAsync(ctx->sNumM.p, &mm, sizeof mm, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ps_prep_from_matches, dim3(1), dim3(kBlock), 0, ctx->stream, (const float *)ctx->sMisc0.p,
                       (const float *)ctx->sMisc1.p, (const PsDMatch *)ctx->sMatches.p, m, pl.pa, rec_ptrs(ctx, cap, pl.pa.mode),
                       (int32_t *)ctx->mvalid.p, (float2 *)ctx->cmax.p);
    PS_HIP(hipGetLastError());
    rc = run_ransac_stage(ctx, pl, 1, cap, (const PsDMatch *)ctx->sMatches.p, (const int32_t *)ctx->sNumM.p, cap,
                          (float *)ctx->sPose.p, (uint8_t *)ctx->sMask.p, (PsRansacStats *)ctx->sStats.p, 2);
    if (rc) return rc;
    std::vector<uint8_t> hmask((size_t)cap);
    PS_HIP(hipMemcpyAsync(pose, ctx->sPose.p, 16 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipMemcpyAsync(&st, ctx->sStats.p, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    if (m > 0) PS_HIP(hipMemcpyAsync(hmask.data(), ctx->sMask.p, (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (countsOut) {
        int32_t mv = 0;
        PS_HIP(hipMemcpyAsync(&mv, ctx->mvalid.p, sizeof mv, hipMemcpyDeviceToHost, ctx->stream));
        PS_HIP(hipStreamSynchronize(ctx->stream));
        if (mv >= pl.minRun)
            PS_HIP(hipMemcpyAsync(countsOut, ctx->counts.p, (size_t)pl.H * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  ctx->stream));
        else
            memset(countsOut, 0, (size_t)pl.H * sizeof(int32_t));
    }
    PS_HIP(hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (int i = 0; i < m; ++i)
        if (hmask[(size_t)i]) inliers[n++] = matches[i];
    *ninl = n;
    if (mask && m > 0) memcpy(mask, hmask.data(), (size_t)m);
    if (trainRange == 0 && m > 0) {
        // train indices beyond the device bitmap: RANSAC::pointInlierRatio (RANSAC.h:56-66) on the host lists
        std::vector<uint8_t> seen((size_t)ncur, 0);
        int ua = 0, ui = 0;
        for (int i = 0; i < m; ++i) {
            uint8_t &s = seen[(size_t)matches[i].trainIdx];
            if (!(s & 1)) { s |= 1; ++ua; }
            if (hmask[(size_t)i] && !(s & 2)) { s |= 2; ++ui; }
        }
        st.pointInlierRatio = (double)ui / (double)ua;
    }
    if (stats) *stats = st;
    return PS_OK;
}

int ps_ransac_rigid3d(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const float *prev, int nprev, const float *cur, int ncur, const PsDMatch *matches, int m,
                      float *pose, PsDMatch *inliers, int *ninl, uint8_t *mask, PsRansacStats *stats)
{
    return ransac_host_entry(ctx, params, cfg, K, prev, nprev, cur, ncur, matches, m, pose, inliers, ninl, mask, stats,
                             nullptr);
}

// Diagnostic twin of ps_ransac_rigid3d that also returns the per-hypothesis inlier counts the
// scoring kernel produced (length = hypotheses actually scored, returned through *numScored).
int ps_debug_ransac_counts(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                           const float *prev, int nprev, const float *cur, int ncur, const PsDMatch *matches, int m,
                           int32_t *counts, int *numScored)
{
    if (!cfg || !counts || !numScored) return PS_ERR_BAD_ARG;
    std::vector<PsDMatch> inl((size_t)(m > 0 ? m : 1));
    float pose[16];
    int ninl = 0;
    PsRansacStats st;
    // the number of scored hypotheses follows the same rule as make_plan
    int H = cfg->numHypotheses;
    if (cfg->estimator == PS_EST_RANSAC && params) {
        int a = ransac_iterations_host(0.20), b = ransac_iterations_host(params->minimalInlierRatioThreshold);
        int most = a > b ? a : b;
        if (most < H) H = most;
        if (H < 1) H = 1;
    } else if (cfg->estimator == PS_EST_USAC && H > (int)kUsacMaxHyp)
        H = (int)kUsacMaxHyp;
    *numScored = H;
    return ransac_host_entry(ctx, params, cfg, K, prev, nprev, cur, ncur, matches, m, pose, inl.data(), &ninl, nullptr,
                             &st, counts);
}

// Diagnostic: bitwise comparison of the shared-reciprocal division with the '/' operator on random inputs.
int ps_debug_fastdiv(PsContext *ctx, uint64_t seed, int blocks, int perThread, uint64_t *mismatches, uint64_t *tested)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!mismatches || !tested || blocks < 1 || perThread < 1) return PS_ERR_BAD_ARG;
    PS_ENSURE(ctx->sMisc2, 16);
    PS_HIP(hipMemsetAsync(ctx->sMisc2.p, 0, 16, ctx->stream));
    hipLaunchKernelGGL(ps_fastdiv_check, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, seed, perThread,
                       (unsigned long long *)ctx->sMisc2.p, (unsigned long long *)ctx->sMisc2.p + 1);
    PS_HIP(hipGetLastError());
    uint64_t h[2] = {0, 0};
    PS_HIP(hipMemcpyAsync(h, ctx->sMisc2.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches = h[0];
    *tested = h[1];
    return PS_OK;
}

// Diagnostic: the exact short forms of the square root / reciprocal / shared-denominator quotients (ps_device_math.h)
// against sqrtf and '/', bit for bit (modes: see ps_mathcheck).  `elements` = how many elements to test: modes 0 / 1 walk
// consecutive float patterns from 1.0f (0x40001000 of them reach past +inf, 0x00800001 cover [1, 2]), modes 2 .. 4 draw
// random operands.
int ps_debug_mathcheck(PsContext *ctx, int mode, uint64_t seed, uint64_t elements, uint64_t *mismatches, uint64_t *tested)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!mismatches || !tested || mode < 0 || mode > 4 || elements < 1 || elements > ((uint64_t)1 << 34)) return PS_ERR_BAD_ARG;
    PS_ENSURE(ctx->sMisc2, 16);
    PS_HIP(hipMemsetAsync(ctx->sMisc2.p, 0, 16, ctx->stream));
    const int perThread = 1024;
    const uint64_t threads = (elements + perThread - 1) / perThread;
    const unsigned blocks = (unsigned)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ps_mathcheck, dim3(blocks), dim3(kBlock), 0, ctx->stream, mode, seed, perThread,
                       (unsigned long long)elements, (unsigned long long *)ctx->sMisc2.p, (unsigned long long *)ctx->sMisc2.p + 1);
    PS_HIP(hipGetLastError());
    uint64_t h[2] = {0, 0};
    PS_HIP(hipMemcpyAsync(h, ctx->sMisc2.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches = h[0];
    *tested = h[1];
    return PS_OK;
}

// Diagnostic: how many evaluations the last fast scoring launch parked for the value-exact code (option "score_stats").
int ps_debug_score_stats(PsContext *ctx, uint64_t *parked, uint64_t *evaluations)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!parked || !evaluations) return PS_ERR_BAD_ARG;
    *parked = *evaluations = 0;
    if (!ctx->dbgCnt.p) return PS_OK;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PS_HIP(hipMemcpyAsync(h, ctx->dbgCnt.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    *parked = h[0];
    *evaluations = h[1];
    return PS_OK;
}

// Diagnostic: all eight counters of the last scoring step (option "score_stats"): [0] evaluations handed to the
// value-exact code, [1] (hypothesis, match) evaluations made (lanes of partially filled wavefronts included; with the
// staged scoring this is what is left of the complete sweep); the rest reserved.
int ps_debug_score_stats_ex(PsContext *ctx, uint64_t *out8)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out8) return PS_ERR_BAD_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!ctx->dbgCnt.p) return PS_OK;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PS_HIP(hipMemcpyAsync(h, ctx->dbgCnt.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) out8[i] = h[i];
    return PS_OK;
}

// Diagnostic: how many hypotheses of every pair survived stages 1 and 2 of the LAST staged scoring step (zeros if that
// call was not staged); out = [2][P].
int ps_debug_stage_survivors(PsContext *ctx, int P, int32_t *out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out || P <= 0) return PS_ERR_BAD_ARG;
    memset(out, 0, (size_t)2 * P * sizeof(int32_t));
    if (ctx->stagedP == 0) return PS_OK; // the last scoring step was not staged
    if (P != ctx->stagedP) // (the counters are laid out [2][P] with the P of the call that wrote them)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_debug_stage_survivors: the last staged scoring step had a different number of pairs");
    if (!ctx->survN.p || ctx->survN.cap < (size_t)2 * P * sizeof(int32_t)) return PS_OK;
    PS_HIP(hipMemcpyAsync(out, ctx->survN.p, (size_t)2 * P * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

// Diagnostic: the order in which stages 1+ of the LAST staged, reordered scoring step swept every pair's matches
// (ps_stage_reorder): perm[p][i] = match of the original record arrays at position i, front[p] = how many of the leading
// positions hold matches every voter rejected and found far off.  PS_ERR_BAD_ARG if the context holds no such order for
// P pairs of `cap` matches.
int ps_debug_stage_order(PsContext *ctx, int P, int cap, int32_t *perm, int32_t *front)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!perm || !front || P <= 0 || cap <= 0) return PS_ERR_BAD_ARG;
    if (P != ctx->reorderedP || cap != ctx->stagedCap || !ctx->permBuf.p ||
        ctx->permBuf.cap < (size_t)P * cap * sizeof(int32_t) || !ctx->prefInfo.p ||
        ctx->prefInfo.cap < (size_t)4 * P * sizeof(int32_t))
        return fail(ctx, PS_ERR_BAD_ARG, "no reordered scoring step of that size in this context");
    std::vector<int32_t> info((size_t)4 * P);
    PS_HIP(hipMemcpyAsync(perm, ctx->permBuf.p, (size_t)P * cap * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipMemcpyAsync(info.data(), ctx->prefInfo.p, info.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < P; ++p) front[p] = info[(size_t)4 * p + 2];
    return PS_OK;
}

// Diagnostic: the shader-clock stamps kernels 2 and 4 of the LAST call wrote (option "stamps"): out16[0..3] = kernel 2
// (start, best[q] built, matches compacted + records, end), out16[4..9] = kernel 4 (start, selection, inlier pass,
// refit, re-selection, end), of work-group 0.  Differences are shader-clock ticks (s_memtime).
int ps_debug_stamps(PsContext *ctx, uint64_t *out16)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out16) return PS_ERR_BAD_ARG;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (!ctx->stamps.p) return PS_OK;
    unsigned long long h[16];
    PS_HIP(hipMemcpyAsync(h, ctx->stamps.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 16; ++i) out16[i] = h[i];
    return PS_OK;
}

// Diagnostic: how many words of the context's keys block are not all-ones once the queued work has drained (the matcher forms
// that merge their query splits with atomicMin rely on kernel 2 putting kNoKey back into every entry it read; see
// PsContext::keysCleanPtr).  *bad must come back 0 after any sequence of calls, failed ones included.
int ps_debug_keys_clean(PsContext *ctx, uint64_t *bad)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!bad) return PS_ERR_BAD_ARG;
    *bad = 0;
    PS_HIP(hipStreamSynchronize(ctx->stream));
    if (!ctx->keys.p || ctx->keysCleanPtr != ctx->keys.p || ctx->keysCleanBytes == 0) return PS_OK; // nothing is claimed to be clean
    PS_ENSURE(ctx->sMisc2, 16);
    PS_HIP(hipMemsetAsync(ctx->sMisc2.p, 0, 16, ctx->stream));
    hipLaunchKernelGGL(ps_count_not_ones, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->keys.p,
                       ctx->keysCleanBytes / sizeof(uint32_t), (unsigned long long *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    unsigned long long h = 0;
    PS_HIP(hipMemcpyAsync(&h, ctx->sMisc2.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    *bad = h;
    return PS_OK;
}

// Diagnostic: device-side trip limits for every inlier count 1..M (see ps_limits_table).
int ps_debug_limits(PsContext *ctx, int estimator, double minRatio, int H, int M, int32_t *out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (M < 1 || !out) return PS_ERR_BAD_ARG;
    SelectArgs sa{};
    sa.estimator = estimator;
    sa.H = H;
    rc = prepare_tables(ctx, estimator, minRatio, H, sa);
    if (rc) return rc;
    PS_ENSURE(ctx->sMisc2, (size_t)M * sizeof(int32_t));
    hipLaunchKernelGGL(ps_limits_table, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, sa, M,
                       (int32_t *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(out, ctx->sMisc2.p, (size_t)M * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

// ---------------------------------------------------------------------------------------------
int ps_predicted_level(int octave, double detDist, double curDist)
{
    // Matcher::matchXYZ, matcher.cpp:639-652,681-692 with scaleFactor 1.2 / nLevels 8 (matcher.h:26-28); host libm
    // exactly as the reference evaluates it.
    const double scaleFactor = 1.2;
    const int nLevels = 8;
    const double logScaleFactor = std::log(scaleFactor);
    double detLevelScaleFactor = std::pow(scaleFactor, octave);
    double curLevelScaleFactor = detLevelScaleFactor * detDist / curDist;
    int curLevel = (int)std::ceil(std::log(curLevelScaleFactor) / logScaleFactor);
    if (curLevel < 0) curLevel = 0;
    if (curLevel > nLevels - 1) curLevel = nLevels - 1;
    return curLevel;
}

int ps_match_xyz(PsContext *ctx, const float *mapPos, const uint8_t *mapDesc, size_t mapDescStep, const int32_t *mapLevel,
                 int nmap, const float *curPos, const uint8_t *curDesc, size_t curDescStep, const int32_t *curLevel, int ncur,
                 double sphereRadius, double acceptRatio, PsDMatch *out, int cap, int *nout)
{
    int rc = bind(ctx);
    if (rc) return rc;
    TimingOff toff(ctx);
    if (nout) *nout = 0;
    if (!nout || nmap < 0 || ncur < 0 || cap < 0 || (cap > 0 && !out) ||
        (nmap > 0 && (!mapPos || !mapDesc || !mapLevel)) || (ncur > 0 && (!curPos || !curDesc || !curLevel)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_match_xyz: bad argument");
    if ((nmap > 0 && mapDescStep < PS_DESC_BYTES) || (ncur > 0 && curDescStep < PS_DESC_BYTES))
        return fail(ctx, PS_ERR_UNSUPPORTED, "descriptor rows must be 32 bytes (ORB/LDB)");
    if (nmap == 0 || ncur == 0) return PS_OK;
    PS_ENSURE(ctx->sMisc0, (size_t)nmap * 12);
    PS_ENSURE(ctx->sMisc1, (size_t)ncur * 12);
    PS_ENSURE(ctx->sDesc, (size_t)(nmap + ncur) * 32);
    PS_ENSURE(ctx->sNk, (size_t)(nmap + ncur) * sizeof(int32_t));
    PS_ENSURE(ctx->sMisc2, (size_t)(2 * nmap + 2) * sizeof(int32_t));
    PS_ENSURE(ctx->sMatches, (size_t)(cap > 0 ? cap : 1) * sizeof(PsDMatch));
    uint8_t *dDesc = (uint8_t *)ctx->sDesc.p;
    int32_t *dLvl = (int32_t *)ctx->sNk.p;
    int32_t *dCnt = (int32_t *)ctx->sMisc2.p, *dOff = dCnt + nmap;
    PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, mapPos, (size_t)nmap * 12, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpyAsync(ctx->sMisc1.p, curPos, (size_t)ncur * 12, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpy2DAsync(dDesc, 32, mapDesc, mapDescStep, 32, (size_t)nmap, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpy2DAsync(dDesc + (size_t)nmap * 32, 32, curDesc, curDescStep, 32, (size_t)ncur, hipMemcpyHostToDevice,
                            ctx->stream));
    PS_HIP(hipMemcpyAsync(dLvl, mapLevel, (size_t)nmap * 4, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpyAsync(dLvl + nmap, curLevel, (size_t)ncur * 4, hipMemcpyHostToDevice, ctx->stream));
    const float bound = sq_bound_f32(sphereRadius); // (float)norm < (double)radius  <=>  squared sum < bound
    const unsigned blocks = (unsigned)((nmap + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL(ps_match_xyz_kernel<false>, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       (const float *)ctx->sMisc0.p, (const uint4 *)dDesc, dLvl, nmap, (const float *)ctx->sMisc1.p,
                       (const uint4 *)(dDesc + (size_t)nmap * 32), dLvl + nmap, ncur, bound, acceptRatio, dCnt,
                       (const int32_t *)nullptr, (PsDMatch *)nullptr, 0);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ps_exclusive_scan, dim3(1), dim3(kBlock), 0, ctx->stream, dCnt, nmap, dOff);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ps_match_xyz_kernel<true>, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       (const float *)ctx->sMisc0.p, (const uint4 *)dDesc, dLvl, nmap, (const float *)ctx->sMisc1.p,
                       (const uint4 *)(dDesc + (size_t)nmap * 32), dLvl + nmap, ncur, bound, acceptRatio, dCnt,
                       (const int32_t *)dOff, (PsDMatch *)ctx->sMatches.p, cap);
    PS_HIP(hipGetLastError());
    int32_t total = 0;
    PS_HIP(hipMemcpyAsync(&total, dOff + nmap, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    *nout = total;
    int ncopy = total < cap ? total : cap;
    if (ncopy > 0) {
        PS_HIP(hipMemcpyAsync(out, ctx->sMatches.p, (size_t)ncopy * sizeof(PsDMatch), hipMemcpyDeviceToHost, ctx->stream));
        PS_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (total > cap) return fail(ctx, PS_ERR_BAD_ARG, "ps_match_xyz: output capacity too small (*nout = needed)");
    return PS_OK;
}

// ---------------------------------------------------------------------------------------------
int ps_umeyama_f32(PsContext *ctx, const float *src, const float *dst, int k, int nsets, float *T, int32_t *valid)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (k < 0 || nsets < 0 || !T || !valid || (k > 0 && nsets > 0 && (!src || !dst)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_umeyama_f32: bad argument");
    if (nsets == 0) return PS_OK;
    size_t bytes = (size_t)nsets * (k > 0 ? k : 1) * 12;
    PS_ENSURE(ctx->sMisc0, bytes);
    PS_ENSURE(ctx->sMisc1, bytes);
    PS_ENSURE(ctx->sPose, (size_t)nsets * 16 * sizeof(float));
    PS_ENSURE(ctx->sMisc2, (size_t)nsets * sizeof(int32_t));
    if (k > 0) {
        PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, src, (size_t)nsets * k * 12, hipMemcpyHostToDevice, ctx->stream));
        PS_HIP(hipMemcpyAsync(ctx->sMisc1.p, dst, (size_t)nsets * k * 12, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(ps_umeyama_sets, dim3((unsigned)nsets), dim3(64), 0, ctx->stream, (const float *)ctx->sMisc0.p,
                       (const float *)ctx->sMisc1.p, k, nsets, (float *)ctx->sPose.p, (int32_t *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(T, ctx->sPose.p, (size_t)nsets * 16 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipMemcpyAsync(valid, ctx->sMisc2.p, (size_t)nsets * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

int ps_kabsch_f64(PsContext *ctx, const double *A, const double *B, int n, int ld, double *T)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (T)
        for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (!T || n < 0 || (n > 0 && (!A || !B || ld < n))) return fail(ctx, PS_ERR_BAD_ARG, "ps_kabsch_f64: bad argument");
    if (n == 0) return PS_OK; // kabschEst.cpp:28
    size_t bytes = (size_t)3 * ld * sizeof(double);
    PS_ENSURE(ctx->sMisc0, bytes);
    PS_ENSURE(ctx->sMisc1, bytes);
    PS_ENSURE(ctx->sMisc2, (16 + (size_t)15 * 1024) * sizeof(double)); // pose + per-wave partial sums
    PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, A, bytes - (size_t)(ld - n) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpyAsync(ctx->sMisc1.p, B, bytes - (size_t)(ld - n) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n <= 16384) { // one wavefront: lowest latency (config 1 has 500 points)
        hipLaunchKernelGGL(ps_kabsch_f64_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)ctx->sMisc0.p,
                           (const double *)ctx->sMisc1.p, n, ld, (double *)ctx->sMisc2.p);
    } else { // G wavefronts, two passes over the points + a finishing wave
        int G = (n + 4095) / 4096;
        if (G > 1024) G = 1024;
        double *part = (double *)ctx->sMisc2.p + 16, *part2 = part + (size_t)6 * 1024;
        hipLaunchKernelGGL(ps_kabsch_f64_sums, dim3((unsigned)G), dim3(64), 0, ctx->stream, (const double *)ctx->sMisc0.p,
                           (const double *)ctx->sMisc1.p, n, ld, part);
        hipLaunchKernelGGL(ps_kabsch_f64_cov, dim3((unsigned)G), dim3(64), 0, ctx->stream, (const double *)ctx->sMisc0.p,
                           (const double *)ctx->sMisc1.p, n, ld, (const double *)part, part2);
        hipLaunchKernelGGL(ps_kabsch_f64_finish, dim3(1), dim3(64), 0, ctx->stream, (const double *)part,
                           (const double *)part2, G, n, (double *)ctx->sMisc2.p);
    }
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(T, ctx->sMisc2.p, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

int ps_keypoints2Dto3D(PsContext *ctx, const float *xy, int n, const uint16_t *depth, int rows, int cols,
                       size_t depthStep, const float *K, double depthImageScale, float *out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n < 0 || rows <= 0 || cols <= 0 || !depth || !K || depthStep < (size_t)cols * 2 || (n > 0 && (!xy || !out)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_keypoints2Dto3D: bad argument");
    if (n == 0) return PS_OK;
    PS_ENSURE(ctx->sMisc0, (size_t)n * 8);
    PS_ENSURE(ctx->sMisc1, (size_t)rows * depthStep);
    PS_ENSURE(ctx->sMisc2, (size_t)n * 12);
    PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    PS_HIP(hipMemcpyAsync(ctx->sMisc1.p, depth, (size_t)rows * depthStep, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ps_backproject, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const float *)ctx->sMisc0.p, n, (const uint8_t *)ctx->sMisc1.p, rows, cols, depthStep, K[0],
                       K[4], K[2], K[5], depthImageScale, (float *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(out, ctx->sMisc2.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

int ps_remove_image_distortion(PsContext *ctx, const float *xy, int n, const float *K, const double *dist5, float *out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n < 0 || !K || !dist5 || (n > 0 && (!xy || !out)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_remove_image_distortion: bad argument");
    if (n == 0) return PS_OK;
    PS_ENSURE(ctx->sMisc0, (size_t)n * 8);
    PS_ENSURE(ctx->sMisc2, (size_t)n * 8);
    PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    DistArgs a;
    for (int i = 0; i < 5; ++i) a.k[i] = dist5[i];
    a.fx = K[0]; a.fy = K[4]; a.cx = K[2]; a.cy = K[5];
    hipLaunchKernelGGL(ps_undistort_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const float *)ctx->sMisc0.p, n, a, (float *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(out, ctx->sMisc2.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

int ps_points3Dto2D(PsContext *ctx, const float *xyz, int n, const float *K, float *uv)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n < 0 || !K || (n > 0 && (!xyz || !uv))) return fail(ctx, PS_ERR_BAD_ARG, "ps_points3Dto2D: bad argument");
    if (n == 0) return PS_OK;
    PS_ENSURE(ctx->sMisc0, (size_t)n * 12);
    PS_ENSURE(ctx->sMisc2, (size_t)n * 8);
    PS_HIP(hipMemcpyAsync(ctx->sMisc0.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ps_project_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const float *)ctx->sMisc0.p, n, K[0], K[4], K[2], K[5], (float *)ctx->sMisc2.p);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(uv, ctx->sMisc2.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PS_HIP(hipStreamSynchronize(ctx->stream));
    return PS_OK;
}

// ---------------------------------------------------------------------------------------------
int ps_vo_pairs_device(PsContext *ctx, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                       const PsFrameSet *frames, const int32_t *pairs, int P, const PsPairResults *out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!frames || !out || P < 0 || (P > 0 && !pairs)) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_pairs_device: bad argument");
    if (P == 0) return PS_OK;
    if (!frames->desc || !frames->pts || !frames->nkpts || frames->maxKpts < 1 || frames->maxKpts > PS_MAX_KPTS)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_pairs_device: bad frame set");
    if (!out->matches || !out->numMatches || !out->inlierMask || !out->pose || !out->stats)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_pairs_device: null output");
    if (cfg && cfg->sampleIdx) return fail(ctx, PS_ERR_BAD_ARG, "explicit sample streams are per call, not per batch");
    const int cap = frames->maxKpts;
    Plan pl;
    rc = make_plan(ctx, params, cfg, K, cap, cap, pl);
    if (rc) return rc;
    if (ctx->timing) {
        ctx->curCall = (int)(ctx->timedCalls % kTimingRing);
        ctx->slotMask[ctx->curCall] = 0;
        ctx->timedCalls++;
    }
    // This call returns with its work still queued.  Whatever happens after the first launch -- success or an error half
    // way (an allocation failure for a later block, a launch failure) -- the end of what WAS queued is marked for a
    // later ps_context_set_stream: the new stream must not touch the shared arena before that work has finished.
    // (prepare_score below may already queue a clearing: the guard stands before it)
    struct Handoff {
        PsContext *c;
        ~Handoff()
        {
            if (!c->handoff && hipEventCreateWithFlags(&c->handoff, hipEventDisableTiming) != hipSuccess) {
                c->handoff = nullptr;
                (void)hipStreamSynchronize(c->stream); // no event to wait on: drain instead
                return;
            }
            if (hipEventRecord(c->handoff, c->stream) == hipSuccess)
                c->handoffPending = true;
            else
                (void)hipStreamSynchronize(c->stream);
        }
    } handoffGuard{ctx};
    rc = prepare_score(ctx, pl, P, cap, false, true, frames->desc);
    if (rc) return rc;
    rc = run_match_stage(ctx, *frames, pairs, P, true, pl.pa, out->matches, out->numMatches, 0);
    if (rc) return rc;
    rc = run_ransac_stage(ctx, pl, P, cap, out->matches, out->numMatches, cap, out->pose, out->inlierMask, out->stats, 2);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Streaming form of Matcher::match (reference src/Matcher/matcher.cpp:452-516): the previous frame's
// descriptors and 3-D points stay resident in HBM (the prevDescriptors / prevFeatures3D members,
// matcher.h:379-384), each push uploads only the new frame.
struct PsVoStream {
    PsContext *ctx = nullptr;
    int cap = 0;
    long long frames = 0;   // frames pushed so far
    int curSlot = 0;        // slot of the most recent frame
    int32_t nkSlot[2] = {0, 0}; // row count of the frame resident in each slot
    Buf desc, pts, meta;    // [2][cap][32], [2][cap][3], int32 {nk0, nk1, prevSlot, curSlot, seedLo, seedHi}
    // One contiguous result block on the device and its pinned host mirror, so a push needs ONE
    // device-to-host copy and ONE synchronisation: [PsRansacStats][pose 16 f32][numMatches i32 + pad]
    // [matches cap x 16 B][mask cap B]
    Buf res;
    uint8_t *hres = nullptr;   // pinned
    uint8_t *hin = nullptr;    // pinned staging of the incoming frame: [cap x 32 B][cap x 12 B][4 x i32]
    uint8_t *hresDev = nullptr, *hinDev = nullptr; // their device views (hipHostGetDevicePointer): the copy kernels' side
    size_t offPose = 0, offNum = 0, offMatches = 0, offMask = 0, resBytes = 0;
    // A push is launch-bound (three copies in, four kernels, one copy out): once the scratch
    // arena has been sized by an ordinary push with the same parameters the sequence is captured into one hipGraph
    // per frame slot and replayed with a single launch.  Everything that changes between pushes travels as data:
    // the frame (full-capacity copies from the pinned staging area), its row count and slot (meta) and the seed.
    bool graphsEnabled = true;
    bool warm = false;          // an un-captured push has run with `key`
    struct Key {
        PsRansacParams prm;
        int estimator, numHypotheses;
        float K[9];
        int options[32];               // every option of the context (kernel variants, the staged scoring's knobs), stamps
        unsigned long long arenaGen;   // PsContext::arenaGen the captured launches' pointers belong to: ANY block of the
                                       // context that is (re)allocated afterwards -- by this stream or by another call on
                                       // the same context -- invalidates the graphs
    } key{};
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    long long graphLaunches = 0;
    // PUTSLAM_HIP_PUSH_TIMING=1: host-side phases of the synchronous push, printed to stderr when the stream is destroyed
    // (staging copy | plan + tables + key | submission | wait for the GPU | results out), microseconds per push
    bool pushTiming = false;
    double pushPhase[5] = {0, 0, 0, 0, 0};
    long long pushTimed = 0;
    struct PsVoAsync *async = nullptr; // the pipelined form's state (ps_stream_async.h); null = synchronous stream
    int asyncResultMode = 0;           // PsStreamResults of the next ps_vo_stream_configure_async
    int asyncFrameLayout = 0;          // PsStreamFrames of the next ps_vo_stream_configure_async
};
static void async_release(PsVoStream *s); // (ps_stream_async.h)
static int async_reset(PsVoStream *s);

int ps_vo_stream_create(PsContext *ctx, int maxKpts, PsVoStream **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out || maxKpts < 1 || maxKpts > PS_MAX_KPTS) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_create: bad argument");
    PsVoStream *s = new PsVoStream();
    s->ctx = ctx;
    s->cap = maxKpts;
    if (const char *v = std::getenv("PUTSLAM_HIP_PUSH_TIMING")) s->pushTiming = std::strtol(v, nullptr, 10) != 0;
    *out = s;
    const size_t cap = (size_t)maxKpts;
    s->offPose = sizeof(PsRansacStats);
    s->offNum = s->offPose + 16 * sizeof(float);
    s->offMatches = s->offNum + 16;
    s->offMask = s->offMatches + cap * sizeof(PsDMatch);
    s->resBytes = s->offMask + cap;
    PS_ENSURE(s->desc, 2 * cap * 32);
    PS_ENSURE(s->pts, 2 * cap * 12);
    PS_ENSURE(s->meta, 8 * sizeof(int32_t));
    PS_ENSURE(s->res, (s->resBytes + 3) & ~(size_t)3);
    PS_HIP(hipHostMalloc((void **)&s->hres, (s->resBytes + 3) & ~(size_t)3, hipHostMallocDefault));
    PS_HIP(hipHostMalloc((void **)&s->hin, cap * 44 + 32, hipHostMallocDefault));
    memset(s->hin, 0, cap * 44 + 32); // rows beyond a frame's count are copied by the captured graph, never read
    if (hipHostGetDevicePointer((void **)&s->hresDev, s->hres, 0) != hipSuccess || hipHostGetDevicePointer((void **)&s->hinDev, s->hin, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->hresDev = s->hinDev = nullptr; // (no device view: the pushes use hipMemcpyAsync)
    }
    PS_HIP(hipMemsetAsync(s->meta.p, 0, 8 * sizeof(int32_t), ctx->stream));
    if (const char *v = std::getenv("PUTSLAM_HIP_NO_GRAPH")) s->graphsEnabled = std::atoi(v) == 0;
    return PS_OK;
}

void ps_vo_stream_destroy(PsVoStream *s)
{
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    if (s->pushTiming && s->pushTimed > 0)
        fprintf(stderr, "[putslam_hip] %lld replayed pushes, host phases in us: staging copy %.1f | plan, tables, key %.1f | submission %.1f | "
                        "wait for the GPU %.1f | results out %.1f\n", s->pushTimed, s->pushPhase[0] / s->pushTimed, s->pushPhase[1] / s->pushTimed,
                s->pushPhase[2] / s->pushTimed, s->pushPhase[3] / s->pushTimed, s->pushPhase[4] / s->pushTimed);
    async_release(s);
    for (hipGraphExec_t &g : s->gexec)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    Buf *all[] = {&s->desc, &s->pts, &s->meta, &s->res};
    for (Buf *b : all) release(*b);
    if (s->hres) (void)hipHostFree(s->hres);
    if (s->hin) (void)hipHostFree(s->hin);
    delete s;
}

int ps_vo_stream_reset(PsVoStream *s)
{
    if (!s) return PS_ERR_BAD_ARG;
    if (s->async) return async_reset(s);
    s->frames = 0;
    s->curSlot = 0;
    return PS_OK;
}

int ps_vo_stream_push(PsVoStream *s, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const uint8_t *desc, size_t descStep, const float *pts, int n, PsDMatch *matches, int *nmatches,
                      uint8_t *inlierMask, float *pose, PsRansacStats *stats)
{
    if (!s) return PS_ERR_BAD_ARG;
    PsContext *ctx = s->ctx;
    int rc = bind(ctx);
    if (rc) return rc;
    TimingOff toff(ctx);
    if (pose) identity16(pose);
    if (nmatches) *nmatches = 0;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->bestHypothesis = -1;
        stats->pointInlierRatio = NAN;
    }
    if (s->async) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_push: the stream is configured for the pipelined form (push_async / push_many)");
    if (n < 0 || n > s->cap || (n > 0 && (!desc || !pts)) || descStep < PS_DESC_BYTES || !pose || !nmatches ||
        (n > 0 && (!matches || !inlierMask)) || !cfg)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_push: bad argument");
    if (cfg->sampleIdx) return fail(ctx, PS_ERR_BAD_ARG, "explicit sample streams are not supported by the streaming call");
    const bool first = s->frames == 0;
    const int slot = first ? 0 : 1 - s->curSlot;
    const int prevSlot = s->curSlot;
    const size_t cap = (size_t)s->cap;
    using PushClock = std::chrono::steady_clock;
    PushClock::time_point tp[6];
    if (s->pushTiming) tp[0] = PushClock::now();
    // incoming frame -> pinned staging -> HBM (asynchronous; the staging area is free again after the
    // synchronisation that ends the previous push)
    uint8_t *hd = s->hin;
    float *hp = reinterpret_cast<float *>(s->hin + cap * 32);
    // meta block exactly as it lies on the device: {nk[slot 0], nk[slot 1], prevSlot, slot, seedLo, seedHi} -- ONE copy per push
    // (round 3 sent the row count, the slot pair and the seed as three copies: each is a node of the captured graph with a few
    // microseconds of its own)
    int32_t *hm = reinterpret_cast<int32_t *>(s->hin + cap * 44);
    for (int i = 0; i < n; ++i) memcpy(hd + (size_t)i * 32, desc + (size_t)i * descStep, 32);
    if (n > 0) memcpy(hp, pts, (size_t)n * 12);
    hm[slot] = n;
    hm[1 - slot] = s->nkSlot[1 - slot];
    hm[2] = prevSlot; // query = previous frame, train = current (matcher.cpp:470-471)
    hm[3] = slot;
    memcpy(&hm[4], &cfg->seed, sizeof(uint64_t));
    if (s->pushTiming) tp[1] = PushClock::now();
    // The stream's state (curSlot, frames) is committed only when the push has succeeded: after a failed push
    // (bad parameters, a HIP error) the resident frame is still the previous one and the next push matches against it.
    auto commit = [&]() {
        s->curSlot = slot;
        s->nkSlot[slot] = n;
        s->frames++;
    };
    // Frame in / results out as ONE kernel each over the mapped pinned staging blocks (ps_copy_segments) instead of three and one
    // hipMemcpyAsync: a copy of this size is a node of its own with 5 - 8 us of latency in the captured graph, the kernel reads
    // the 88 KB of a 2000-keypoint frame over the link in 4 (option "stream_copy_kernels" = 0: the copies of rounds 1 - 4).
    const bool copyKernels = ctx->streamCopyKernels != 0 && s->hinDev != nullptr && s->hresDev != nullptr;
    auto copy_in = [&](size_t rows) -> int {
        if (copyKernels) {
            CopySegs up{};
            int k = 0;
            if (rows > 0) {
                up.src[k] = s->hinDev;
                up.dst[k] = (uint8_t *)s->desc.p + (size_t)slot * cap * 32;
                up.bytes[k++] = rows * 32;
                up.src[k] = s->hinDev + cap * 32;
                up.dst[k] = (uint8_t *)s->pts.p + (size_t)slot * cap * 12;
                up.bytes[k++] = rows * 12;
            }
            up.src[k] = s->hinDev + cap * 44;
            up.dst[k] = s->meta.p;
            up.bytes[k++] = 6 * sizeof(int32_t);
            up.n = k;
            const unsigned groups = (unsigned)((rows * 32 / 16 + 255) / 256);
            hipLaunchKernelGGL(ps_copy_segments, dim3(groups < 1 ? 1 : (groups > 64 ? 64 : groups)), dim3(256), 0, ctx->stream, up);
            PS_HIP(hipGetLastError());
            return PS_OK;
        }
        if (rows > 0) {
            PS_HIP(hipMemcpyAsync((uint8_t *)s->desc.p + (size_t)slot * cap * 32, hd, rows * 32, hipMemcpyHostToDevice,
                                  ctx->stream));
            PS_HIP(hipMemcpyAsync((float *)s->pts.p + (size_t)slot * cap * 3, hp, rows * 12, hipMemcpyHostToDevice,
                                  ctx->stream));
        }
        PS_HIP(hipMemcpyAsync(s->meta.p, hm, 6 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        return PS_OK;
    };
    if (first) { // detectInitFeatures (matcher.cpp:17-64): nothing to match against yet
        rc = copy_in((size_t)n);
        if (rc) return rc;
        PS_HIP(hipStreamSynchronize(ctx->stream));
        commit();
        *nmatches = -1;
        return PS_OK;
    }
    PsFrameSet fs;
    fs.desc = (const uint8_t *)s->desc.p;
    fs.pts = (const float *)s->pts.p;
    fs.nkpts = (const int32_t *)s->meta.p;
    fs.numFrames = 2;
    fs.maxKpts = s->cap;
    fs.descFrameStride = fs.ptsFrameStride = 0;
    Plan pl;
    rc = make_plan(ctx, params, cfg, K, s->cap, s->cap, pl);
    if (rc) return rc;
    pl.ma.seedDev = reinterpret_cast<const uint64_t *>((const int32_t *)s->meta.p + 4);
    rc = prepare_score(ctx, pl, 1, s->cap);
    if (rc) return rc;
    // (outside the capture: a captured push then holds no clearing node, and a replay finds the block as its capture did)
    PS_ENSURE(ctx->keys, (size_t)s->cap * sizeof(uint32_t));
    rc = keys_clean(ctx, (size_t)s->cap * sizeof(uint32_t));
    if (rc) return rc;
    uint8_t *dres = (uint8_t *)s->res.p;
    auto enqueue = [&](size_t rows) -> int {
        int r = copy_in(rows);
        if (r) return r;
        r = run_match_stage(ctx, fs, (const int32_t *)s->meta.p + 2, 1, true, pl.pa, (PsDMatch *)(dres + s->offMatches),
                            (int32_t *)(dres + s->offNum), 0);
        if (r) return r;
        r = run_ransac_stage(ctx, pl, 1, s->cap, (const PsDMatch *)(dres + s->offMatches),
                             (const int32_t *)(dres + s->offNum), s->cap, (float *)(dres + s->offPose), dres + s->offMask,
                             (PsRansacStats *)dres, 2);
        if (r) return r;
        if (copyKernels) {
            CopySegs down{};
            down.src[0] = dres;
            down.dst[0] = s->hresDev;
            down.bytes[0] = (s->resBytes + 3) & ~(size_t)3;
            down.n = 1;
            const unsigned groups = (unsigned)((s->resBytes / 16 + 255) / 256);
            hipLaunchKernelGGL(ps_copy_segments, dim3(groups < 1 ? 1 : (groups > 32 ? 32 : groups)), dim3(256), 0, ctx->stream, down);
            PS_HIP(hipGetLastError());
            return PS_OK;
        }
        PS_HIP(hipMemcpyAsync(s->hres, dres, s->resBytes, hipMemcpyDeviceToHost, ctx->stream));
        return PS_OK;
    };
    PsVoStream::Key key;
    memset(&key, 0, sizeof key);
    // field by field: the caller's struct may carry indeterminate padding bytes, the key is compared with memcmp
    key.prm.verbose = params->verbose;
    key.prm.errorVersion = params->errorVersion;
    key.prm.errorVersionVO = params->errorVersionVO;
    key.prm.errorVersionMap = params->errorVersionMap;
    key.prm.inlierThresholdEuclidean = params->inlierThresholdEuclidean;
    key.prm.inlierThresholdReprojection = params->inlierThresholdReprojection;
    key.prm.inlierThresholdMahalanobis = params->inlierThresholdMahalanobis;
    key.prm.minimalInlierRatioThreshold = params->minimalInlierRatioThreshold;
    key.prm.minimalNumberOfMatches = params->minimalNumberOfMatches;
    key.prm.usedPairs = params->usedPairs;
    key.prm.iterationCount = params->iterationCount;
    {
        static_assert(sizeof kOptions / sizeof kOptions[0] + 1 <= sizeof key.options / sizeof key.options[0], "key.options too small");
        int n = 0;
        for (const OptDesc &o : kOptions) key.options[n++] = ctx->*(o.field);
        key.options[n++] = ctx->stampsOn;
    }
    key.estimator = cfg->estimator;
    key.numHypotheses = cfg->numHypotheses;
    if (K) memcpy(key.K, K, sizeof key.K);
    key.arenaGen = ctx->arenaGen; // (make_plan / prepare_score above may already have grown a block: then no replay)
    const bool sameKey = s->warm && memcmp(&key, &s->key, sizeof key) == 0;
    if (!sameKey) { // new parameters: the next ordinary push re-sizes scratch and tables, graphs are rebuilt after it
        for (hipGraphExec_t &g : s->gexec)
            if (g) {
                (void)hipGraphExecDestroy(g);
                g = nullptr;
            }
    }
    bool launched = false;
    if (s->pushTiming) tp[2] = PushClock::now();
    if (s->graphsEnabled && sameKey) {
        if (!s->gexec[slot]) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int r = enqueue(cap);
                hipError_t e2 = hipStreamEndCapture(ctx->stream, &graph);
                if (r == PS_OK && e2 == hipSuccess && graph &&
                    hipGraphInstantiate(&s->gexec[slot], graph, nullptr, nullptr, 0) != hipSuccess)
                    s->gexec[slot] = nullptr;
                if (r != PS_OK || e2 != hipSuccess) s->gexec[slot] = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!s->gexec[slot]) {
                s->graphsEnabled = false; // capture is not available here: stay on ordinary launches
                (void)hipGetLastError();
                ctx->err.clear();
            }
        }
        if (s->gexec[slot]) {
            PS_HIP(hipGraphLaunch(s->gexec[slot], ctx->stream));
            s->graphLaunches++;
            launched = true;
        }
    }
    if (!launched) {
        rc = enqueue((size_t)n);
        if (rc) return rc;
        key.arenaGen = ctx->arenaGen; // the blocks as this ordinary push left them
        memcpy(&s->key, &key, sizeof key); // (bytewise, padding included: the key is compared with memcmp)
        s->warm = true;
    }
    if (s->pushTiming) tp[3] = PushClock::now();
    PS_HIP(hipStreamSynchronize(ctx->stream));
    if (s->pushTiming) tp[4] = PushClock::now();
    commit();
    int32_t nm = 0;
    memcpy(&nm, s->hres + s->offNum, sizeof nm);
    memcpy(pose, s->hres + s->offPose, 16 * sizeof(float));
    if (stats) memcpy(stats, s->hres, sizeof *stats);
    if (nm > 0) {
        memcpy(matches, s->hres + s->offMatches, (size_t)nm * sizeof(PsDMatch));
        memcpy(inlierMask, s->hres + s->offMask, (size_t)nm);
    }
    *nmatches = nm;
    if (s->pushTiming && launched) {
        tp[5] = PushClock::now();
        for (int i = 0; i < 5; ++i) s->pushPhase[i] += std::chrono::duration<double, std::micro>(tp[i + 1] - tp[i]).count();
        s->pushTimed++;
    }
    return PS_OK;
}

} // extern "C"

#include "ps_stream_async.h"
