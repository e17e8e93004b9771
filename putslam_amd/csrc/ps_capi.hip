// ps_capi.hip -- the device translation unit of libputslam_hip.so (include/putslam_hip.h): the kernels (ps_kernels.h and
// friends), the plan of a call and every launch.  PsContext itself -- stream, scratch arena, option table, stop-table builders,
// timing record -- lives in ps_context.cpp / ps_internal.h, host-only; so do the batch queue and the environment default.
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

static_assert(kPsReorderMargin == kReorderMargin && kPsReorderTopMax == kReorderTopMax, "ps_internal.h mirrors ps_score_fast.h");

namespace {

constexpr int kWidePairs = 16; // batches of at most this many pairs run kernel 2 with 1024-thread work-groups
// Tuning constants of the staged scoring that were per-context knobs until round 5 (debug.list_r3 / list_g3 / reorder_c2div /
// reorder_gran); every A/B since round 3 put their other values within 1 % of these (profiles/r03n, r04b), so they are constants:
// work-groups the last stage's match range is split over; where the stages' ranges are cut (multiples of kReorderGran matches, the
// second cut at 1 / kReorderC2div of the range); stage 3 runs one looping work-group per eight possible ones (list_groups).
constexpr int kListRsplit3 = 4, kReorderGran = 64, kReorderC2div = 16;
// where the staged scoring takes over from complete scoring (prepare_score): batch size in work units = pairs x
// (ceil(H / 256) - 1) x frame capacity, threshold = base + perRow x frame capacity (profiles/r05c/staged_crossover.txt)
struct StagedFrom {
    double base, perRow;
};
constexpr StagedFrom kStagedFromEuclidFixed = {7.8e5, 250.0}, kStagedFromReprojFixed = {1.3e6, 500.0},
                     kStagedFromEuclidAdaptive = {6.0e4, 0.0}, kStagedFromReprojAdaptive = {4.5e4, 0.0};
// ... and where it does when the context is one of several launch chains that run side by side (option "side_by_side": the
// chains of a PsBatchQueue, the lanes of the pipelined stream): the other chains' kernels fill the gaps between the staged form's
// dependent, chip-underfilling launches, so what is left of its price is the work of the extra launches -- the crossover lies
// 2 - 5 times lower (profiles/r06u/concurrent_crossover.txt: four chains, 500 ... 4000 keypoints x H = 1024 ... 16384 x 2 ... 64
// pairs on the C++ demo's frames, 73 % inliers; profiles/r06u/bench_data_crossover.txt: bench.py's kind of data with 70 % / 40 % true
// correspondences, where the staged form abandons less -- the constants are set by the latter: batches of 32 / 64 pairs of the
// bench workload's shape gain 1.26 / 1.58 x with the reprojection error, 1.27 / 1.58 x with the Euclidean one)
constexpr StagedFrom kStagedFromEuclidFixedSbs = {4.5e5, 75.0}, kStagedFromReprojFixedSbs = {3.0e5, 90.0},
                     kStagedFromEuclidAdaptiveSbs = {2.4e4, 0.0}, kStagedFromReprojAdaptiveSbs = {2.4e4, 0.0};
} // namespace

namespace {

// The matcher forms that merge with atomicMin start from an all-ones keys block (see PsContext::keysCleanPtr).
int keys_clean(PsContext *ctx, size_t bytes)
{
    if (ctx->keysCleanPtr == ctx->keys.p && ctx->keysCleanBytes >= bytes) return PS_OK;
    PS_HIP(hipMemsetAsync(ctx->keys.p, 0xFF, ctx->keys.cap, ctx->stream));
    ctx->keysCleanPtr = ctx->keys.p;
    ctx->keysCleanBytes = ctx->keys.cap;
    return PS_OK;
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
    double stagedFrom = sf.base + sf.perRow * (double)cap;
    if (ctx->sideBySide >= 2) {
        const StagedFrom sb = with_euclid_fast(ctx, pl.mode) ? (pl.sa.estimator == PS_EST_FIXED ? kStagedFromEuclidFixedSbs : kStagedFromEuclidAdaptiveSbs)
                                                             : (pl.sa.estimator == PS_EST_FIXED ? kStagedFromReprojFixedSbs : kStagedFromReprojAdaptiveSbs);
        const double sbs = sb.base + sb.perRow * (double)cap;
        // (two chains hide half of one another's gaps: half way between the two crossovers, on the logarithmic scale)
        stagedFrom = ctx->sideBySide >= 3 ? sbs : std::sqrt(sbs * stagedFrom);
    }
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
                                                  : auto3) : all;
        return want < all ? want : all;
    };
    // the last stage: work-groups its match range is split over (their counts add up in counts[]; a short survivor list
    // swept by one wavefront per SIMD pays the full latency of every record load, 0.25 us per match)
    auto list_rsplit = [&](int stage) { return (pl.reorder && stage == kStages) ? kListRsplit3 : 1; };
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
        st.gran = kReorderGran;
        st.c2div = kReorderC2div;
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

} // namespace

extern "C" {

void psi_kernel_attributes(void)
{
    // the cross-check kernel keeps best[q] for up to PS_MAX_KPTS queries in LDS (64 KiB of the CU's 160 KiB)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<true, 1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ps_crosscheck_prep<false, 1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PS_MAX_KPTS * 4);
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

#include "ps_diag.h" // ps_debug_*: the parity tests' diagnostics (no reference counterpart)

// ---------------------------------------------------------------------------------------------
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
// the 72-byte per-pair record of the multi-GPU gather (include/putslam_hip.h: ps_pack_records_device)
namespace {
__global__ void ps_pack_records_kernel(const float *__restrict__ pose, const PsRansacStats *__restrict__ stats, int valid, int pairs,
                                       float *__restrict__ rec)
{
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= pairs) return;
    float *r = rec + (size_t)p * PS_RECORD_FLOATS;
    if (p < valid) {
        for (int i = 0; i < 16; ++i) r[i] = pose[(size_t)p * 16 + i];
        r[16] = (float)stats[p].numInliers;
        r[17] = (float)stats[p].numMatchesIn;
    } else {
        for (int i = 0; i < PS_RECORD_FLOATS; ++i) r[i] = 0.0f;
    }
}
} // namespace

int ps_pack_records_device(PsContext *ctx, void *hipStream, const float *pose, const PsRansacStats *stats, int valid, int pairs,
                           float *records)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (pairs < 0 || valid < 0 || valid > pairs || (pairs > 0 && !records) || (valid > 0 && (!pose || !stats)))
        return fail(ctx, PS_ERR_BAD_ARG, "ps_pack_records_device: bad argument");
    if (pairs == 0) return PS_OK;
    hipStream_t st = hipStream ? (hipStream_t)hipStream : ctx->stream;
    hipLaunchKernelGGL(ps_pack_records_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, pose, stats, valid, pairs, records);
    PS_HIP(hipGetLastError());
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
    {
        // (frame strides are checked here, before anything is planned or allocated; run_match_stage checks them again for its
        // other callers)
        const size_t ds = frames->descFrameStride ? frames->descFrameStride : (size_t)frames->maxKpts * 32;
        const size_t ps = frames->ptsFrameStride ? frames->ptsFrameStride : (size_t)frames->maxKpts * 12;
        if ((ds & 15) != 0 || ds < (size_t)frames->maxKpts * 32 || (ps & 3) != 0 || ps < (size_t)frames->maxKpts * 12 || ((uintptr_t)frames->desc & 15) != 0)
            return fail(ctx, PS_ERR_BAD_ARG, "frame set: descFrameStride must be a multiple of 16 and >= maxKpts x 32 (desc 16-byte aligned), "
                                             "ptsFrameStride a multiple of 4 and >= maxKpts x 12");
    }
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

#include "ps_stream_push.h" // ps_vo_stream_create / _push: the synchronous streaming form

} // extern "C"

#include "ps_stream_async.h"
