// ps_internal.h -- what the translation units of libputslam_hip.so share beyond the public C ABI (include/putslam_hip.h).
// Not installed, not part of the ABI: the psi_ names may change between versions.
//
// The library is ONE device translation unit (ps_capi.hip: the kernels, the plan and every launch) and host-only ones compiled
// as plain C++ (ps_context.cpp: context, scratch arena, option table, stop-table builders, timing record; ps_batch_queue.cpp;
// ps_env.cpp).  A change to the option table or to the queue recompiles in a second, not with the 54 kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "putslam_hip.h"

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};


constexpr int kMaxTimed = 8;   // kernels timed per call
constexpr int kTimingRing = 128; // calls kept (HIP events on the launch stream around every kernel)

// (mirrors of psdev::kReorderMargin / kReorderTopMax, ps_score_fast.h: the device translation unit checks them with static_asserts)
constexpr int kPsReorderMargin = 16, kPsReorderTopMax = 16;
constexpr unsigned kUsacMaxHyp = 850000u; // USAC_wrapper.cpp:70

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
    // launch chains that run side by side with this one (option "side_by_side"; 0 = the context runs alone): a PsBatchQueue and
    // the pipelined stream set it on their private contexts.  Chains that overlap hide one another's launch gaps, so the staged
    // scoring's dependent launches cost throughput next to nothing and its crossover lies at far smaller batches (prepare_score)
    int sideBySide = 0;
    // staged scoring: stages 1+ sweep the matches in the order ps_stage_reorder writes (those the prefix's best hypotheses
    // reject first: hypotheses end sooner, ps_score_fast.h).  1 = always, 0 = never (original order), 2 (default) = for the
    // fixed schedule only: under the adaptive schedules the trip limit usually ends the scoring inside the prefix, and the
    // extra launch (6 us per call) buys nothing (option "reorder")
    int reorder = 2;
    int genSplit = 1;    // staged scoring: stage 0 as two launches -- the prefix's models once, then the sweep with the match range
                         // split over twice as many work-groups (option "gensplit" = 0: one launch, every part repeats the
                         // sample -> SVD chain)
    int singleRest = 1;  // adaptive schedules, no reordering: one stage after the prefix instead of three (option "singlerest")
    int pretest = 1;     // stage 1: one-direction pre-test on the all-reject front (option "pretest")
    int listGroups2 = 0; // work-groups per pair of stage 2 (option "list_g2"; 0 = automatic)
    int forcePrefix = 0; // option "prefix": hypotheses stage 0 scores completely under the fixed schedule (64 .. 256; 0 = default)
    int bail = 1;        // option "bail": a pair whose prefix leaves nothing to abandon is swept in ONE stage (ps_stage_reorder)
    int streamCopyKernels = 1; // option "stream_copy_kernels": ps_vo_stream_push moves its frame in / results out with a copy
                               // kernel over mapped pinned memory (1) or with hipMemcpyAsync (0: rounds 1 - 4)
    int streamAhead = -1; // option "stream_ahead": places of the pipelined stream beyond one per lane (chunks queued behind the running
                          // ones); -1 = six places in all
    int modelRoomMiB = 0; // option "model_room_mib": room for the staged scoring's parked models (0 = 256 MiB adaptive / 2 GiB fixed)
    int reorderTop = 8, reorderMargin = kPsReorderMargin; // (options "reorder_top" / "reorder_margin")
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


inline int fail(PsContext *c, int code, const char *what, hipError_t e = hipSuccess)
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

inline int ensure(PsContext *ctx, Buf &b, size_t bytes)
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


inline void release(Buf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}


inline int bind(PsContext *ctx)
{
    if (!ctx) return PS_ERR_BAD_ARG;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, PS_ERR_HIP, "hipSetDevice", e);
    ctx->err.clear();
    return PS_OK;
}

// ---- ps_context.cpp
// RANSAC::computeRANSACIteration (RANSAC.cpp:457-461) / USAC<T>::updateStandardStopping (USAC.h:944-971) with the host's libm, and the
// threshold tables the device binary-searches instead of evaluating log / pow
int ransac_iterations_host(double inlierRatio, double successProbability = 0.98, int numberOfPairs = 3);
unsigned usac_stopping_host(double prob_good_model);
void build_ransac_table(double minRatio, int H, std::vector<float> &tab, int &iter0, float &tiny);
void build_usac_table(int H, std::vector<double> &tab);
// every settable option's current value, in table order (the key of ps_vo_stream_push's captured graphs); returns the count
int psi_options_snapshot(const PsContext *ctx, int *out, int cap);

extern "C" {
// Error text of a context (what ps_last_error returns), set from another translation unit.
void psi_set_error(PsContext *ctx, const char *what);
// Every settable option of `src` (kernel variants, staged-scoring knobs) copied to `dst`: lanes of the pipelined stream and
// chains of a batch queue run what their parent context would.
void psi_copy_options(PsContext *dst, const PsContext *src);
// GPU_MAX_HW_QUEUES as the process environment held it when the library was loaded (after the library's own default, ps_env.cpp):
// the number of hardware queues the HIP runtime gives the process' streams; 0 = the variable held no number.
int psi_hw_queues_seen(void);
// 1 if the library set the variable itself (it was unset when the library was loaded).
int psi_hw_queues_defaulted(void);
// (device translation unit) launch attributes of the kernels that need them; called once per context
void psi_kernel_attributes(void);
}
