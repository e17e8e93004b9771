// ps_context.cpp -- PsContext: one HIP stream + one grow-only scratch arena per instance (the reference's Matcher / RANSAC objects
// are per-thread instances with no shared state, PUTSLAM.cpp:566,570), its option table, the stop-table builders and the timing
// record.  Host-only translation unit: no kernels, no launches (ps_capi.hip has those).
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ps_internal.h"

// RANSAC::computeRANSACIteration (reference src/TransformEst/RANSAC.cpp:457-461) evaluated with the
// host's libm exactly as the reference evaluates it; the int conversion (UB there for huge
// quotients) saturates.
int ransac_iterations_host(double inlierRatio, double successProbability, int numberOfPairs)
{
    double v = std::log(1 - successProbability) / std::log(1 - std::pow(inlierRatio, numberOfPairs));
    if (!(v < 2147483647.0)) return INT_MAX;
    if (v < 0) return 0;
    return (int)v;
}

// USAC<T>::updateStandardStopping (reference include/putslam/USAC/USAC.h:944-971) as a function of the
// good-model probability, with confThreshold 0.99 and maxHypotheses 850000 (USAC_wrapper.cpp:66,70).
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


namespace {

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
    {"side_by_side", "SIDE_BY_SIDE", &PsContext::sideBySide, 0, 16, "side_by_side: 0 (the context runs alone) or the number of launch chains it shares the chip with"},
    {"reorder", "REORDER", &PsContext::reorder, 0, 2, "reorder: 0, 1 or 2"},
    {"qsplit", "QSPLIT", &PsContext::forceQsplit, 0, 1024, "qsplit: 0 (automatic) .. 1024", true},
    {"msplit", "MSPLIT", &PsContext::forceMsplit, 0, 1024, "msplit: 0 (automatic) .. 1024", true},
    // staged scoring (ps_score_fast.h): the twins of its launch forms ...
    {"gensplit", "GENSPLIT", &PsContext::genSplit, 0, 1, "gensplit: 0 (stage 0 as one launch) or 1 (models, then the sweep)", true},
    {"singlerest", "SINGLEREST", &PsContext::singleRest, 0, 1, "singlerest: 0 (three stages) or 1 (one stage after the prefix, adaptive schedules)", true},
    {"pretest", "PRETEST", &PsContext::pretest, 0, 1, "pretest: 0 or 1 (stage 1's one-direction pre-test)", true},
    // ... and its tuning knobs
    {"list_g2", "LISTG2", &PsContext::listGroups2, 0, 64, "list_g2: 0 (automatic) .. 64 work-groups per pair of stage 2", true},
    {"prefix", "PREFIX", &PsContext::forcePrefix, 0, 256, "prefix: 0 (default) or 64, 128, 192, 256 hypotheses of stage 0 (fixed schedule)", true},
    {"reorder_top", "REORDER_TOP", &PsContext::reorderTop, 1, kPsReorderTopMax, "reorder_top: 1 .. 16 voters", true},
    {"reorder_margin", "REORDER_MARGIN", &PsContext::reorderMargin, 1, 4096, "reorder_margin: 1 .. 4096 matches", true},
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

int psi_options_snapshot(const PsContext *ctx, int *out, int cap)
{
    int n = 0;
    for (const OptDesc &o : kOptions)
        if (n < cap) out[n++] = ctx->*(o.field);
    return n;
}

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
    psi_kernel_attributes(); // (the cross-check kernel's LDS: ps_capi.hip)
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


} // extern "C"
