// ps_diag.h -- the diagnostics of the C ABI (ps_debug_*, include/putslam_hip.h): what the parity tests read beyond the outputs
// -- per-hypothesis counts, stop tables, the exactness checks of the short division / root forms, staged-scoring survivors,
// stamps.  No reference counterpart.  Included by ps_capi.hip inside its extern "C" block (the kernels and ransac_host_entry are
// that file's).
#pragma once

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

