// ps_score_euclid.h -- kernel 3 for the Euclidean metrics (errorVersion 0 and 4), decision-exact instead of value-exact.
//
// RANSAC::computeMatchInlierRatioEuclidean (reference src/TransformEst/RANSAC.cpp:251-281) decides, for every
// (hypothesis, match), whether  |R cur + t - prev|  stays below inlierThresholdEuclidean (times prev.z when errorVersion
// is ADAPTIVE_ERROR, :268-271).  Every configuration the reference ships runs this metric (errorVersionVO = 0 in
// resources/putslammatcherOpenCVParameters.xml:30-31 and in all configs/*).  ps_ransac_score<0/4> (ps_kernels.h)
// reproduces every intermediate value bit for bit: 19 vector instructions per evaluation.  Only the DECISION has to
// equal the reference's; this kernel takes it from a cheap evaluation with a proven error band and hands the evaluations
// inside the band to the value-exact inlier_test<MODE>():
//
//   fast evaluation: one lane = one hypothesis, TWO matches per packed instruction (v_pk_fma_f32: .x = match 2k,
//   .y = match 2k+1; the model is a per-lane scalar used for both halves, the match pair arrives wave-uniform in SGPR
//   pairs from a pair-interleaved record):
//       errorVersion 0:  d_i = fma(R_i0, c_x, fma(R_i1, c_y, fma(R_i2, c_z, t_i))) - p_i          (4 packed / row)
//       errorVersion 4:  d'_i = fma(R_i0, c'_x, fma(R_i1, c'_y, fma(R_i2, c'_z, fma(t_i, w, -p'_i))))   with the record
//                        normalised by the match's own depth, w = fl(1 / p_z), c' = fl(c w), p' = fl(p w): the adaptive
//                        threshold thr p_z becomes the constant thr                                (4 packed / row)
//       s~ = fma(d_0, d_0, fma(d_1, d_1, d_2 d_2))                                                 (3 packed)
//   = 15 packed instructions per TWO evaluations; the decision costs four more per two evaluations (below): 9.5 vector
//   instructions per evaluation instead of 19, and -- what decides the speed of this loop -- 8 scalar instructions per
//   packed step instead of 28: a SIMD issues one scalar instruction every four cycles, and the first form of this kernel
//   (two v_cmp per evaluation, masks combined and tested with s_or / s_andn2 / s_cbranch per match) was bound by them:
//   1.02 ms per 499 pairs with 21 vector instructions per step, 1.20 ms with six s_mov more (profiles/r03a).
//
//   error band (u = 2^-24).  Per hypothesis  S >= sum_j |R_ij| |c_j| + |t_i|  for every row and every point of the
//   pair (S = 1.001 (rho cmax + tau): rho = largest row sum of |R|, tau = largest |t_i|, cmax = the pair's largest
//   coordinate).  With E_i the real-valued residual:
//     reference (no FMA, order a0 + (a1 + a2) + t, then - p):   |d_ref_i - E_i| <= 4.02 u S + 1.01 u |d_ref_i|
//     errorVersion 0 here (3 FMA roundings, then - p):          |d~_i - E_i|    <= 3.01 u S + 1.01 u |d~_i| ...
//       => |d~ - d_ref|_2 <= a + 2.01 u r_ref,   a = 12.5 u S  >= sqrt3 * 7.03 u S
//     errorVersion 4 here (inputs within 2.01 u, 4 FMA roundings, everything divided by p_z >= 0.1):
//       |p_z d'_i - E_i| <= 6.03 u (S + cmax)  =>  |p_z d' - d_ref|_2 <= a4 + 1.01 u r_ref,  a4 = 17.5 u (S + cmax)
//   Both sums of squares carry at most three roundings: s = r^2 (1 + theta), |theta| <= 3.01 u.  The reference's test is
//   s_ref < B with B = sq_bound_f32(thr) (the smallest float whose correctly rounded root reaches the double threshold:
//   sqrt(B) within [1 - 1.01 u, 1 + 2.02 u] of thr).  Chaining the inequalities (header of make_plan's EuclidConsts):
//       s~ < lo = ((Tb (1 - 6u) - a)^2) (1 - 8u)   =>  s_ref < B        (certain inlier; never when Tb (1 - 6u) <= a)
//       s~ > hi = ((Tb (1 + 6u) + a (1 + 4u))^2) (1 + 8u)  =>  s_ref >= B   (certain outlier)
//   with Tb = sqrt(B) for errorVersion 0 and, for errorVersion 4, Tb = thr (7u instead of 6u) and a = 10.001 a4
//   (1 / p_z <= 10 (1 + u) after the depth filter of RANSAC.cpp:65-74).  lo and hi do not depend on the match: they are
//   two per-lane constants computed once in the prologue, so the loop has no band arithmetic at all.
//
//   decision without compares.  Per lane  c = 0.98 / (hi (1 + 2^-21) - lo)  and  K = fl(c hi) (1 + 2^-22)  (K >= c hi,
//   K - c lo <= 0.99); one clamped packed FMA gives the indicator of both matches of the step
//       ind = clamp01(fl(K - c s~)):   s~ >= hi  =>  K - c s~ <= 0 + ...  =>  ind = 0 exactly only if s~ >= K / c >= hi
//                                      ind = 1 exactly  <=>  K - c s~ >= 1 - 2^-25  =>  s~ <= (K - 0.99..) / c <= lo
//   so an indicator that is exactly 0 (1) is a certain outlier (inlier), and every other value of [0, 1] marks an
//   evaluation inside [lo, hi].  acc += ind (packed) counts; tf = fma(-ind, ind, ind) is +0 exactly for ind in {0, 1} and
//   a positive float for every other ind in (0, 1) (ind - ind^2 >= 2^-25 there, no cancellation to zero), and
//   v_or3_b32 collects its bits.  Every 64 matches a lane whose collector is still zero adds its (exact, small-integer)
//   float sum to the count; a lane with a marked block has that block recounted by inlier_test<MODE>() on the
//   value-exact records, the wave's 64 lanes taking one match each (no queue, no LDS traffic besides the model).
//   NaN cannot occur under the bounds checked below (finite model, |values| <= 1e15).
//
// A wavefront whose bounds do not hold (non-finite model, coordinates beyond 1e15, threshold outside [1e-10, 1e10]) runs
// the value-exact loop.  Counts equal ps_ransac_score<0/4>'s for every hypothesis (tests/test_gpu_score_variants.py,
// tests/test_gpu_band_edges.py, the fuzz slice).
#pragma once

#include "ps_score_fast.h"

namespace psdev {

constexpr int kEuclidRecFloats0 = 12; // per match PAIR: (c_x c_x')(c_y c_y')(c_z c_z')(p_x p_x')(p_y p_y')(p_z p_z')
constexpr int kEuclidRecFloats4 = 16; // (c'_x ..)(c'_y ..)(c'_z ..)(w w')(p'_x ..)(p'_y ..)(p'_z ..)(pad)

PS_D v2f_t splat(float v) { return v2f_t{v, v}; }

constexpr int kEuclidBlock = 64; // matches between two checks of the indicators (32 packed steps)

#ifndef PS_EUCLID_WAVES
#define PS_EUCLID_WAVES 7
#endif
// One pair record as the SGPR pairs the packed instructions take.
template <int MODE> struct EuclidRec {
    float2 g[MODE == PS_ADAPTIVE_ERROR ? 7 : 6];
};
template <int MODE> PS_D EuclidRec<MODE> load_euclid_rec(const float2 *__restrict__ g)
{
    EuclidRec<MODE> r;
#pragma unroll
    for (int i = 0; i < (MODE == PS_ADAPTIVE_ERROR ? 7 : 6); ++i) r.g[i] = g[i];
    return r;
}

// ------------------------------------------------------------------------------------------
// Pruned scoring (PRUNE = true): hypotheses that provably cannot matter are abandoned.
//
// The selection of kernel 4 consumes the counts sequentially: hypothesis i matters only if its count is a RECORD,
// count_i > max_{j<i} count_j (strict '>' first-best, RANSAC.cpp:438-455; the arg-max of the fixed schedule takes the
// lowest index among equals, the same rule), and only while i is below the adaptive trip limit, which never grows
// (RANSAC.cpp:450-453, USAC.h:944-971).  So the step is scored in two launches:
//   launch A (PRUNE = false)  the first kPrefix = 256 hypotheses of every pair, completely;
//   launch B (PRUNE = true)   the hypotheses from kPrefix on.  Every work-group first replays the selection over the
//       prefix (wave_replay_prefix: the same record walk as ps_select_refit) and gets  B0 = the best count so far and
//       L0 = the trip limit after the prefix (valid as a bound once the prefix holds a record: from record to record
//       the limit only shrinks).  Hypotheses >= L0 are never consumed: their work-groups return at once
//       (with the reference's own <= 487-iteration schedule that is nearly every pair: the limit after 256 hypotheses is
//       typically 2 ... 30).  A hypothesis below L0 is abandoned as soon as  count so far + matches left <= B0 : it can
//       no longer become a record; its partial count (<= B0, hence never selected, never a record) is what it stores.
//   Lanes are re-packed: every two 64-match blocks the work-group counts its live hypotheses and, when they fit into
//   fewer wavefronts, compacts them (slot + count through LDS, the model re-read from the LDS copy), so that whole
//   wavefronts retire.  With 75 ... 90 % inliers a bad sample is abandoned after 10 ... 25 % of the matches and a good one
//   that is not better than the prefix's best after 50 ... 70 %.
// Outputs of kernel 4 are unchanged bit for bit (tests/test_gpu_prune.py: pruned vs unpruned vs oracle); what changes is
// the meaning of counts[] for abandoned hypotheses (a lower bound <= B0 instead of the count), which is why the
// diagnostic ps_debug_ransac_counts and small batches run unpruned.
// ------------------------------------------------------------------------------------------
template <int MODE, bool PRUNE>
__global__ __launch_bounds__(kBlock, PS_EUCLID_WAVES) void ps_ransac_score_euclid(
    const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float2 *__restrict__ recG,
    const int32_t *__restrict__ mvalid, const float2 *__restrict__ pairBound, ModelArgs ma, ScoreConsts k,
    EuclidConsts ec, SelectArgs sa, int hBase, int hCount, int H, int cap, int minRun, int msplit,
    int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg)
{
    // This launch scores hypotheses [hBase, hBase + hCount) of every pair; counts[] has H entries per pair.
    static_assert(MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR, "the Euclidean metrics");
    constexpr int RF = MODE == PS_ADAPTIVE_ERROR ? kEuclidRecFloats4 : kEuclidRecFloats0;
    __shared__ float s_mdl[12][kBlock];
    __shared__ int s_pref[2];
    __shared__ int s_alive[kBlock / 64];
    __shared__ uint32_t s_list[kBlock];

    const unsigned hb = (unsigned)((hCount + kBlock - 1) / kBlock);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bx = L % hb, by = (L / hb) % (unsigned)msplit;
    const int p = (int)(L / (hb * (unsigned)msplit));
    const int M = mvalid[p];
    if (M < minRun) return; // too few matches: kernel 4 returns identity (RANSAC.cpp:77-80)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform: says so to the compiler)
    const int hFirst = hBase + (int)bx * kBlock; // hypothesis of this work-group's slot 0
    int hEnd = hBase + hCount;                   // hypotheses of this launch end here ...
    int best0 = 0;
    if (PRUNE) { // ... or at the trip limit the prefix leaves (msplit == 1 in this form)
        if (wv == 0) {
            int b, l;
            wave_replay_prefix(counts + (size_t)p * H, hBase < H ? hBase : H, sa, M, b, l);
            if (lane == 0) {
                s_pref[0] = b;
                s_pref[1] = l;
            }
        }
        __syncthreads();
        // (LDS loads count as divergent for the compiler: readfirstlane keeps the work-group-uniform values in SGPRs and
        // the branches on them scalar)
        best0 = __builtin_amdgcn_readfirstlane(s_pref[0]);
        // the limit only shrinks from record to record, but the FIRST record may raise it above its initial value
        // (RANSAC.cpp:30 starts from computeRANSACIteration(0.20); a first ratio below 0.2 gives more): without a record
        // in the prefix nothing can be cut
        if (best0 > 0) hEnd = min(hEnd, __builtin_amdgcn_readfirstlane(s_pref[1]));
        if (hFirst >= hEnd) return; // never consumed by the selection
    }
    const size_t rbase = (size_t)p * cap;
    // the match range is split on match PAIRS (the packed loop takes two matches per step)
    const int npair = (M + 1) >> 1;
    const int m0 = 2 * (int)(((long long)npair * by) / msplit);
    int m1 = 2 * (int)(((long long)npair * (by + 1)) / msplit);
    m1 = m1 < M ? m1 : M;

    int slot = tid; // which of the work-group's 256 hypotheses this lane is scoring (changes when lanes are re-packed)
    const int h = hFirst + tid;
    Rigid mdl, inv;
    set_identity(mdl);
    set_identity(inv);
    bool valid = false;
    if (h < hEnd) valid = gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)h, mdl);
    if (ma.models && by == 0 && h < hEnd) store_model(ma, (size_t)p * H + h, mdl);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_mdl[3 * i + j][tid] = mdl.R[i][j];
        s_mdl[9 + i][tid] = mdl.t[i];
    }

    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float2 *__restrict__ pg = recG + (size_t)p * ((size_t)((cap + 1) >> 1) * (RF / 2));
    const float cmax = pairBound[p].x;
    int32_t *__restrict__ cout = counts + (size_t)p * H;

    float rho = 0.0f, tau = 0.0f;
    model_norms(mdl, rho, tau);
    const float S = (rho * cmax + tau) * 1.001f;
    // (comparisons are false for NaN: a non-finite model or cmax sends the wavefront to the value-exact loop)
    const bool boundsOk = ec.enabled != 0 && S >= 1.0e-20f && S <= 1.0e15f && cmax <= 1.0e15f;
    int cnt = 0;
    // PRUNE: one decision for the work-group (its wavefronts meet at barriers)
    const bool fastOk = PRUNE ? (__syncthreads_and(boundsOk ? 1 : 0) != 0) : wave_all(boundsOk);

    if (!fastOk) {
        for (int m = m0; m < m1; ++m) {
            const float4 A = pa[m], B = pb[m];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, A, cnt);
        }
        if (h < hEnd) {
            if (!valid) cnt = 0; // model not computed -> iteration skipped (RANSAC.cpp:107)
            if (msplit == 1)
                cout[h] = cnt;
            else if (cnt)
                atomicAdd(&cout[h], cnt);
        }
        return;
    }

    // ---- per-lane registers of the hot loop, (re)built from a model: the limits lo / hi of the squared residual (see
    // the header: a = 12.5 u S for errorVersion 0, 10.001 * 17.5 u (S + cmax) for errorVersion 4, both with 1 % slack
    // for their own float roundings), the indicator's c and K, and the model splat over both halves
    v2f_t negc, kk, r00, r01, r02, t0, r10, r11, r12, t1, r20, r21, r22, t2;
    auto build = [&](const Rigid &md) {
        float rh = 0.0f, ta = 0.0f;
        model_norms(md, rh, ta);
        const float Sl = (rh * cmax + ta) * 1.001f;
        const float a = MODE == PS_ADAPTIVE_ERROR ? (Sl + cmax) * (177.0f * kEpsU) : Sl * (12.7f * kEpsU);
        const float x = ec.tbLo - a;
        const float lo = x > 0.0f ? (x * x) * (1.0f - 8.0f * kEpsU) : -1.0f;
        const float y = ec.tbHi + a * (1.0f + 4.0f * kEpsU);
        const float hi = (y * y) * (1.0f + 8.0f * kEpsU);
        // indicator  ind = clamp01(K - c s~):  exactly 0 only for s~ >= hi, exactly 1 only for s~ <= lo
        const float den = __builtin_fmaf(hi, 4.76837158203125e-07f /* 2^-21 */, hi) - lo;
        const float cInd = 0.98f / den;
        const float kInd = (cInd * hi) * (1.0f + 2.384185791015625e-07f /* 2^-22 */);
        negc = splat(-cInd);
        kk = splat(kInd);
        r00 = splat(md.R[0][0]); r01 = splat(md.R[0][1]); r02 = splat(md.R[0][2]); t0 = splat(md.t[0]);
        r10 = splat(md.R[1][0]); r11 = splat(md.R[1][1]); r12 = splat(md.R[1][2]); t1 = splat(md.t[1]);
        r20 = splat(md.R[2][0]); r21 = splat(md.R[2][1]); r22 = splat(md.R[2][2]); t2 = splat(md.t[2]);
    };
    auto model_of_slot = [&](int sl, Rigid &md) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][sl];
            md.t[i] = s_mdl[9 + i][sl];
        }
    };
    build(mdl);
    unsigned long long parked = 0;
    unsigned dbgBlocks = 0, dbgRepack = 0; // (statistics: 64-match blocks this wave computed, re-packings)
    const int m1e = m1 & ~1;              // whole pairs; an odd last match is scored by the value-exact code below
    const int tailMatches = m1 - m1e;     // 0 or 1
    bool active = h < hEnd;               // PRUNE: this lane still scores a live hypothesis
    if (PRUNE && active && !valid) {      // an invalid sample scores 0 (RANSAC.cpp:107): nothing to do for it
        cout[h] = 0;
        active = false;
    }
    int liveWaves = kBlock / 64;          // PRUNE: wavefronts that still hold live hypotheses (work-group uniform)

    for (int blk = m0; blk < m1e; blk += kEuclidBlock) {
        const int bend = blk + kEuclidBlock < m1e ? blk + kEuclidBlock : m1e;
        if (!PRUNE || wv < liveWaves) {
            ++dbgBlocks;
            v2f_t acc = {0.0f, 0.0f}; // inliers of this block (even / odd matches)
            uint32_t frac = 0u;       // != 0 as soon as one indicator of the block was neither 0 nor 1
            const float2 *__restrict__ g = pg + (size_t)(blk >> 1) * (RF / 2);
            // one packed step = the two matches of one pair record
            auto step = [&](const EuclidRec<MODE> &rec) {
                const float2 *e = rec.g;
                v2f_t dx, dy, dz;
                if (MODE == PS_ADAPTIVE_ERROR) {
                    const v2f_t cx = {e[0].x, e[0].y}, cy = {e[1].x, e[1].y}, cz = {e[2].x, e[2].y}, w = {e[3].x, e[3].y};
                    const v2f_t px = {e[4].x, e[4].y}, py = {e[5].x, e[5].y}, pz = {e[6].x, e[6].y};
                    dx = pk_fma(r00, cx, pk_fma(r01, cy, pk_fma(r02, cz, pk_fma(t0, w, -px))));
                    dy = pk_fma(r10, cx, pk_fma(r11, cy, pk_fma(r12, cz, pk_fma(t1, w, -py))));
                    dz = pk_fma(r20, cx, pk_fma(r21, cy, pk_fma(r22, cz, pk_fma(t2, w, -pz))));
                } else {
                    const v2f_t cx = {e[0].x, e[0].y}, cy = {e[1].x, e[1].y}, cz = {e[2].x, e[2].y};
                    const v2f_t px = {e[3].x, e[3].y}, py = {e[4].x, e[4].y}, pz = {e[5].x, e[5].y};
                    dx = pk_fma(r00, cx, pk_fma(r01, cy, pk_fma(r02, cz, t0))) - px;
                    dy = pk_fma(r10, cx, pk_fma(r11, cy, pk_fma(r12, cz, t1))) - py;
                    dz = pk_fma(r20, cx, pk_fma(r21, cy, pk_fma(r22, cz, t2))) - pz;
                }
                const v2f_t ss = pk_fma(dx, dx, pk_fma(dy, dy, dz * dz));
                v2f_t ind;
                asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(ind) : "v"(ss), "v"(negc), "v"(kk));
                acc = acc + ind;
                const v2f_t tf = pk_fma(-ind, ind, ind); // ind - ind^2: +0 for 0 and 1, > 0 for every other value of [0, 1]
                asm("v_or3_b32 %0, %0, %1, %2" : "+v"(frac) : "v"(tf.x), "v"(tf.y));
            };
            // two steps per trip, both records requested before the first is used (scalar loads; with few waves per
            // SIMD -- small H -- their latency is otherwise exposed once per step)
            int m = blk;
            for (; m + 4 <= bend; m += 4, g += RF) {
                const EuclidRec<MODE> ra = load_euclid_rec<MODE>(g), rb = load_euclid_rec<MODE>(g + RF / 2);
                step(ra);
                step(rb);
            }
            if (m < bend) step(load_euclid_rec<MODE>(g));
            // block epilogue: lanes whose indicators were all 0 / 1 take the sum; the others are recounted value-exactly,
            // the wave's 64 lanes taking the block's (at most) 64 matches of one such hypothesis at a time
            const bool clean = frac == 0u || (PRUNE && !active);
            if (clean) cnt += (int)(acc.x + acc.y);
            unsigned long long todo = __builtin_amdgcn_ballot_w64(!clean);
            while (todo != 0ull) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int t = PRUNE ? __builtin_amdgcn_readlane(slot, l) : wv * 64 + l;
                Rigid md;
                model_of_slot(t, md);
                const int mm = blk + lane;
                bool in = false;
                if (mm < bend) {
                    const float4 A = pa[mm], B = pb[mm];
                    in = inlier_test<MODE>(md, md, k, A, B, A);
                }
                const int c = __popcll(__builtin_amdgcn_ballot_w64(in));
                if (lane == l) cnt += c;
                parked += (unsigned long long)(bend - blk);
            }
        }
        // ---- PRUNE checkpoint every second block: abandon what cannot become a record any more, re-pack the rest
        if (PRUNE && (((blk - m0) / kEuclidBlock) & 1) == 1 && bend < m1e) {
            const int left = (m1e - bend) + tailMatches;
            if (active && cnt + left <= best0) { // count_i <= best of the earlier hypotheses whatever the rest brings
                cout[hFirst + slot] = cnt;
                active = false;
            }
            const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
            if (lane == 0) s_alive[wv] = __popcll(am);
            __syncthreads();
            int total = 0, before = 0;
#pragma unroll
            for (int i = 0; i < kBlock / 64; ++i) {
                const int n = i < liveWaves ? s_alive[i] : 0; // (wavefronts beyond liveWaves have ended)
                if (i < wv) before += n;
                total += n;
            }
            total = __builtin_amdgcn_readfirstlane(total);
            before = __builtin_amdgcn_readfirstlane(before);
            if (total == 0) break; // (work-group uniform)
            const int need = (total + 63) >> 6;
            if (need < liveWaves) {
                if (active) s_list[before + __popcll(am & ((1ull << lane) - 1ull))] = ((uint32_t)cnt << 8) | (uint32_t)slot;
                __syncthreads();
                active = tid < total;
                if (active) {
                    const uint32_t e = s_list[tid];
                    slot = (int)(e & 255u);
                    cnt = (int)(e >> 8);
                    Rigid md;
                    model_of_slot(slot, md);
                    build(md);
                }
                liveWaves = need;
                ++dbgRepack;
                if (wv >= need) {
                    // this wavefront holds no live hypothesis any more: it ends here and frees its registers for other
                    // work-groups (S_BARRIER waits only for the wavefronts of the group that have not terminated)
                    if (lane == 0) s_alive[wv] = 0;
                    if (dbg != nullptr && lane == 0) {
                        atomicAdd(&dbg[0], parked);
                        atomicAdd(&dbg[1], (unsigned long long)(m1 - m0) * 64ull);
                        atomicAdd(&dbg[2], (unsigned long long)dbgBlocks);
                        atomicAdd(&dbg[3], (unsigned long long)((m1e - m0 + kEuclidBlock - 1) / kEuclidBlock));
                    }
                    return;
                }
            }
            __syncthreads(); // s_alive / s_list are reused at the next checkpoint
        }
    }
    if (m1e < m1 && (!PRUNE || active)) { // odd last match of the range
        Rigid md;
        model_of_slot(slot, md);
        const float4 A = pa[m1e], B = pb[m1e];
        score_accumulate<MODE, false>(md, md, k, A, B, A, cnt);
    }
    if (dbg != nullptr && lane == 0) {
        atomicAdd(&dbg[0], parked);
        atomicAdd(&dbg[1], (unsigned long long)(m1 - m0) * 64ull);
        // wave-blocks computed / wave-blocks of an unpruned sweep / re-packings (ps_debug_score_stats_ex)
        atomicAdd(&dbg[2], (unsigned long long)dbgBlocks);
        atomicAdd(&dbg[3], (unsigned long long)((m1e - m0 + kEuclidBlock - 1) / kEuclidBlock));
        if (wv == 0) atomicAdd(&dbg[4], (unsigned long long)dbgRepack);
    }
    if (PRUNE) {
        if (active) cout[hFirst + slot] = cnt; // (invalid samples were stored as 0 above)
    } else if (h < hEnd) {
        if (!valid) cnt = 0; // model not computed -> iteration skipped (RANSAC.cpp:107)
        if (msplit == 1)
            cout[h] = cnt;
        else if (cnt)
            atomicAdd(&cout[h], cnt);
    }
}

} // namespace psdev
