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

// KIND (ps_score_fast.h): 0 = the hypotheses [hBase, hBase + hCount) completely (plain launch, stage 0), 1 = stage 1,
// 2 = stages 2+ of the staged scoring.
// One pass of a work-group (ps_score_fast.h, score_fast_pass): 256 hypotheses, or one pass over the survivor list.
template <int MODE, int KIND>
PS_D bool score_euclid_pass(const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float2 *__restrict__ recG,
                            const float2 *__restrict__ pairBound, const ModelArgs &ma, const ScoreConsts &k,
                            const EuclidConsts &ec, const SelectArgs &sa, const StageArgs &st, int H, int cap, int msplit,
                            int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg, const unsigned bx,
                            const unsigned by, const int p, const int M)
{
    constexpr int RF = MODE == PS_ADAPTIVE_ERROR ? kEuclidRecFloats4 : kEuclidRecFloats0;
    __shared__ float s_mdl[12][kBlock];
    __shared__ int s_tot[kBlock]; // kind 2, split match range: the counts of the range's parts meet here
    __shared__ int s_pref[2];

    int tid = threadIdx.x;
    // (kind 2 calls this in a loop: without the empty asm the compiler computes everything that hangs on the thread index --
    // a dozen LDS row addresses -- once before the loop and, short of registers, keeps it in scratch memory, the inlier
    // counter of the hot loop with it)
    if (KIND == 2) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform by construction: keep it and what hangs on it scalar)
    const size_t rbase = (size_t)p * cap;
    int32_t *__restrict__ cout = counts + (size_t)p * H;
    int h = st.hBase + (int)bx * kBlock + tid;
    int hEnd = st.hBase + st.hCount; // this lane scores hypothesis h if h < hEnd
    // the match range is split on match PAIRS (the packed loop takes two matches per step)
    const int npair = (M + 1) >> 1;
    int m0 = 2 * (int)(((long long)npair * by) / msplit);
    int m1 = 2 * (int)(((long long)npair * (by + 1)) / msplit);
    m1 = m1 < M ? m1 : M;
    int best0 = 0;
    constexpr bool LIST = KIND == 2; // stage >= 2: hypotheses from the survivor list, models from HBM
    constexpr bool pruned = KIND >= 1;
    int cover = kBlock, slot = tid, part = 0; // kind 2: hypotheses per pass, the lane's place, its part of the match range
    int mStageEnd = m1;
    if (LIST) {
        s_tot[tid] = 0;
        __syncthreads(); // (a part with a short range must not add into a slot another wavefront has yet to clear)
    }
    if (pruned) { // (msplit == 1 in these stages; the cuts are multiples of 64)
        int hLimit;
        stage_prefix(cout, st.hBase, sa, M, s_pref, best0, hLimit, // (stage >= 1: hBase = size of the prefix)
                     st.prefInfo != nullptr ? st.prefInfo + 4 * p : nullptr);
        stage_range(st, M, best0, m0, m1);
        // a block with hypotheses that have no model slot (ModelArgs::modelH: long caps) is swept in one piece: nothing of it
        // is parked or listed
        {
            const int blockEnd = st.hBase + ((int)bx + 1) * kBlock;
            if (!LIST && ma.models != nullptr && (blockEnd < hEnd ? blockEnd : hEnd) > ma.modelH) m1 = M;
        }
        mStageEnd = m1;
        if (m0 >= m1) return true; // an earlier stage finished the pair's matches
        if (!LIST) {
            hEnd = hEnd < hLimit ? hEnd : hLimit; // beyond the trip limit: never consumed by the selection
            if (st.hBase + (int)bx * kBlock >= hEnd) return true;
        } else {
            if (msplit > 1) { // the LAST stage may split its range over work-groups too: their counts meet in counts[]
                const int blen = (((m1 - m0 + msplit - 1) / msplit) + 63) & ~63;
                m0 += (int)by * blen;
                m1 = m1 < m0 + blen ? m1 : m0 + blen;
                if (m0 >= m1) return false;
            }
            const int n = st.countIn[p];
            cover = list_cover(n);
            part = __builtin_amdgcn_readfirstlane(tid / cover);
            slot = tid - part * cover;
            const int i = (int)bx * cover + slot;
            hEnd = 0x7FFFFFFF;
            h = i < n ? st.listIn[(size_t)p * st.listStride + i] : 0x7FFFFFFF; // (h >= hEnd: idle lane)
            // this wavefront's part of the stage's match range (whole blocks of 64 matches; a part may be empty)
            const int parts = kBlock / cover;
            const int plen = ((m1 - m0 + parts * 64 - 1) / (parts * 64)) * 64;
            m0 += part * plen;
            m1 = m1 < m0 + plen ? m1 : m0 + plen;
            m1 = m1 < m0 ? m0 : m1; // (an empty part: nothing to sweep, no odd last match either)
        }
    }

    // a wavefront without a hypothesis of its own (ps_score_fast.h; not in a pass with a split match range: barrier to come)
    if (cover == kBlock && hFirstOfWave(h, lane) >= hEnd) return false;

    // A launch with a split match range is a small one (a handful of pairs): one wavefront per SIMD, nothing hides a load.
    // The pair's bound and this work-group's records were written by the previous launch (HBM latency, 2 x 1.5 us exposed
    // behind the sample -> SVD chain): their lines are requested here, in front of the chain -- the bound itself, and one
    // dword of every 64-byte line of the record range (a vector load: 64 lines per instruction), whose only use is to be
    // waited for behind the chain.
    const float cmaxEarly = (KIND == 0) ? pairBound[p].x : 0.0f;
    float warm = 0.0f;
    if (KIND == 0 && msplit > 1) {
        const float *w0 = reinterpret_cast<const float *>(recG + (size_t)p * ((size_t)((cap + 1) >> 1) * (RF / 2))) +
                          (size_t)(m0 >> 1) * RF;
        const int words = ((m1 - m0 + 1) >> 1) * RF;
        const int at = tid * 16; // (the work-group's 256 lanes cover 16 KiB; a split range is far shorter)
        if (at < words) warm = w0[at];
    }

    Rigid mdl, inv;
    set_identity(mdl);
    set_identity(inv);
    bool valid = false;
    if (LIST) {
        if (h < hEnd) {
            load_model(ma, (size_t)p * ma.modelH + h, mdl); // parked by stage 1 (only valid samples survive it; h < modelH: a
                                                            // hypothesis beyond is swept completely by stage 1 and never listed)
            valid = true;
        }
    } else if (KIND == 0 && st.validMask != nullptr && !st.genOnly) {
        // stage 0, second launch: models and validity read back (ps_score_fast.h)
        const unsigned long long vm = uniform64(st.validMask[(size_t)p * ((st.hCount + 63) >> 6) + (bx * (kBlock / 64) + wv)]);
        valid = h < hEnd && lane_in(vm);
        if (h < hEnd) load_model(ma, (size_t)p * ma.modelH + h, mdl); // (launches of this form have a slot for every hypothesis)
    } else {
        if (h < hEnd) valid = gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)h, mdl);
        // (stage 1 parks only the models of its survivors, at the end: the abandoned majority is never read again)
        if (ma.models && by == 0 && !pruned) {
            const int hs = stage_hypothesis_again(false, st, (int)bx * kBlock, tid, p, H); // (ps_score_fast.h)
            if (hs < hEnd && hs < ma.modelH) store_model(ma, (size_t)p * ma.modelH + hs, mdl);
        }
        if (KIND == 0 && st.genOnly) { // stage 0, first launch: models and validity only
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
            if (lane == 0) st.validMask[(size_t)p * ((st.hCount + 63) >> 6) + (bx * (kBlock / 64) + wv)] = vm;
            return false;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_mdl[3 * i + j][tid] = mdl.R[i][j];
        s_mdl[9 + i][tid] = mdl.t[i];
    }

    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float2 *__restrict__ pg = recG + (size_t)p * ((size_t)((cap + 1) >> 1) * (RF / 2));
    const float cmax = (KIND == 0) ? cmaxEarly : pairBound[p].x;
    asm volatile("" ::"v"(warm)); // (the warm-up load's destination stays allocated until here)
    // position of the hot record -> match of the original record arrays (stages after ps_stage_reorder; the cold paths)
    const int32_t *__restrict__ pperm = (pruned && st.perm != nullptr) ? st.perm + rbase : nullptr;
    auto orig = [&](int m) { return pperm != nullptr ? pperm[m] : m; };

    float rho = 0.0f, tau = 0.0f;
    model_norms(mdl, rho, tau);
    const float S = (rho * cmax + tau) * 1.001f;
    // (comparisons are false for NaN: a non-finite model or cmax sends the wavefront to the value-exact loop)
    const bool boundsOk = ec.enabled != 0 && S >= 1.0e-20f && S <= 1.0e15f && cmax <= 1.0e15f;
    int cnt = 0;

    if (!wave_all(boundsOk)) {
        for (int m = m0; m < m1; ++m) {
            const int mo = orig(m);
            const float4 A = pa[mo], B = pb[mo];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, A, cnt);
        }
    } else {
        // per-lane limits (see the header): a = 12.5 u S (errorVersion 0) or 10.001 * 17.5 u (S + cmax) (errorVersion 4),
        // both with 1 % slack for their own float roundings
        const float a = MODE == PS_ADAPTIVE_ERROR ? (S + cmax) * (177.0f * kEpsU) : S * (12.7f * kEpsU);
        const float x = ec.tbLo - a;
        const float lo = x > 0.0f ? (x * x) * (1.0f - 8.0f * kEpsU) : -1.0f;
        const float y = ec.tbHi + a * (1.0f + 4.0f * kEpsU);
        const float hi = (y * y) * (1.0f + 8.0f * kEpsU);
        // indicator  ind = clamp01(K - c s~):  exactly 0 for s~ >= hi, exactly 1 for s~ <= lo, fractional in between
        // (see "decision without compares" in the header); c, K rounded to the safe sides
        const float den = __builtin_fmaf(hi, 4.76837158203125e-07f /* 2^-21 */, hi) - lo;
        const float cInd = 0.98f / den;
        const float kInd = (cInd * hi) * (1.0f + 2.384185791015625e-07f /* 2^-22 */);
        const v2f_t negc = splat(-cInd), kk = splat(kInd);
        const v2f_t r00 = splat(mdl.R[0][0]), r01 = splat(mdl.R[0][1]), r02 = splat(mdl.R[0][2]), t0 = splat(mdl.t[0]);
        const v2f_t r10 = splat(mdl.R[1][0]), r11 = splat(mdl.R[1][1]), r12 = splat(mdl.R[1][2]), t1 = splat(mdl.t[1]);
        const v2f_t r20 = splat(mdl.R[2][0]), r21 = splat(mdl.R[2][1]), r22 = splat(mdl.R[2][2]), t2 = splat(mdl.t[2]);
        unsigned long long parked = 0;
        const int m1e = m1 & ~1; // whole pairs; an odd last match is scored by the value-exact code below

        for (int blk = m0; blk < m1e; blk += kEuclidBlock) {
            const int bend = blk + kEuclidBlock < m1e ? blk + kEuclidBlock : m1e;
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
            const bool clean = frac == 0u;
            if (clean) cnt += (int)(acc.x + acc.y);
            unsigned long long todo = __builtin_amdgcn_ballot_w64(!clean);
            while (todo != 0ull) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int t = wv * 64 + l;
                Rigid md;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][t];
                    md.t[i] = s_mdl[9 + i][t];
                }
                const int mm = blk + lane;
                bool in = false;
                if (mm < bend) {
                    const int mo = orig(mm);
                    const float4 A = pa[mo], B = pb[mo];
                    in = inlier_test<MODE>(md, md, k, A, B, A);
                }
                const int c = __popcll(__builtin_amdgcn_ballot_w64(in));
                if (lane == l) cnt += c;
                parked += (unsigned long long)(bend - blk);
            }
        }
        if (m1e < m1) { // odd last match of the range
            const int mo = orig(m1e);
            const float4 A = pa[mo], B = pb[mo];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, A, cnt);
        }
        if (dbg != nullptr && lane == 0) {
            atomicAdd(&dbg[0], parked);
            atomicAdd(&dbg[1], (unsigned long long)(m1 - m0) * 64ull);
        }
    }
    int tidE = tid; // (a fresh copy for the epilogue's LDS addresses: kept alive across the loops they went to scratch memory)
    asm volatile("" : "+v"(tidE));
    if (LIST && cover < kBlock) { // (work-group uniform) the parts of the match range add up
        if (part > 0 && h < hEnd) atomicAdd(&s_tot[slot], cnt);
        __syncthreads();
        if (part == 0) cnt += s_tot[slot];
    }
    if (pruned) {
        h = stage_hypothesis_again(LIST, st, (int)bx * cover, slot, p, H);
        const bool mine = h < hEnd && part == 0;
        if (LIST && msplit > 1) { // (last stage, range split over work-groups: nothing survives it, the counts add up)
            if (mine && valid && cnt) atomicAdd(&cout[h], cnt);
            return false;
        }
        const int cnt0 = (LIST && mine) ? cout[h] : 0; // count so far
        const int total = valid ? cnt0 + cnt : 0; // model not computed -> iteration skipped (RANSAC.cpp:107)
        if (mine) cout[h] = total;
        // still able to become a record?  (count so far + matches left > best count of the earlier hypotheses)
        const bool alive = mine && valid && total + (M - mStageEnd) > best0;
        if (!LIST && ma.models && h < ma.modelH && (alive || (mine && mStageEnd >= M))) { // survivors (or: this stage was the whole sweep)
            Rigid md;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][tidE];
                md.t[i] = s_mdl[9 + i][tidE];
            }
            store_model(ma, (size_t)p * ma.modelH + h, md);
        }
        if (st.stage < kStages && mStageEnd < M) stage_append(alive, h, st.listOut, st.countOut, p, st.listStride);
        return false;
    }
    h = stage_hypothesis_again(false, st, (int)bx * kBlock, tid, p, H);
    if (h < hEnd) {
        if (!valid) cnt = 0; // model not computed -> iteration skipped (RANSAC.cpp:107)
        if (msplit == 1)
            cout[h] = cnt;
        else if (cnt)
            atomicAdd(&cout[h], cnt);
    }
    return false;
}

// (kind 2: five wavefronts per SIMD, see ps_ransac_score_fast; LOOP: stage 1 of an adaptive schedule with a long cap, same place.
// Kind 0 -- the plain launch and stage 0 -- asks for seven: at eight the register allocator spilled 34 scalar registers into
// vector lanes and left a 20-byte private segment behind that no instruction touched; at seven: 8 spills, no
// private segment, the same times within 1 % in every regime, profiles/r05i/ab_seven_waves.txt)
template <int MODE, int KIND = 0, bool LOOP = false>
__global__ __launch_bounds__(kBlock, (KIND == 2 || LOOP) ? 5 : (KIND == 0 ? 7 : 8)) void ps_ransac_score_euclid(
    const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float2 *__restrict__ recG,
    const int32_t *__restrict__ mvalid, const float2 *__restrict__ pairBound, ModelArgs ma, ScoreConsts k,
    EuclidConsts ec, SelectArgs sa, StageArgs st, int H, int cap, int minRun, int msplit, int32_t *__restrict__ counts,
    unsigned long long *__restrict__ dbg)
{
    static_assert(MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR, "the Euclidean metrics");
    // hypotheses of this launch: [hBase, hBase + hCount) (plain launch: [0, H); stages 0 / 1), or a survivor list swept by
    // hCount / 256 work-groups per pair (stages 2+)
    static_assert(!LOOP || KIND == 1, "the looping form is stage 1's");
    const unsigned blocks = (unsigned)((st.hCount + kBlock - 1) / kBlock);
    const unsigned hb = LOOP ? (unsigned)st.loopGroups : blocks; // work-groups per pair and part of the match range
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bx = L % hb, by = (L / hb) % (unsigned)msplit;
    const int p = (int)(L / (hb * (unsigned)msplit));
    const int M = mvalid[p];
    if (M < minRun) return; // too few matches: kernel 4 returns identity (RANSAC.cpp:77-80)
    if (LOOP) {
        for (unsigned b = bx; b < blocks; b += hb) {
            if (b != bx) __syncthreads(); // (the pass before is done with the work-group's LDS)
            if (score_euclid_pass<MODE, KIND>(recA, recB, recG, pairBound, ma, k, ec, sa, st, H, cap, msplit, counts, dbg, b, by, p, M))
                break; // (uniform over the work-group)
        }
    } else if (KIND == 2) {
        const int n = st.countIn[p];
        const int cover = list_cover(n);
        for (unsigned b = bx; (int)b * cover < n; b += hb) {
            if (b != bx) __syncthreads(); // (the pass before is done with the work-group's LDS)
            score_euclid_pass<MODE, KIND>(recA, recB, recG, pairBound, ma, k, ec, sa, st, H, cap, msplit, counts, dbg, b, by, p, M);
        }
    } else
        score_euclid_pass<MODE, KIND>(recA, recB, recG, pairBound, ma, k, ec, sa, st, H, cap, msplit, counts, dbg, bx, by, p, M);
}

} // namespace psdev
