// ps_score_mfma.h -- kernel 3 for the reprojection metric with the two rigid transforms on the matrix cores.
//
// Same decision-exact scheme as ps_ransac_score_fast (ps_score_fast.h: cheap evaluation + proven error band, in-band
// evaluations parked and re-done by the value-exact code), different mapping.  The cheap evaluation spends 18 of its 24
// VALU instructions on the two transforms  X~ = fx (R p + t)_x ...  -- a [hypotheses x 4] by [4 x matches] product per
// component.  v_mfma_f32_16x16x4_f32 computes it as an f32 FMA chain (exact f32, no reduced precision; the error bound
// only needs "at most five roundings per term", which any chain order satisfies) at the f32 vector rate, on a pipe that
// runs beside the VALU:
//
//   A operand  = 16 matches x (x, y, z, 1)           one VGPR: lane l holds coordinate l>>4 of match m0 + (l & 15)
//   B operand  = (r0, r1, r2, t) x 16 hypotheses      one VGPR per model row: lane l holds coefficient l>>4 of hypothesis l & 15
//   D          = 16 x 16 transformed components       4 VGPRs: lane l holds matches m0 + 4 (l>>4) + {0..3} of hypothesis l & 15
//
// so a lane is (hypothesis l & 15 of a 16-hypothesis group, match quarter l >> 4) and the four accumulator registers
// are four matches: six MFMAs (3 components x 2 directions) feed four evaluations per lane, whose projection, offsets,
// squares and band limits run as v_pk_*_f32 over match pairs.  Per 256 evaluations: 6 MFMA + 52 VALU instead of 96 VALU.
// A wave owns 64 hypotheses = four groups that share the A operands and the per-match offsets of a 16-match tile; the
// inlier counts stay per lane and the four match quarters of a hypothesis are summed once at the end (two shuffles).
// The prologue (sample -> Umeyama -> inverse, one hypothesis per lane) is unchanged; it parks the folded model rows in
// LDS, from where the B operands are gathered.
//
// Error band: as in ps_score_fast.h with the FMA-chain depth of the numerators and of the denominator raised by one
// (the product with the constant 1 of the fourth K slot): eta = 16 u fmaxK S, zeta = 10 u S.
#pragma once

#include <type_traits>

#include "ps_score_fast.h"

namespace psdev {

typedef float v4f_t __attribute__((ext_vector_type(4)));

PS_D v4f_t mfma16(float a, float b)
{
    const v4f_t z = {0.0f, 0.0f, 0.0f, 0.0f};
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, z, 0, 0, 0);
}

// Value-exact evaluation of the (match, hypothesis) pairs one wave has parked, one lane each.  Out of line: it is cold
// code with sixteen call sites (every accumulator register of every hypothesis group can park), inlined it multiplied
// the kernel to 11 600 instructions.
template <int MODE>
__device__ __attribute__((noinline)) void drain_parked16(const float (*s_mdl)[kBlock], const uint32_t *s_qw, int *s_cnt,
                                                         const float4 *__restrict__ pa, const float4 *__restrict__ pb,
                                                         const float4 *__restrict__ pc, const ScoreConsts &k, int qn,
                                                         int lane, int hw)
{
    for (int e = lane; e < qn; e += 64) {
        const uint32_t ent = s_qw[e];
        const int t = hw + (int)(ent & 63u), m = (int)(ent >> 6);
        Rigid md, iv;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) md.R[i][jj] = s_mdl[3 * i + jj][t];
            md.t[i] = s_mdl[9 + i][t];
        }
        inverse_rigid_general(md, iv);
        const float4 A = pa[m], B = pb[m], C = pc[m];
        if (inlier_test<MODE>(md, iv, k, A, B, C)) atomicAdd(&s_cnt[t], 1);
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock, 3) void ps_ransac_score_mfma(
    const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float4 *__restrict__ recC,
    const float4 *__restrict__ recP, const float *__restrict__ recE4, const int32_t *__restrict__ mvalid,
    const float2 *__restrict__ pairBound, ModelArgs ma, ScoreConsts k, FastConsts fc, int H, int cap, int capE, int minRun,
    int msplit, int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg)
{
    static_assert(MODE == PS_REPROJECTION_ERROR, "the matrix-core path covers the reprojection metric");
    __shared__ float s_mdl[12][kBlock];      // exact model of every hypothesis (for the parked evaluations)
    __shared__ float s_fold[24][kBlock];     // folded rows: [(direction * 3 + component) * 4 + coefficient][hypothesis]
    __shared__ float s_band[3][kBlock];      // a1, b1, rcap of every hypothesis
    __shared__ uint32_t s_q[kBlock / 64][kQueueCap];
    __shared__ int s_cnt[kBlock];

    const unsigned hb = (unsigned)((H + kBlock - 1) / kBlock);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bx = L % hb, by = (L / hb) % (unsigned)msplit;
    const int p = (int)(L / (hb * (unsigned)msplit));
    const int M = mvalid[p];
    if (M < minRun) return; // too few matches: kernel 4 returns identity (RANSAC.cpp:77-80)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = (int)bx * kBlock + tid;
    const size_t rbase = (size_t)p * cap;
    const int tiles = (M + 15) >> 4; // 16-match tiles; the match range of this work-group in whole tiles
    const int tl0 = (int)(((long long)tiles * by) / msplit), tl1 = (int)(((long long)tiles * (by + 1)) / msplit);

    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float4 *__restrict__ pc = recC + rbase;
    const float2 pbnd = pairBound[p];
    const float cmax = pbnd.x, umax = pbnd.y;

    // ---- prologue: one hypothesis per lane ----
    Rigid mdl, inv;
    set_identity(mdl);
    bool valid = false;
    if (h < H) valid = gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)h, mdl);
    if (ma.models && by == 0 && h < H) store_model(ma, (size_t)p * H + h, mdl);
    inverse_rigid_general(mdl, inv);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_mdl[3 * i + j][tid] = mdl.R[i][j];
        s_mdl[9 + i][tid] = mdl.t[i];
    }
    s_cnt[tid] = 0;
    float rho = 0.0f, tau = 0.0f;
    model_norms(mdl, rho, tau);
    model_norms(inv, rho, tau);
    const float S = (rho * cmax + tau) * 1.001f;
    const bool boundsOk = fc.enabled != 0 && S * fc.fmaxK <= kDivHi && S >= 1.0e-20f && umax <= 1.0e7f;
    int cnt = 0;

    if (!wave_all(boundsOk)) {
        // value-exact loop, one hypothesis per lane (non-finite models, coordinates or offsets beyond the bounds)
        const int mEnd = tl1 * 16 < M ? tl1 * 16 : M;
        for (int m = tl0 * 16; m < mEnd; ++m) {
            const float4 A = pa[m], B = pb[m], C = pc[m];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, C, cnt);
        }
    } else {
        {
            const float sx[3] = {k.fx, k.fy, 1.0f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    s_fold[(c * 4) + j][tid] = sx[c] * mdl.R[c][j];       // direction 0: current point -> previous image
                    s_fold[((3 + c) * 4) + j][tid] = sx[c] * inv.R[c][j]; // direction 1: previous point -> current image
                }
                s_fold[(c * 4) + 3][tid] = sx[c] * mdl.t[c];
                s_fold[((3 + c) * 4) + 3][tid] = sx[c] * inv.t[c];
            }
            const float Qin = 1.02f * (umax + fc.thrUp + 0.016f * fc.fmaxK + 1.0f);
            const float kap1 = S * (1.02f * kEpsU * (16.0f * fc.fmaxK + 10.0f * Qin));
            const float kap0 = 1.02f * kEpsU * (8.0f * Qin + 2.0f * fc.cmaxK + 2.0f * fc.thrUp + 2.0f + umax);
            const float rcap2 = fc.thrUp > kap0 ? ((fc.thrUp - kap0) / kap1) * 0.99999f : -1.0f;
            const float up4 = 1.0f + 4.0f * kEpsU;
            s_band[0][tid] = (fc.cIn * kap1) * up4;            // a1
            s_band[1][tid] = (fc.cHi * kap1) * up4;            // b1
            s_band[2][tid] = fminf(8192.0f / S, rcap2);        // rcap: |Z~| >= 2^-13 S (= 102 zeta)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the wave reads back what its own lanes have just parked
        // uniform band constants (kap0 does not depend on the hypothesis)
        const float Qin = 1.02f * (umax + fc.thrUp + 0.016f * fc.fmaxK + 1.0f);
        const float kap0 = 1.02f * kEpsU * (8.0f * Qin + 2.0f * fc.cmaxK + 2.0f * fc.thrUp + 2.0f + umax);
        const float up4 = 1.0f + 4.0f * kEpsU;
        const float a0 = (fc.bIn0 - fc.cIn * kap0) - 4.0f * kEpsU * fc.bIn0;
        const float b0 = (fc.thr2Up + fc.cHi * kap0) * up4;

        const int j = lane & 15, kk = lane >> 4, hw = wv * 64;
        float Bop[4][6], a1[4], b1[4], rc[4];
        int cg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hh = hw + 16 * g + j;
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) Bop[g][c6] = s_fold[c6 * 4 + kk][hh];
            a1[g] = s_band[0][hh];
            b1[g] = s_band[1][hh];
            rc[g] = s_band[2][hh];
            cg[g] = 0;
        }
        int qn = 0;
        unsigned long long parked = 0;
        auto drain = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain_parked16<MODE>(s_mdl, s_q[wv], s_cnt, pa, pb, pc, k, qn, lane, hw);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            parked += (unsigned long long)qn;
            qn = 0;
        };

        const unsigned long long execAll = __builtin_amdgcn_ballot_w64(true);
        const float *__restrict__ fcur = reinterpret_cast<const float *>(recB + rbase);
        const float *__restrict__ fprev = reinterpret_cast<const float *>(recP + rbase);
        const float4 *__restrict__ e4 = reinterpret_cast<const float4 *>(recE4 + (size_t)p * capE * 4);
        // One 16-match tile for the wave's four hypothesis groups.  The uncertain masks of all 16 (group, register)
        // decisions are kept in SGPRs and handled behind ONE branch per tile, so the hot path of a tile is a single
        // basic block (24 MFMAs + ~210 VALU) in which the scheduler can run the matrix instructions of the later groups
        // beside the vector work of the earlier ones instead of stalling on every result.
        auto tile = [&](int tl, auto tailTag) {
            constexpr bool TAIL = decltype(tailTag)::value;
            const int mt = tl * 16;
            int row = mt + j;
            if (TAIL) row = row < M ? row : M - 1;
            const float aCur = fcur[row * 4 + kk], aPrev = fprev[row * 4 + kk];
            // offsets of this lane's four matches mt + 4 kk + {0..3}: (cx - uOld), (cx - uNew), (cy - vOld), (cy - vNew)
            const float4 *eq = e4 + ((size_t)(mt >> 2) + kk) * 4;
            const float4 eXo = eq[0], eXn = eq[1], eYo = eq[2], eYn = eq[3];
            // matches of the last tile beyond M take no part: neither inlier nor uncertain
            unsigned long long mV[4] = {execAll, execAll, execAll, execAll};
            if (TAIL) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) mV[r4] = __builtin_amdgcn_ballot_w64(mt + 4 * kk + r4 < M);
            }
            unsigned long long mUa[4][4];
            unsigned long long anyU = 0ull;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f_t Xe = mfma16(aCur, Bop[g][0]), Ye = mfma16(aCur, Bop[g][1]), Ze = mfma16(aCur, Bop[g][2]);
                const v4f_t Xn = mfma16(aPrev, Bop[g][3]), Yn = mfma16(aPrev, Bop[g][4]), Zn = mfma16(aPrev, Bop[g][5]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int r0 = 2 * pr, r1 = 2 * pr + 1;
                    const v2f_t re = {__builtin_amdgcn_rcpf(Ze[r0]), __builtin_amdgcn_rcpf(Ze[r1])};
                    const v2f_t rn = {__builtin_amdgcn_rcpf(Zn[r0]), __builtin_amdgcn_rcpf(Zn[r1])};
                    const v2f_t due = pk_fma(v2f_t{Xe[r0], Xe[r1]}, re, v2f_t{eXo[r0], eXo[r1]});
                    const v2f_t dve = pk_fma(v2f_t{Ye[r0], Ye[r1]}, re, v2f_t{eYo[r0], eYo[r1]});
                    const v2f_t dun = pk_fma(v2f_t{Xn[r0], Xn[r1]}, rn, v2f_t{eXn[r0], eXn[r1]});
                    const v2f_t dvn = pk_fma(v2f_t{Yn[r0], Yn[r1]}, rn, v2f_t{eYn[r0], eYn[r1]});
                    const v2f_t se = pk_fma(due, due, dve * dve);
                    const v2f_t sn = pk_fma(dun, dun, dvn * dvn);
                    const v2f_t rm = {fmaxf(fabsf(re.x), fabsf(rn.x)), fmaxf(fabsf(re.y), fabsf(rn.y))};
                    const v2f_t lo2 = pk_fma(v2f_t{-a1[g], -a1[g]}, rm, v2f_t{a0, a0});
                    const v2f_t hi2 = pk_fma(v2f_t{b1[g], b1[g]}, rm, v2f_t{b0, b0});
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r4 = 2 * pr + q;
                        const float so = q ? se.y : se.x, sN = q ? sn.y : sn.x;
                        const float rmq = q ? rm.y : rm.x, loq = q ? lo2.y : lo2.x, hiq = q ? hi2.y : hi2.x;
                        const uint32_t uo = __builtin_bit_cast(uint32_t, so), un = __builtin_bit_cast(uint32_t, sN);
                        const float sm = __builtin_bit_cast(float, max(uo, un));
                        unsigned long long mZ = __builtin_amdgcn_ballot_w64(rmq <= rc[g]);
                        if (TAIL) mZ &= mV[r4];
                        const unsigned long long mIn = __builtin_amdgcn_ballot_w64(sm < loq) & mZ;
                        unsigned long long mOut = __builtin_amdgcn_ballot_w64(sm > hiq) & mZ;
                        if (TAIL) mOut |= ~mV[r4];
                        add_mask(cg[g], mIn);
                        mUa[g][r4] = execAll & ~(mIn | mOut);
                        anyU |= mUa[g][r4];
                    }
                }
            }
            if (anyU != 0ull) { // cold: park the uncertain (match, hypothesis) pairs of this tile
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const unsigned long long mU = mUa[g][r4];
                        if (mU != 0ull) {
                            const int n = __popcll(mU);
                            if (qn + n > kQueueCap) drain();
                            if ((mU >> lane) & 1ull)
                                s_q[wv][qn + __popcll(mU & ((1ull << lane) - 1ull))] =
                                    ((uint32_t)(mt + 4 * kk + r4) << 6) | (uint32_t)(16 * g + j);
                            qn += n;
                        }
                    }
            }
        };
        const int fullEnd = (tl1 * 16 <= M) ? tl1 : tl1 - 1; // only the very last tile of the pair can be partial
        for (int tl = tl0; tl < fullEnd; ++tl) tile(tl, std::false_type{});
        if (fullEnd < tl1 && fullEnd >= tl0) tile(fullEnd, std::true_type{});
        drain();
        // the four match quarters of every hypothesis: lanes j, j + 16, j + 32, j + 48
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int v = cg[g];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kk == 0) atomicAdd(&s_cnt[hw + 16 * g + j], v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt = s_cnt[tid];
        if (dbg != nullptr && lane == 0) {
            atomicAdd(&dbg[0], parked);
            atomicAdd(&dbg[1], (unsigned long long)(tl1 - tl0) * 16ull * 64ull);
        }
    }
    if (h < H) {
        if (!valid) cnt = 0; // model not computed -> iteration skipped (RANSAC.cpp:107)
        if (msplit == 1)
            counts[(size_t)p * H + h] = cnt;
        else if (cnt)
            atomicAdd(&counts[(size_t)p * H + h], cnt);
    }
}

} // namespace psdev
