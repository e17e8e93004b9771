// ps_score_mfma.h -- kernel 3 for the reprojection metric with the two rigid transforms AND the offset products on the
// matrix cores, in split f16.
//
// Same decision-exact scheme as ps_ransac_score_fast (ps_score_fast.h: cheap evaluation + proven error band, in-band
// evaluations parked and re-done by the value-exact code).  The cheap evaluation there spends 11 of its 16 packed
// instructions on        X~ = fx (R p + t)_x,  Y~,  Z~        and        A~ = X~ + (cx - u_real) Z~,  B~ = Y~ + (cy - v_real) Z~.
// With k = c - real folded into the MATCH side these are three plain dot products over eight "coordinates":
//
//       A~ = [ p_x  p_y  p_z  1 | k p_x  k p_y  k p_z  k ] . [ fx R_0j  fx t_0 | R_2j  t_2 ]
//       B~ = [ p      1         | k' p         k'        ] . [ fy R_1j  fy t_1 | R_2j  t_2 ]
//       Z~ = [ p      1 ] . [ R_2j  t_2 ]
//
// i.e. [32 matches x K] by [K x 32 hypotheses] products.  An f32 MFMA has the vector ALU's rate and would only move the
// work (profiles/variants/ps_score_mfma_f32.h.txt: 8 % slower); v_mfma_f32_32x32x16_f16 is 16 times faster, and f16 is
// enough when every operand is SPLIT: v 2^e = hi + lo, hi = f16(v 2^e), lo = f16(v 2^e - hi), and a product becomes
// the four slots hi hi + hi lo + lo hi + lo lo of the K dimension (4 coordinates x 4 slots = one 32x32x16 MFMA; the
// eight-coordinate rows are two MFMAs chained through the accumulator).  Powers of two 2^e (per pair for the match side,
// per hypothesis for the model side) keep every operand in f16's range; they cancel in the test, which is homogeneous.
//
//   A operand (match side, kernel 2 writes it: RecPtrs::H)   lane l: match tile + (l & 31), K-block l >> 5 = coordinates 2(l>>5), +1
//   B operand (model side, built in the prologue, via LDS)   lane l: hypothesis group + (l & 31), same K-block
//   D (v16f)                                                  lane l: hypothesis (l & 31), matches tile + (r&3) + 8(r>>2) + 4(l>>5)
//
// so a lane is (hypothesis, match half): per-hypothesis band constants live in registers, register pairs are adjacent
// matches for v_pk_*_f32, and NO per-match data is needed by the vector part.
//
// Decisions come from SIGN BITS, not compares: per direction  ni = s~ - w (b w - cL)  (one FMA rounding: the sign is exact)
// and  no = upper limit - s~;  v_alignbit_b32 gathers the 16 signs of a lane's accumulator registers into two bit
// fields, and inlier / certain / uncertain are three integer instructions per 16 evaluations:
//       in = inE & inN;   certain = in | outE | outN;   count += popcount(in);   uncertain = ~certain
// (logic instructions issue at 2.4 cycles against 4.3 for a compare, and no SGPR masks have to be kept).  That needs every
// value of the hot loop to be finite -- see boundsOk.  Per 1024 evaluations: 10 MFMA + ~265 VALU against 23 x 16 = 368.
//
// What it achieves (499 pairs, 2000 keypoints, H = 4096; profiles/r02i): 1.61 ms against 1.69 ms for ps_ransac_score_fast
// single-chain, 249 k against 253 k frame-pairs/s in the three-chain timed region: a tested option
// (PUTSLAM_HIP_SCORE=mfma), not the default.
// Why not more (profiles/microbench/mfma_valu_coissue.hip, valu_dep.hip):
//   * on one gfx950 SIMD a v_mfma_f32_32x32x16_f16 stream and a v_pk_fma_f32 stream do NOT overlap (one wave or two:
//     the times add); 10 MFMAs cost 400 cycles per 1024 evaluations on top of the vector work;
//   * 230-256 VGPRs (two accumulator sets of 48 registers, two tiles of match operands) leave two waves per SIMD; the
//     prologue (0.25 ms of dependent arithmetic) is hidden much worse than at the fast kernel's seven;
//   * see SOURCE-OPERAND HAZARD below: 12-24 registers are spent on keeping MFMA source operands untouched.
//
// Error band.  u = 2^-24, S as in ps_score_fast.h.  Against the real A* = fx E_x + (cx - u_real) E_z:
//   operands   a_j = p_j, 1, fl(fl(cx - u_real) p_j), fl(cx - u_real);  b_j = fl(fx R_0j), fl(fx t_0), R_2j, t_2:
//                 sum |a_j b_j - alpha_j beta_j|              <= u S (1.01 fmaxK + 2.01 Umax)
//   split      |v - hi - lo| <= 2^-22 |v| + 2^-25 (scaled):  <= 8.04 u sum|a_j b_j|  +  2^-7 / s1          (s1 = the accumulator's scale)
//   MFMA       one v_mfma_f32_32x32x16_f16 returns c + sum a_k b_k within  gamma (|c| + sum |a_k b_k|).  The ISA does not
//              state gamma.  profiles/microbench/mfma_f16_probe*.hip show the adder of gfx950: each K-block of 8 products
//              is aligned to its largest exponent sum and truncated one bit below that product's last f32 place
//              (< 7 u pmax), the two block sums and c are aligned with three guard bits and rounded to nearest even:
//              gamma <= 8.75 u; 3 M adversarial dot products stay below 5.4 u (mfma_f16_acc.hip, also run by
//              tests/test_gpu_mfma_accuracy.py on the device under test).  This kernel ASSUMES gamma = 12 u; two chained:
//                 <= 2.006 gamma 1.002 sum|a_j b_j|          <= 24.2 u sum|a_j b_j|
//   together   |A~ - A*| <= u S (33.3 fmaxK + 34.3 Umax) + 2^-6 / s1        (ps_score_fast.h's chain: 5.02 fmaxK + 5.03 Umax)
//              |Z~ - E_z| <= 20.2 u S + 2^-7 / s2
//   with the reference's own distance to A* (ps_score_fast.h: 8.03 fmaxK + 5.02 Umax + 3.02 cmaxK, and 5 u S for e_z):
//       E = 1.05 u (42 fmaxK + 40 Umax + 4 cmaxK) S + 2^-6 / s1,     |e_z - Z~| <= 26 u S + 2^-7 / s2,
//       G = sqrt2 E + T' (26 u S + 2^-7 / s2)
//   and the same two limits  s~ < w (T^2 (1 - 20u) w - 2 T' G)  /  s~ > w (T'^2 w + 2 T' G) + G^2 (1 + 1e-4), evaluated on the
//   SCALED accumulators (all scale factors are powers of two: no further rounding).  The band is ~3 times wider than the
//   f32 chain's: 0.21 % instead of 0.07 % of the evaluations are parked.
#pragma once

#include <type_traits>

#include "ps_score_fast.h"

namespace psdev {

typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));


PS_D v16f_t mfma_h(const uint4 &a, const uint4 &b, const v16f_t &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h_t, a), __builtin_bit_cast(v8h_t, b), c, 0, 0, 0);
}

// Value-exact evaluation of the (match, hypothesis) pairs one wave has parked, one lane each.  Out of line: cold code
// with several call sites.
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

// One K-block (two coefficients) of a model-side operand row: slots (hi, lo, hi, lo) per coefficient; the register
// image of the block is (w0, w0, w1, w1).
PS_D uint2 model_block(float c0, float c1)
{
    uint32_t h0, l0, h1, l1;
    split_f16(c0, h0, l0);
    split_f16(c1, h1, l1);
    return make_uint2(h0 | (l0 << 16), h1 | (l1 << 16));
}

template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void ps_ransac_score_mfma(
    const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float4 *__restrict__ recC,
    const uint4 *__restrict__ recH, const int2 *__restrict__ pairScale, const int32_t *__restrict__ mvalid,
    const float2 *__restrict__ pairBound, ModelArgs ma, ScoreConsts k, FastConsts fc, int H, int cap, int capH, int minRun,
    int msplit, int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg)
{
    static_assert(MODE == PS_REPROJECTION_ERROR, "the matrix-core path covers the reprojection metric");
    __shared__ float s_mdl[12][kBlock];      // exact model of every hypothesis (for the parked evaluations)
    __shared__ uint2 s_bop[2][3][2][kBlock]; // model-side operands: [direction][row X Y Z][K-block][hypothesis]
    __shared__ float s_band[4][kBlock];      // cL'' and G2'' of both directions
    __shared__ uint32_t s_q[kBlock / 64][kQueueCap];
    __shared__ int s_cnt[kBlock];
    // (46 KiB in all.  A trial with an eight times larger queue -- 74.75 KiB, queue and counters above the 64-KiB mark --
    // gave wrong counts now and then whenever two work-groups shared a CU, although the launch is accepted: data of one
    // work-group above 64 KiB is not safe from a co-resident work-group on this stack, gfx950 / ROCm 7.2.)

    const unsigned hb = (unsigned)((H + kBlock - 1) / kBlock);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bx = L % hb, by = (L / hb) % (unsigned)msplit;
    const int p = (int)(L / (hb * (unsigned)msplit));
    const int M = mvalid[p];
    if (M < minRun) return; // too few matches: kernel 4 returns identity (RANSAC.cpp:77-80)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = (int)bx * kBlock + tid;
    const size_t rbase = (size_t)p * cap;
    const int tiles = (M + 31) >> 5; // 32-match tiles; the match range of this work-group in whole tiles
    const int tl0 = (int)(((long long)tiles * by) / msplit), tl1 = (int)(((long long)tiles * (by + 1)) / msplit);

    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float4 *__restrict__ pc = recC + rbase;
    const float2 pbnd = pairBound[p];
    const float cmax = pbnd.x, umax = pbnd.y;
    const int2 psc = pairScale[p];
    const int eP = psc.x, kap = psc.y;

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
    // the thresholds in the scaled domain: s1 / s2 = 2^-kappa for both directions and every hypothesis
    const float bIn2 = ldexpf(fc.bIn0, -2 * kap), thr2 = ldexpf(fc.thr2Up, -2 * kap);
    // The hot loop takes its decisions from SIGN BITS, so nothing in it may be NaN or infinite: under these bounds
    // every operand is finite, |accumulator| < 2^31, s~ < 2^63, thr2 w^2 < 2^103.  (Comparisons are false for NaN: a
    // non-finite model, cmax or umax -- kernel 2 reports non-finite records that way -- sends the wavefront to the
    // value-exact loop.)
    const bool boundsOk = fc.enabled != 0 && S * fc.fmaxK <= kDivHi && S >= 1.0e-20f && umax <= 1.0e7f &&
                          fc.fmaxK <= 1048576.0f && bIn2 >= 1.0e-30f && thr2 <= 1.0e12f && cmax <= 1.0e12f;
    int cnt = 0;

    if (!wave_all(boundsOk)) {
        // value-exact loop, one hypothesis per lane (non-finite models, coordinates, offsets or thresholds beyond the bounds)
        const int mEnd = tl1 * 32 < M ? tl1 * 32 : M;
        for (int m = tl0 * 32; m < mEnd; ++m) {
            const float4 A = pa[m], B = pb[m], C = pc[m];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, C, cnt);
        }
    } else {
        {
            const float lam = 1.05f * kEpsU * (42.0f * fc.fmaxK + 40.0f * umax + 4.0f * fc.cmaxK);
            const float up4 = 1.0f + 4.0f * kEpsU;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                // direction 0: current point -> previous image (the model); 1: previous point -> current image (its inverse)
                const Rigid &m = d ? inv : mdl;
                float mx = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) mx = fmaxf(mx, fabsf(m.R[i][j]));
                    mx = fmaxf(mx, fabsf(m.t[i]));
                }
                // every scaled coefficient stays below 2^14: |fx R| 2^(eZ - kappa) <= |R| 2^eZ because 2^kappa >= fx, fy
                const int eZ = 14 - exp_ceil(mx), eX = eZ - kap;
                const float sx[3] = {k.fx, k.fy, 1.0f};
                const int ex[3] = {eX, eX, eZ};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float b0 = ldexpf(sx[c] * m.R[c][0], ex[c]), b1 = ldexpf(sx[c] * m.R[c][1], ex[c]);
                    const float b2 = ldexpf(sx[c] * m.R[c][2], ex[c]), b3 = ldexpf(sx[c] * m.t[c], ex[c]);
                    s_bop[d][c][0][tid] = model_block(b0, b1);
                    s_bop[d][c][1][tid] = model_block(b2, b3);
                }
                // band of this direction, in the units of its scaled accumulators (s1 for A~, B~; s2 for Z~)
                const float r1 = ldexpf(1.0f, -(eP + eX)), r2 = ldexpf(1.0f, -(eP + eZ)); // 1 / s1, 1 / s2
                const float E = lam * S + 0.015625f * r1;
                const float zerr = (26.0f * kEpsU) * S + 0.0078125f * r2;
                const float G = ((1.4143f * E + (fc.thrUp * zerr) * up4) * up4) * up4;
                const float cL = (2.0f * fc.thrUp * G) * (1.00001f * up4); // 2 T' G, rounded up
                const float G2 = (G * G) * 1.0001f;
                // scaled: cL'' = cL s1 2^-kappa, G2'' = G2 s1^2 (powers of two: exact)
                s_band[2 * d][tid] = ldexpf(cL, eP + eX - kap);
                s_band[2 * d + 1][tid] = ldexpf(G2, 2 * (eP + eX));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the wave reads back what its own lanes have just parked

        const int j = lane & 31, kb = lane >> 5, hw = wv * 64;
        float cLs[2][2], G2s[2][2];
        int cg[2] = {0, 0};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int hh = hw + 32 * g + j;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                cLs[g][d] = s_band[2 * d][hh];
                G2s[g][d] = s_band[2 * d + 1][hh];
            }
        }
        int qn = 0;
        unsigned long long parked = 0;
        v16f_t eA, eB, eZ, nA, nB, nZ; // the two accumulator sets of the main loop
        auto drain = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain_parked16<MODE>(s_mdl, s_q[wv], s_cnt, pa, pb, pc, k, qn, lane, hw);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            parked += (unsigned long long)qn;
            qn = 0;
        };

        const uint4 *__restrict__ hp = recH + (size_t)p * 6 * capH * 2;
        const size_t kind = (size_t)capH * 2;
        const v16f_t zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const v2f_t bIn2v = {bIn2, bIn2}, thr2v = {thr2, thr2};

        // the six match-side operands of one 32-match tile (rows beyond M, last tile of the pair only, hold whatever the
        // arena holds: a row only reaches its own 32 results, which the validity words drop)
        auto loadA = [&](int tl, uint4(&A)[2][3]) {
            const uint4 *row = hp + ((size_t)(tl * 32 + j)) * 2 + kb;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int c = 0; c < 3; ++c) A[d][c] = row[(size_t)(3 * d + c) * kind];
        };
        // the five MFMAs of one (tile, hypothesis group, direction): Z~, A~ = X + kx Z, B~ = Y + ky Z
        auto issue = [&](const uint4(&A)[3], int g, int d, v16f_t &aA, v16f_t &aB, v16f_t &aZ, uint4(&Bk)[3]) {
            const int hh = hw + 32 * g + j;
            const uint2 wX = s_bop[d][0][kb][hh], wY = s_bop[d][1][kb][hh], wZ = s_bop[d][2][kb][hh];
            Bk[0] = make_uint4(wX.x, wX.x, wX.y, wX.y);
            Bk[1] = make_uint4(wY.x, wY.x, wY.y, wY.y);
            Bk[2] = make_uint4(wZ.x, wZ.x, wZ.y, wZ.y);
            const uint4 &bX = Bk[0], &bY = Bk[1], &bZ = Bk[2];
            aZ = mfma_h(A[0], bZ, zero);
            aA = mfma_h(A[1], bZ, mfma_h(A[0], bX, zero));
            aB = mfma_h(A[2], bZ, mfma_h(A[0], bY, zero));
        };
        // one direction of a register pair: sign(ni) set <=> s~ below the lower limit, sign(no) set <=> s~ above the upper
        auto evalPair = [&](const v16f_t &aA, const v16f_t &aB, const v16f_t &aZ, int r, v2f_t cl, v2f_t g2, v2f_t &nI,
                            v2f_t &nO) {
            const v2f_t A2 = {aA[r], aA[r + 1]}, B2 = {aB[r], aB[r + 1]}, Z2 = {aZ[r], aZ[r + 1]};
            const v2f_t ss = pk_fma(A2, A2, B2 * B2);
            const v2f_t w = __builtin_elementwise_max(Z2, -Z2);
            const v2f_t t = pk_fma(bIn2v, w, -cl);
            nI = pk_fma(-t, w, ss);                             // s~ - w (b w - cL): one rounding, the sign is exact
            nO = pk_fma(pk_fma(thr2v, w, cl), w, g2) - ss;      // limit - s~
        };
        // SOURCE-OPERAND HAZARD (found the hard way, tests/test_gpu_batch.py::test_full_size_properties): the compiler
        // re-uses the A / B source registers of a v_mfma_f32_32x32x16_f16 for something else in the very next
        // instruction.  With ONE wave on the SIMD that is fine; with TWO waves sharing the matrix pipe (this kernel's
        // occupancy) the MFMA -- in particular the second of a pair chained through the accumulator -- reads them later
        // than that, and hypothesis counts came out wrong and different from run to run.  keep() reserves the registers
        // of a model-side operand set up to the point where it is placed: behind the vector code that has consumed the
        // results of the MFMAs the set fed (a consumed result = a finished MFMA).  The match-side sets (Acur / Anext)
        // outlive their MFMAs by construction.
        typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
        auto keep = [&](const uint4(&Bk)[3]) {
            asm volatile("" ::"v"(__builtin_bit_cast(u4_t, Bk[0])), "v"(__builtin_bit_cast(u4_t, Bk[1])),
                         "v"(__builtin_bit_cast(u4_t, Bk[2])));
        };
        // One direction of a (tile, group): the sign bits of its 16 accumulator registers gathered into two bit fields
        // (bit 15 - r belongs to register r): inB = s~ below the lower limit, outB = s~ above the upper limit.
        auto evalRegs = [&](const v16f_t &aA, const v16f_t &aB, const v16f_t &aZ, int g, int d, auto r0Tag, auto r1Tag,
                            uint32_t &inB, uint32_t &outB) {
            constexpr int R0 = decltype(r0Tag)::value, R1 = decltype(r1Tag)::value;
            const v2f_t cl = {cLs[g][d], cLs[g][d]}, g2 = {G2s[g][d], G2s[g][d]};
#pragma unroll
            for (int r = R0; r < R1; r += 2) {
                v2f_t nI, nO;
                evalPair(aA, aB, aZ, r, cl, g2, nI, nO);
                const float i0 = nI.x, i1 = nI.y, o0 = nO.x, o1 = nO.y;
                inB = __builtin_amdgcn_alignbit(inB, __builtin_bit_cast(uint32_t, i0), 31);
                outB = __builtin_amdgcn_alignbit(outB, __builtin_bit_cast(uint32_t, o0), 31);
                inB = __builtin_amdgcn_alignbit(inB, __builtin_bit_cast(uint32_t, i1), 31);
                outB = __builtin_amdgcn_alignbit(outB, __builtin_bit_cast(uint32_t, o1), 31);
            }
        };
        typedef std::integral_constant<int, 0> R0_t;
        typedef std::integral_constant<int, 2> R2_t;
        typedef std::integral_constant<int, 16> R16_t;
        // head: the first register pair (once a result has been read its MFMAs -- issued in order -- are all done);
        // tail: the other seven
        auto evalHead = [&](const v16f_t &aA, const v16f_t &aB, const v16f_t &aZ, int g, int d, uint32_t &inB, uint32_t &outB) {
            inB = 0u;
            outB = 0u;
            evalRegs(aA, aB, aZ, g, d, R0_t{}, R2_t{}, inB, outB);
        };
        auto evalTail = [&](const v16f_t &aA, const v16f_t &aB, const v16f_t &aZ, int g, int d, uint32_t &inB, uint32_t &outB) {
            evalRegs(aA, aB, aZ, g, d, R2_t{}, R16_t{}, inB, outB);
        };
        // The decision for the 16 matches x 64 lanes of a (tile, group) from the bit fields of both directions.
        // (fA, fB, fZ: the accumulator set in flight while this runs)
        auto decide = [&](uint32_t inE, uint32_t outE, uint32_t inN, uint32_t outN, int g, int mt, const v16f_t &fA,
                          const v16f_t &fB, const v16f_t &fZ, auto tailTag) {
            constexpr bool TAIL = decltype(tailTag)::value;
            uint32_t in = inE & inN;            // certain inlier: below the lower limit in both directions
            uint32_t c = in | outE | outN;      // certain either way
            if (TAIL) { // matches of the last tile beyond M take no part: neither inlier nor uncertain
                uint32_t vb = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) vb = (vb << 1) | (uint32_t)(mt + (r & 3) + 8 * (r >> 2) + 4 * kb < M);
                in &= vb;
                c |= ~vb;
            }
            cg[g] += __popc(in);
            uint32_t u = ~c & 0xFFFFu;
            unsigned long long mk = __builtin_amdgcn_ballot_w64(u != 0u);
            if (mk != 0ull) { // cold: park the uncertain (match, hypothesis) pairs of this group
                do {
                    const int n = __popcll(mk);
                    if (qn + n > kQueueCap) {
                        // The drain is a real call, and the callee -- compiled on its own -- knows nothing about the MFMAs
                        // this wave has in flight (the other direction's set, issued at the start of this step): if it
                        // saved and restored a register one of them is about to write, that result would be lost.  A vector
                        // instruction that reads the last register of each of the three accumulators makes the
                        // compiler's hazard logic wait for them (they are live across the call anyway).
                        const float w0 = fA[15], w1 = fB[15], w2 = fZ[15];
                        const int sink = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w0)) ^
                                         __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w1)) ^
                                         __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w2));
                        asm volatile("" ::"s"(sink));
                        drain();
                    }
                    if (u != 0u) {
                        const int b = 31 - __clz((int)u), r = 15 - b;
                        s_q[wv][qn + __popcll(mk & ((1ull << lane) - 1ull))] =
                            ((uint32_t)(mt + (r & 3) + 8 * (r >> 2) + 4 * kb) << 6) | (uint32_t)(32 * g + j);
                        u &= ~(1u << b);
                    }
                    qn += n;
                    mk = __builtin_amdgcn_ballot_w64(u != 0u);
                } while (mk != 0ull);
            }
        };

        // Full tiles, software-pipelined by direction: while the vector ALU turns one direction's accumulators into sign
        // words, the next direction's MFMAs are already in flight (two accumulator sets, ping-pong).  Two waves per SIMD.
        // Measured alternatives (profiles/variants/README.md): one accumulator set at three waves per SIMD 1.64 ms
        // against 1.47; the first form of this kernel (compare-based masks in SGPRs, no pipelining) 1.90.
        const int fullEnd = (tl1 * 32 <= M) ? tl1 : tl1 - 1; // only the very last tile of the pair can be partial
        uint4 BE[3], BN[3]; // the model-side operand sets of the two directions
        auto evalDir = [&](const v16f_t &aA, const v16f_t &aB, const v16f_t &aZ, int g, int d, uint32_t &inB, uint32_t &outB) {
            evalHead(aA, aB, aZ, g, d, inB, outB);
            evalTail(aA, aB, aZ, g, d, inB, outB);
        };
        if (tl0 < fullEnd) {
            uint4 Acur[2][3], Anext[2][3];
            loadA(tl0, Acur);
            issue(Acur[0], 0, 0, eA, eB, eZ, BE);
            for (int tl = tl0; tl < fullEnd; ++tl) {
                const int mt = tl * 32;
                uint32_t inE, outE, inN, outN;
                loadA(tl + 1 < fullEnd ? tl + 1 : tl, Anext);
                // every step: issue the other direction's MFMAs, then turn this direction's results into bits; once they
                // have been read, the operand set their MFMAs used may go (keep)
                issue(Acur[1], 0, 1, nA, nB, nZ, BN);
                evalDir(eA, eB, eZ, 0, 0, inE, outE);
                keep(BE);
                __builtin_amdgcn_sched_barrier(0);
                issue(Acur[0], 1, 0, eA, eB, eZ, BE);
                evalDir(nA, nB, nZ, 0, 1, inN, outN);
                keep(BN);
                decide(inE, outE, inN, outN, 0, mt, eA, eB, eZ, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
                issue(Acur[1], 1, 1, nA, nB, nZ, BN);
                evalDir(eA, eB, eZ, 1, 0, inE, outE);
                keep(BE);
                __builtin_amdgcn_sched_barrier(0);
                // (not after the last tile: MFMAs left in flight would land in registers the code behind the loop -- the
                // out-of-line drain in particular -- has already taken for something else)
                if (tl + 1 < fullEnd) issue(Anext[0], 0, 0, eA, eB, eZ, BE);
                evalDir(nA, nB, nZ, 1, 1, inN, outN);
                keep(BN);
                decide(inE, outE, inN, outN, 1, mt, eA, eB, eZ, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Acur[d][c] = Anext[d][c];
            }
        }
        if (fullEnd < tl1 && fullEnd >= tl0) { // the partial tile, not pipelined
            uint4 At[2][3];
            loadA(fullEnd, At);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint32_t inE, outE, inN, outN;
                issue(At[0], g, 0, eA, eB, eZ, BE);
                issue(At[1], g, 1, nA, nB, nZ, BN);
                evalDir(eA, eB, eZ, g, 0, inE, outE);
                evalDir(nA, nB, nZ, g, 1, inN, outN);
                keep(BE);
                keep(BN);
                __builtin_amdgcn_sched_barrier(0);
                decide(inE, outE, inN, outN, g, fullEnd * 32, eA, eB, eZ, std::true_type{});
            }
        }
        drain();
        // the two match halves of every hypothesis: lanes j and j + 32
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int v = cg[g];
            v += __shfl_xor(v, 32, 64);
            if (kb == 0) atomicAdd(&s_cnt[hw + 32 * g + j], v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt = s_cnt[tid];
        if (dbg != nullptr && lane == 0) {
            atomicAdd(&dbg[0], parked);
            atomicAdd(&dbg[1], (unsigned long long)(tl1 - tl0) * 32ull * 64ull);
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
