// ps_score_fast.h -- kernel 3 for the reprojection metric, decision-exact instead of value-exact.
//
// RANSAC::computeInlierRatioReprojection (reference src/TransformEst/RANSAC.cpp:325-375) decides, for every
// (hypothesis, match), whether two reprojection errors stay below inlierThresholdReprojection.  Only that
// DECISION has to equal the reference's; ps_ransac_score<1> (ps_kernels.h) reproduces every intermediate VALUE
// bit for bit (two rigid transforms without FMA, four IEEE quotients, two cv::norm tests: 61 VALU instructions
// per evaluation).  This kernel takes the decision from a cheap evaluation with a proven error band and hands the
// evaluations that fall inside the band to the value-exact code:
//
//   fast evaluation (21 VALU instructions, 16 of them v_pk_*_f32 over the two directions): both transforms as FMA
//   chains with the camera constants folded into the model rows (X~ = fx (R p + t)_x, ..., Z~ = (R p + t)_z) and the
//   test taken in the MULTIPLIED domain, without any division:
//       reference:  dx = fl(fl(fl(e_x fx) / e_z) + cx) - u_real,   inlier <=> dx^2 + dy^2 < boundR  (both directions)
//       here:       A~ = X~ + (cx - u_real) Z~  ~  dx * e_z,        s~ = A~^2 + B~^2  against  T^2 Z~^2,  T = sqrt(boundR)
//   (offsets c - real come from a per-match record; the reciprocal form of this kernel, 24 instructions with two
//   v_rcp_f32 and a depth floor, is kept in profiles/variants/ps_score_fast_rcp.h.txt: same speed within 1 %, 0.11 % instead
//   of 0.07 % of the evaluations parked, and it needs v_rcp_f32's 1-ulp accuracy as an assumption).
//
//   error band.  u = 2^-24.  For one hypothesis let S >= sum_j |R_ij| |p_j| + |t_i| for every row of the model and of
//   its inverse and every point of the pair (S = 1.001 (rho * cmax + tau), rho = largest row sum of |R|, tau = largest
//   |t|, cmax = the pair's largest coordinate); then |e_z| <= S.  Against the real number A* = fx E_x + (cx - u_real) E_z
//   (E = the real-valued transform):
//     reference:  dx e_z = (1+d4)[(1+d3)(1+d2) c0 + (1+d3) cx e_z - u_real e_z],  c0 = fl(e_x fx), e_x within g4 S of E_x
//                 => |dx e_z - A*| <= u S (8.03 fmaxK + 5.02 Umax + 3.02 cmaxK)      (Umax = the pair's largest |c - real|)
//     here:       A~ = fl(fma(k, Z~, X~)), X~ within 4.01 u fmaxK S of fx E_x, Z~ within 3.01 u S of E_z, k = fl(cx - u_real)
//                 => |A~ - A*|     <= u S (5.02 fmaxK + 5.03 Umax)
//   so each of the four products the comparison is about is within  E = lambda S,  lambda = 1.05 u (14 fmaxK + 11 Umax +
//   4 cmaxK), of the fast value, and |e_z - Z~| <= 8 u S.  With w = |Z~|, G = sqrt2 E + T 8 u S (triangle inequality on
//   the 2-vector (dx e_z, dy e_z) and on the right-hand side T |e_z|):
//       s~ < w (T^2 (1 - 16u) w - 2 T' G)                   =>  the reference's test passes      (certain inlier)
//       s~ > w (T'^2 w + 2 T' G) + G^2 (1 + 1e-4)           =>  it fails                          (certain outlier:
//                                                                had it passed, s~ would be below this limit)
//   with T' = T (1 + 1e-5).  Round 4 evaluates both with per-HYPOTHESIS coefficients: 2 T' G w <= T' G (eps w^2 + 1 / eps) for
//   any eps > 0, so  (T^2 (1 - 20u) - T' G eps) q - T' G / eps - G^2  and  (T'^2 + T' G eps) q + T' G / eps + G^2 (1 + 1e-4),  q = Z~^2,
//   are valid (slightly wider) limits: one packed FMA each on q, constants rounded to the safe side (rebuild()).  (Round 3 had a
//   per-match band 2 T' G |Z~| + G^2: two plain FMAs with the |.| source modifier more per evaluation.)  No depth floor is
//   needed: the errors are absolute, not divided by Z~; for |Z~| -> 0 the lower limit turns negative (never "inlier") and the
//   upper one tends to its constant term, the noise level of s~ itself (then "uncertain", not "outlier").
//   Anything else -- inside the band, NaN -- is "uncertain".
//
//   uncertain evaluations are parked as (match, lane) in a wave-private LDS queue and evaluated later by the
//   value-exact inlier_test<MODE>() (the model is parked in LDS at the start), densely packed: one lane per parked
//   evaluation instead of one whole wavefront per uncertain match.  Counts therefore equal ps_ransac_score<1>'s
//   for every hypothesis (tests: test_hypothesis_counts_bit_exact, test_score_variants_*, the fuzz slice).
//
// A wavefront whose bounds do not hold (non-finite model, S fmaxK > 2^40, threshold outside [1e-10, 1e15]) runs
// the value-exact loop of ps_kernels.h instead.
#pragma once

#include "ps_kernels.h"

namespace psdev {

struct FastConsts {
    float fmaxK;  // max(|fx|, |fy|, 1)
    float cmaxK;  // max(|cx|, |cy|)
    float thrUp;  // sqrt(boundR) (1 + 1e-5), rounded up
    float bIn0;   // boundR (1 - 20u), rounded down
    float cIn;    // 2 sqrt2 sqrt(boundR) (1 + 1e-5), rounded up
    float thr2Up; // thrUp^2, rounded up
    float cHi;    // (2 c + c^2) thrUp with c = sqrt2 (1 + 1e-5), rounded up
    int enabled;  // thresholds and camera constants inside the range the bounds were derived for
};

// Limits of the decision-exact Euclidean test (derivation: ps_score_euclid.h), used by ps_ransac_score_euclid<0/4> and by
// the Euclidean term of ps_ransac_score_fast<2>.
struct EuclidConsts {
    float tbLo;  // errorVersion 0 / 2: sqrt(B) (1 - 6u) rounded down; errorVersion 4: thr (1 - 7u) rounded down
    float tbHi;  // ... (1 + 6u) / (1 + 7u) rounded up
    int enabled; // threshold inside the range the bounds were derived for
};

constexpr int kQueueCap = 256; // parked evaluations per wave (drained by the wave itself when full)
constexpr float kEpsU = 5.9604644775390625e-08f; // 2^-24

typedef float v2f_t __attribute__((ext_vector_type(2)));

// Both folded models of one hypothesis, packed across the two directions for v_pk_fma_f32:
// .x = current point -> previous image (model), .y = previous point -> current image (inverse model).
// (twelve named pairs, not arrays of pairs: the compiler merged array neighbours into 4-vectors and took them apart again
// through scratch memory)
struct FastModel {
    v2f_t r00, r01, r02, t0; // fx * row 0, fx * t_0
    v2f_t r10, r11, r12, t1; // fy * row 1, fy * t_1
    v2f_t r20, r21, r22, t2; // row 2, t_2
};

PS_D void make_fast(const Rigid &m, const Rigid &inv, float fx, float fy, FastModel &f)
{
    f.r00 = v2f_t{fx * m.R[0][0], fx * inv.R[0][0]};
    f.r01 = v2f_t{fx * m.R[0][1], fx * inv.R[0][1]};
    f.r02 = v2f_t{fx * m.R[0][2], fx * inv.R[0][2]};
    f.r10 = v2f_t{fy * m.R[1][0], fy * inv.R[1][0]};
    f.r11 = v2f_t{fy * m.R[1][1], fy * inv.R[1][1]};
    f.r12 = v2f_t{fy * m.R[1][2], fy * inv.R[1][2]};
    f.r20 = v2f_t{m.R[2][0], inv.R[2][0]};
    f.r21 = v2f_t{m.R[2][1], inv.R[2][1]};
    f.r22 = v2f_t{m.R[2][2], inv.R[2][2]};
    f.t0 = v2f_t{fx * m.t[0], fx * inv.t[0]};
    f.t1 = v2f_t{fy * m.t[1], fy * inv.t[1]};
    f.t2 = v2f_t{m.t[2], inv.t[2]};
}

// largest row sum of |R| and largest |t| (NaN anywhere makes the sums NaN: the caller's comparison then fails)
PS_D void model_norms(const Rigid &m, float &rho, float &tau)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float rs = (fabsf(m.R[i][0]) + fabsf(m.R[i][1])) + fabsf(m.R[i][2]);
        rho = rs > rho ? rs : (rs == rs ? rho : rs); // propagate NaN
        const float ta = fabsf(m.t[i]);
        tau = ta > tau ? ta : (ta == ta ? tau : ta);
    }
}

PS_D v2f_t pk_fma(v2f_t a, v2f_t b, v2f_t c) { return __builtin_elementwise_fma(a, b, c); }
// The two limits of the reprojection test from TWO register pairs, K = (kLo, kHi) and G = (G2p, -G2p), with the halves
// broadcast by op_sel:   lower = kLo * q - G2p   (K.x, q, G.y)      upper = kHi * q + G2p   (K.y, q, G.x)
// (inline asm: left to the register allocator the broadcasts became v_mov pairs inside the hot loop in two of the builds)
PS_D v2f_t limit_lower(v2f_t K, v2f_t q, v2f_t G)
{
    v2f_t r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(K), "v"(q), "v"(G));
    return r;
}
PS_D v2f_t limit_upper(v2f_t K, v2f_t q, v2f_t G)
{
    v2f_t r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(K), "v"(q), "v"(G));
    return r;
}

// The two directions are .x: predictedOld - realOld and .y: predictedNew - realNew.  Operand pairs (cur, prev) of the three
// coordinates and the offset pairs (cx - uOld, cx - uNew), (cy - vOld, cy - vNew).
// Division-free form: A~ = X~ + k Z~ ~ (predicted - real) * depth; returns s~ = A~^2 + B~^2 of the two directions and
// Z~ of the two projected depths.
PS_D v2f_t fast_sq2(const FastModel &f, v2f_t px, v2f_t py, v2f_t pz, v2f_t kx, v2f_t ky, v2f_t &Zout)
{
    const v2f_t X = pk_fma(f.r00, px, pk_fma(f.r01, py, pk_fma(f.r02, pz, f.t0)));
    const v2f_t Y = pk_fma(f.r10, px, pk_fma(f.r11, py, pk_fma(f.r12, pz, f.t1)));
    const v2f_t Z = pk_fma(f.r20, px, pk_fma(f.r21, py, pk_fma(f.r22, pz, f.t2)));
    const v2f_t Au = pk_fma(kx, Z, X);
    const v2f_t Bv = pk_fma(ky, Z, Y);
    Zout = Z;
    return pk_fma(Au, Au, Bv * Bv);
}


// ------------------------------------------------------------------------------------------
// Staged scoring: hypotheses that provably cannot matter are abandoned between launches.
//
// The selection of kernel 4 consumes the counts sequentially: hypothesis i matters only if its count is a RECORD,
// count_i > max_{j<i} count_j (strict '>' first-best, RANSAC.cpp:438-455; the arg-max of the fixed schedule takes the
// lowest index among equals, the same rule), and only while i is below the adaptive trip limit, which from record to
// record only shrinks (RANSAC.cpp:450-453, USAC.h:944-971).  So a large batch is scored in stages:
//   stage 0   the first 256 hypotheses of every pair (64 for the adaptive schedules, whose limit after a few dozen
//             hypotheses is usually below that already), all matches (the plain launch);
//   stage 1   every later hypothesis below the trip limit L0 the prefix leaves, matches [0, c1);
//   stage 2   the SURVIVORS of stage 1, matches [c1, c2);        stage 3   the survivors of stage 2, matches [c2, M).
// Every work-group of stages 1-3 first replays the selection over the prefix (wave_replay_prefix: the record walk of
// ps_select_refit) and gets B0 = the best count so far and L0.  After its match range a hypothesis survives if
// count so far + matches left > B0; otherwise it can no longer become a record and keeps its partial count
// (<= B0: never selected, never a record).  Survivors are appended (one atomic per wavefront) to the next stage's
// per-pair list, which the next launch reads back densely: 256 live hypotheses per work-group again, full wavefronts,
// no barrier and no re-packing inside the hot loops -- they are the plain loops.  The models travel through HBM
// (ma.models, 48 B per hypothesis), the partial counts through counts[].
//   c1 = about 1.15 (M - B0) + 32 matches: where a hypothesis without inliers has run out of chances (with 75 % inliers
//   a bad sample is abandoned after a quarter of the matches); c2 halves the rest.  With the reference's own <= 487-iteration
//   schedule L0 is typically 2 ... 30, far below the prefix: stages 1-3 return at once.
// An in-kernel form (checkpoints + re-packing of live lanes through LDS inside one launch) was built first and kept
// the wave-steps it saved (29 %) from showing up as time: barriers, rebuilds and 300 bytes of scratch per lane made the
// loop 20 % slower before anything was abandoned (profiles/r03c/README.md).
// Outputs of kernel 4 are unchanged bit for bit (tests/test_gpu_prune.py: staged vs complete vs oracle); what changes is
// the meaning of counts[] for abandoned hypotheses (a lower bound <= B0 instead of the count), which is why the
// diagnostic ps_debug_ransac_counts and small batches score completely.
// ------------------------------------------------------------------------------------------
constexpr int kPrefixFixed = kBlock; // hypotheses scored completely by stage 0: fixed schedule (arg-max over all H) ...
constexpr int kPrefixAdaptive = 64;  // ... and RANSAC / USAC schedules (the reference consumes 3 of 487 on good data)
constexpr int kStages = 3;      // pruned stages after the prefix
constexpr int kReorderTopMax = 16; // ps_stage_reorder: at most this many of the prefix's best hypotheses vote on the order
constexpr int kReorderMargin = 16; // reordered sweep: stage 1 ends this many matches (rounded up to 64) after the point where
                                   // a hypothesis that rejects all the leading matches is out

struct StageArgs {
    int stage;                 // 0 = the hypotheses [hBase, hBase + hCount) completely (the plain launch: [0, H)); 1 .. kStages
                               // = pruned stages
    int hBase, hCount;         // stage 0 / 1: the hypothesis range of this launch
    int listStride;            // entries per pair of the survivor lists (= hypotheses with a model slot: only those are listed)
    const int32_t *listIn;     // stage >= 2: survivors of the previous stage, [P][listStride] hypothesis indices ...
    const int32_t *countIn;    //             ... and how many per pair
    int32_t *listOut;          // stage < kStages: where this stage's survivors go
    int32_t *countOut;
    const int32_t *perm;       // stage >= 1 after ps_stage_reorder: [P][cap], match of the ORIGINAL record arrays at each position
                               // of the reordered hot record (null: the hot record is in the original order)
    const float2 *frontRec;    // after ps_stage_reorder (reprojection metrics): [P][cap / 2][5] the matches ALL voters reject, two
                               // per record: (cur.x) (cur.y) (cur.z) (cx - uOld) (cy - vOld) of matches 2k | 2k + 1 -- the
                               // operands of the one-direction pre-test of stage 1 (null: no pre-test)
    int margin, c2div;         // reordered sweep: where stages 1 and 2 end (stage_range)
    unsigned long long *validMask; // stage 0 in two launches (models once, then the sweep split over many work-groups): [P][hCount/64]
    int genOnly;               // 1 = this launch only generates the models of [hBase, hBase + hCount), parks them and writes
                               // validMask; 0 with validMask set = this launch reads both back instead of generating
    int single;                // 1 = stage 1 sweeps ALL matches and is the only stage after the prefix (adaptive schedules
                               // without reordering: what their trip limit leaves is little, two more launches cost more)
    int gran;                  // the stage cuts are multiples of this (64: the Euclidean kernel recounts blocks of 64 matches;
                               // 8 for the reprojection kernels, which only need even cuts)
    int loopGroups;            // stage 1 of an adaptive schedule with a long cap (USAC's 850 000: 3320 blocks of 256 hypotheses per
                               // pair, of which the trip limit leaves a handful): this many work-groups per pair take the blocks
                               // bx, bx + loopGroups, ... and stop at the first block beyond the limit (0: one work-group per block)
    const int32_t *prefInfo;   // after ps_stage_reorder: [P][4] = (best count, trip limit) the prefix leaves, how many matches at
                               // the front of the reordered record ALL voters reject, reserved (null: every work-group of
                               // the stages replays the prefix itself)
};

// Replay of the sequential selection over counts[0 .. n) by one wavefront (the rule of ps_select_refit part (1)):
// best = the best count among the consumed ones, limit = the trip limit afterwards (a.H for the fixed schedule).
// bestIdx = the hypothesis that holds `best` (the first of equals), -1 without a record.
PS_D void wave_replay_prefix(const int32_t *__restrict__ cnts, int n, const SelectArgs &a, int M, int &best, int &limit,
                             int &bestIdx)
{
    const int lane = threadIdx.x & 63;
    bestIdx = -1;
    if (a.estimator == PS_EST_FIXED) {
        // (count, lowest index first) as one key: counts are <= PS_MAX_KPTS < 2^15, prefixes <= 2^16 hypotheses
        int b = 0;
        for (int i = lane; i < n; i += 64) b = max(b, (cnts[i] << 16) | (0xFFFF - i));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = max(b, __shfl_xor(b, o, 64));
        best = b >> 16;
        if (best > 0) bestIdx = 0xFFFF - (b & 0xFFFF);
        limit = a.H;
        return;
    }
    int pos = 0;
    best = 0;
    limit = a.iter0;
    for (;;) {
        const int lim = limit < n ? limit : n;
        unsigned found = 0xFFFFFFFFu;
        for (int i = pos + lane; i < lim; i += 64)
            if (cnts[i] > best) {
                found = (unsigned)i;
                break;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned other = (unsigned)__shfl_xor((int)found, o, 64);
            found = other < found ? other : found;
        }
        if (found == 0xFFFFFFFFu) break;
        best = cnts[found];
        bestIdx = (int)found;
        pos = (int)found + 1;
        limit = a.estimator == PS_EST_USAC ? usac_limit(a, (unsigned)best, (unsigned)M)
                                           : ransac_limit(a, (float)best / (float)M); // ratio as RANSAC.cpp:280
    }
}

// Match range [lo, hi) of pruned stage `stage` (1 .. kStages) for a pair with M matches whose prefix reached best0.
// reordered: the stages sweep the record ps_stage_reorder wrote -- the matches the prefix's best hypothesis rejects first.
// Every hypothesis that is not better than that one rejects (almost all of) them too and is out a few matches later.
PS_D void stage_range(const StageArgs &st, int M, int best0, int &lo, int &hi)
{
    const int stage = st.stage;
    const bool reordered = st.perm != nullptr;
    int c1 = M, c2 = M;
    if (best0 > 0 && !st.single) {
        const int miss = M - best0; // a hypothesis is out once it has missed this many matches
        if (reordered) {
            const int g1 = st.gran - 1; // (gran is a power of two)
            c1 = (miss + st.margin + g1) & ~g1;
            if (c1 >= M - M / 8) c1 = M;
            if (c1 < M) {
                // (stage 2 ends where it would with stage 1 cut at a multiple of 64, whatever stage 1's granularity)
                const int c64 = (c1 + 63) & ~63;
                const int step = ((M - c64) / st.c2div + 63) & ~63;
                c2 = c64 + (step > 0 ? step : 64); // (never an empty stage 2: it hands the survivors on)
                if (c2 >= M - M / 16) c2 = M;
            }
        } else {
            c1 = (miss + miss / 7 + 32 + 63) & ~63;
            if (c1 >= M - M / 8) c1 = M; // nothing worth a second launch
            if (c1 < M) {
                c2 = (c1 + (M - c1) / 2 + 63) & ~63;
                if (c2 >= M - M / 16) c2 = M;
            }
        }
    }
    static_assert(kStages == 3, "three pruned stages");
    lo = stage == 1 ? 0 : (stage == 2 ? c1 : c2); // (selects, not an indexed array: that would live in scratch memory)
    hi = stage == 1 ? c1 : (stage == 2 ? c2 : M);
}

// hypothesis of lane 0 of the calling lane's wavefront (h is consecutive over the lanes, or 0x7FFFFFFF for an idle lane
// of a survivor list: the wave's first lane then tells whether the whole wave is idle)
PS_D int hFirstOfWave(int h, int lane) { return __builtin_amdgcn_readfirstlane(h); }

// What every work-group of a pruned stage needs before it starts: the prefix's best count and trip limit (work-group
// uniform, in SGPRs) -- wave 0 replays, the others wait.
PS_D void stage_prefix(const int32_t *__restrict__ cnts, int nPrefix, const SelectArgs &sa, int M, int *s_pref, int &best0,
                       int &hLimit, const int32_t *__restrict__ prefInfo = nullptr)
{
    if (prefInfo != nullptr) { // replayed once per pair by ps_stage_reorder (scalar loads, no barrier)
        best0 = prefInfo[0];
        hLimit = best0 > 0 ? prefInfo[1] : sa.H;
        return;
    }
    if ((threadIdx.x >> 6) == 0) {
        int b, l, bi;
        wave_replay_prefix(cnts, nPrefix, sa, M, b, l, bi);
        if ((threadIdx.x & 63) == 0) {
            s_pref[0] = b;
            s_pref[1] = l;
        }
    }
    __syncthreads();
    // (LDS loads count as divergent for the compiler: readfirstlane keeps the values in SGPRs and the branches scalar)
    best0 = __builtin_amdgcn_readfirstlane(s_pref[0]);
    // the limit only shrinks from record to record, but the FIRST record may raise it above its initial value
    // (RANSAC.cpp:30 starts from computeRANSACIteration(0.20); a first ratio below 0.2 gives more): without a record in
    // the prefix nothing can be cut
    hLimit = best0 > 0 ? __builtin_amdgcn_readfirstlane(s_pref[1]) : sa.H;
}

// Appends the hypotheses of the lanes with `alive` to the next stage's list of pair p (one atomic per wavefront).
PS_D void stage_append(bool alive, int h, int32_t *__restrict__ listOut, int32_t *__restrict__ countOut, int p, int stride)
{
    const unsigned long long am = __builtin_amdgcn_ballot_w64(alive);
    if (am == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __builtin_ctzll(am)) base = atomicAdd(&countOut[p], __popcll(am));
    base = __builtin_amdgcn_readlane(base, __builtin_ctzll(am));
    if (alive) listOut[(size_t)p * stride + base + lanes_below(am)] = h;
}

// The hypothesis of a lane, derived AGAIN at the end of a staged launch from values that cost no vector register in between
// (the scalar work-group offset passes through an empty asm, so that the compiler cannot keep the first derivation -- or
// the 64-bit addresses built on it -- alive across the loops: with the vectorisers on, at seven waves per SIMD, it kept
// them in scratch memory; kept for the registers it still saves).
// base = first hypothesis (or list entry) of the work-group's pass, slot = the lane's place in it.
PS_D int stage_hypothesis_again(bool list, const StageArgs &st, int base, int slot, int p, int H)
{
    asm volatile("" : "+s"(base));
    if (!list) return st.hBase + base + slot;
    const int i = base + slot;
    return i < st.countIn[p] ? st.listIn[(size_t)p * st.listStride + i] : 0x7FFFFFFF;
}

// Stages 2+: hypotheses one pass of a work-group takes from the survivor list.  After the reordered stage 1 the lists are
// short and the ranges left are long: a pair with at most 64 (128) survivors is swept by ONE work-group whose four
// wavefronts each take a quarter (two halves) of the match range for the same hypotheses and add their counts up in LDS --
// a single wavefront alone on its SIMD waits out every record load (stage 3 of the bench step: 21 survivors per pair, 157 us
// that way, profiles/r03k).  Longer lists: 256 hypotheses per pass as before.  The launches carry a few work-groups per
// pair (StageArgs::hCount / 256) that loop over the list: thousands of work-groups that find nothing to do still cost
// their launch (30 ... 50 us per empty stage of 7485 work-groups).
PS_D int list_cover(int n) { return n <= 64 ? 64 : (n <= 128 ? 128 : kBlock); }

// Two record forms.  BIG (launches that fill the chip several times over): the packed 40-byte match record (RecPtrs::F: the
// loop is sensitive to the scalar-cache footprint of the records every wave streams).  Small launches are bound by the
// latency of ONE prologue and of cold record loads: the three 16-byte records, whose loads go out side by side.
// MODE = EUCLIDEAN_AND_REPROJECTION_ERROR (RANSAC.cpp:377-436: both reprojection errors AND the Euclidean residual below their
// thresholds) adds the Euclidean term of ps_score_euclid.h to every evaluation: the unfolded current -> previous transform as
// three FMA chains (plain instructions: only one direction has a Euclidean test), the squared residual against the same
// per-lane limits lo / hi; "inlier" needs both certain, one certain "outlier" suffices, anything else is parked and decided by
// inlier_test<2>().  40 vector instructions per evaluation instead of 61 + the Euclidean part of the value-exact kernel.
// KIND: what the launch scores (StageArgs).  0 = the hypotheses [hBase, hBase + hCount) completely -- the plain launch over
// [0, H) and stage 0 of the staged scoring; 1 = stage 1 (generates its models, first match range); 2 = stages 2+ (models and
// hypothesis list read back, a loop over the list).
// Registers: the library is built with the SLP vectoriser and VectorCombine off (Makefile).  With them the scalar prologue
// (sample -> Umeyama -> SVD -> general inverse) was turned into packed instructions whose aligned register pairs and
// shuffles cost 20 registers: 72 VGPRs + 12 ... 19 values per lane in scratch memory at seven waves per SIMD (0.4 GB of HBM
// traffic per 499 pairs, profiles/r03i) against 56 VGPRs, no scratch memory and eight waves without (profiles/r03l).  The
// epilogues still derive the hypothesis index again instead of keeping it across the loops.
// One pass of a work-group: the hypotheses [hBase + bx * 256, + 256) (kinds 0 / 1) or one pass over the survivor list (kind 2).
template <int MODE, bool BIG, int KIND>
PS_D bool score_fast_pass(const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                          const float4 *__restrict__ recE, const float2 *__restrict__ recF,
                          const float2 *__restrict__ pairBound, const ModelArgs &ma, const ScoreConsts &k,
                          const FastConsts &fc, const EuclidConsts &ec, const SelectArgs &sa, const StageArgs &st, int H, int cap,
                          int msplit, int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg, const unsigned bx,
                          const unsigned by, const int p, const int M)
{
    constexpr bool EUCLID = MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR;
    __shared__ float s_mdl[12][kBlock];
    __shared__ uint32_t s_q[kBlock / 64][kQueueCap];
    __shared__ int s_cnt[kBlock];
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
    // (the match range is split on match PAIRS: the loop takes two matches per trip, only the pair's last range can be odd)
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
    if (pruned) { // (msplit == 1 in these stages)
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
        if (!LIST && st.hBase + (int)bx * kBlock >= hEnd) return true;
    }

    // (a wavefront without a hypothesis of its own has nothing to do -- the 64-hypothesis prefix of the adaptive
    // schedules fills one of the four; a pass with a split match range has no such wavefront, and a barrier to come)
    if (cover == kBlock && hFirstOfWave(h, lane) >= hEnd) return false;

    Rigid mdl;
    set_identity(mdl);
    bool valid = false;
    if (LIST) {
        if (h < hEnd) {
            load_model(ma, (size_t)p * ma.modelH + h, mdl); // parked by stage 1 (only valid samples survive it; h < modelH: a
                                                            // hypothesis beyond is swept completely by stage 1 and never listed)
            valid = true;
        }
    } else if (KIND == 0 && st.validMask != nullptr && !st.genOnly) {
        // stage 0, second launch: the models were generated once by the first (the sweep is split over msplit work-groups,
        // each of which would otherwise repeat the sample -> SVD chain: 63 % of stage 0's instructions with four parts)
        const unsigned long long vm = uniform64(st.validMask[(size_t)p * ((st.hCount + 63) >> 6) + (bx * (kBlock / 64) + wv)]);
        valid = h < hEnd && lane_in(vm);
        if (h < hEnd) load_model(ma, (size_t)p * ma.modelH + h, mdl); // (launches of this form have a slot for every hypothesis)
    } else {
        if (h < hEnd) valid = gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)h, mdl);
        // (stage 1 parks only the models of its survivors, at the end: the abandoned majority is never read again)
        if (ma.models && by == 0 && !pruned) {
            const int hs = stage_hypothesis_again(false, st, (int)bx * kBlock, tid, p, H);
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
    s_cnt[tid] = 0;

    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float4 *__restrict__ pc = recC + rbase;
    const float4 *__restrict__ pe = recE + rbase;
    const float2 *__restrict__ pf = recF + rbase * 5;
    // position of the hot record -> match of the original record arrays (stages after ps_stage_reorder; the cold paths)
    const int32_t *__restrict__ pperm = (pruned && st.perm != nullptr) ? st.perm + rbase : nullptr;
    auto orig = [&](int m) { return pperm != nullptr ? pperm[m] : m; };
    const float2 pbnd = pairBound[p];
    const float cmax = pbnd.x, umax = pbnd.y;

    // The model leaves the registers here: everything below reads it back from LDS.
    // Everything the hot loop keeps in registers per hypothesis -- the folded models and the band coefficients -- is
    // (re)built by rebuild() from the model parked in LDS: once before the loop and again after every in-loop drain, so that
    // none of it has to stay alive across the drain's register-hungry code (kept alive it was spilled to scratch on
    // every launch: 40 dwords per lane, 349 MB of writes per 499 pairs, profiles/r02d).  The size S the bounds hang on
    // comes out of the same pass (one general inverse per hypothesis, not two).
    int cnt = 0;
    {
        FastModel F;
        // per-hypothesis coefficients of the two limits (see rebuild), held as TWO register pairs whose halves the packed FMAs
        // broadcast through op_sel (limit_lower / limit_upper)
        v2f_t KL, GG; // KL = (kLo, kHi), GG = (G2p, -G2p)
        float S;
        float U[3][4];          // EUCLID: the unfolded model (R | t), current point -> previous frame
        float loE = 0.0f, hiE = 0.0f; // EUCLID: per-lane limits of the squared residual (ps_score_euclid.h)
        // uniform part of the band: E = lambda S, |e_z - Z~| <= 8 u S, G = (sqrt2 lambda + 8 u T') S = g S
        const float lam = 1.05f * kEpsU * (14.0f * fc.fmaxK + 11.0f * umax + 4.0f * fc.cmaxK);
        const float g = 1.4143f * lam + 8.0f * kEpsU * fc.thrUp;
        auto rebuild = [&]() {
            Rigid md, iv;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][tid];
                md.t[i] = s_mdl[9 + i][tid];
            }
            // md: current point -> previous image (estimatedOldPosition, RANSAC.cpp:346);
            // iv: previous point -> current image (estimatedNewPosition, RANSAC.cpp:348)
            inverse_rigid_general(md, iv);
            make_fast(md, iv, k.fx, k.fy, F);
            float rho2 = 0.0f, tau2 = 0.0f;
            model_norms(md, rho2, tau2);
            model_norms(iv, rho2, tau2);
            S = (rho2 * cmax + tau2) * 1.001f;
            const float G = S * g;
            // The limits  T^2 (1 - 16u) w^2 - 2 T' G w  and  T'^2 w^2 + 2 T' G w + G^2 (1 + 1e-4)  (w = |Z~|) with the linear term
            // bounded by the quadratic one:  2 T' G w <= T' G (eps w^2 + 1 / eps)  for ANY eps > 0 (AM-GM), so
            //     lower limit >= (T^2 (1 - 16u) - T' G eps) q - T' G / eps - G^2 (1 + 1e-4),   q = Z~^2
            //     upper limit <= (T'^2          + T' G eps) q + T' G / eps + G^2 (1 + 1e-4)
            // are valid limits too -- a little wider than the exact ones except at w = 1 / eps -- and have per-HYPOTHESIS
            // coefficients: one packed FMA per limit on q, no |Z~| and no per-match band (round 3: two plain v_fma_f32 with the
            // |.| modifier per evaluation; 23 -> 21 vector instructions, 19 -> 17 per pre-tested pair of matches).
            // eps = 2 / cmax puts the tangent point in the middle of the pair's depth range; the band is 0.1 % of the
            // evaluations wide either way.  Roundings: 1e-6 / 1e-5 relative slack on every constant covers the few float
            // roundings of the constants themselves (<= 6e-8 each), the rounding of q (u) and of the limit's FMA (u).
            const float eps = 2.0f / cmax;
            const float TG = (fc.thrUp * G) * 1.00001f;
            const float cQ = (TG * eps) * 1.00001f;
            const float G2p = ((TG / eps) * 1.00001f + (G * G) * 1.0001f) * 1.00002f;
            // (kLo is negative for a band wider than the threshold: then never "certainly inside")
            KL = v2f_t{fc.bIn0 * 0.999999f - cQ, fc.thr2Up * 1.000001f + cQ};
            GG = v2f_t{G2p, -G2p};
            if (EUCLID) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) U[i][j] = md.R[i][j];
                    U[i][3] = md.t[i];
                }
                float rhoE = 0.0f, tauE = 0.0f;
                model_norms(md, rhoE, tauE);
                const float aE = ((rhoE * cmax + tauE) * 1.001f) * (12.7f * kEpsU);
                const float xl = ec.tbLo - aE;
                loE = xl > 0.0f ? (xl * xl) * (1.0f - 8.0f * kEpsU) : -1.0f;
                const float yh = ec.tbHi + aE * (1.0f + 4.0f * kEpsU);
                hiE = (yh * yh) * (1.0f + 8.0f * kEpsU);
            }
        };
        rebuild();
        // (comparisons are false for NaN: a non-finite model, cmax or umax sends the wavefront to the value-exact loop)
        const bool boundsOk = fc.enabled != 0 && S * fc.fmaxK <= kDivHi && S >= 1.0e-20f && umax <= 1.0e7f &&
                              (!EUCLID || (ec.enabled != 0 && cmax <= 1.0e15f));
        if (!wave_all(boundsOk)) {
            Rigid md, iv;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][tid];
                md.t[i] = s_mdl[9 + i][tid];
            }
            inverse_rigid_general(md, iv);
            for (int m = m0; m < m1; ++m) {
                const int mo = orig(m);
                const float4 A = pa[mo], B = pb[mo], C = pc[mo];
                score_accumulate<MODE, false>(md, iv, k, A, B, C, cnt);
            }
        } else {
        int qn = 0;                      // parked evaluations of this wave (wave-uniform)
        unsigned long long parked = 0;

        // value-exact evaluation of the parked (match, lane) pairs, one lane each
        auto drain = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int e = lane; e < qn; e += 64) {
                const uint32_t ent = s_q[wv][e];
                const int t = wv * 64 + (int)(ent & 63u), m = orig((int)(ent >> 6));
                Rigid md, iv;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) md.R[i][j] = s_mdl[3 * i + j][t];
                    md.t[i] = s_mdl[9 + i][t];
                }
                inverse_rigid_general(md, iv);
                const float4 A = pa[m], B = pb[m], C = pc[m];
                if (inlier_test<MODE>(md, iv, k, A, B, C)) atomicAdd(&s_cnt[t], 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            parked += (unsigned long long)qn;
            qn = 0;
        };

        const unsigned long long execAll = __builtin_amdgcn_ballot_w64(true);
        // one match: its record as the five operand pairs (cur.x, prev.x) (cur.y, prev.y) (cur.z, prev.z) (cx - uOld, cx - uNew)
        // (cy - vOld, cy - vNew)
        // (held as five 64-bit scalars = the aligned register pairs the packed instructions take)
        struct Rec {
            unsigned long long q[5];
        };
        auto pair64 = [](float lo, float hi) {
            return (unsigned long long)__builtin_bit_cast(uint32_t, lo) | ((unsigned long long)__builtin_bit_cast(uint32_t, hi) << 32);
        };
        // (BIG: rp = the packed record of match m, kept as a running pointer -- recomputed from the index every trip it cost a
        // multiply, a sign extension, a 64-bit shift and a 64-bit add per record on the scalar unit, which the probe of round 4
        // showed to be nearly as loaded as the vector ALU in this loop)
        auto load_rec = [&](int m, const unsigned long long *__restrict__ rp) {
            Rec r;
            if (BIG) {
#pragma unroll
                for (int i = 0; i < 5; ++i) r.q[i] = rp[i];
            } else {
                const float4 A = pa[m], B = pb[m], E = pe[m];
                r.q[0] = pair64(B.x, A.x); r.q[1] = pair64(B.y, A.y); r.q[2] = pair64(B.z, A.z);
                r.q[3] = pair64(E.x, E.y); r.q[4] = pair64(E.z, E.w);
            }
            return r;
        };
        // decides one match for the 64 hypotheses of the wavefront: counts the certain inliers, returns the DECIDED lanes
        // (the undecided ones are worked out only when a trip has any: two scalar instructions less per evaluation)
        auto eval = [&](const Rec &r) -> unsigned long long {
            v2f_t Z;
            const v2f_t e0 = __builtin_bit_cast(v2f_t, r.q[0]), e1 = __builtin_bit_cast(v2f_t, r.q[1]),
                        e2 = __builtin_bit_cast(v2f_t, r.q[2]), e3 = __builtin_bit_cast(v2f_t, r.q[3]),
                        e4 = __builtin_bit_cast(v2f_t, r.q[4]);
            const v2f_t ss = fast_sq2(F, e0, e1, e2, e3, e4, Z);
            // the match's current and previous point (wave-uniform)
            const float cxm = e0.x, cym = e1.x, czm = e2.x, pxm = e0.y, pym = e1.y, pzm = e2.y;
            // limits  kLo Z~^2 - G2p  and  kHi Z~^2 + G2p  with the per-hypothesis coefficients of rebuild()
            const v2f_t q = Z * Z;
            const v2f_t lo = limit_lower(KL, q, GG);
            const v2f_t hi = limit_upper(KL, q, GG);
            // (scalar copies: comparisons on vector-element expressions; any NaN makes all four comparisons false)
            const float se = ss.x, sn = ss.y, loe = lo.x, lon = lo.y, hie = hi.x, hin = hi.y;
            unsigned long long mIn =
                __builtin_amdgcn_ballot_w64(se < loe) & __builtin_amdgcn_ballot_w64(sn < lon);
            unsigned long long mOut =
                __builtin_amdgcn_ballot_w64(se > hie) | __builtin_amdgcn_ballot_w64(sn > hin);
            if (EUCLID) {
                const float d0 = __builtin_fmaf(U[0][0], cxm, __builtin_fmaf(U[0][1], cym, __builtin_fmaf(U[0][2], czm, U[0][3]))) - pxm;
                const float d1 = __builtin_fmaf(U[1][0], cxm, __builtin_fmaf(U[1][1], cym, __builtin_fmaf(U[1][2], czm, U[1][3]))) - pym;
                const float d2 = __builtin_fmaf(U[2][0], cxm, __builtin_fmaf(U[2][1], cym, __builtin_fmaf(U[2][2], czm, U[2][3]))) - pzm;
                const float sE = __builtin_fmaf(d0, d0, __builtin_fmaf(d1, d1, d2 * d2));
                mIn &= __builtin_amdgcn_ballot_w64(sE < loE);
                mOut |= __builtin_amdgcn_ballot_w64(sE > hiE);
            }
            add_mask(cnt, mIn);
            return mIn | mOut;
        };
        // Stage 1 on the reordered record starts with the matches EVERY voter rejected (the wrong correspondences): nearly
        // every hypothesis rejects them too, and one direction of the test is enough to know -- current point -> previous
        // image alone.  ps_stage_reorder lays that direction's operands out two MATCHES per record (StageArgs::frontRec), so
        // one packed chain decides a pair of matches in 19 vector instructions instead of 2 x 23: a lane whose squared error is
        // above the upper limit is a certain outlier -- the values are the .x lanes of eval() bit for bit (same operations, same
        // order; the model's .x halves are broadcast), so eval() would say the same -- and every other lane (under 1 %: mostly
        // the hypotheses that were sampled from that very match) is parked for the value-exact test like an undecided one.
        // Neither the matches' full records nor the other direction are touched.
        auto bx2 = [](float v) { return v2f_t{v, v}; };
        auto not_out2 = [&](const float2 *__restrict__ fr, unsigned long long &ua, unsigned long long &ub) {
            const unsigned long long *__restrict__ e = reinterpret_cast<const unsigned long long *>(fr);
            const unsigned long long q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4];
            const v2f_t cx = __builtin_bit_cast(v2f_t, q0), cy = __builtin_bit_cast(v2f_t, q1), cz = __builtin_bit_cast(v2f_t, q2),
                        kx = __builtin_bit_cast(v2f_t, q3), ky = __builtin_bit_cast(v2f_t, q4);
            const v2f_t X = pk_fma(bx2(F.r00.x), cx, pk_fma(bx2(F.r01.x), cy, pk_fma(bx2(F.r02.x), cz, bx2(F.t0.x))));
            const v2f_t Y = pk_fma(bx2(F.r10.x), cx, pk_fma(bx2(F.r11.x), cy, pk_fma(bx2(F.r12.x), cz, bx2(F.t1.x))));
            const v2f_t Z = pk_fma(bx2(F.r20.x), cx, pk_fma(bx2(F.r21.x), cy, pk_fma(bx2(F.r22.x), cz, bx2(F.t2.x))));
            const v2f_t A = pk_fma(kx, Z, X), B = pk_fma(ky, Z, Y);
            const v2f_t ss = pk_fma(A, A, B * B);
            const v2f_t q = Z * Z;
            const v2f_t hi = limit_upper(KL, q, GG); // (eval()'s limit, bit for bit)
            const float sa = ss.x, sb = ss.y, ha = hi.x, hb2 = hi.y;
            ua = __builtin_amdgcn_ballot_w64(sa > ha); // decided = certainly out (NaN: not "above": parked)
            ub = __builtin_amdgcn_ballot_w64(sb > hb2);
        };
        // (only stage 1 sweeps the front.  profiles/r03n: stage 1 640 -> 563 us with errorVersion 1, -11 % of the scoring step
        // with errorVersion 2; the pre-test leaves 0.9 % of its trips with a lane to park.)
        constexpr bool PRE = KIND == 1;
        int mFront = m0; // matches [m0, mFront): the all-reject front of the reordered record
        // (the launch passes the front record as recE, which the packed-record build does not read otherwise: only loads
        // through a __restrict__ kernel argument become scalar loads -- through the pointer inside StageArgs they were
        // vector loads of one address by 64 lanes)
        const float2 *__restrict__ pfr = reinterpret_cast<const float2 *>(recE) + (size_t)p * ((size_t)((cap + 1) >> 1) * 5);
        if (PRE && st.frontRec != nullptr && st.prefInfo != nullptr) {
            const int f = __builtin_amdgcn_readfirstlane(st.prefInfo[4 * p + 2]); // (uniform: keep the loop's branch scalar)
            mFront = f < m1 ? f : m1;
        }
        // Two matches per trip, both records requested before the first is used: a wavefront that is alone on its SIMD (the
        // stages with few hypotheses, a single pair) otherwise waits out one scalar load per match.  The undecided lanes of
        // both are parked afterwards, in ONE copy of the parking code (it holds the drain).
        const unsigned long long *__restrict__ rp = reinterpret_cast<const unsigned long long *>(pf) + 5 * (size_t)m0;
        const float2 *__restrict__ fr2 = pfr + 5 * (size_t)(m0 >> 1); // (the front record of the pair of matches m, m + 1)
        // (whole pairs only: no "is there a second match" select on the scalar unit in every trip; the odd last match of a
        // pair's last range goes to the value-exact code for every lane, below)
        const int mPairs = m0 + ((m1 - m0) & ~1);
        int frontParkTrips = 0; // trips of the pre-tested front that left a lane to the value-exact test (wave-uniform)
        for (int m = m0; m < mPairs; m += 2, rp += 10, fr2 += (PRE ? 5 : 0)) {
            unsigned long long ua, ub;
            if (PRE && m + 2 <= mFront) { // (m0 = 0 in stage 1: m is even)
                not_out2(fr2, ua, ub);
            } else {
                Rec ra = load_rec(m, rp), rb = load_rec(m + 1, rp + 5);
                // (an empty asm that takes both records: the compiler otherwise sinks the second load below the first evaluation)
                asm volatile("" : "+s"(ra.q[0]), "+s"(ra.q[1]), "+s"(ra.q[2]), "+s"(ra.q[3]), "+s"(ra.q[4]), "+s"(rb.q[0]),
                             "+s"(rb.q[1]), "+s"(rb.q[2]), "+s"(rb.q[3]), "+s"(rb.q[4]));
                ua = eval(ra);
                ub = eval(rb);
            }
            if ((ua & ub) != execAll) { // (ua, ub: the lanes eval() / the pre-test decided)
                if (PRE && m + 2 <= mFront) ++frontParkTrips; // (statistics: counted here, in the rare path, not per trip)
#pragma nounroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned long long mU = execAll & ~(j ? ub : ua);
                    if (mU == 0ull) continue;
                    const int n = __popcll(mU);
                    if (qn + n > kQueueCap) {
                        drain();
                        rebuild();
                    }
                    if (lane_in(mU)) s_q[wv][qn + lanes_below(mU)] = ((uint32_t)(m + j) << 6) | (uint32_t)lane;
                    qn += n;
                }
            }
        }
        if (mPairs < m1) { // the odd last match: every lane's evaluation is handed to the value-exact code
            const int n = __popcll(execAll);
            if (qn + n > kQueueCap) drain();
            s_q[wv][qn + lanes_below(execAll)] = ((uint32_t)mPairs << 6) | (uint32_t)lane;
            qn += n;
        }
        drain();
        cnt += s_cnt[tid];
        if (dbg != nullptr && lane == 0) {
            atomicAdd(&dbg[0], parked);
            atomicAdd(&dbg[1], (unsigned long long)(m1 - m0) * 64ull);
            if (PRE) { // (ps_debug_score_stats_ex: [2] / [3] = pre-test trips without / with a lane left to the value-exact test)
                const int frontTrips = mFront > m0 ? ((mFront < mPairs ? mFront : mPairs) - m0) >> 1 : 0;
                atomicAdd(&dbg[3], (unsigned long long)frontParkTrips);
                atomicAdd(&dbg[2], (unsigned long long)(frontTrips - frontParkTrips));
            }
        }
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

// (kind 2 sweeps short lists with few wavefronts: five per SIMD are plenty, and its loop over the list keeps the kernel
// arguments alive across the passes -- at seven or eight they went to scratch memory and the inlier counter with them)
// LOOP (kind 1 only): StageArgs::loopGroups work-groups per pair walk the blocks of 256 hypotheses and stop at the first one beyond
// the trip limit -- a pass returns true when its block, and with it every later one, has nothing to do.
template <int MODE, bool BIG, int KIND = 0, bool LOOP = false>
__global__ __launch_bounds__(kBlock, (KIND == 2 || LOOP) ? 5 : 8) void ps_ransac_score_fast(
    const float4 *__restrict__ recA, const float4 *__restrict__ recB, const float4 *__restrict__ recC,
    const float4 *__restrict__ recE, const float2 *__restrict__ recF, const int32_t *__restrict__ mvalid,
    const float2 *__restrict__ pairBound,
    ModelArgs ma, ScoreConsts k, FastConsts fc, EuclidConsts ec, SelectArgs sa, StageArgs st, int H, int cap, int minRun,
    int msplit, int32_t *__restrict__ counts, unsigned long long *__restrict__ dbg)
{
    static_assert(MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR,
                  "the metrics with a reprojection test");
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
            if (score_fast_pass<MODE, BIG, KIND>(recA, recB, recC, recE, recF, pairBound, ma, k, fc, ec, sa, st, H, cap, msplit,
                                                 counts, dbg, b, by, p, M))
                break; // (uniform over the work-group: the exits that return true are taken by all of its waves)
        }
    } else if (KIND == 2) {
        const int n = st.countIn[p];
        const int cover = list_cover(n);
        for (unsigned b = bx; (int)b * cover < n; b += hb) {
            if (b != bx) __syncthreads(); // (the pass before is done with the work-group's LDS)
            score_fast_pass<MODE, BIG, KIND>(recA, recB, recC, recE, recF, pairBound, ma, k, fc, ec, sa, st, H, cap, msplit, counts,
                                             dbg, b, by, p, M);
        }
    } else
        score_fast_pass<MODE, BIG, KIND>(recA, recB, recC, recE, recF, pairBound, ma, k, fc, ec, sa, st, H, cap, msplit, counts, dbg,
                                         bx, by, p, M);
}

// ------------------------------------------------------------------------------------------
// Between stage 0 and stage 1: the hot record of every pair in the order that ends hypotheses soonest.
//
// A hypothesis is out once it has missed M - B0 matches.  The prefix's best hypothesis misses exactly that many; nearly
// every other hypothesis misses THOSE matches too (they are the wrong correspondences) and then a few more.  In the original
// order the wrong correspondences are spread over the whole list, and a hypothesis that ends up five inliers short of B0 is
// only out at the very end; with the matches the prefix's best hypothesis rejects moved to the front, it is out at its first
// miss among the remaining ones.  Any order is valid -- a count is a sum over all matches and the bound "count so far +
// matches left <= B0" holds for every subset -- so this changes no output, only where hypotheses stop.
// One hypothesis's verdict is a noisy guide (it rejects marginal matches others accept, which leaves those others slack):
// the nTopMax best hypotheses of the prefix VOTE, and the matches are ordered by how many of them reject each (a stable
// counting sort: all-reject first -- the wrong correspondences --, then the marginal ones, the solid ones last).
// One work-group per pair: replay the prefix, pick the voters, read their models (parked by stage 0), test every match
// value-exactly against each, sort, and write
//   F2 / G2  the hot record of the stage kernels (RecPtrs::F or ::G layout),   perm  position -> original match
// (the cold paths -- value-exact recounts, the sampler of gen_model -- keep using the original arrays through perm).
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void ps_stage_reorder(const float4 *__restrict__ recA, const float4 *__restrict__ recB,
                                                           const float4 *__restrict__ recC, const float2 *__restrict__ recF,
                                                           const int32_t *__restrict__ mvalid, ModelArgs ma, ScoreConsts k,
                                                           SelectArgs sa, int prefix, int nTopMax, int bailGran, int bailMargin,
                                                           int H, int cap, int minRun,
                                                           const int32_t *__restrict__ counts, float2 *__restrict__ recF2,
                                                           int32_t *__restrict__ perm, int32_t *__restrict__ prefInfo,
                                                           float2 *__restrict__ frontRec, unsigned *__restrict__ bailCount)
{
    constexpr bool EUCLID_REC = MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR; // hot record = RecPtrs::G
    constexpr int RF = MODE == PS_ADAPTIVE_ERROR ? 16 : 12;
    constexpr int kMaxChunks = PS_MAX_KPTS / kBlock;
    __shared__ int s_top[kReorderTopMax + 1];                              // [0] = how many, then the hypotheses
    __shared__ uint8_t s_rej[PS_MAX_KPTS];                                 // per match: how many of them reject it
    __shared__ float s_vote[kReorderTopMax][24];                           // the voters' models and their inverses
    __shared__ int s_tab[(kReorderTopMax + 2) * kMaxChunks * (kBlock / 64)]; // bucket-major counts -> start positions
    const int p = blockIdx.x;
    const int M = mvalid[p];
    if (M < minRun) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t rbase = (size_t)p * cap;
    const int32_t *__restrict__ cnts = counts + (size_t)p * H;
    if (wv == 0) {
        // the limit the prefix leaves: with every later hypothesis beyond it the stages return at once and read nothing
        int b, l, bi;
        wave_replay_prefix(cnts, prefix, sa, M, b, l, bi);
        if (lane == 0) { // what every work-group of the stages needs (stage_prefix)
            prefInfo[4 * p] = b;
            prefInfo[4 * p + 1] = l;
            prefInfo[4 * p + 2] = 0; // (the all-reject front: known after the sort)
            prefInfo[4 * p + 3] = 0; // (1: the vote was skipped, see below)
        }
        int n = 0;
        const bool idle = b > 0 && l <= prefix; // (without a record nothing can be cut: the stages sweep everything)
        // Nothing to gain from an order (option "bail", Euclidean metrics): the prefix's best hypothesis misses so many
        // matches that stage 1 sweeps the whole list whatever the order (stage_range puts its cut at M) -- hopeless data, no
        // pair accepted.  The vote (nTop value-exact tests per match) is skipped and the record is handed on in its
        // original order.  The reprojection metrics always vote: their pre-test front pays even when nothing can be abandoned.
        bool bail = false;
        if (bailGran > 0 && b > 0) {
            const int c1 = (M - b + bailMargin + bailGran - 1) & ~(bailGran - 1);
            bail = c1 >= M - M / 8; // (the rule of stage_range for the reordered sweep)
        }
        if (b > 0 && !idle && !bail) {
            // the nTopMax best counts of the prefix (first of equals first): (count, 0xFFFF - index) keys, lane i holds
            // hypotheses i, i + 64, ...
            int key[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j;
                key[j] = (i < prefix && cnts[i] > 0) ? ((cnts[i] << 16) | (0xFFFF - i)) : 0;
            }
            for (; n < nTopMax; ++n) {
                int m = max(max(key[0], key[1]), max(key[2], key[3]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
                if (m == 0) break;
                if (lane == 0) s_top[1 + n] = 0xFFFF - (m & 0xFFFF);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (key[j] == m) key[j] = 0;
            }
        }
        if (lane == 0) {
            s_top[0] = idle ? -1 : n; // (no record: no votes, one bucket, the original order)
            if (bail) prefInfo[4 * p + 3] = 1;
            if (bailCount != nullptr) { // what the host's "nothing to gain" policy looks at (ps_capi.hip, prepare_score)
                atomicAdd(&bailCount[0], 1u);
                if (bail || b <= 0) atomicAdd(&bailCount[1], 1u);
            }
        }
    }
    __syncthreads();
    const int nTop = s_top[0];
    if (nTop < 0) return; // nothing beyond the prefix will be scored
    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float4 *__restrict__ pc = recC + rbase;
    // the voters' models and inverses, one thread each, side by side (one global round trip instead of one per voter)
    if (tid < nTop) {
        Rigid mdl, inv;
        load_model(ma, (size_t)p * ma.modelH + s_top[1 + tid], mdl); // parked by stage 0
        inverse_rigid_general(mdl, inv);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                s_vote[tid][3 * i + j] = mdl.R[i][j];
                s_vote[tid][12 + 3 * i + j] = inv.R[i][j];
            }
            s_vote[tid][9 + i] = mdl.t[i];
            s_vote[tid][21 + i] = inv.t[i];
        }
    }
    __syncthreads();
    for (int m = tid; m < M; m += kBlock) { // (the thread's own matches)
        const float4 A = pa[m], B = pb[m], C = pc[m];
        int rej = 0;
        bool far = (MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) && nTop >= 1;
        for (int t = 0; t < nTop; ++t) {
            Rigid mdl, inv;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    mdl.R[i][j] = s_vote[t][3 * i + j];
                    inv.R[i][j] = s_vote[t][12 + 3 * i + j];
                }
                mdl.t[i] = s_vote[t][9 + i];
                inv.t[i] = s_vote[t][21 + i];
            }
            rej += inlier_test<MODE>(mdl, inv, k, A, B, C) ? 0 : 1;
            if (MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
                // FAR off in the direction stage 1's pre-test looks at (current point -> previous image): more than four
                // thresholds (a heuristic for the order only; plain float arithmetic)
                float ex, ey, ez, pu, pv;
                xform(mdl, B.x, B.y, B.z, ex, ey, ez);
                project(ex, ey, ez, k.fx, k.fy, k.cx, k.cy, pu, pv);
                const float du = pu - C.x, dv = pv - C.y;
                far = far && (du * du + dv * dv > 16.0f * (float)k.boundR);
            }
        }
        // bucket of the counting sort: 0 = every voter rejects it AND it is far off for each (the wrong correspondences
        // proper), then 1 + (nTop - rejections): rejected by all but close for some voter (marginal true correspondences:
        // in doubt for every good hypothesis), ..., accepted by all
        s_rej[m] = (uint8_t)(far && rej == nTop ? 0 : 1 + nTop - rej);
    }
    // Stable counting sort by bucket.  Counts per (bucket, chunk of 256 matches, wave) -> exclusive prefix in that order
    // -> position = start + lanes below in the same bucket.
    const int nB = nTop + 2; // buckets
    const int nChunks = (M + kBlock - 1) / kBlock;
    for (int c = 0; c < nChunks; ++c) {
        const int m = c * kBlock + tid;
        const int d = m < M ? (int)s_rej[m] : -1;
        for (int b = 0; b < nB; ++b) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(d == b);
            if (lane == 0) s_tab[(b * nChunks + c) * (kBlock / 64) + wv] = __popcll(bal);
        }
    }
    __syncthreads();
    if (wv == 0) {
        const int n = nB * nChunks * (kBlock / 64);
        int carry = 0;
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            const int v = e < n ? s_tab[e] : 0;
            int inc = v; // inclusive scan over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o, 64);
                if (lane >= o) inc += u;
            }
            if (e < n) s_tab[e] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
    }
    __syncthreads();
    // matches every voter rejects and finds far off (bucket 0; empty without a voter and for the Euclidean metrics)
    const int front = s_tab[(1 * nChunks) * (kBlock / 64)];
    if (tid == 0) prefInfo[4 * p + 2] = front;
    float *__restrict__ frOut =
        (EUCLID_REC || frontRec == nullptr) ? nullptr : reinterpret_cast<float *>(frontRec) + (size_t)p * ((size_t)((cap + 1) >> 1) * 10);
    const float *__restrict__ gIn = reinterpret_cast<const float *>(recF) + (size_t)p * ((size_t)((cap + 1) >> 1) * RF);
    float *__restrict__ gOut = reinterpret_cast<float *>(recF2) + (size_t)p * ((size_t)((cap + 1) >> 1) * RF);
    for (int c = 0; c < nChunks; ++c) {
        const int m = c * kBlock + tid;
        const int d = m < M ? (int)s_rej[m] : -1;
        int pos = 0;
        for (int b = 0; b < nB; ++b) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(d == b);
            if (d == b) pos = s_tab[(b * nChunks + c) * (kBlock / 64) + wv] + lanes_below(bal);
        }
        if (m < M) {
            perm[rbase + pos] = m;
            if (EUCLID_REC) {
                const float *src = gIn + (size_t)(m >> 1) * RF + (m & 1);
                float *dst = gOut + (size_t)(pos >> 1) * RF + (pos & 1);
#pragma unroll
                for (int i = 0; i < RF; i += 2) dst[i] = src[i];
            } else {
                const float2 *src = recF + 5 * (rbase + m);
                float2 *dst = recF2 + 5 * (rbase + pos);
#pragma unroll
                for (int i = 0; i < 5; ++i) dst[i] = src[i];
                if (frOut != nullptr && pos < front) { // one direction's operands, two matches per record (stage 1's pre-test)
                    float *fd = frOut + (size_t)(pos >> 1) * 10 + (pos & 1);
#pragma unroll
                    for (int i = 0; i < 5; ++i) fd[2 * i] = src[i].x;
                }
            }
        }
    }
    if (EUCLID_REC && (M & 1)) { // the unwritten second half of the last pair record repeats the first (finish_pair_records)
        __syncthreads();
        if (tid == 0) {
            float *g = gOut + (size_t)(M >> 1) * RF;
            for (int i = 0; i < RF; i += 2) g[i + 1] = g[i];
        }
    }
}

} // namespace psdev
