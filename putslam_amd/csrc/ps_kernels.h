// ps_kernels.h -- HIP kernels of the PUTSLAM front-end hot path for gfx950 (wave64, 256 CUs).
//
// Data layout in HBM (one "frame set" + one "pair batch", see include/putslam_hip.h):
//   desc   [F][cap][32 B]      binary descriptors (ORB / LDB, 256 bit)
//   pts    [F][cap][3] f32     back-projected 3-D points (Eigen::Vector3f storage)
//   keys   [P][cap]    u32     per train row: (hamming << 16) | nearest query   (scratch; all-ones again after kernel 2)
//   recA/B/C/D [P][cap] 16 B   per depth-valid match: prev xyz + squared Euclid bound |
//                              cur xyz | projections of prev and cur | indices    (scratch)
//   counts [P][H]      i32     inlier count of every hypothesis                   (scratch)
// Nothing of size N*N or H*M ever touches HBM: distances live in VGPRs, models in VGPRs.
//
// Kernel 1  ps_hamming_nn      train -> nearest query (the N x N popcount sweep)
// Kernel 2  ps_crosscheck_prep OpenCV cross-check, ordered compaction, depth filter, records
// Kernel 3  ps_ransac_score    one lane = one 3-point hypothesis: Umeyama fit + score all M matches
// Kernel 4  ps_select_refit    replay of the sequential best-model rule, refit, final inliers
#pragma once

#include "ps_device_math.h"

#include "../../include/putslam_hip.h"

namespace psdev {

constexpr int kBlock = 256;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

// XCD-aware work-group order.  Work-groups are dealt round-robin over the 8 XCDs (each with its own
// 4 MiB L2), so linear ids L and L+8 share an L2.  This bijection gives every XCD a CONTIGUOUS range of
// logical ids, so the work-groups that sweep the same frame pair (same query rows / same match records)
// run on one XCD and re-read them from its L2 instead of each XCD fetching its own copy.  Speed only:
// nothing depends on where a work-group actually lands.
PS_D unsigned xcd_remap(unsigned L, unsigned total)
{
    const unsigned xcd = L & 7u, idx = L >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
}

// Phase stamps of the latency study (option "stamps", ps_debug_stamps): thread 0 of work-group 0 writes the shader clock
// (s_memtime) at the phase boundaries of kernels 2 and 4 into a private buffer -- never into outputs.  stamps == nullptr in
// every ordinary launch (one scalar compare per boundary).
PS_D void phase_stamp(unsigned long long *stamps, int i)
{
    if (stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0) stamps[i] = __builtin_readcyclecounter();
}

// v_bcnt_u32_b32: popcount(x) + acc in one VALU op (the compiler emits bcnt + add otherwise).
PS_D uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

PS_D uint32_t ham_key(const uint4 &a, const uint4 &b, const uint4 &x, const uint4 &y, uint32_t q)
{
    uint32_t d = bcnt_acc(a.x ^ x.x, 0u);
    d = bcnt_acc(a.y ^ x.y, d);
    d = bcnt_acc(a.z ^ x.z, d);
    d = bcnt_acc(a.w ^ x.w, d);
    d = bcnt_acc(b.x ^ y.x, d);
    d = bcnt_acc(b.y ^ y.y, d);
    d = bcnt_acc(b.z ^ y.z, d);
    d = bcnt_acc(b.w ^ y.w, d);
    return (d << 16) | q; // lexicographic (distance, query): min == strict '<' scan in ascending q
}

// ------------------------------------------------------------------------------------------
// Kernel 1.  cv::batchDistance(train, query, K=1) inside BFMatcher's cross-check
// (reference call site src/Matcher/matcherOpenCV.cpp:203): for each train row the nearest
// query row, ties to the lowest query index.
// One lane owns TPL train descriptors in VGPRs (8 dwords each).  The query rows are
// wave-uniform, so they are fetched with scalar loads (s_load_dwordx8) into SGPRs and feed
// v_xor/v_bcnt directly: no LDS traffic, no bank conflicts, 18 VALU ops per pair.
// grid = tiles * qsplit * P work-groups in XCD-aware order.  qsplit > 1 splits the query range across workgroups (few pairs,
// many CUs) and merges with atomicMin on the packed key.
// ------------------------------------------------------------------------------------------
template <int TPL>
__global__ __launch_bounds__(kBlock) void ps_hamming_nn(const uint4 *__restrict__ desc,
                                                        const int32_t *__restrict__ nkpts,
                                                        const int32_t *__restrict__ pairs, int cap, int fstride, int tiles,
                                                        int qsplit, uint32_t *__restrict__ keys)
{
    const unsigned perPair = (unsigned)(tiles * qsplit);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int p = (int)(L / perPair);
    const int inner = (int)(L - (unsigned)p * perPair);
    const int fq = pairs[2 * p], ft = pairs[2 * p + 1]; // query = previous frame, train = current
    const int nq = nkpts[fq], nt = nkpts[ft];
    const int tile = inner / qsplit, qs = inner - tile * qsplit;
    const int t0 = tile * (kBlock * TPL);
    if (t0 >= nt) return;
    const int q0 = (int)(((long long)nq * qs) / qsplit), q1 = (int)(((long long)nq * (qs + 1)) / qsplit);

    // (fstride: uint4 units between consecutive frames' descriptor blocks -- cap * 2 for the dense frame set, the packed
    // frame's stride when descriptors and points of a frame lie together, PsFrameSet::descFrameStride)
    const uint4 *__restrict__ tdesc = desc + (size_t)ft * fstride;
    const uint4 *__restrict__ qdesc = desc + (size_t)fq * fstride;

    uint4 a[TPL], b[TPL];
    uint32_t best[TPL];
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
        int t = t0 + i * kBlock + threadIdx.x;
        int tc = t < nt ? t : nt - 1;
        a[i] = tdesc[2 * tc];
        b[i] = tdesc[2 * tc + 1];
        best[i] = kNoKey;
    }
    int q = q0;
    for (; q + 4 <= q1; q += 4) {
        uint4 x0 = qdesc[2 * q], y0 = qdesc[2 * q + 1];
        uint4 x1 = qdesc[2 * q + 2], y1 = qdesc[2 * q + 3];
        uint4 x2 = qdesc[2 * q + 4], y2 = qdesc[2 * q + 5];
        uint4 x3 = qdesc[2 * q + 6], y3 = qdesc[2 * q + 7];
#pragma unroll
        for (int i = 0; i < TPL; ++i) {
            best[i] = min(best[i], ham_key(a[i], b[i], x0, y0, (uint32_t)q));
            best[i] = min(best[i], ham_key(a[i], b[i], x1, y1, (uint32_t)q + 1));
            best[i] = min(best[i], ham_key(a[i], b[i], x2, y2, (uint32_t)q + 2));
            best[i] = min(best[i], ham_key(a[i], b[i], x3, y3, (uint32_t)q + 3));
        }
    }
    for (; q < q1; ++q) {
        uint4 x0 = qdesc[2 * q], y0 = qdesc[2 * q + 1];
#pragma unroll
        for (int i = 0; i < TPL; ++i) best[i] = min(best[i], ham_key(a[i], b[i], x0, y0, (uint32_t)q));
    }
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
        int t = t0 + i * kBlock + threadIdx.x;
        if (t < nt) {
            if (qsplit == 1)
                keys[(size_t)p * cap + t] = best[i];
            else
                atomicMin(&keys[(size_t)p * cap + t], best[i]);
        }
    }
}

// ---- workgroup-wide ordered compaction helper (4 waves) ----
// Returns the exclusive prefix of `flag` over the 256 threads; `total` = number of set flags.
template <int BLOCK = kBlock> PS_D int block_scan_flag(bool flag, int &total, int *wsum)
{
    unsigned long long bal = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        int s = wsum[i];
        if (i < w) off += s;
        total += s;
    }
    __syncthreads();
    return off + pre;
}

// Two ordered compactions with one pair of barriers: exclusive prefixes of flagA and flagB over the work-group
// (flagB implies nothing about flagA); the wave sums travel packed (A in the low 16 bits, B in the high ones).
template <int BLOCK = kBlock>
PS_D void block_scan_flags2(bool flagA, bool flagB, int &posA, int &totalA, int &posB, int &totalB, int *wsum)
{
    const unsigned long long ba = __ballot(flagA), bb = __ballot(flagB);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) wsum[w] = __popcll(ba) | (__popcll(bb) << 16);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        const int sv = wsum[i];
        if (i < w) off += sv;
        tot += sv;
    }
    __syncthreads();
    posA = (off & 0xFFFF) + __popcll(ba & below);
    posB = (off >> 16) + __popcll(bb & below);
    totalA = tot & 0xFFFF;
    totalB = tot >> 16;
}

// The same scans with ONE barrier per call: the wave sums alternate between two slots (`parity` = the call's number, uniform
// over the work-group).  A slot is written again two calls later, behind the barrier of the call in between, which every
// thread reaches only after it has read the slot.
template <int BLOCK = kBlock> PS_D int block_scan_flag_alt(bool flag, int &total, int *wsum2, int parity)
{
    int *wsum = wsum2 + (parity & 1) * (BLOCK / 64);
    unsigned long long bal = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        int s = wsum[i];
        if (i < w) off += s;
        total += s;
    }
    return off + pre;
}

template <int BLOCK = kBlock>
PS_D void block_scan_flags2_alt(bool flagA, bool flagB, int &posA, int &totalA, int &posB, int &totalB, int *wsum2, int parity)
{
    int *wsum = wsum2 + (parity & 1) * (BLOCK / 64);
    const unsigned long long ba = __ballot(flagA), bb = __ballot(flagB);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) wsum[w] = __popcll(ba) | (__popcll(bb) << 16);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        const int sv = wsum[i];
        if (i < w) off += sv;
        tot += sv;
    }
    posA = (off & 0xFFFF) + __popcll(ba & below);
    posB = (off >> 16) + __popcll(bb & below);
    totalA = tot & 0xFFFF;
    totalB = tot >> 16;
}

// workgroup-wide maxima of two non-negative floats with one pair of barriers
template <int BLOCK = kBlock> PS_D void block_max2(float &a, float &b, float2 *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = fmaxf(a, __shfl_down(a, o, 64));
        b = fmaxf(b, __shfl_down(b, o, 64));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(a, b);
    __syncthreads();
    float2 r = red[0];
#pragma unroll
    for (int i = 1; i < BLOCK / 64; ++i) {
        r.x = fmaxf(r.x, red[i].x);
        r.y = fmaxf(r.y, red[i].y);
    }
    __syncthreads();
    a = r.x;
    b = r.y;
}

// workgroup-wide maximum of a non-negative float (4 waves)
template <int BLOCK = kBlock> PS_D float block_max(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < BLOCK / 64; ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}

struct PrepArgs {
    float fx, fy, cx, cy;   // camera matrix entries (cv::Mat CV_32FC1, RGBD.cpp:93-96)
    double thrE;            // inlierThresholdEuclidean
    int mode;               // RANSAC::ERROR_VERSION
    int cap;
    int ptsStride;          // floats between consecutive frames' point blocks (PsFrameSet::ptsFrameStride / 4; cap * 3 when dense)
    int32_t *zeroCounts;    // optional: the first zeroH counts of every pair (row stride zeroStride) to clear for kernel 3's
    int zeroH;              // split-range atomics (saves a memset launch)
    int zeroStride;
    int32_t *zeroSurvA, *zeroSurvB; // optional: per-pair survivor counters of the staged scoring to clear (ps_score_fast.h)
    int skipE, skipF;       // the reprojection kernels' record forms no launch of this call will read (RecPtrs::E / ::F)
};

// Where kernel 2 puts the records of the depth-valid matches (scratch arena, [P][cap] each unless noted).
struct RecPtrs {
    float4 *A;  // prev xyz + squared Euclid bound
    float4 *B;  // cur xyz, w = 1
    float4 *C;  // projections of prev and cur (realOld u v, realNew u v)
    int4 *D;    // (index in the match list, queryIdx, trainIdx, 0)
    float4 *E;  // offsets c - real of the decision-exact scoring paths: (cx - uOld, cx - uNew, cy - vOld, cy - vNew)
#ifdef PS_STREAM_DIAG
    float4 *S;  // -DPS_STREAM_DIAG builds only: a shadow block kernel 2 writes every record a second time into (7 x 16 B per match;
                // nothing reads it): doubles kernel 2's write traffic for the A/B of profiles/r06*/records_traffic_ab.txt
#endif
    float2 *F;  // the fast scoring kernel's packed match record for large launches, 40 B = 5 float2 per match, laid out as
                // the SGPR pairs its packed instructions take (one s_load_dwordx8 + one dwordx2, no repacking; 8 B less
                // scalar-cache footprint per match than prev + cur + offsets):
                // (cur.x, prev.x) (cur.y, prev.y) (cur.z, prev.z) (cx - uOld, cx - uNew) (cy - vOld, cy - vNew)
    float *G;   // the Euclidean fast scoring kernel's pair-interleaved record (ps_score_euclid.h), [P][ceil(cap/2)][12 or 16]:
                // match 2k in the even floats, match 2k+1 in the odd ones; null unless that kernel will run (it then
                // replaces E and F, which only the reprojection kernels read)
};

// Builds the records of depth-valid match number v of pair p; returns the largest |offset| of E.
// `fixedBound` = sq_bound_f32(inlierThresholdEuclidean), the same for every match unless the mode is ADAPTIVE (the caller
// computes it once: the exact squared-domain bound costs two or three square roots).
PS_D float write_records(const PrepArgs &a, const RecPtrs &r, int p, int v, int srcIdx, int q, int t, float px, float py,
                         float pz, float cx_, float cy_, float cz_, float fixedBound)
{
    // Euclid rule of RANSAC.cpp:268-272: thr = inlierThresholdEuclidean (* prev.z in ADAPTIVE mode),
    // turned into its exact squared-domain bound.
    const float bound = a.mode == PS_ADAPTIVE_ERROR ? sq_bound_f32(a.thrE * (double)pz) : fixedBound;
    const size_t slot = (size_t)p * a.cap + (size_t)v;
    r.A[slot] = make_float4(px, py, pz, bound);
    r.B[slot] = make_float4(cx_, cy_, cz_, 1.0f);
    r.D[slot] = make_int4(srcIdx, q, t, 0);
    // the projections (four divisions) are read by the reprojection metrics only
    const bool euclidOnly = a.mode == PS_EUCLIDEAN_ERROR || a.mode == PS_ADAPTIVE_ERROR;
    float ou = 0.0f, ov = 0.0f, nu = 0.0f, nv = 0.0f;
    if (!euclidOnly) {
        project(px, py, pz, a.fx, a.fy, a.cx, a.cy, ou, ov);    // realOld  (RANSAC.cpp:357-358)
        project(cx_, cy_, cz_, a.fx, a.fy, a.cx, a.cy, nu, nv); // realNew  (RANSAC.cpp:352-353)
        r.C[slot] = make_float4(ou, ov, nu, nv);
    }
    if (r.G != nullptr) {
        // Euclidean metrics: the pair-interleaved operands of ps_ransac_score_euclid; errorVersion 4 normalises the match
        // by its own depth (the adaptive threshold thr * prev.z becomes the constant thr, ps_score_euclid.h)
        const bool adaptive = a.mode == PS_ADAPTIVE_ERROR;
        const int rf = adaptive ? 16 : 12;
        float *g = r.G + ((size_t)p * (size_t)((a.cap + 1) >> 1) + (size_t)(v >> 1)) * rf + (v & 1);
        if (adaptive) {
            const float w = 1.0f / pz;
            g[0] = cx_ * w; g[2] = cy_ * w; g[4] = cz_ * w; g[6] = w;
            g[8] = px * w; g[10] = py * w; g[12] = pz * w; g[14] = 0.0f;
        } else {
            g[0] = cx_; g[2] = cy_; g[4] = cz_;
            g[6] = px; g[8] = py; g[10] = pz;
        }
        return 0.0f;
    }
    if (euclidOnly) return 0.0f;
    // offsets of the decision-exact scoring paths (ps_score_fast.h): predicted - real = quotient + (c - real)
    // (laid out as the two v_pk_fma_f32 operand pairs: u offsets of both directions, then v offsets)
    const float4 e = make_float4(a.cx - ou, a.cx - nu, a.cy - ov, a.cy - nv);
    if (!a.skipE) r.E[slot] = e;
    if (!a.skipF) {
        float2 *f = r.F + 5 * slot;
        f[0] = make_float2(cx_, px); f[1] = make_float2(cy_, py); f[2] = make_float2(cz_, pz);
        f[3] = make_float2(e.x, e.y); f[4] = make_float2(e.z, e.w);
    }
#ifdef PS_STREAM_DIAG
    if (r.S != nullptr) { // (diagnostic build: every record once more, into a block nothing reads)
        float4 *sh = r.S + 7 * slot;
        sh[0] = make_float4(px, py, pz, bound); sh[1] = make_float4(cx_, cy_, cz_, 1.0f); sh[2] = make_float4(ou, ov, nu, nv);
        sh[3] = make_float4((float)srcIdx, (float)q, (float)t, 0.0f); sh[4] = make_float4(cx_, px, cy_, py);
        sh[5] = make_float4(cz_, pz, e.x, e.y); sh[6] = make_float4(e.z, e.w, 0.0f, 0.0f);
    }
#endif
    // (a NaN offset reports an infinite bound: the decision-exact kernels then leave the pair to the value-exact code)
    const bool num = e.x == e.x && e.y == e.y && e.z == e.z && e.w == e.w;
    return num ? fmaxf(fmaxf(fabsf(e.x), fabsf(e.y)), fmaxf(fabsf(e.z), fabsf(e.w))) : INFINITY;
}

// An odd number of depth-valid matches leaves the second half of the last pair record unwritten: it repeats the first
// (finite, inside the pair's bounds; the scoring kernel does not count it).  Called by one thread after the records of
// the pair are visible to it.
PS_D void finish_pair_records(const PrepArgs &a, const RecPtrs &r, int p, int M)
{
    if (r.G == nullptr || !(M & 1)) return;
    const int rf = a.mode == PS_ADAPTIVE_ERROR ? 16 : 12;
    float *g = r.G + ((size_t)p * (size_t)((a.cap + 1) >> 1) + (size_t)(M >> 1)) * rf;
    for (int i = 0; i < rf; i += 2) g[i + 1] = g[i];
}

// ------------------------------------------------------------------------------------------
// Kernel 2.  Cross-check + ordered compaction (BFMatcher with crossCheck=true) followed by the
// depth filter of RANSAC::estimateTransformation (RANSAC.cpp:65-74) and record building.
// One workgroup per pair; best[q] lives in LDS (cap x 4 B).
// ------------------------------------------------------------------------------------------
// BLOCK = 1024 is the low-latency form for a handful of pairs (one work-group per pair either way).
template <bool WITH_RECORDS, int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK) void ps_crosscheck_prep(const float *__restrict__ pts,
                                                             const int32_t *__restrict__ nkpts,
                                                             const int32_t *__restrict__ pairs,
                                                             uint32_t *__restrict__ keys, PrepArgs a,
                                                             PsDMatch *__restrict__ matches,
                                                             int32_t *__restrict__ numMatches, RecPtrs rec,
                                                             int32_t *__restrict__ mvalid, float2 *__restrict__ cmaxOut,
                                                             unsigned long long *__restrict__ stamps)
{
    extern __shared__ __align__(16) uint32_t s_best[];
    phase_stamp(stamps, 0);
    __shared__ int s_wsum[2 * (BLOCK / 64)];
    __shared__ float2 s_red[BLOCK / 64];
    float cm = 0.0f; // largest |coordinate| among this pair's depth-valid matches
    float um = 0.0f; // largest |c - projection| among them (NaN offsets are skipped by fmaxf; they only ever score "out")
    const int p = blockIdx.x;
    const int cap = a.cap;
    const int fq = pairs[2 * p], ft = pairs[2 * p + 1];
    const int nq = nkpts[fq], nt = nkpts[ft];
    // (the thread's first two keys are on their way while best[] is initialised: they come from the matcher's launch)
    uint32_t k0 = kNoKey, k1 = kNoKey;
    if ((int)threadIdx.x < nt) k0 = keys[(size_t)p * cap + threadIdx.x];
    if ((int)threadIdx.x + BLOCK < nt) k1 = keys[(size_t)p * cap + threadIdx.x + BLOCK];
    for (int q = threadIdx.x; q < nq; q += BLOCK) s_best[q] = kNoKey;
    __syncthreads();
    // step 2 of the cross-check: query q keeps the closest train row among those that chose it,
    // ties to the lowest train index (strict '<' while scanning t ascending).
    for (int t = threadIdx.x; t < nt; t += BLOCK) {
        uint32_t key = t < BLOCK ? k0 : (t < 2 * BLOCK ? k1 : keys[(size_t)p * cap + t]);
        if (key != kNoKey) {
            uint32_t q = key & 0xFFFFu, d = key >> 16;
            atomicMin(&s_best[q], (d << 16) | (uint32_t)t);
            keys[(size_t)p * cap + t] = kNoKey; // the block is all-ones at rest: the next call's matcher merges with atomicMin
        }
    }
    __syncthreads();
    phase_stamp(stamps, 1); // best[q] built
    const float *pp = pts + (size_t)fq * a.ptsStride; // (floats between consecutive frames' point blocks: cap * 3 when dense)
    const float *cp = pts + (size_t)ft * a.ptsStride;
    int base = 0, vbase = 0;
    // one candidate per thread and trip; the next trip's key and points are fetched before this trip's scan, so that the
    // gather's latency (the points were written by another launch: nothing is cached) runs beside it instead of in front of
    // every trip (a single pair with 1024 threads makes two trips)
    struct Cand {
        uint32_t key;
        float px, py, pz, cx, cy, cz;
    };
    auto fetch = [&](int q, Cand &c) {
        c.key = (q < nq) ? s_best[q] : kNoKey;
        c.px = c.py = c.pz = c.cx = c.cy = c.cz = 0.0f;
        if (WITH_RECORDS && c.key != kNoKey) {
            const int t = (int)(c.key & 0xFFFFu);
            c.px = pp[3 * q]; c.py = pp[3 * q + 1]; c.pz = pp[3 * q + 2];
            c.cx = cp[3 * t]; c.cy = cp[3 * t + 1]; c.cz = cp[3 * t + 2];
        }
    };
    const float fixedBound = (WITH_RECORDS && a.mode != PS_ADAPTIVE_ERROR) ? sq_bound_f32(a.thrE) : 0.0f;
    Cand cur;
    fetch((int)threadIdx.x, cur);
    int trip = 0;
    for (int q0 = 0; q0 < nq; q0 += BLOCK, ++trip) {
        const int q = q0 + threadIdx.x;
        Cand nxt;
        nxt.key = kNoKey;
        nxt.px = nxt.py = nxt.pz = nxt.cx = nxt.cy = nxt.cz = 0.0f;
        if (q0 + BLOCK < nq) fetch(q + BLOCK, nxt);
        const uint32_t key = cur.key;
        const bool has = key != kNoKey;
        const int t = (int)(key & 0xFFFFu);
        const float px = cur.px, py = cur.py, pz = cur.pz, cx_ = cur.cx, cy_ = cur.cy, cz_ = cur.cz;
        const bool ok = WITH_RECORDS && has && depth_ok(px, py, pz) && depth_ok(cx_, cy_, cz_);
        // both ordered compactions (cross-check survivors; depth-valid ones among them) with one barrier
        int pos, total, vpos, vtotal;
        block_scan_flags2_alt<BLOCK>(has, ok, pos, total, vpos, vtotal, s_wsum, trip);
        if (has) {
            PsDMatch m;
            m.queryIdx = q;
            m.trainIdx = t;
            m.imgIdx = 0;
            m.distance = (float)(key >> 16);
            matches[(size_t)p * cap + base + pos] = m;
        }
        if (WITH_RECORDS) {
            if (ok) {
                um = fmaxf(um, write_records(a, rec, p, vbase + vpos, base + pos, q, t, px, py, pz, cx_, cy_, cz_, fixedBound));
                cm = fmaxf(cm, fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), fmaxf(fabsf(cx_), fmaxf(fabsf(cy_), fabsf(cz_))))));
            }
            vbase += vtotal;
        }
        base += total;
        cur = nxt;
    }
    phase_stamp(stamps, 2); // matches compacted, records written
    if (WITH_RECORDS) {
        float c = cm, u = um;
        block_max2<BLOCK>(c, u, s_red);
        if (threadIdx.x == 0) cmaxOut[p] = make_float2(c, u);
        if (threadIdx.x == 0) finish_pair_records(a, rec, p, vbase); // (block_max ends with a barrier: the records are visible)
        if (a.zeroCounts)
            for (int i = threadIdx.x; i < a.zeroH; i += BLOCK) a.zeroCounts[(size_t)p * a.zeroStride + i] = 0;
    }
    if (threadIdx.x == 0) {
        numMatches[p] = base;
        if (WITH_RECORDS) mvalid[p] = vbase;
        if (WITH_RECORDS && a.zeroSurvA) {
            a.zeroSurvA[p] = 0;
            a.zeroSurvB[p] = 0;
        }
    }
    phase_stamp(stamps, 3); // bounds, split operands, counters cleared
}

// Depth filter + records for a caller-supplied match list (stand-alone RANSAC entry point).
__global__ __launch_bounds__(kBlock) void ps_prep_from_matches(const float *__restrict__ prev,
                                                               const float *__restrict__ cur,
                                                               const PsDMatch *__restrict__ matches, int m, PrepArgs a,
                                                               RecPtrs rec, int32_t *__restrict__ mvalid,
                                                               float2 *__restrict__ cmaxOut)
{
    __shared__ int s_wsum[kBlock / 64];
    __shared__ float s_red[kBlock / 64];
    float cm = 0.0f, um = 0.0f;
    int vbase = 0;
    const float fixedBound = a.mode != PS_ADAPTIVE_ERROR ? sq_bound_f32(a.thrE) : 0.0f;
    for (int i0 = 0; i0 < m; i0 += kBlock) {
        const int i = i0 + threadIdx.x;
        float px = 0, py = 0, pz = 0, cx_ = 0, cy_ = 0, cz_ = 0;
        int q = 0, t = 0;
        bool ok = false;
        if (i < m) {
            q = matches[i].queryIdx;
            t = matches[i].trainIdx;
            px = prev[3 * q]; py = prev[3 * q + 1]; pz = prev[3 * q + 2];
            cx_ = cur[3 * t]; cy_ = cur[3 * t + 1]; cz_ = cur[3 * t + 2];
            ok = depth_ok(px, py, pz) && depth_ok(cx_, cy_, cz_);
        }
        int vtotal;
        int vpos = block_scan_flag(ok, vtotal, s_wsum);
        if (ok) {
            um = fmaxf(um, write_records(a, rec, 0, vbase + vpos, i, q, t, px, py, pz, cx_, cy_, cz_, fixedBound));
            cm = fmaxf(cm, fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), fmaxf(fabsf(cx_), fmaxf(fabsf(cy_), fabsf(cz_))))));
        }
        vbase += vtotal;
    }
    float c = block_max(cm, s_red);
    float u = block_max(um, s_red);
    if (threadIdx.x == 0) {
        mvalid[0] = vbase;
        cmaxOut[0] = make_float2(c, u);
        finish_pair_records(a, rec, 0, vbase);
    }
    if (a.zeroCounts)
        for (int i = threadIdx.x; i < a.zeroH; i += kBlock) a.zeroCounts[i] = 0;
    if (threadIdx.x == 0 && a.zeroSurvA) { // the staged scoring's survivor counters (one pair), as ps_crosscheck_prep does
        a.zeroSurvA[0] = 0;
        a.zeroSurvB[0] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Hypothesis model: sample -> gather 3 correspondences -> float Umeyama (+ general inverse when a
// reprojection mode needs it).  Used identically by kernel 3 (all hypotheses) and kernel 4 (the
// selected one), so both see the same bits.
// ------------------------------------------------------------------------------------------
struct ModelArgs {
    uint64_t seed;
    const uint32_t *raw;     // optional explicit draws, H x 3 (device)
    const uint64_t *seedDev; // optional: the seed lives in device memory (captured graphs replay with a new seed)
    float *models;           // optional [P][modelH][12]: kernel 3 parks hypothesis models here and kernel 4 reads the
                             // winner's instead of repeating its sample -> Umeyama -> Jacobi-SVD chain (small batches:
                             // the chain is ~10 us of latency on the critical path of a single frame pair)
    int modelH;              // hypotheses per pair that have a slot: [0, modelH).  Under a long cap (USAC's 850 000) only the
                             // leading ones are parked -- the schedules end long before -- and a hypothesis beyond is swept in
                             // one piece by stage 1 and, should it win, rebuilt by kernel 4 (the rare path)
};

PS_D void store_model(const ModelArgs &ma, size_t slot, const Rigid &m)
{
    float4 *d = reinterpret_cast<float4 *>(ma.models + slot * 12);
    d[0] = make_float4(m.R[0][0], m.R[0][1], m.R[0][2], m.R[1][0]);
    d[1] = make_float4(m.R[1][1], m.R[1][2], m.R[2][0], m.R[2][1]);
    d[2] = make_float4(m.R[2][2], m.t[0], m.t[1], m.t[2]);
}
PS_D void load_model(const ModelArgs &ma, size_t slot, Rigid &m)
{
    const float4 *d = reinterpret_cast<const float4 *>(ma.models + slot * 12);
    const float4 a = d[0], b = d[1], c = d[2];
    m.R[0][0] = a.x; m.R[0][1] = a.y; m.R[0][2] = a.z; m.R[1][0] = a.w;
    m.R[1][1] = b.x; m.R[1][2] = b.y; m.R[2][0] = b.z; m.R[2][1] = b.w;
    m.R[2][2] = c.x; m.t[0] = c.y; m.t[1] = c.z; m.t[2] = c.w;
}
PS_D uint64_t base_seed(const ModelArgs &ma) { return ma.seedDev ? *ma.seedDev : ma.seed; }

PS_D bool gen_model(const float4 *__restrict__ recA, const float4 *__restrict__ recB, size_t rbase, uint32_t M,
                    const ModelArgs &ma, uint64_t pairSeed, uint32_t h, Rigid &mdl)
{
    uint32_t idx[3];
    sample_triplet(pairSeed, ma.raw, h, M, idx);
    float src[3][3], dst[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float4 A = recA[rbase + idx[j]]; // previous frame = dst
        float4 B = recB[rbase + idx[j]]; // current frame  = src   (umeyama(cur, prev), RANSAC.cpp:225-226)
        dst[j][0] = A.x; dst[j][1] = A.y; dst[j][2] = A.z;
        src[j][0] = B.x; src[j][1] = B.y; src[j][2] = B.z;
    }
    return umeyama3(src, dst, mdl);
}

struct ScoreConsts {
    float fx, fy, cx, cy;
    double boundR; // squared-domain bound of inlierThresholdReprojection (double, cv::norm)
    float bLo, bHi; // float pre-filter band around boundR: f < bLo => surely < boundR, f > bHi => surely >= boundR
};

// ---- IEEE division without the range fix-ups, two numerators sharing one reciprocal -------------
// hipcc expands a correctly rounded float a/b into
//   div_scale(b), div_scale(a), rcp, fma, fma, mul, fma, fma, fma, div_fmas, div_fixup        (11 VALU ops)
// The two div_scale ops rescale operands only when an exponent is extreme (|b| or |a| outside roughly
// 2^+-96, a quotient that would be subnormal, a subnormal denominator); div_fmas is a plain fma when no
// scaling happened and div_fixup returns its first operand unless an operand is 0 / inf / NaN or the
// quotient leaves the normal range.  Inside the window checked by div_window_ok() none of that can
// trigger, so the sequence below IS the compiler's sequence with the no-op instructions removed and
// returns the same bits.  x*fx/z and y*fy/z share z, hence one rcp + refinement for both quotients:
// 13 VALU ops instead of 22.  Outside the window the caller uses the ordinary '/' operator.
constexpr float kDivLo = 9.094947017729282e-13f; // 2^-40
constexpr float kDivHi = 1.099511627776e12f;     // 2^+40

PS_D void div2_shared(float a0, float a1, float b, float &q0, float &q1)
{
    float r0 = __builtin_amdgcn_rcpf(b);
    float e0 = __builtin_fmaf(-b, r0, 1.0f);
    float r1 = __builtin_fmaf(e0, r0, r0);
    float m0 = a0 * r1;
    float m1 = a1 * r1;
    float f0 = __builtin_fmaf(-b, m0, a0);
    float f1 = __builtin_fmaf(-b, m1, a1);
    float g0 = __builtin_fmaf(f0, r1, m0);
    float g1 = __builtin_fmaf(f1, r1, m1);
    float h0 = __builtin_fmaf(-b, g0, a0);
    float h1 = __builtin_fmaf(-b, g1, a1);
    q0 = __builtin_fmaf(h0, r1, g0);
    q1 = __builtin_fmaf(h1, r1, g1);
}

// wave-uniform "every active lane satisfies p": compares the ballot with the exec mask directly, so the
// predicate never has to be materialised in a VGPR (hipcc's __all() costs a v_cndmask + v_cmp_ne)
PS_D bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }

PS_D float min3_abs(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
PS_D float max3_abs(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Wave-level forms: every comparison is balloted on its own (a ballot of a single compare IS the compare's
// SGPR result) and the masks are combined with scalar ANDs; balloting a compound predicate would first
// materialise it in a VGPR (v_cndmask + v_cmp_ne).
// HI_HOISTED: the upper end of the window has been established for every match of the pair from
// per-hypothesis norms and the pair's largest coordinate (hi_bound_holds), only the lower end is checked here.
template <bool HI_HOISTED>
PS_D bool wave_div_window_ok(float a0, float a1, float b, float c0, float c1, float d)
{
    unsigned long long m = __builtin_amdgcn_ballot_w64(min3_abs(a0, a1, b) >= kDivLo) &
                           __builtin_amdgcn_ballot_w64(min3_abs(c0, c1, d) >= kDivLo);
    if (!HI_HOISTED)
        m &= __builtin_amdgcn_ballot_w64(max3_abs(a0, a1, b) <= kDivHi) &
             __builtin_amdgcn_ballot_w64(max3_abs(c0, c1, d) <= kDivHi);
    return m == __builtin_amdgcn_ballot_w64(true);
}

// |R x + t| <= 3 max|R_ij| max|x_i| + max|t_i| for every point of the pair (coordinates bounded by cmax),
// with 1 % slack for the float roundings of the transform and of the multiplication by fx / fy: if that
// bound times max(|fx|, |fy|, 1) stays below 2^40 no numerator or denominator of this hypothesis can leave the
// window at its upper end.  NaN / inf anywhere makes the comparison false (the per-match check then applies).
PS_D bool hi_bound_holds(const Rigid &m, float cmax, float fmaxK)
{
    float r = 0.0f, t = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) r = fmaxf(r, fabsf(m.R[i][j]));
        t = fmaxf(t, fabsf(m.t[i]));
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) finite = finite && (m.R[i][j] == m.R[i][j]);
        finite = finite && (m.t[i] == m.t[i]);
    }
    const float e = (3.0f * r * cmax + t) * 1.01f;
    return finite && (e * fmaxK <= kDivHi);
}
// Lanes of a wave-uniform 64-bit mask without lane masks in vector registers ((1 << lane) - 1 and its complement are four
// registers the compiler hoists out of every loop): how many set bits lie below the calling lane (v_mbcnt_lo / _hi),
// and whether the calling lane's own bit is set (v_cndmask with the mask as the selector).
PS_D int lanes_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// a 64-bit value that is the same in every lane, moved to scalar registers (loaded through a pointer the compiler cannot
// prove uniform it sits in vector registers)
PS_D unsigned long long uniform64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
PS_D bool lane_in(unsigned long long mask)
{
    int x;
    asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(x) : "s"(mask));
    return x != 0;
}
// cnt += 1 in the lanes of `mask` (v_addc_co_u32 with the mask as carry-in: one VALU op)
PS_D void add_mask(int &cnt, unsigned long long mask)
{
    asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(cnt) : "s"(mask) : "vcc");
}

// true when every magnitude of both (numerator, numerator, denominator) triples lies in [2^-40, 2^40]
// (a NaN operand is ignored by min3/max3; it then yields NaN on either division path: same outcome)
PS_D bool div_window_ok(float a0, float a1, float b, float c0, float c1, float d)
{
    return min3_abs(a0, a1, b) >= kDivLo && min3_abs(c0, c1, d) >= kDivLo && max3_abs(a0, a1, b) <= kDivHi &&
           max3_abs(c0, c1, d) <= kDivHi;
}

// Reference form of the inlier test of one match under one model (RANSAC.cpp:251-281,325-436): used by
// kernel 4 for the selected hypothesis.  Kernel 3 evaluates the same arithmetic through score_accumulate().
template <int MODE>
PS_D bool inlier_test(const Rigid &mdl, const Rigid &inv, const ScoreConsts &k, const float4 &A, const float4 &B,
                      const float4 &C)
{
    float ex, ey, ez;
    xform(mdl, B.x, B.y, B.z, ex, ey, ez); // estimatedOldPosition = R * features[train] + t
    bool in = true;
    if (MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
        float d0 = ex - A.x, d1 = ey - A.y, d2 = ez - A.z;
        float s = d0 * d0 + (d1 * d1 + d2 * d2);
        in = s < A.w;
    }
    if (MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
        float nx, ny, nz;
        xform(inv, A.x, A.y, A.z, nx, ny, nz); // estimatedNewPosition = Rinv * prev[query] + tinv
        float pnu, pnv, pou, pov;
        project(nx, ny, nz, k.fx, k.fy, k.cx, k.cy, pnu, pnv);
        project(ex, ey, ez, k.fx, k.fy, k.cx, k.cy, pou, pov);
        float dxn = pnu - C.z, dyn = pnv - C.w; // predictedNew - realNew
        float dxo = pou - C.x, dyo = pov - C.y; // predictedOld - realOld
        double e0 = (double)dxn * (double)dxn + (double)dyn * (double)dyn;
        double e1 = (double)dxo * (double)dxo + (double)dyo * (double)dyo;
        in = in && (e0 < k.boundR) && (e1 < k.boundR);
    }
    return in;
}

// Kernel 3's form of the same test: cnt += inlier.  Identical values, cheaper instruction stream:
//  * the two quotients of each projected point share one reciprocal inside the checked division window
//    (div2_shared), the whole wavefront falling back to '/' otherwise;
//  * cv::norm's double comparison is settled in float outside the band [bLo, bHi] around boundR: the float
//    sums f0, f1 are within 2^-22 of the exact values and the band's half-width is 2^-21; the larger of the
//    two (as unsigned bit patterns: sums of squares are >= +0 and any NaN sorts above +inf) decides both tests
//    at once; inside the band, or on NaN, the wavefront evaluates in double;
//  * the count is updated inside each branch, so no boolean has to be carried across the branches in a VGPR.
template <int MODE, bool HI_HOISTED>
PS_D void score_accumulate(const Rigid &mdl, const Rigid &inv, const ScoreConsts &k, const float4 &A, const float4 &B,
                           const float4 &C, int &cnt)
{
    float ex, ey, ez;
    xform(mdl, B.x, B.y, B.z, ex, ey, ez);
    bool in = true;
    if (MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
        float d0 = ex - A.x, d1 = ey - A.y, d2 = ez - A.z;
        float s = d0 * d0 + (d1 * d1 + d2 * d2);
        in = s < A.w;
    }
    if (MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR) {
        float nx, ny, nz;
        xform(inv, A.x, A.y, A.z, nx, ny, nz);
        float pnu, pnv, pou, pov;
        const float a0 = nx * k.fx, a1 = ny * k.fy, c0 = ex * k.fx, c1 = ey * k.fy;
        if (wave_div_window_ok<HI_HOISTED>(a0, a1, nz, c0, c1, ez)) {
            float q0, q1, q2, q3;
            div2_shared(a0, a1, nz, q0, q1);
            div2_shared(c0, c1, ez, q2, q3);
            pnu = q0 + k.cx; pnv = q1 + k.cy;
            pou = q2 + k.cx; pov = q3 + k.cy;
        } else {
            pnu = a0 / nz + k.cx; pnv = a1 / nz + k.cy;
            pou = c0 / ez + k.cx; pov = c1 / ez + k.cy;
        }
        float dxn = pnu - C.z, dyn = pnv - C.w;
        float dxo = pou - C.x, dyo = pov - C.y;
        // pre-filter only (the decision inside the band is taken in double below): one product may be fused, the
        // float sums then sit within 2^-23 of the exact values, inside the 2^-22 the band allows for
        float f0 = __builtin_fmaf(dxn, dxn, dyn * dyn), f1 = __builtin_fmaf(dxo, dxo, dyo * dyo);
        uint32_t u0, u1;
        memcpy(&u0, &f0, 4);
        memcpy(&u1, &f1, 4);
        uint32_t um = u0 > u1 ? u0 : u1;
        float fm;
        memcpy(&fm, &um, 4);
        const unsigned long long mIn = __builtin_amdgcn_ballot_w64(fm < k.bLo);
        const unsigned long long mOut = __builtin_amdgcn_ballot_w64(fm > k.bHi);
        if ((mIn | mOut) == __builtin_amdgcn_ballot_w64(true)) {
            if (MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR)
                add_mask(cnt, mIn & __builtin_amdgcn_ballot_w64(in));
            else
                add_mask(cnt, mIn);
            return;
        }
        // cold side: keep the eight f64 instructions from being speculated above the branch
        asm volatile("" : "+v"(dxn), "+v"(dyn), "+v"(dxo), "+v"(dyo));
        double e0 = (double)dxn * (double)dxn + (double)dyn * (double)dyn;
        double e1 = (double)dxo * (double)dxo + (double)dyo * (double)dyo;
        in = in && (e0 < k.boundR) && (e1 < k.boundR);
    }
    cnt += in ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// Kernel 3.  RANSAC loop body of RANSAC.cpp:93-135 for ALL hypotheses at once.
// lane = hypothesis (model in VGPRs); the match records are wave-uniform and arrive through
// scalar loads.  grid = ceil(H/256) * msplit * P work-groups in XCD-aware order; msplit > 1 splits the
// match range and merges the integer counts with atomicAdd.  __launch_bounds__(256, 6): 6 waves per SIMD
// (80 VGPRs; the one-off Jacobi-SVD prologue spills 3-5 dwords) measured 2-3 % faster than 5 waves, 8 is slower.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock, 6) void ps_ransac_score(const float4 *__restrict__ recA,
                                                          const float4 *__restrict__ recB,
                                                          const float4 *__restrict__ recC,
                                                          const int32_t *__restrict__ mvalid,
                                                          const float2 *__restrict__ cmaxArr, ModelArgs ma,
                                                          ScoreConsts k, int H, int cap, int minRun, int msplit,
                                                          int32_t *__restrict__ counts)
{
    const unsigned hb = (unsigned)((H + kBlock - 1) / kBlock);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bx = L % hb, by = (L / hb) % (unsigned)msplit;
    const int p = (int)(L / (hb * (unsigned)msplit));
    const int M = mvalid[p];
    if (M < minRun) return; // too few matches: kernel 4 returns identity (RANSAC.cpp:77-80)
    const int h = (int)bx * kBlock + threadIdx.x;
    const size_t rbase = (size_t)p * cap;
    const int m0 = (int)(((long long)M * by) / msplit);
    const int m1 = (int)(((long long)M * (by + 1)) / msplit);

    Rigid mdl, inv;
    bool valid = false;
    if (h < H) {
        valid = gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)h, mdl);
        if (ma.models && by == 0 && h < ma.modelH) store_model(ma, (size_t)p * ma.modelH + h, mdl);
        if (MODE == PS_REPROJECTION_ERROR || MODE == PS_EUCLIDEAN_AND_REPROJECTION_ERROR)
            inverse_rigid_general(mdl, inv);
    }
    int cnt = 0;
    if (MODE == PS_MAHALANOBIS_ERROR) valid = false; // dead metric in the reference: scores 0
    const float4 *__restrict__ pa = recA + rbase;
    const float4 *__restrict__ pb = recB + rbase;
    const float4 *__restrict__ pc = recC + rbase;
    if (MODE == PS_EUCLIDEAN_ERROR || MODE == PS_ADAPTIVE_ERROR) {
#pragma unroll 4
        for (int m = m0; m < m1; ++m) { // straight-line body: the unrolled loop batches the scalar record loads
            float4 A = pa[m], B = pb[m];
            score_accumulate<MODE, false>(mdl, inv, k, A, B, A, cnt);
        }
    } else {
        // wave-uniform branches inside: not unrollable.  The upper end of the division window is settled here,
        // once per hypothesis, whenever the pair's coordinate bound allows it.
        const float cmax = cmaxArr[p].x;
        const float fmaxK = fmaxf(fmaxf(fabsf(k.fx), fabsf(k.fy)), 1.0f);
        const bool hoisted = wave_all(hi_bound_holds(mdl, cmax, fmaxK) && hi_bound_holds(inv, cmax, fmaxK));
        if (hoisted) {
            for (int m = m0; m < m1; ++m) {
                float4 A = pa[m], B = pb[m], C = pc[m];
                score_accumulate<MODE, true>(mdl, inv, k, A, B, C, cnt);
            }
        } else {
            for (int m = m0; m < m1; ++m) {
                float4 A = pa[m], B = pb[m], C = pc[m];
                score_accumulate<MODE, false>(mdl, inv, k, A, B, C, cnt);
            }
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

// ------------------------------------------------------------------------------------------
// Wave-level N-point Umeyama (refit, RANSAC.cpp:153): the canonical summation order is
// 64 strided partials (lane l takes elements l, l+64, ...) and a stride-1,2,4,..,32 tree.
// getter(j, src[3], dst[3]) yields the j-th correspondence.  All 64 lanes return the model.
// ------------------------------------------------------------------------------------------
// (the strides 1 .. 8 stay inside a row of 16 lanes: DPP row shifts, no trip through the LDS crossbar; the partial sums the
// result depends on -- lanes that are multiples of twice the stride -- read what __shfl_down would hand them; strides 16 and
// 32 combine the four row sums, read as scalars: (r0 + r16) + (r32 + r48), the tree's own association)
PS_D float wave_tree_sum(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, false));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x102, 0xF, 0xF, false));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0xF, false));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108, 0xF, 0xF, false));
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r16 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r32 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r48 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r16) + (r32 + r48);
#else
    return v;
#endif
}

// The correspondences come from two places: j < split through getA (kernel 4: operands parked in LDS), the rest through getB
// (global memory).  Two loops, not one getter with a branch inside: the compiler merged the branch's two sources into generic
// pointers and FLAT loads (four per trip, waited for together: 350 cycles a trip) -- and, before that, into arrays in scratch
// memory (600 cycles a trip; the two passes were 8 of kernel 4's 26 us for a single pair).  A getter yields
// u = (src.x, src.y, src.z, dst.x), v = (dst.y, dst.z, -, -).  The LDS loop requests the lane's next element before it adds
// the current one; each lane still adds its elements in ascending order.
template <typename GetA, typename GetB> PS_D bool wave_umeyama(int k, int split, GetA getA, GetB getB, Rigid &mdl)
{
    const int lane = threadIdx.x & 63;
    const float one_over_n = 1.0f / (float)k;
    const int k1 = k < split ? k : split;
    // the lane's first element behind the split: the smallest j = lane (mod 64) with j >= k1
    const int jB = lane + (((k1 > lane ? k1 - lane : 0) + 63) >> 6 << 6);
    float ss[3] = {0.0f, 0.0f, 0.0f}, ds[3] = {0.0f, 0.0f, 0.0f};
    auto add1 = [&](const float4 &u, const float4 &v) {
        ss[0] = ss[0] + u.x; ss[1] = ss[1] + u.y; ss[2] = ss[2] + u.z;
        ds[0] = ds[0] + u.w; ds[1] = ds[1] + v.x; ds[2] = ds[2] + v.y;
    };
    {
        float4 u, v, un, vn;
        if (lane < k1) getA(lane, u, v);
        for (int j = lane; j < k1; j += 64) {
            if (j + 64 < k1) getA(j + 64, un, vn);
            add1(u, v);
            u = un;
            v = vn;
        }
        for (int j = jB; j < k; j += 64) {
            getB(j, u, v);
            add1(u, v);
        }
    }
    float sm[3], dm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sm[c] = wave_tree_sum(ss[c]) * one_over_n;
        dm[c] = wave_tree_sum(ds[c]) * one_over_n;
    }
    float acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = 0.0f;
    auto add2 = [&](const float4 &u, const float4 &v) {
        const float s[3] = {u.x, u.y, u.z}, d[3] = {u.w, v.x, v.y};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[r][c] = acc[r][c] + (d[r] - dm[r]) * (s[c] - sm[c]);
    };
    {
        float4 u, v, un, vn;
        if (lane < k1) getA(lane, u, v);
        for (int j = lane; j < k1; j += 64) {
            if (j + 64 < k1) getA(j + 64, un, vn);
            add2(u, v);
            u = un;
            v = vn;
        }
        for (int j = jB; j < k; j += 64) {
            getB(j, u, v);
            add2(u, v);
        }
    }
    float sigma[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) sigma[r][c] = one_over_n * wave_tree_sum(acc[r][c]);
    bool ok = umeyama_finish(sigma, sm, dm, mdl);
    if (!ok) set_identity(mdl);
    return ok;
}

struct SelectArgs {
    int estimator;       // PsEstimator
    int mode;            // RANSAC::ERROR_VERSION
    int H, cap;
    int minMatches;      // minimalNumberOfMatches (RANSAC) or 8 (USAC_wrapper.cpp:120)
    int iter0;           // initial trip limit: min(computeRANSACIteration(0.20), H) or min(850000, H)
    double minRatio;     // minimalInlierRatioThreshold
    const float *ransacTab;  // non-increasing; ransacTab[k] = smallest ratio r with iterations(r) <= k
    int ransacTabN;
    float ransacTiny;        // ratios <= this make the reference's quotient -inf (limit 0)
    const double *usacTab;   // non-increasing; usacTab[k] = smallest p with stopping(p) <= k+1
    int usacTabN;
    int trainRange;      // train indices are < trainRange (size of the uniqueness bitmaps, 0 = skip)
    int stageCap;        // inlier correspondences staged in LDS for the refit and the re-selection (8 words each)
};

// First k of [0, n) with !(x < tab[k]) -- n if there is none -- for a non-increasing table, found by the whole wavefront: every
// lane probes one of 64 evenly spaced entries and the ballot narrows the range 64-fold per step (850 000 entries: four
// dependent loads instead of a binary search's twenty, 0.5 us each from a cold table).  Wave-uniform arguments and result.
template <typename T> PS_D int table_first_not_below(const T *__restrict__ tab, int n, T x)
{
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;
        const int kq = lo + lane * step;
        const bool valid = kq < hi;
        const bool below = valid && (x < tab[valid ? kq : lo]);
        const int t = __popcll(__ballot(below));     // the probes that are still "below" form a prefix of the lanes
        const int nvalid = __popcll(__ballot(valid));
        const int nlo = t > 0 ? lo + (t - 1) * step + 1 : lo;
        const int nhi = t < nvalid ? lo + t * step : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}
// min(Kcap, computeRANSACIteration(r)) through the host-built threshold table (RANSAC.cpp:450-461).
PS_D int ransac_limit(const SelectArgs &a, float r)
{
    if (r <= a.ransacTiny) return 0;
    return table_first_not_below<float>(a.ransacTab, a.ransacTabN, r); // first k with r >= tab[k]
}
// min(H, updateStandardStopping(inliers, M, 3)) through the host-built table (USAC.h:944-971).
PS_D int usac_limit(const SelectArgs &a, unsigned c, unsigned M)
{
    double n_in = 1.0, n_pts = 1.0;
#pragma unroll
    for (unsigned i = 0; i < 3; ++i) {
        n_in *= (double)(unsigned)(c - i);
        n_pts *= (double)(unsigned)(M - i);
    }
    double pgood = n_in / n_pts;
    const int lo = table_first_not_below<double>(a.usacTab, a.usacTabN, pgood);
    int stop = lo + 1;
    return stop < a.H ? stop : a.H;
}

PS_D void store_pose(float *pose, const Rigid &m)
{
    // column-major 4x4 (Eigen::Matrix4f storage)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) pose[4 * c + r] = m.R[r][c];
        pose[4 * c + 3] = 0.0f;
    }
    pose[12] = m.t[0]; pose[13] = m.t[1]; pose[14] = m.t[2]; pose[15] = 1.0f;
}

// ------------------------------------------------------------------------------------------
// Kernel 4.  One workgroup per pair:
//  (1) replay of the sequential selection over counts[0..H): strict '>' first-best with the
//      adaptive trip limit (RANSAC.cpp:87,438-455 / USAC.h:326,409-414,498-509) or plain arg-max;
//  (2) inliers of the selected hypothesis, in match order;
//  (3) Umeyama refit on them and the Euclidean re-selection (RANSAC.cpp:152-158);
//  (4) ratio gate (RANSAC.cpp:161-164 / USAC_wrapper.cpp:139-141), pointInlierRatio (RANSAC.h:56-66).
// ------------------------------------------------------------------------------------------
template <int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK) void ps_select_refit(const float4 *__restrict__ recA,
                                                          const float4 *__restrict__ recB,
                                                          const float4 *__restrict__ recC,
                                                          const int4 *__restrict__ recD,
                                                          const int32_t *__restrict__ mvalid,
                                                          const int32_t *__restrict__ counts,
                                                          const PsDMatch *__restrict__ matches,
                                                          const int32_t *__restrict__ numMatches, int matchStride,
                                                          ModelArgs ma, ScoreConsts k, SelectArgs a,
                                                          int32_t *__restrict__ idxList, float *__restrict__ poseOut,
                                                          uint8_t *__restrict__ maskOut,
                                                          PsRansacStats *__restrict__ statsOut,
                                                          unsigned long long *__restrict__ stamps,
                                                          const unsigned *__restrict__ bailDev, unsigned *__restrict__ bailHost)
{
    phase_stamp(stamps, 4);
    // the staged scoring's "nothing to gain" counters (ps_stage_reorder) on their way to the host's policy: two plain stores
    // into mapped host memory by one thread of the launch that follows every stage (nobody waits for them)
    if (bailHost != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        bailHost[0] = bailDev[0];
        bailHost[1] = bailDev[1];
    }
    extern __shared__ __align__(16) uint32_t s_bits[]; // two bitmaps over train indices: 2 * ceil(trainRange/32) words
    __shared__ int s_wsum[2 * (BLOCK / 64)];
    __shared__ unsigned long long s_red[BLOCK / 64];
    __shared__ int s_sel[4];
    __shared__ Rigid s_model;
    __shared__ int s_uniq[2];

    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int words = (a.trainRange + 31) >> 5;
    // [stageCap][8]: cur xyz (src), prev xyz (dst), the match's Euclidean bound, its index among the depth-valid matches
    float *s_stage = reinterpret_cast<float *>(s_bits + ((2 * words + 3) & ~3));
    const int M = mvalid[p];
    const int nIn = numMatches[p];
    const size_t rbase = (size_t)p * a.cap;
    const int32_t *cnts = counts + (size_t)p * a.H;
    uint8_t *mask = maskOut + (size_t)p * matchStride;
    int32_t *list = idxList + rbase;

    for (int i = tid; i < nIn; i += BLOCK) mask[i] = 0;

    const bool run = !(M < a.minMatches || M < 3);
    int bestIdx = -1, bestCount = 0, trips = 0;
    // The inlier pass' first records do not depend on the selection: their loads (written by another launch: HBM latency)
    // are in flight while the schedule is replayed; every later trip fetches the next one's before its scan.
    const bool needInv = (a.mode == PS_REPROJECTION_ERROR || a.mode == PS_EUCLIDEAN_AND_REPROJECTION_ERROR);
    const bool needC = needInv;
    float4 cA, cB, cC;
    auto fetch = [&](int i, float4 &A, float4 &B, float4 &C) {
        A = B = C = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < M) {
            A = recA[rbase + i];
            B = recB[rbase + i];
            if (needC) C = recC[rbase + i];
        }
    };
    if (run) fetch(tid, cA, cB, cC);

    if (run) {
        if (a.estimator == PS_EST_FIXED) {
            unsigned long long key = 0ull;
            auto take = [&](int i, int c) {
                unsigned long long kk = ((unsigned long long)(unsigned)c << 32) | (0xFFFFFFFFu - (unsigned)i);
                key = kk > key ? kk : key;
            };
            int i = tid;
            for (; i + 7 * BLOCK < a.H; i += 8 * BLOCK) { // eight loads in flight: the counts come from another launch (HBM latency)
                int c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) c[u] = cnts[i + u * BLOCK];
#pragma unroll
                for (int u = 0; u < 8; ++u) take(i + u * BLOCK, c[u]);
            }
            for (; i < a.H; i += BLOCK) take(i, cnts[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                unsigned long long other = __shfl_down(key, o, 64);
                key = other > key ? other : key;
            }
            if (lane == 0) s_red[wv] = key;
            __syncthreads();
            if (tid == 0) {
                unsigned long long b = s_red[0];
                for (int i = 1; i < BLOCK / 64; ++i) b = s_red[i] > b ? s_red[i] : b;
                int c = (int)(b >> 32);
                s_sel[0] = c > 0 ? (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFu)) : -1;
                s_sel[1] = c;
                s_sel[2] = a.H;
            }
            __syncthreads();
        } else if (a.estimator == PS_EST_RANSAC && a.H <= 8 * 64 && a.ransacTabN <= 8 * 64) {
            // The reference's own regime (at most 487 iterations): every wavefront replays the schedule by itself, counts and
            // stop table in registers (eight of each per lane, index = 64 u + lane), a round = eight ballots and scalar code.
            // No barrier, no table search through memory: the work-group form below spends 0.8 us per record on its
            // dependent table loads alone (3.9 of the single pair's 91 us, profiles/r04g).
            int c[8];
            float tb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = u * 64 + lane;
                c[u] = idx < a.H ? cnts[idx] : 0;
                tb[u] = idx < a.ransacTabN ? a.ransacTab[idx] : 0.0f;
            }
            int pos = 0, limit = a.iter0, best = 0, bIdx = -1;
            for (;;) {
                int fIdx = -1, fCnt = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = u * 64 + lane;
                    const unsigned long long mk = __ballot(idx >= pos && idx < limit && c[u] > best);
                    if (fIdx < 0 && mk != 0ull) { // the first index in [pos, limit) that beats the best
                        const int l = __ffsll((long long)mk) - 1;
                        fIdx = u * 64 + l;
                        fCnt = __builtin_amdgcn_readlane(c[u], l);
                    }
                }
                if (fIdx < 0) break;
                bIdx = fIdx;
                best = fCnt;
                pos = bIdx + 1;
                const float r = (float)best / (float)M; // ratio as RANSAC.cpp:280
                // ransac_limit(): the first k with r >= tab[k] of a non-increasing table = the number of k with r < tab[k]
                int lim = 0;
                if (!(r <= a.ransacTiny)) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        lim += __popcll(__ballot(u * 64 + lane < a.ransacTabN && r < tb[u]));
                }
                limit = lim;
            }
            bestIdx = bIdx;
            bestCount = best;
            trips = (bIdx + 1 > limit) ? bIdx + 1 : limit;
        } else {
            // Sequential replay.  State changes only at "records" (count > best so far), so the
            // workgroup repeatedly finds the first index in [pos, limit) that beats the best.
            // The range is searched in windows that double (2 Ki, 4 Ki, ... indices; eight loads of a thread in flight): the
            // first record lies a few indices behind `pos` as a rule, and a thread that had no hit of its own used to walk the
            // whole range alone, one dependent load at a time -- under USAC's initial limit of 850 000 that was 480 us for a
            // single pair (profiles/r04g/single_pair_schedules.txt).  A window never reaches beyond `limit`: the same indices
            // are read as before (counts beyond the trip limit may be stale under the staged scoring).
            int pos = 0, limit = a.iter0, best = 0, bIdx = -1;
            for (;;) {
                unsigned f = 0xFFFFFFFFu;
                int span = 8 * BLOCK;
                for (int w0 = pos; w0 < limit;) {
                    const int w1 = (limit - w0 > span) ? w0 + span : limit;
                    unsigned found = 0xFFFFFFFFu;
                    for (int i0 = w0 + tid; i0 < w1 && found == 0xFFFFFFFFu; i0 += 8 * BLOCK) {
                        int c[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * BLOCK;
                            c[u] = i < w1 ? cnts[i] : (int)0x80000000;
                        }
#pragma unroll
                        for (int u = 7; u >= 0; --u)
                            if (c[u] > best) found = (unsigned)(i0 + u * BLOCK); // (descending: the smallest index stays)
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        unsigned other = __shfl_down(found, o, 64);
                        found = other < found ? other : found;
                    }
                    if (lane == 0) s_red[wv] = found;
                    __syncthreads();
                    f = (unsigned)s_red[0];
                    for (int i = 1; i < BLOCK / 64; ++i) f = (unsigned)s_red[i] < f ? (unsigned)s_red[i] : f;
                    __syncthreads();
                    if (f != 0xFFFFFFFFu) break;
                    w0 = w1;
                    if (span < 512 * BLOCK) span *= 2;
                }
                if (f == 0xFFFFFFFFu) break;
                bIdx = (int)f;
                best = cnts[bIdx];
                pos = bIdx + 1;
                if (a.estimator == PS_EST_USAC)
                    limit = usac_limit(a, (unsigned)best, (unsigned)M);
                else
                    limit = ransac_limit(a, (float)best / (float)M); // ratio as RANSAC.cpp:280
            }
            if (tid == 0) {
                s_sel[0] = bIdx;
                s_sel[1] = best;
                s_sel[2] = (bIdx + 1 > limit) ? bIdx + 1 : limit;
            }
            __syncthreads();
        }
        if (!(a.estimator == PS_EST_RANSAC && a.H <= 8 * 64 && a.ransacTabN <= 8 * 64)) {
            bestIdx = s_sel[0];
            bestCount = s_sel[1];
            trips = s_sel[2];
        }
    }

    phase_stamp(stamps, 5); // (1) selection replayed
    // ---- (2) inliers of the selected hypothesis ----
    Rigid mdl, inv;
    set_identity(mdl);
    set_identity(inv);
    int kin = 0;
    if (run && bestIdx >= 0) {
        if (ma.models && bestIdx < ma.modelH)
            load_model(ma, (size_t)p * ma.modelH + bestIdx, mdl); // parked by kernel 3 (an invalid sample parks the identity)
        else
            gen_model(recA, recB, rbase, (uint32_t)M, ma, base_seed(ma) + (uint64_t)p, (uint32_t)bestIdx, mdl);
        if (needInv) inverse_rigid_general(mdl, inv);
        int trip = 0;
        for (int i0 = 0; i0 < M; i0 += BLOCK, ++trip) {
            const int i = i0 + tid;
            float4 nA, nB, nC;
            fetch(i + BLOCK, nA, nB, nC);
            const float4 A = cA, B = cB, C = cC;
            bool in = false;
            if (i < M) {
                switch (a.mode) {
                case PS_EUCLIDEAN_ERROR: in = inlier_test<PS_EUCLIDEAN_ERROR>(mdl, inv, k, A, B, C); break;
                case PS_ADAPTIVE_ERROR: in = inlier_test<PS_ADAPTIVE_ERROR>(mdl, inv, k, A, B, C); break;
                case PS_REPROJECTION_ERROR: in = inlier_test<PS_REPROJECTION_ERROR>(mdl, inv, k, A, B, C); break;
                case PS_EUCLIDEAN_AND_REPROJECTION_ERROR:
                    in = inlier_test<PS_EUCLIDEAN_AND_REPROJECTION_ERROR>(mdl, inv, k, A, B, C);
                    break;
                default: in = false; break;
                }
            }
            int total;
            int pos = block_scan_flag_alt<BLOCK>(in, total, s_wsum, trip);
            if (in) {
                const int slot = kin + pos;
                list[slot] = i;
                if (slot < a.stageCap) { // the refit's and the re-selection's operands, parked in LDS while they are in registers
                    float *d = s_stage + 8 * slot;
                    reinterpret_cast<float4 *>(d)[0] = make_float4(B.x, B.y, B.z, A.x);
                    reinterpret_cast<float4 *>(d)[1] = make_float4(A.y, A.z, A.w, __int_as_float(i));
                }
            }
            kin += total;
            cA = nA; cB = nB; cC = nC;
        }
        __syncthreads(); // list / staged points visible to the whole workgroup
    }

    phase_stamp(stamps, 6); // (2) winner's model, inlier pass, ordered compaction
    const float ratioF = run ? (float)bestCount / (float)M : 0.0f;
    const double ratioD = (double)ratioF;
    bool accepted = run && !(ratioD < a.minRatio);
    int nfinal = 0;

    // pointInlierRatio (RANSAC.h:56-66) needs the unique trainIdx among ALL input matches and among the final inliers.
    // The first does not depend on the refit: the other wavefronts build it while wave 0 runs the refit's serial
    // Umeyama + Jacobi-SVD chain (14 of this kernel's 35 us for a single pair, profiles/r03e/latency_stamps.json).
    for (int i = tid; i < 2 * words; i += BLOCK) s_bits[i] = 0u;
    if (tid == 0) s_uniq[0] = s_uniq[1] = 0;
    __syncthreads();
    const PsDMatch *mm = matches + (size_t)p * matchStride;
    int ua = 0, ui = 0;
    auto all_matches_pass = [&](int first, int stride) {
        for (int i = first; i < nIn; i += stride) {
            const int t = mm[i].trainIdx;
            if (t >= 0 && t < a.trainRange) {
                const uint32_t bit = 1u << (t & 31);
                const uint32_t old = atomicOr(&s_bits[t >> 5], bit);
                ua += (old & bit) ? 0 : 1;
            }
        }
    };

    // a final inlier: its flag in the caller's mask and -- pointInlierRatio's numerator (RANSAC.h:56-66) -- its train index
    // in the second bitmap, where it is found (round 3 walked all input matches once more for this and read the mask back)
    auto note_inlier = [&](const int4 &d) { // d = (index in the match list, queryIdx, trainIdx, 0)
        mask[d.x] = 1;
        if (d.z >= 0 && d.z < a.trainRange) {
            const uint32_t bit = 1u << (d.z & 31);
            const uint32_t old = atomicOr(&s_bits[words + (d.z >> 5)], bit);
            ui += (old & bit) ? 0 : 1;
        }
    };
    if (run && a.estimator != PS_EST_USAC) {
        // ---- (3) refit on the best inliers (wave 0), re-selection with the Euclid/adaptive rule ----
        if (wv != 0) all_matches_pass(tid - 64, BLOCK - 64);
        if (wv == 0) {
            Rigid ref;
            wave_umeyama(kin, a.stageCap,
                         [&](int j, float4 &u, float4 &v) { // parked by the inlier pass
                             const float4 *q = reinterpret_cast<const float4 *>(s_stage + 8 * j);
                             u = q[0];
                             v = q[1];
                         },
                         [&](int j, float4 &u, float4 &v) { // more inliers than the LDS holds: from the records
                             const int i = list[j];
                             const float4 A = recA[rbase + i], B = recB[rbase + i];
                             u = make_float4(B.x, B.y, B.z, A.x);
                             v = make_float4(A.y, A.z, 0.0f, 0.0f);
                         },
                         ref);
            if (lane == 0) s_model = ref;
        }
        __syncthreads();
        phase_stamp(stamps, 7); // (3a) refit: wave-level Umeyama + Jacobi SVD
        mdl = s_model;
        // (two loops, one per source of the operands: a branch inside one loop was compiled into generic pointers and FLAT
        // loads; the order of the matches does not matter here)
        auto reselect = [&](const float4 &A, const float4 &B, int i) {
            const bool in = inlier_test<PS_EUCLIDEAN_ERROR>(mdl, inv, k, A, B, A); // s < A.w : plain or adaptive bound
            if (in && accepted) note_inlier(recD[rbase + i]);
            nfinal += in ? 1 : 0;
        };
        const int kLds = kin < a.stageCap ? kin : a.stageCap;
        for (int j = tid; j < kLds; j += BLOCK) { // parked by the inlier pass
            const float4 *q = reinterpret_cast<const float4 *>(s_stage + 8 * j);
            const float4 u = q[0], v = q[1];
            reselect(make_float4(u.w, v.x, v.y, v.z), make_float4(u.x, u.y, u.z, 0.0f), __float_as_int(v.w));
        }
        for (int j = a.stageCap + tid; j < kin; j += BLOCK) {
            const int i = list[j];
            reselect(recA[rbase + i], recB[rbase + i], i);
        }
        { // (each lane counted its own: one exchange at the end instead of a scan per trip)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nfinal += __shfl_down(nfinal, o, 64);
            if (lane == 0) s_wsum[wv] = nfinal;
            __syncthreads();
            nfinal = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) nfinal += s_wsum[w];
        }
        if (!accepted) nfinal = 0; // identity + inliers cleared (RANSAC.cpp:161-164)
    } else if (run) {
        // USAC: no refit (USAC_wrapper.cpp:204-222 commented out); the loop's inliers are returned even
        // when the ratio gate replaces the pose by identity (USAC_wrapper.cpp:139-141).
        for (int j = tid; j < kin; j += BLOCK) note_inlier(recD[rbase + list[j]]);
        nfinal = kin;
        all_matches_pass(tid, BLOCK);
    } else
        all_matches_pass(tid, BLOCK);

    phase_stamp(stamps, 8); // (3b) Euclidean re-selection, mask, (4) unique trainIdx among the final inliers
    { // one LDS atomic per wave and tally (the compiler's own form walks the 64 lanes one by one)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ua += __shfl_down(ua, o, 64);
            ui += __shfl_down(ui, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&s_uniq[0], ua);
            atomicAdd(&s_uniq[1], ui);
        }
    }
    __syncthreads();

    if (tid == 0) {
        Rigid out = mdl;
        if (!accepted) set_identity(out);
        store_pose(poseOut + (size_t)p * 16, out);
        PsRansacStats st;
        st.numMatchesIn = nIn;
        st.numMatchesValid = M;
        st.bestHypothesis = bestIdx;
        st.bestInlierCount = bestCount;
        st.iterationsRun = trips;
        st.numInliers = nfinal;
        st.accepted = accepted ? 1 : 0;
        st.bestInlierRatio = ratioF;
        st.pointInlierRatio = (double)s_uniq[1] / (double)s_uniq[0];
        statsOut[p] = st;
    }
    phase_stamp(stamps, 9); // (4) pointInlierRatio, pose and statistics stored
}

// ------------------------------------------------------------------------------------------
// N2: guided map matching, core of Matcher::matchXYZ (reference src/Matcher/matcher.cpp:694-746).
// One wavefront per map feature j; lanes sweep the current frame's keypoints 64 at a time.
//   candidate i : |mapPos[j] - curPos[i]| < sphereRadius (float norm against a double radius, exact
//                 squared-domain bound) and |predictedLevel[i] - level[j]| <= 1          (:699-711)
//   value(i)    : popcount of the per-byte SATURATING difference mapDesc[j] - curDesc[i]
//                 (cv::Mat subtraction of CV_8U + NORM_HAMMING, :719-721 -- not XOR Hamming)
//   best        : smallest value, first index on ties                                      (:714-727)
//   accepted    : acceptRatio * value <= best (double)                                     (:734-746)
// WRITE = false counts the accepted candidates per map feature, WRITE = true emits them at the
// offsets of the exclusive scan, so the output is ordered by (j, i) like the reference's push_back loop.
// ------------------------------------------------------------------------------------------
PS_D uint32_t satdiff_popc256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    uint32_t v = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int d = (int)((aw[w] >> (8 * k)) & 0xFFu) - (int)((bw[w] >> (8 * k)) & 0xFFu);
            d = d < 0 ? 0 : d;
            v += (uint32_t)__popc((unsigned)d);
        }
    return v;
}

template <bool WRITE>
__global__ __launch_bounds__(kBlock) void ps_match_xyz_kernel(const float *__restrict__ mapPos,
                                                              const uint4 *__restrict__ mapDesc,
                                                              const int32_t *__restrict__ mapLevel, int nmap,
                                                              const float *__restrict__ curPos,
                                                              const uint4 *__restrict__ curDesc,
                                                              const int32_t *__restrict__ curLevel, int ncur,
                                                              float radiusBound, double acceptRatio,
                                                              int32_t *__restrict__ counts,
                                                              const int32_t *__restrict__ offsets,
                                                              PsDMatch *__restrict__ out, int cap)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (j >= nmap) return;
    const float mx = mapPos[3 * j], my = mapPos[3 * j + 1], mz = mapPos[3 * j + 2];
    const int lj = mapLevel[j];
    const uint4 a0 = mapDesc[2 * j], a1 = mapDesc[2 * j + 1];

    // sweep 1: best value among the candidates (packed (value, index) minimum = first index on ties)
    unsigned long long best = ~0ull;
    constexpr int UN = 4; // position / level loads of four 64-keypoint groups are issued before any is used
    const int lastc = ncur - 1;
    for (int i0 = 0; i0 < ncur; i0 += 64 * UN) {
        float px[UN], py[UN], pz[UN];
        int lv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + 64 * u + lane, ic = i < ncur ? i : lastc;
            px[u] = curPos[3 * ic]; py[u] = curPos[3 * ic + 1]; pz[u] = curPos[3 * ic + 2];
            lv[u] = curLevel[ic];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + 64 * u + lane;
            float d0 = mx - px[u], d1 = my - py[u], d2 = mz - pz[u];
            float s = d0 * d0 + (d1 * d1 + d2 * d2);
            int li = lv[u];
            if (i < ncur && s < radiusBound && li - 1 <= lj && lj <= li + 1) {
                uint32_t v = satdiff_popc256(a0, a1, curDesc[2 * i], curDesc[2 * i + 1]);
                unsigned long long key = ((unsigned long long)v << 32) | (unsigned)i;
                best = key < best ? key : best;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_down(best, o, 64);
        best = other < best ? other : best;
    }
    best = __shfl(best, 0, 64);
    if (best == ~0ull) { // no candidate: nothing is emitted for this map feature
        if (!WRITE && lane == 0) counts[j] = 0;
        return;
    }
    const float bestVal = (float)(uint32_t)(best >> 32);

    // sweep 2: every candidate within the accept ratio of the best, ascending i
    int n = 0;
    const int base = WRITE ? offsets[j] : 0;
    for (int i0 = 0; i0 < ncur; i0 += 64 * UN) {
        float px[UN], py[UN], pz[UN];
        int lv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + 64 * u + lane, ic = i < ncur ? i : lastc;
            px[u] = curPos[3 * ic]; py[u] = curPos[3 * ic + 1]; pz[u] = curPos[3 * ic + 2];
            lv[u] = curLevel[ic];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + 64 * u + lane;
            bool acc = false;
            float value = 0.f;
            float d0 = mx - px[u], d1 = my - py[u], d2 = mz - pz[u];
            float s = d0 * d0 + (d1 * d1 + d2 * d2);
            int li = lv[u];
            if (i < ncur && s < radiusBound && li - 1 <= lj && lj <= li + 1) {
                value = (float)satdiff_popc256(a0, a1, curDesc[2 * i], curDesc[2 * i + 1]);
                acc = acceptRatio * (double)value <= (double)bestVal;
            }
            unsigned long long bal = __ballot(acc);
            if (WRITE && acc) {
                int pos = base + n + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < cap) {
                    PsDMatch m;
                    m.queryIdx = j;
                    m.trainIdx = i;
                    m.imgIdx = -1; // default-constructed cv::DMatch (matcher.cpp:741)
                    m.distance = value;
                    out[pos] = m;
                }
            }
            n += __popcll(bal);
        }
    }
    if (!WRITE && lane == 0) counts[j] = n;
}

// exclusive scan of counts[0..n) by one work-group; total written to offsets[n]
__global__ __launch_bounds__(kBlock) void ps_exclusive_scan(const int32_t *__restrict__ counts, int n,
                                                            int32_t *__restrict__ offsets)
{
    __shared__ int s_w[kBlock / 64];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i0 = 0; i0 < n; i0 += kBlock) {
        const int i = i0 + threadIdx.x;
        int v = i < n ? counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int off = s_carry;
        for (int w = 0; w < wv; ++w) off += s_w[w];
        if (i < n) offsets[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = s_carry;
}

// ------------------------------------------------------------------------------------------
// Stand-alone fits
// ------------------------------------------------------------------------------------------
// ps_umeyama_f32: one wavefront per point set.
__global__ __launch_bounds__(64) void ps_umeyama_sets(const float *__restrict__ src, const float *__restrict__ dst,
                                                      int k, int nsets, float *__restrict__ T,
                                                      int32_t *__restrict__ valid)
{
    const int s = blockIdx.x;
    if (s >= nsets) return;
    const float *sp = src + (size_t)s * k * 3;
    const float *dp = dst + (size_t)s * k * 3;
    Rigid m;
    auto get = [&](int j, float4 &u, float4 &v) {
        u = make_float4(sp[3 * j], sp[3 * j + 1], sp[3 * j + 2], dp[3 * j]);
        v = make_float4(dp[3 * j + 1], dp[3 * j + 2], 0.0f, 0.0f);
    };
    bool ok = wave_umeyama(k, 0, get, get, m);
    if ((threadIdx.x & 63) == 0) {
        store_pose(T + (size_t)s * 16, m);
        valid[s] = ok ? 1 : 0;
    }
}

PS_D double wave_tree_sum_f64(double v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = v + __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

PS_D double det3_lu(const double (&Ain)[3][3])
{
    // Eigen's dynamic-size determinant goes through PartialPivLU (kabschEst.cpp:53).
    double m[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m[i][j] = Ain[i][j];
    double det = 1.0;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        int piv = kx;
        double big = fabs(m[kx][kx]);
#pragma unroll
        for (int r = kx + 1; r < 3; ++r)
            if (fabs(m[r][kx]) > big) {
                big = fabs(m[r][kx]);
                piv = r;
            }
        if (big == 0.0) return 0.0;
#pragma unroll
        for (int r = kx + 1; r < 3; ++r)
            if (r == piv) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double t = m[kx][c];
                    m[kx][c] = m[r][c];
                    m[r][c] = t;
                }
                det = -det;
            }
        det *= m[kx][kx];
#pragma unroll
        for (int r = kx + 1; r < 3; ++r) {
            double f = m[r][kx] / m[kx][kx];
#pragma unroll
            for (int c = kx + 1; c < 3; ++c) m[r][c] -= f * m[kx][c];
        }
    }
    return det;
}

// KabschEst::computeTransformation (reference src/TransformEst/kabschEst.cpp:24-68), double precision.
// A, B: n x 3 column-major with leading dimension ld.  One wavefront.
__global__ __launch_bounds__(64) void ps_kabsch_f64_kernel(const double *__restrict__ A, const double *__restrict__ B,
                                                           int n, int ld, double *__restrict__ T)
{
    const int lane = threadIdx.x & 63;
    double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
    for (int i = lane; i < n; i += 64) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sa[c] += A[(size_t)c * ld + i];
            sb[c] += B[(size_t)c * ld + i];
        }
    }
    double cA[3], cB[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        cA[c] = wave_tree_sum_f64(sa[c]) / (double)n;
        cB[c] = wave_tree_sum_f64(sb[c]) / (double)n;
    }
    double acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = 0.0;
    for (int i = lane; i < n; i += 64) {
        double a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = A[(size_t)c * ld + i] - cA[c];
            b[c] = B[(size_t)c * ld + i] - cB[c];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[r][c] += a[r] * b[c];
    }
    double Am[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Am[r][c] = wave_tree_sum_f64(acc[r][c]);
    double V[3][3], W[3][3], S[3];
    jacobi_svd3<double>(Am, V, S, W); // V = svd.matrixU(), W = svd.matrixV()  (kabschEst.cpp:48-49)
    double det = det3_lu(Am);
    double dsg = (det != 0.0) ? det : 1.0;
    double d = (double)((dsg > 0.0) - (dsg < 0.0));
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = (W[i][0] * V[j][0] + W[i][1] * V[j][1]) + (W[i][2] * d) * V[j][2];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double ti = (R[i][0] * (-cA[0]) + (R[i][1] * (-cA[1]) + R[i][2] * (-cA[2]))) + cB[i];
#pragma unroll
                for (int j = 0; j < 3; ++j) T[4 * j + i] = R[i][j];
                T[12 + i] = ti;
            }
        }
    }
}

// Large point sets: the same arithmetic spread over G wavefronts (one 64-lane work-group each, strided like the
// single-wave kernel's lanes), partial sums combined in wave order by every wave / by the finishing wave.  The
// summation tree is a function of (n, G) only, so results are reproducible; versus the oracle's sequential sums the
// bound is the 1e-12 of the single-wave form.
//   pass 1: part[g][0..5]  = column sums of A, B over the wave's points
//   pass 2: part2[g][0..8] = sum (a - cA)(b - cB)^T over the wave's points, cA / cB from all part[] in order
//   finish: one wave adds part2[] in order, SVD, pose
__global__ __launch_bounds__(64) void ps_kabsch_f64_sums(const double *__restrict__ A, const double *__restrict__ B, int n,
                                                         int ld, double *__restrict__ part)
{
    const int lane = threadIdx.x & 63, g = blockIdx.x, G = gridDim.x;
    double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
    for (long long i = (long long)g * 64 + lane; i < n; i += (long long)G * 64) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sa[c] += A[(size_t)c * ld + i];
            sb[c] += B[(size_t)c * ld + i];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double va = wave_tree_sum_f64(sa[c]), vb = wave_tree_sum_f64(sb[c]);
        if (lane == 0) {
            part[(size_t)g * 6 + c] = va;
            part[(size_t)g * 6 + 3 + c] = vb;
        }
    }
}

PS_D void kabsch_means(const double *__restrict__ part, int G, int n, double (&cA)[3], double (&cB)[3])
{
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int g = 0; g < G; ++g) // every lane of every wave: the same order, the same values
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] += part[(size_t)g * 6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        cA[c] = s[c] / (double)n;
        cB[c] = s[3 + c] / (double)n;
    }
}

__global__ __launch_bounds__(64) void ps_kabsch_f64_cov(const double *__restrict__ A, const double *__restrict__ B, int n,
                                                        int ld, const double *__restrict__ part,
                                                        double *__restrict__ part2)
{
    const int lane = threadIdx.x & 63, g = blockIdx.x, G = gridDim.x;
    double cA[3], cB[3];
    kabsch_means(part, G, n, cA, cB);
    double acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = 0.0;
    for (long long i = (long long)g * 64 + lane; i < n; i += (long long)G * 64) {
        double a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = A[(size_t)c * ld + i] - cA[c];
            b[c] = B[(size_t)c * ld + i] - cB[c];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[r][c] += a[r] * b[c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = wave_tree_sum_f64(acc[r][c]);
            if (lane == 0) part2[(size_t)g * 9 + 3 * r + c] = v;
        }
}

__global__ __launch_bounds__(64) void ps_kabsch_f64_finish(const double *__restrict__ part, const double *__restrict__ part2,
                                                           int G, int n, double *__restrict__ T)
{
    const int lane = threadIdx.x & 63;
    double cA[3], cB[3];
    kabsch_means(part, G, n, cA, cB);
    double Am[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int g = 0; g < G; ++g) v += part2[(size_t)g * 9 + 3 * r + c];
            Am[r][c] = v;
        }
    double V[3][3], W[3][3], S[3];
    jacobi_svd3<double>(Am, V, S, W);
    double det = det3_lu(Am);
    double dsg = (det != 0.0) ? det : 1.0;
    double d = (double)((dsg > 0.0) - (dsg < 0.0));
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double R[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) R[j] = (W[i][0] * V[j][0] + W[i][1] * V[j][1]) + (W[i][2] * d) * V[j][2];
            double ti = (R[0] * (-cA[0]) + (R[1] * (-cA[1]) + R[2] * (-cA[2]))) + cB[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) T[4 * j + i] = R[j];
            T[12 + i] = ti;
        }
    }
}

// RGBD::keypoints2Dto3D / point2Dto3D / roundSize (reference src/RGBD/RGBD.cpp:10-16,30-65).
PS_D int round_size(double x, int size)
{
    if (x < 0)
        x = 0;
    else if (x > size - 1)
        x = size; // sic: the reference clamps to size, not size-1
    return (int)round(x);
}
__global__ __launch_bounds__(kBlock) void ps_backproject(const float *__restrict__ xy, int n,
                                                         const uint8_t *__restrict__ depth, int rows, int cols,
                                                         size_t step, float fx, float fy, float cx, float cy,
                                                         double scale, float *__restrict__ out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x = xy[2 * i], y = xy[2 * i + 1];
    int uR = round_size((double)x, cols), vR = round_size((double)y, rows);
    size_t off = (size_t)vR * step + (size_t)uR * 2;
    unsigned dv = 0;
    // cv::Mat::at is plain pointer arithmetic: u == cols reads the first pixel of the next row;
    // past the last row the reference reads out of bounds, here that is depth 0 (= missing).
    if (off + 2 <= (size_t)rows * step) dv = (unsigned)depth[off] | ((unsigned)depth[off + 1] << 8);
    float Z = (float)(((double)dv) / scale);
    float u = (x - cx) / fx;
    float v = (y - cy) / fy;
    out[3 * i] = u * Z;
    out[3 * i + 1] = v * Z;
    out[3 * i + 2] = Z;
}

// RGBD::removeImageDistortion (reference src/RGBD/RGBD.cpp:254-314): cv::undistortPoints with R = P = I
// (OpenCV 3.x cvUndistortPoints: 5 fixed-point iterations of the Brown model, double precision) and
// u = x_n * fx + cx in float.
struct DistArgs {
    double k[5]; // k1 k2 p1 p2 k3
    float fx, fy, cx, cy;
};
__global__ __launch_bounds__(kBlock) void ps_undistort_kernel(const float *__restrict__ xy, int n, DistArgs a,
                                                              float *__restrict__ out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double fx = (double)a.fx, fy = (double)a.fy, cx = (double)a.cx, cy = (double)a.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = a.k[0], k1 = a.k[1], k2 = a.k[2], k3 = a.k[3], k4 = a.k[4];
    const double z = 0.0; // k[5..11] of OpenCV's 12-coefficient form: absent in the reference's configs
    double x = (double)xy[2 * i], y = (double)xy[2 * i + 1];
    double x0 = x = (x - cx) * ifx;
    double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; ++j) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((z * r2 + z) * r2 + z) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + z * r2 + z * r2 * r2;
        double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + z * r2 + z * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    float xn = (float)x, yn = (float)y;
    out[2 * i] = xn * a.fx + a.cx;
    out[2 * i + 1] = yn * a.fy + a.cy;
}

__global__ __launch_bounds__(kBlock) void ps_project_kernel(const float *__restrict__ xyz, int n, float fx, float fy,
                                                            float cx, float cy, float *__restrict__ uv)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float u, v;
    project(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], fx, fy, cx, cy, u, v);
    uv[2 * i] = u;
    uv[2 * i + 1] = v;
}

// Diagnostic: counts inputs inside the division window for which div2_shared differs from the '/' operator.
__global__ __launch_bounds__(kBlock) void ps_fastdiv_check(uint64_t seed, int perThread, unsigned long long *mismatch,
                                                           unsigned long long *tested)
{
    const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long bad = 0, n = 0;
    for (int i = 0; i < perThread; ++i) {
        uint64_t r0 = mix64(seed ^ mix64(tid * 0x10001ull + (uint64_t)i));
        uint64_t r1 = mix64(r0);
        // random sign, exponent in [2^-40, 2^40), random mantissa
        auto mk = [](uint32_t bits) {
            uint32_t sign = bits & 0x80000000u;
            uint32_t ex = 87u + ((bits >> 23) & 0xFFu) % 80u;
            uint32_t v = sign | (ex << 23) | (bits & 0x007FFFFFu);
            float f;
            memcpy(&f, &v, 4);
            return f;
        };
        float a0 = mk((uint32_t)r0), a1 = mk((uint32_t)(r0 >> 32)), b = mk((uint32_t)r1);
        if (i & 1) { // the regime of the kernel: metres times focal length over depth
            a0 = a0 * 0.0f + (float)((int)((r0 >> 8) & 0xFFFF) - 32768) * 0.37f;
            b = 0.1f + (float)((r1 >> 8) & 0xFFFF) * 1e-4f;
        }
        if (!div_window_ok(a0, a1, b, a0, a1, b)) continue;
        float q0, q1;
        div2_shared(a0, a1, b, q0, q1);
        float e0 = a0 / b, e1 = a1 / b;
        uint32_t x0, y0, x1, y1;
        memcpy(&x0, &q0, 4); memcpy(&y0, &e0, 4); memcpy(&x1, &q1, 4); memcpy(&y1, &e1, 4);
        bad += (x0 != y0) + (x1 != y1);
        n += 2;
    }
    atomicAdd(mismatch, bad);
    atomicAdd(tested, n);
}

// Diagnostic (ps_debug_mathcheck): the exact short forms of ps_device_math.h against the operators, bit for bit.
//   mode 0  sqrt_ge1(x) vs sqrtf(x) for EVERY float pattern from 1.0f up to +inf and the NaN patterns just above it
//           (element e of the sweep = pattern 0x3F800000 + e; 0x40001000 elements cover them)
//   mode 1  rcp_exact_in_window(x) vs 1.0f / x for every float of [1, 2] (0x00800001 elements)
//   mode 2  unit_sign(y) vs y / |y| for random y of every exponent (+-inf and NaN included; NaN == NaN here)
//   mode 3  JacobiSVD's pair 1 / d, u / d with d = sqrt(1 + u * u): div_with on random u inside the window
//   mode 4  nine numerators over one denominator (scale_down3 / inverse_rigid_general): div_with on random operands of
//           the window [2^-40, 2^40]
__global__ __launch_bounds__(kBlock) void ps_mathcheck(int mode, uint64_t seed, int perThread, unsigned long long total,
                                                       unsigned long long *mismatch, unsigned long long *tested)
{
    const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, n = 0;
    auto bits = [](float f) { return __builtin_bit_cast(uint32_t, f); };
    auto same = [&](float a, float b) { return bits(a) == bits(b) || (a != a && b != b); };
    auto in_window = [](uint32_t r) { // random sign, exponent in [2^-40, 2^40), random mantissa
        const uint32_t ex = 87u + ((r >> 23) & 0xFFu) % 80u;
        return __builtin_bit_cast(float, (r & 0x80000000u) | (ex << 23) | (r & 0x007FFFFFu));
    };
    for (int i = 0; i < perThread; ++i) {
        const uint64_t e = (uint64_t)i * nthreads + tid; // (consecutive threads take consecutive elements)
        if (mode <= 1 && e >= total) break;
        const uint64_t r0 = mix64(seed ^ mix64(e));
        const uint64_t r1 = mix64(r0);
        if (mode == 0) {
            const float x = __builtin_bit_cast(float, (uint32_t)(0x3F800000ull + e));
            bad += same(sqrt_ge1(x), sqrtf(x)) ? 0 : 1;
            n += 1;
        } else if (mode == 1) {
            const float x = __builtin_bit_cast(float, (uint32_t)(0x3F800000ull + e));
            bad += same(rcp_exact_in_window(x), 1.0f / x) ? 0 : 1;
            n += 1;
        } else if (mode == 2) {
            float y = __builtin_bit_cast(float, (uint32_t)r0);
            if (!(fabsf(y) >= 5.877471754111438e-39f) && y == y) y = 1.0f; // (the caller's guard: 2 |y| >= FLT_MIN)
            bad += same(unit_sign(y), y / fabsf(y)) ? 0 : 1;
            n += 1;
        } else if (mode == 3) {
            float u = in_window((uint32_t)r0);
            if (i & 1) u = (float)((int)((r0 >> 40) & 0xFFFF) - 32768) * 1.0e-3f + 1.0e-6f; // the regime of the SVD: |u| ~ 0 .. 30
            const float d = sqrt_ge1(u * u + 1.0f);
            if (!(fabsf(u) >= kExactDivLo && d <= kExactDivHi)) continue;
            const float q = rcp_refined(d);
            bad += (same(div_with(1.0f, d, q), 1.0f / d) ? 0 : 1) + (same(div_with(u, d, q), u / d) ? 0 : 1);
            n += 2;
        } else {
            const float b = in_window((uint32_t)r1);
            const float q = rcp_refined(b);
            uint64_t r = r0;
#pragma unroll 1
            for (int k = 0; k < 9; ++k) {
                const float a = in_window((uint32_t)r);
                bad += same(div_with(a, b, q), a / b) ? 0 : 1;
                r = mix64(r);
            }
            n += 9;
        }
    }
    atomicAdd(mismatch, bad);
    atomicAdd(tested, n);
}

// Diagnostic: tabulates the device-side trip limits so tests can compare them with the direct
// libm evaluation (RANSAC.cpp:457-461, USAC.h:944-971) for every (count, M).
// (the limit functions search their table as a wavefront with uniform arguments, as the replays call them: the wave takes
// its 64 counts one after the other)
// ------------------------------------------------------------------------------------------
// Host <-> device transfers as a kernel over mapped pinned host memory (the streaming entry points: ps_vo_stream_push's frame in /
// results out, the pipelined form's meta block).
constexpr int kCopySegs = 5;
struct CopySegs {
    const void *src[kCopySegs];
    void *dst[kCopySegs];
    unsigned long long bytes[kCopySegs]; // multiples of 4
    int n;
};

// Host <-> device transfer as a kernel: every segment is swept grid-stride, 16 bytes per lane where source, destination and
// length allow it, 4 bytes otherwise.  One side of every segment is mapped pinned host memory: the accesses go over the link.
__global__ void __launch_bounds__(256) ps_copy_segments(CopySegs cs)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < cs.n; ++k) {
        const size_t bytes = cs.bytes[k];
        if ((((size_t)cs.src[k] | (size_t)cs.dst[k] | bytes) & 15) == 0) {
            const uint4 *__restrict__ s = (const uint4 *)cs.src[k];
            uint4 *__restrict__ d = (uint4 *)cs.dst[k];
            for (size_t i = t0; i < bytes / 16; i += stride) d[i] = s[i];
        } else {
            const uint32_t *__restrict__ s = (const uint32_t *)cs.src[k];
            uint32_t *__restrict__ d = (uint32_t *)cs.dst[k];
            for (size_t i = t0; i < bytes / 4; i += stride) d[i] = s[i];
        }
    }
}


// Small chunks of the pipelined stream (chunkFrames <= 4: ps_stream_async.h, "mini" chunks): everything that changes from chunk to
// chunk travels as DATA through this block, so that a place's whole chunk -- frames in, kernels 1 - 4, results out -- is one
// captured hipGraph.  The host fills it in the place's pinned meta block; ps_mini_copy_in copies it to the device meta block,
// which the kernels behind it read (row counts, pair list, seed).
constexpr int kMiniFrames = 4;
struct MiniMeta {
    int32_t pairs[2 * kMiniFrames]; // the chunk's pairs in the place's private frame set: slot 0 = the previous frame (halo), 1 .. n = the chunk's
    int32_t nk[kMiniFrames + 1];    // row counts of the private frames
    int32_t n, first, pad;          // frames in the chunk; 1 = its first frame has no predecessor (slot 0 is not copied)
    unsigned long long seed;        // hypothesis seed of the chunk's first pair (cfg->seed + its number in the stream)
    const uint8_t *src[kMiniFrames + 1]; // device views of the pinned packed frames [cap x 32 B][cap x 12 B]: [0] the halo, [1 .. n] the chunk's
};

// Frames of a mini chunk from pinned host memory into the place's private frame set (packed layout, `stride` bytes per frame),
// and the meta block with them.  grid = (kMiniFrames + 1) x groups work-groups: frame f is swept by `groups` of them.  The reads go
// over the link: 88 KB per 2000-keypoint frame, a few microseconds (the synchronous ps_vo_stream_push moves its frame the same way).
__global__ void __launch_bounds__(256) ps_mini_copy_in(const MiniMeta *__restrict__ hm, MiniMeta *__restrict__ dm, uint8_t *__restrict__ frames,
                                                       int cap, unsigned long long stride, int groups)
{
    if (blockIdx.x == 0 && threadIdx.x < sizeof(MiniMeta) / 4)
        reinterpret_cast<uint32_t *>(dm)[threadIdx.x] = reinterpret_cast<const uint32_t *>(hm)[threadIdx.x];
    const int f = (int)(blockIdx.x / (unsigned)groups), g = (int)(blockIdx.x % (unsigned)groups);
    const int n = hm->n, first = hm->first;
    if (f > n || (f == 0 && first)) return;
    const int rows = hm->nk[f];
    // (the frame's address comes out of memory: told to be a global address, or the compiler emits FLAT loads for it)
    typedef unsigned __attribute__((ext_vector_type(4))) u4v_t;
    typedef const u4v_t __attribute__((address_space(1))) *gu4_t;
    typedef const uint32_t __attribute__((address_space(1))) *gu32_t;
    const uintptr_t src = reinterpret_cast<uintptr_t>(hm->src[f]);
    uint8_t *__restrict__ dst = frames + (size_t)f * stride;
    const size_t t0 = (size_t)g * blockDim.x + threadIdx.x, step = (size_t)groups * blockDim.x;
    { // descriptors: rows x 32 bytes, 16 per lane
        gu4_t sv = (gu4_t)src;
        u4v_t *__restrict__ dv = reinterpret_cast<u4v_t *>(dst);
        for (size_t i = t0; i < (size_t)rows * 2; i += step) dv[i] = sv[i];
    }
    { // points: rows x 12 bytes behind the cap x 32 descriptor bytes, 4 per lane
        gu32_t sv = (gu32_t)(src + (size_t)cap * 32);
        uint32_t *__restrict__ dv = reinterpret_cast<uint32_t *>(dst + (size_t)cap * 32);
        for (size_t i = t0; i < (size_t)rows * 3; i += step) dv[i] = sv[i];
    }
}

// Results of a chunk of the pipelined stream written straight into the lane's mapped pinned block, only what the caller asked
// for (PsStreamResults): mode 1 = the INLIER matches of every pair in input order (what Matcher::match hands back,
// matcher.cpp:452-516: `inlierMatches`) + pose + stats + match count; mode 2 = pose + stats + match count.  One work-group per
// pair; ordered compaction by one ballot-prefix scan per 256 matches.  The writes go over the host link: 12 KB instead of 34 KB
// per 2000-keypoint pair in mode 1, 108 bytes in mode 2.
__global__ __launch_bounds__(kBlock) void ps_pack_results_to_host(const PsDMatch *__restrict__ matches,
                                                                  const int32_t *__restrict__ numMatches,
                                                                  const uint8_t *__restrict__ mask, const float *__restrict__ pose,
                                                                  const PsRansacStats *__restrict__ stats, int cap, int mode,
                                                                  PsDMatch *__restrict__ hMatches, float *__restrict__ hPose,
                                                                  PsRansacStats *__restrict__ hStats, int32_t *__restrict__ hNum)
{
    __shared__ int s_w[2 * (kBlock / 64)];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = numMatches[p];
    if (tid < 16) hPose[(size_t)p * 16 + tid] = pose[(size_t)p * 16 + tid];
    if (tid == 0) {
        hStats[p] = stats[p];
        hNum[p] = n;
    }
    if (mode != 1) return;
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(matches + (size_t)p * cap);
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(hMatches + (size_t)p * cap);
    const uint8_t *__restrict__ mk = mask + (size_t)p * cap;
    int base = 0, trip = 0;
    for (int i0 = 0; i0 < n; i0 += kBlock, ++trip) {
        const int i = i0 + tid;
        const bool in = i < n && mk[i] != 0;
        int pos, total, posB, totalB;
        block_scan_flags2_alt<kBlock>(in, false, pos, total, posB, totalB, s_w, trip);
        if (in) dst[base + pos] = src[i];
        base += total;
    }
}

// Diagnostic: words of the keys block that are not kNoKey (the matcher's atomicMin merge relies on an all-ones block at rest).
__global__ void ps_count_not_ones(const uint32_t *__restrict__ keys, size_t n, unsigned long long *__restrict__ bad)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += keys[i] != kNoKey;
    if (c) atomicAdd(bad, c);
}

__global__ void ps_limits_table(SelectArgs a, int M, int32_t *__restrict__ out)
{
    const int c0 = (int)(blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) + 1; // the wave's first count
    const int lane = threadIdx.x & 63;
    int mine = 0;
    for (int j = 0; j < 64; ++j) {
        const int c = c0 + j;
        if (c > M) break; // (uniform)
        const int lim = (a.estimator == PS_EST_USAC) ? usac_limit(a, (unsigned)c, (unsigned)M)
                                                     : ransac_limit(a, (float)c / (float)M);
        if (lane == j) mine = lim;
    }
    if (c0 + lane <= M) out[c0 + lane - 1] = mine;
}

} // namespace psdev
