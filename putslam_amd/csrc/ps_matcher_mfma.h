// ps_matcher_mfma.h -- kernel 1 of the path on the matrix cores of gfx950.
//
// cv::batchDistance(train, query, K=1) inside BFMatcher's cross-check (reference call site
// src/Matcher/matcherOpenCV.cpp:198-206): for each train row the nearest query row under the 256-bit
// Hamming distance, ties to the lowest query index.
//
// The N x N x 256 popcount sweep is an exact small-integer contraction: with every descriptor bit b
// written as s = 4 - 8b (+4 / -4), sum_k s_q[k] * s_t[k] = 16 * (256 - 2 * hamming).  +-4 are FP4 (e2m1) values, so
// v_mfma_f32_32x32x64_f8f6f4 with FP4 operands (the densest matrix format of CDNA4: K = 64 per 32-cycle
// instruction) computes a 32 x 32 tile of distances in 4 instructions; every partial sum is an integer below
// 2^13, hence exact in the f32 accumulator whatever the internal summation order.  The accumulator is
// initialised with 2^23 + 4127 - (row of the tile), so each entry ends as
//       2^23 + D,   D = 8223 - 32 * hamming - row = 32 * (256 - hamming) + (31 - row)   in [0, 8223]
// Every value (and every partial sum) lies in [2^23, 2^24), where f32 has unit spacing: the mantissa field of
// the result IS the integer D and the bit patterns order like integers.  Within a tile the nearest query, lowest
// row first, is the MAXIMUM pattern: v_max3_i32 over the 16 accumulator registers a lane holds (its column = its
// train descriptor).  Across tiles (ascending query index) a later tile wins only with a strictly smaller
// distance, i.e. when its maximum exceeds the running best with the row field saturated (best | 31): one v_or,
// one compare, two selects per tile.  Query rows beyond the frame start from -1e30 (a negative pattern).
//
// Operand traffic: the previous frame (query side) of every pair is expanded once per call into the
// fragment-major FP4 image  Xq[pair][tile of 32 rows][k-step 0..3][lane 0..63][16 B]  by ps_expand_query_fp4
// (N work, not N^2); every wave expands its own TT train tiles in registers once and keeps them as B operands for
// the whole sweep, and streams the query tiles as A operands: four contiguous 1-KiB loads per tile, one tile ahead,
// straight into registers (the four waves of a work-group read the same tiles within a few hundred cycles: L1 / L2
// hits).  The first form staged the query tiles through LDS (double-buffered, one barrier per tile, shared by the
// four waves): 0.211 ms against 0.186 ms per 499 pairs -- the barrier tied the waves of a work-group together, and
// with one wave of each of two work-groups per SIMD the matrix and the vector phases of all of them coincided
// (that form left the tree in round 4; the A/B is on file in profiles/r02h).
// Any consistent assignment of descriptor bits to (k-step, lane half, element) is valid because both
// operands use the same one: lane (r = lane & 31, h = lane >> 5) holds, for k-step s, the 32 bits of dword
// 2s + h of row r.
//
// The integer VALU kernel ps_hamming_nn (ps_kernels.h) is kept as the tested twin
// (PUTSLAM_HIP_MATCHER=valu / ps_context_set_option("matcher", 0)).
#pragma once

#include "ps_kernels.h"

namespace psdev {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 32;            // rows of one MFMA tile
constexpr int kTileU4 = 256;             // uint4 (16 B) pieces of one expanded tile: 4 k-steps x 64 lanes = 4 KiB
constexpr int kWavesPerWG = kBlock / 64; // 4

// 8 descriptor bits -> 8 FP4 nibbles (bit i -> nibble i): 0 -> 0x6 = +4.0, 1 -> 0xE = -4.0
PS_D uint32_t fp4_from_byte(uint32_t x)
{
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 3) | 0x66666666u;
}
PS_D v4i_t fp4_from_dword(uint32_t w)
{
    v4i_t r;
    r.x = (int)fp4_from_byte(w & 0xFFu);
    r.y = (int)fp4_from_byte((w >> 8) & 0xFFu);
    r.z = (int)fp4_from_byte((w >> 16) & 0xFFu);
    r.w = (int)fp4_from_byte(w >> 24);
    return r;
}

// row of the 32 x 32 tile held in accumulator register `reg` by lane half h (C/D layout of every
// 32x32 MFMA of gfx950: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
PS_D int tile_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------
// Query-side expansion: one work-group per (pair, chunk of tiles).  Thread o of a tile writes piece
// o = s * 64 + h * 32 + r (the layout the MFMA kernel reads back linearly), i.e. dword 2s + h of row r.
// Rows beyond the frame's count are written as zeros (0.0 in FP4); the MFMA kernel masks them anyway.
// ------------------------------------------------------------------------------------------
// (fstride in all three kernels: dwords between consecutive frames' descriptor blocks -- cap * 8 for the dense frame set)
__global__ __launch_bounds__(kBlock) void ps_expand_query_fp4(const uint32_t *__restrict__ desc,
                                                              const int32_t *__restrict__ nkpts,
                                                              const int32_t *__restrict__ pairs, int cap, int fstride, int tpf,
                                                              int chunks, uint4 *__restrict__ Xq,
                                                              uint32_t *__restrict__ keysInit)
{
    // grid = chunks x P work-groups in one dimension (no 65535 limit on the number of pairs)
    const int p = (int)(blockIdx.x / (unsigned)chunks), chunk = (int)(blockIdx.x % (unsigned)chunks);
    if (keysInit) // the matcher merges query splits with atomicMin: start every key at "no query" (saves a memset launch)
        for (int t = chunk * kBlock + threadIdx.x; t < cap; t += chunks * kBlock) keysInit[(size_t)p * cap + t] = kNoKey;
    const int fq = pairs[2 * p];
    const int nq = nkpts[fq];
    const uint32_t *__restrict__ q32 = desc + (size_t)fq * fstride;
    uint4 *__restrict__ out = Xq + (size_t)p * tpf * kTileU4;
    const int o = threadIdx.x, s = o >> 6, h = (o >> 5) & 1, r = o & 31;
    for (int tile = chunk; tile < tpf; tile += chunks) {
        const int row = tile * kTileRows + r;
        v4i_t e = {0, 0, 0, 0};
        if (row < nq) e = fp4_from_dword(q32[(size_t)row * 8 + 2 * s + h]);
        out[(size_t)tile * kTileU4 + o] = make_uint4((uint32_t)e.x, (uint32_t)e.y, (uint32_t)e.z, (uint32_t)e.w);
    }
}

PS_D v16f_t mfma_fp4(const v4i_t &a, const v4i_t &b, const v16f_t &c)
{
    // FP4 operands occupy the first 4 of the 8 operand registers (cbsz = blgp = 4 selects e2m1); scale
    // operands 0 with op_sel 0 select the unscaled form (scale 2^0 on both sides)
    const v8i_t a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
    const v8i_t b8 = __builtin_shufflevector(b, b, 0, 1, 2, 3, -1, -1, -1, -1);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
}

// The same instruction in its block-scaled form with the uniform scales 2^9 (A side) and 2^0 (B side): E8M0 bytes 136 and
// 127 in every byte of the scale operands, so that the result does not depend on which byte a lane's block takes.
PS_D v16f_t mfma_fp4s(const v4i_t &a, const v4i_t &b, const v16f_t &c)
{
    const v8i_t a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
    const v8i_t b8 = __builtin_shufflevector(b, b, 0, 1, 2, 3, -1, -1, -1, -1);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, (int)0x88888888u, 0, (int)0x7F7F7F7Fu);
}

PS_D int fbits(float f) { return __builtin_bit_cast(int, f); }
PS_D int imax3(int a, int b, int c) { return max(max(a, b), c); }
// maximum of the 16 accumulator entries as integer bit patterns (valid entries are >= +0.0f)
PS_D int max16(const v16f_t &d)
{
    int m = imax3(fbits(d[0]), fbits(d[1]), fbits(d[2]));
    m = imax3(m, fbits(d[3]), fbits(d[4]));
    m = imax3(m, fbits(d[5]), fbits(d[6]));
    m = imax3(m, fbits(d[7]), fbits(d[8]));
    m = imax3(m, fbits(d[9]), fbits(d[10]));
    m = imax3(m, fbits(d[11]), fbits(d[12]));
    m = imax3(m, fbits(d[13]), fbits(d[14]));
    return max(m, fbits(d[15]));
}

constexpr float kMfmaNoRow = -1.0e30f; // accumulator start of a query row beyond the frame: never the maximum
constexpr int kMfmaBias = 8223;        // 32 * 256 + 31
constexpr float kMfmaBase = 8388608.0f; // 2^23: unit spacing up to 2^24

// ------------------------------------------------------------------------------------------
// grid = groups * qsplit * P work-groups in XCD-aware order; a work-group owns 4 * TT train tiles (one
// wave each TT of them) and sweeps the query tiles [T0, T1) of its split.  qsplit > 1 merges with atomicMin
// on the packed key (hamming << 16 | query), the same key kernel 2 reads from ps_hamming_nn.
// ------------------------------------------------------------------------------------------
// Every wave loads its own query tiles straight into registers (the first form staged them through LDS with a barrier
// per tile, shared by the work-group's four waves: 14 % slower, profiles/r02h).
#ifndef PS_MFMA_WAVES
#define PS_MFMA_WAVES 1
#endif
template <int TT>
__global__ __launch_bounds__(kBlock, PS_MFMA_WAVES) void ps_hamming_mfma(const uint32_t *__restrict__ desc,
                                                          const int32_t *__restrict__ nkpts,
                                                          const int32_t *__restrict__ pairs, int cap, int fstride, int tpf,
                                                          int groups, int qsplit, const uint4 *__restrict__ Xq,
                                                          uint32_t *__restrict__ keys)
{
    const unsigned perPair = (unsigned)(groups * qsplit);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int p = (int)(L / perPair);
    const int inner = (int)(L - (unsigned)p * perPair);
    const int g = inner / qsplit, qs = inner - g * qsplit;
    const int fq = pairs[2 * p], ft = pairs[2 * p + 1]; // query = previous frame, train = current
    const int nq = nkpts[fq], nt = nkpts[ft];
    if (g * (kWavesPerWG * TT * kTileRows) >= nt) return; // whole work-group beyond the train rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile0 = (g * kWavesPerWG + wave) * TT;
    const int nqTiles = (nq + kTileRows - 1) / kTileRows;
    const int T0 = (int)(((long long)nqTiles * qs) / qsplit), T1 = (int)(((long long)nqTiles * (qs + 1)) / qsplit);

    // B operands: this wave's train tiles, expanded in registers (rows beyond nt repeat the last row; never stored)
    const uint32_t *__restrict__ t32 = desc + (size_t)ft * fstride;
    v4i_t B[TT][4];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        int t = (tile0 + i) * kTileRows + r;
        t = t < nt ? t : nt - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) B[i][s] = fp4_from_dword(t32[(size_t)t * 8 + 2 * s + h]);
    }
    v16f_t C;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) C[reg] = kMfmaBase + (float)(kMfmaBias - 4096 - tile_row(reg, h));
    int best[TT], bestT[TT];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        best[i] = 0; // below every valid entry (their patterns are >= 0x4B000000)
        bestT[i] = 0;
    }

    const uint4 *__restrict__ xq = Xq + (size_t)p * tpf * kTileU4;
    // Every wave fetches its own copy of the query tile (4 x 16 B per lane, served by L1 / L2 for the other waves of the
    // work-group) one tile ahead into registers: no LDS staging, no barrier, the waves of a work-group drift apart freely.
    uint4 an[4];
    if (T0 < T1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) an[s] = xq[(size_t)T0 * kTileU4 + s * 64 + lane];
    }
    for (int T = T0; T < T1; ++T) {
        v4i_t A[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            A[s].x = (int)an[s].x; A[s].y = (int)an[s].y; A[s].z = (int)an[s].z; A[s].w = (int)an[s].w;
        }
        if (T + 1 < T1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) an[s] = xq[(size_t)(T + 1) * kTileU4 + s * 64 + lane];
        }
        v16f_t Cin = C;
        if (T * kTileRows + kTileRows > nq) { // last, partial query tile: rows beyond nq can never win
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (T * kTileRows + tile_row(reg, h) >= nq) Cin[reg] = kMfmaNoRow;
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            v16f_t acc = mfma_fp4(A[0], B[i][0], Cin);
            acc = mfma_fp4(A[1], B[i][1], acc);
            acc = mfma_fp4(A[2], B[i][2], acc);
            acc = mfma_fp4(A[3], B[i][3], acc);
            const int m = max16(acc);
            if (m > (best[i] | 31)) { // strictly smaller distance only: the earlier (lower) query tile keeps a tie
                best[i] = m;
                bestT[i] = T;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int t = (tile0 + i) * kTileRows + r;
        uint32_t key = kNoKey;
        if (best[i] > 0) {
            const int v = kMfmaBias - (best[i] & 0x7FFFFF); // 32 * hamming + row of the tile
            key = ((uint32_t)(v >> 5) << 16) | (uint32_t)(bestT[i] * kTileRows + (v & 31));
        }
        const uint32_t other = (uint32_t)__shfl_xor((int)key, 32, 64); // the other half's 16 rows of every tile
        key = other < key ? other : key;
        if (h == 0 && t < nt) {
            if (qsplit == 1)
                keys[(size_t)p * cap + t] = key;
            else
                atomicMin(&keys[(size_t)p * cap + t], key);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused form (default since round 3): no FP4 image in HBM and no expansion launch.  The work-group expands the query
// tiles itself, kFuseChunk tiles at a time, into a double-buffered LDS image in the same fragment-major layout
// (thread o of the group writes piece o of every tile of the chunk: one descriptor dword -> 32 nibbles -> one
// ds_write_b128), and all four waves read their A operands from it (four ds_read_b128 per tile, conflict-free).  One barrier
// per chunk -- the first LDS form of this kernel had one per tile and lost 14 % to it (profiles/r02h); the expansion of
// chunk c + 1 is issued before the MFMAs of chunk c, so its vector work sits in their shadow.  Every work-group of a
// pair expands the whole query range it sweeps (28 vector instructions per tile and wave): the four groups of a
// 2000-row frame repeat each other's expansion, which costs about what the separate launch did (0.036 ms per 499 pairs)
// and removes 0.29 GB of traffic per step and one launch.
// ------------------------------------------------------------------------------------------
// The MFMA runs in its block-scaled form with an add/max epilogue (10 instead of 13 vector instructions per 32 x 32 tile);
// the 8 bits -> 8 nibbles expansion goes through a 256-entry table in LDS (2 vector instructions + one LDS read per byte
// instead of 7).  (The unscaled / table-free sub-forms measured in profiles/r03h left the tree in round 4.)  On gfx950 MFMA and vector instructions of a SIMD do not overlap -- not across waves and, as the
// software-pipelined epilogue tried here showed (profiles/r03h: 0.232 ms either way), not inside one wave either -- so this
// kernel's time is its 512 matrix cycles PLUS its ~90 vector instructions per query tile and wave, and only removing
// vector instructions shortens it.
constexpr int kFuseChunk = 4; // query tiles per chunk: 2 x 4 x 4 KiB = 32 KiB of LDS per work-group

// (Round 4 tried eight wavefronts per work-group -- twice the train rows per work-group, half as many work-groups expanding
// every query tile, 145 instead of 165 VGPRs: 0.213 against 0.190 ms per 499 pairs, profiles/r04b/ab_short_forms_and_options.txt
// row "w8"; the patch is profiles/variants/matcher_8_waves.patch.txt.)
template <int TT>
__global__ __launch_bounds__(kBlock, PS_MFMA_WAVES) void ps_hamming_mfma_fused(const uint32_t *__restrict__ desc,
                                                                const int32_t *__restrict__ nkpts,
                                                                const int32_t *__restrict__ pairs, int cap, int fstride, int tpf,
                                                                int groups, int qsplit, uint32_t *__restrict__ keys)
{
    __shared__ uint4 s_a[2][kFuseChunk][kTileU4];
    __shared__ uint32_t s_lut[256]; // byte -> its 8 FP4 nibbles
    s_lut[threadIdx.x] = fp4_from_byte(threadIdx.x);
    const unsigned perPair = (unsigned)(groups * qsplit);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int p = (int)(L / perPair);
    const int inner = (int)(L - (unsigned)p * perPair);
    const int g = inner / qsplit, qs = inner - g * qsplit;
    const int fq = pairs[2 * p], ft = pairs[2 * p + 1]; // query = previous frame, train = current
    const int nq = nkpts[fq], nt = nkpts[ft];
    if (g * (kWavesPerWG * TT * kTileRows) >= nt) return; // whole work-group beyond the train rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile0 = (g * kWavesPerWG + wave) * TT;
    const int nqTiles = (nq + kTileRows - 1) / kTileRows;
    const int T0 = (int)(((long long)nqTiles * qs) / qsplit), T1 = (int)(((long long)nqTiles * (qs + 1)) / qsplit);

    // B operands: this wave's train tiles, expanded in registers (rows beyond nt repeat the last row; never stored)
    const uint32_t *__restrict__ t32 = desc + (size_t)ft * fstride;
    v4i_t B[TT][4];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        int t = (tile0 + i) * kTileRows + r;
        t = t < nt ? t : nt - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) B[i][s] = fp4_from_dword(t32[(size_t)t * 8 + 2 * s + h]);
    }
    v16f_t C;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) C[reg] = kMfmaBase + (float)((1 << 21) + 31 - tile_row(reg, h));
    int best[TT];
#pragma unroll
    for (int i = 0; i < TT; ++i) best[i] = 0; // below every valid entry (their patterns are >= 0x4B000000)

    // expansion role of this thread: piece o = s * 64 + h * 32 + r of a tile = dword 2 s + h of row r
    const uint32_t *__restrict__ q32 = desc + (size_t)fq * fstride;
    const int es = tid >> 6, eh = (tid >> 5) & 1, er = tid & 31;
    // The last query tile of a frame may be partial (rows beyond nq must never win: their accumulators start from -1e30).
    // It is taken out of the main loop: left inside, the compiler turns the rare masking into 16 compares + 16 selects for
    // EVERY tile (50 of the 126 vector instructions per tile this kernel had, profiles/r03h).
    const int lastT = (nq & (kTileRows - 1)) ? nqTiles - 1 : -1;
    const bool hasPartial = lastT >= T0 && lastT < T1;
    const int Tm = hasPartial ? lastT : T1; // main loop: whole tiles [T0, Tm)
    auto expand_tile = [&](uint4 *dst, int T) {
        const int row = T * kTileRows + er;
        v4i_t e = {0, 0, 0, 0}; // rows beyond the frame: 0.0 in FP4 (masked by the accumulator start anyway)
        if (row < nq) {
            const uint32_t w = q32[(size_t)row * 8 + 2 * es + eh];
            e.x = (int)s_lut[w & 0xFFu];
            e.y = (int)s_lut[(w >> 8) & 0xFFu];
            e.z = (int)s_lut[(w >> 16) & 0xFFu];
            e.w = (int)s_lut[w >> 24];
        }
        dst[tid] = make_uint4((uint32_t)e.x, (uint32_t)e.y, (uint32_t)e.z, (uint32_t)e.w);
    };
    auto expand_chunk = [&](int buf, int Tc) {
#pragma unroll
        for (int j = 0; j < kFuseChunk; ++j)
            if (Tc + j < Tm) expand_tile(s_a[buf][j], Tc + j);
    };
    // one query tile against this wave's TT train tiles
    auto tile_body = [&](int T, const uint4 *__restrict__ tile, const v16f_t &Cin) {
        v4i_t A[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 a = tile[s * 64 + lane];
            A[s].x = (int)a.x; A[s].y = (int)a.y; A[s].z = (int)a.z; A[s].w = (int)a.w;
        }
        // block-scaled MFMA: every product carries 2^9, the accumulator's mantissa is 2^14 (256 - hamming) + (31 - row);
        // adding the tile's 16352 - 32 T turns it into 2^14 (256 - hamming) + (16383 - query row), whose maximum over rows
        // AND tiles is the nearest query, lowest index first: 8 v_max3 + add + max per 32 x 32 tile
        const int cT = 16352 - 32 * T;
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            v16f_t acc = mfma_fp4s(A[0], B[i][0], Cin);
            acc = mfma_fp4s(A[1], B[i][1], acc);
            acc = mfma_fp4s(A[2], B[i][2], acc);
            acc = mfma_fp4s(A[3], B[i][3], acc);
            best[i] = max(best[i], max16(acc) + cT);
        }
    };
    __syncthreads(); // the table is complete
    if (T0 < Tm) expand_chunk(0, T0);
    __syncthreads();
    int c = 0;
    for (int Tc = T0; Tc < Tm; Tc += kFuseChunk, ++c) {
        if (Tc + kFuseChunk < Tm) expand_chunk((c + 1) & 1, Tc + kFuseChunk);
#pragma unroll 1
        for (int j = 0; j < kFuseChunk; ++j) {
            if (Tc + j >= Tm) break;
            tile_body(Tc + j, s_a[c & 1][j], C);
        }
        __syncthreads(); // chunk c + 1 is complete, chunk c may be overwritten
    }
    if (hasPartial) { // (work-group uniform)
        expand_tile(s_a[0][0], lastT);
        __syncthreads();
        v16f_t Cin = C;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            if (lastT * kTileRows + tile_row(reg, h) >= nq) Cin[reg] = kMfmaNoRow;
        tile_body(lastT, s_a[0][0], Cin);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int t = (tile0 + i) * kTileRows + r;
        uint32_t key = kNoKey;
        if (best[i] > 0) {
            const int d = best[i] & 0x7FFFFF; // 2^14 (256 - hamming) + (16383 - query row)
            key = ((uint32_t)(256 - (d >> 14)) << 16) | (uint32_t)(16383 - (d & 0x3FFF));
        }
        const uint32_t other = (uint32_t)__shfl_xor((int)key, 32, 64); // the other half's 16 rows of every tile
        key = other < key ? other : key;
        if (h == 0 && t < nt) {
            if (qsplit == 1)
                keys[(size_t)p * cap + t] = key;
            else
                atomicMin(&keys[(size_t)p * cap + t], key);
        }
    }
}

} // namespace psdev
