// mfma_valu_coissue.hip -- on one gfx950 SIMD, does a wave's v_mfma_f32_32x32x16_f16 stream overlap with another wave's
// (or its own) v_pk_fma_f32 stream?  512-thread work-groups put two waves on every SIMD: waves 0-3 run role A, waves
// 4-7 role B.  Roles: 0 idle, 1 MFMA only, 2 VALU only, 3 MFMA and VALU interleaved in one wave (5 MFMA : 48 pk_fma).
// Accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form, as the library is built) or in AGPRs (default).
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] -o mfma_valu_coissue mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000;

__global__ __launch_bounds__(512) void k(float *out, int roleA, int roleB, float seed)
{
    const int role = (threadIdx.x >> 6) < 4 ? roleA : roleB;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(seed * (threadIdx.x + i));
        b[i] = (_Float16)(1.0f + seed * i);
    }
    f16v c0, c1, c2, c3, c4;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = c4[r] = seed * r;
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f2{1.0f + seed * i, 1.0f - seed * i};
    const f2 m = {1.0000001f, 0.9999999f}, d = {seed, -seed};
    if (role == 1) {
        for (int it = 0; it < ITERS; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c4, 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
        }
    } else if (role == 4) { // the same arithmetic unpacked: 96 v_fma_f32
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(m.x), "v"(d.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].y) : "v"(m.y), "v"(d.y));
                }
        }
    } else if (role == 5) { // 5 MFMA + 96 v_fma_f32 interleaved in one wave
        for (int it = 0; it < ITERS; ++it) {
#define FMA16 _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(m.x), "v"(d.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].y) : "v"(m.y), "v"(d.y)); }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            FMA16
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            FMA16
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            FMA16
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            FMA16
            c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c4, 0, 0, 0);
            FMA16 FMA16
        }
    } else if (role == 6) { // integer / logic class: 96 v_and_b32 + v_alignbit mixed
        unsigned u[8];
        for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(0x7fffffffu ^ (unsigned)it));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                }
        }
        for (int i = 0; i < 8; ++i) v[i].x += (float)u[i];
    } else if (role == 3) {
        for (int it = 0; it < ITERS; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
            c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c4, 0, 0, 0);
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i] + c4[i];
    for (int i = 0; i < 8; ++i) r += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main()
{
    float *out;
    CHK(hipMalloc(&out, 256 * 512 * 4));
    const int roles[][2] = {{1, 0}, {2, 0}, {1, 1}, {2, 2}, {1, 2}, {3, 0}, {3, 3}, {4, 0}, {4, 4}, {1, 4}, {5, 0}, {5, 5}, {6, 0}, {6, 6}, {1, 6}};
    const char *names[] = {"idle", "MFMA (5/iter)", "VALU (48 pk_fma/iter)", "MFMA+pk_fma interleaved", "VALU (96 v_fma/iter)",
                           "MFMA+v_fma interleaved", "VALU (96 and + 96 alignbit)"};
    for (auto &rl : roles) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        k<<<256, 512>>>(out, rl[0], rl[1], 1e-6f);
        CHK(hipEventRecord(e0));
        k<<<256, 512>>>(out, rl[0], rl[1], 1e-6f);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("wave A: %-24s wave B: %-24s %.3f ms = %.0f cycles/iteration @2.4GHz\n", names[rl[0]], names[rl[1]], ms,
               ms * 1e-3 * 2.4e9 / ITERS);
    }
    return 0;
}
