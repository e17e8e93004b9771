// mfma_fp4_rate.hip -- what rate does v_mfma_f32_32x32x64_f8f6f4 with FP4 operands (cbsz = blgp = 4) reach on gfx950, for
// independent and for dependent (chained through the accumulator) streams, one to three waves per SIMD, plain and
// block-scaled?  The matcher (ps_matcher_mfma.h) runs four-deep dependent chains at three waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_fp4_rate mfma_fp4_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int ITERS = 4000;

// MODE 0: 4 independent accumulators; 1: one chain (every MFMA depends on the previous); 2: two chains interleaved;
// 3: as 0 with the block-scaled form; 4: chains of four restarted from a constant (the matcher's pattern), two interleaved
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int seed)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = 0x66666666 ^ (int)((threadIdx.x * 2654435761u + i) & 0x88888888u) ^ seed;
        b[i] = 0x66666666 ^ (int)((threadIdx.x * 40503u + 7 * i) & 0x88888888u);
    }
    v16f c0, c1, c2, c3, z;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = z[r] = 8388608.0f + r;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 0, 0, 0);
        } else if (MODE == 1) {
            for (int q = 0; q < 4; ++q) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
        } else if (MODE == 2) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
        } else if (MODE == 3) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, (int)0x88888888u, 0, (int)0x7F7F7F7Fu);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, (int)0x88888888u, 0, (int)0x7F7F7F7Fu);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, (int)0x88888888u, 0, (int)0x7F7F7F7Fu);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, (int)0x88888888u, 0, (int)0x7F7F7F7Fu);
        } else {
            // two chains of four from a constant start, then a 10-instruction vector epilogue on each (max over the registers)
            v16f p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, z, 4, 4, 0, 0, 0, 0);
            v16f q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, z, 4, 4, 0, 0, 0, 0);
            for (int s = 0; s < 3; ++s) {
                p = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, p, 4, 4, 0, 0, 0, 0);
                q = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, q, 4, 4, 0, 0, 0, 0);
            }
            float m = p[0], n = q[0];
            for (int r = 1; r < 16; ++r) {
                m = fmaxf(m, p[r]);
                n = fmaxf(n, q[r]);
            }
            c0[0] = fmaxf(c0[0], m);
            c1[0] = fmaxf(c1[0], n);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int wavesPerSimd, float *d)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int blocks = 256 * wavesPerSimd; // one 256-thread group = one wave per SIMD of a CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmaPerWave = (MODE == 4 ? 8.0 : 4.0) * ITERS;
    const double perSimd = mfmaPerWave * wavesPerSimd;
    const double cyc = ms * 1e-3 * 2.4e9 / perSimd;
    const double pf = perSimd * 1024.0 * 131072.0 / (ms * 1e-3) / 1e15;
    printf("%-44s waves/SIMD %d  %8.3f ms  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f PFLOP/s\n", name, wavesPerSimd, ms, cyc, pf);
    return 0;
}

int main()
{
    float *d;
    CHK(hipMalloc(&d, 256 * 3 * 256 * sizeof(float)));
    for (int w = 1; w <= 3; ++w) {
        run<0>("4 independent accumulators", w, d);
        run<1>("one dependent chain", w, d);
        run<2>("two dependent chains interleaved", w, d);
        run<3>("4 independent, block-scaled form", w, d);
        run<4>("2 x (chain of 4 + 16-register max epilogue)", w, d);
    }
    return 0;
}
