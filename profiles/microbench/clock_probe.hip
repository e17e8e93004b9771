// clock_probe.hip -- in-kernel shader clock under a sustained VALU load (MI355X_MICROARCH.md "DVFS give-back" item 6):
// clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, sampled around a long loop of the given instruction mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint64_t *out, int iters, uint32_t seed)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3));
    float s = 1.0000001f, t = 0.9999999f;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&v[i & 6]) : "v"(*(double *)&v[(i + 2) & 6]));
            if (KIND == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i];
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = t1 - t0;
        out[3 * blockIdx.x + 1] = r1 - r0;
        out[3 * blockIdx.x + 2] = (uint64_t)__float_as_uint(r);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    uint64_t *out;
    CHK(hipMalloc(&out, (size_t)blocks * 24));
    std::vector<uint64_t> h((size_t)blocks * 3);
    const char *names[3] = {"v_fma_f32", "v_pk_mul_f32", "v_bcnt+v_xor"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            int iters = 400000; // ~ tens of ms per launch
            if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk;
        for (int b = 0; b < blocks; ++b) clk.push_back((double)h[3 * b] / (double)h[3 * b + 1] * 100.0);
        std::sort(clk.begin(), clk.end());
        double cyc = (double)h[0] / (400000.0 * 8 * (kind == 2 ? 2 : 1));
        printf("%-14s in-kernel clock median %.0f MHz (min %.0f max %.0f); %.2f shader cycles per wave-instruction at 8 waves/SIMD\n",
               names[kind], clk[clk.size() / 2], clk.front(), clk.back(), cyc / 8.0 * 1.0);
    }
    return 0;
}
