// valu_dep.hip -- dependent-issue cost of VALU instructions on gfx950 at LOW occupancy: cycles per instruction per wave
// for chains of 1, 2, 4 independent streams of v_pk_fma_f32 / v_fma_f32 / v_and_b32 / v_alignbit_b32 with 1, 2, 4, 8 waves
// per SIMD.  (valu_rates.hip measures the throughput with the SIMD kept full; a 230-register kernel runs two waves per
// SIMD and sees these numbers instead.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_dep valu_dep.hip && ./valu_dep
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

template <int KIND, int CH> __global__ __launch_bounds__(256) void k(float *out, float seed, unsigned long long *cyc)
{
    f2 a[4];
    uint32_t u[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = f2{1.0f + seed * (threadIdx.x + i), 1.0f - seed * i};
        u[i] = threadIdx.x * 7 + i;
    }
    const f2 m = {1.0000001f, 0.9999999f}, c = {seed, -seed};
    const uint32_t um = 0x7fffffffu ^ (uint32_t)seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8 / CH; ++rep) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                if (KIND == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(um));
                if (KIND == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(um));
                if (KIND == 4) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(um));
                if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 4; ++i) r += a[i].x + a[i].y + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int CH> int run(const char *name, float *out, unsigned long long *cyc)
{
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        k<KIND, CH><<<256 * wps, 256>>>(out, 1e-9f, cyc);
        CHK(hipEventRecord(e0));
        k<KIND, CH><<<256 * wps, 256>>>(out, 1e-9f, cyc);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long c;
        CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-16s chains %d  waves/SIMD %d  %.3f ms  %.2f counter ticks/instr/wave  %.2f cyc/instr/SIMD@2.4GHz\n", name, CH, wps, ms,
               (double)c / (ITERS * 8.0), ms * 1e-3 * 2.4e9 / (ITERS * 8.0 * wps));
    }
    return 0;
}

int main()
{
    float *out;
    unsigned long long *cyc;
    CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHK(hipMalloc(&cyc, 8));
    run<0, 1>("v_pk_fma_f32", out, cyc);
    run<0, 2>("v_pk_fma_f32", out, cyc);
    run<0, 4>("v_pk_fma_f32", out, cyc);
    run<5, 1>("v_pk_mul_f32", out, cyc);
    run<1, 1>("v_fma_f32", out, cyc);
    run<1, 2>("v_fma_f32", out, cyc);
    run<1, 4>("v_fma_f32", out, cyc);
    run<2, 1>("v_and_b32", out, cyc);
    run<2, 4>("v_and_b32", out, cyc);
    run<3, 1>("v_alignbit_b32", out, cyc);
    run<3, 4>("v_alignbit_b32", out, cyc);
    run<4, 1>("v_or3_b32", out, cyc);
    run<4, 4>("v_or3_b32", out, cyc);
    return 0;
}
