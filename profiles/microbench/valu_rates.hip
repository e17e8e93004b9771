// valu_rates.hip -- measures the sustained issue rate of the VALU instruction kinds that the two
// sweeps of the hot path are made of (gfx950).  Used to price the kernels against what the vector
// ALUs can actually issue (DESIGN.md "Roofline"), because the path is VALU-bound, not HBM-bound.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;  // independent dependency chains per lane

#define KERNEL_U32(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        uint32_t v[CHAINS];                                                                \
        for (int i = 0; i < CHAINS; ++i) v[i] = seed + threadIdx.x * 7 + i;                \
        uint32_t s = seed | 1;                                                             \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(s)); \
        }                                                                                  \
        uint32_t r = 0;                                                                    \
        for (int i = 0; i < CHAINS; ++i) r ^= v[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                           \
    }

#define KERNEL_F32(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        float v[CHAINS];                                                                   \
        for (int i = 0; i < CHAINS; ++i) v[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3)); \
        float s = 1.0000001f;                                                              \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(s)); \
        }                                                                                  \
        float r = 0;                                                                       \
        for (int i = 0; i < CHAINS; ++i) r += v[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);                          \
    }

#define KERNEL_F64(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        double v[CHAINS];                                                                  \
        for (int i = 0; i < CHAINS; ++i) v[i] = 1.0 + 1e-3 * (double)(threadIdx.x + i + (seed & 3)); \
        double s = 1.0000001;                                                              \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(s)); \
        }                                                                                  \
        double r = 0;                                                                      \
        for (int i = 0; i < CHAINS; ++i) r += v[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)__double_as_longlong(r);           \
    }

#define KERNEL_PK(NAME, ASM)                                                               \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        typedef float f2 __attribute__((ext_vector_type(2)));                              \
        f2 v[CHAINS];                                                                      \
        for (int i = 0; i < CHAINS; ++i) v[i] = f2{1.0f + 1e-3f * (float)(threadIdx.x + i), 1.5f + (seed & 3)}; \
        f2 s = f2{1.0000001f, 0.9999999f};                                                 \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(s)); \
        }                                                                                  \
        float r = 0;                                                                       \
        for (int i = 0; i < CHAINS; ++i) r += v[i].x + v[i].y;                             \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);                          \
    }

#define KERNEL_F32_3(NAME, ASM)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        float v[CHAINS];                                                                   \
        for (int i = 0; i < CHAINS; ++i) v[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3)); \
        float s = 1.0000001f, t = 0.9999999f + (float)(seed & 1);                          \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(s), "v"(t)); \
        }                                                                                  \
        float r = 0;                                                                       \
        for (int i = 0; i < CHAINS; ++i) r += v[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);                          \
    }
#define KERNEL_F32_S(NAME, ASM)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)              \
    {                                                                                      \
        float v[CHAINS];                                                                   \
        for (int i = 0; i < CHAINS; ++i) v[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3)); \
        float s = 1.0000001f + (float)(seed & 1);                                          \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(v[i]) : "s"(s)); \
        }                                                                                  \
        float r = 0;                                                                       \
        for (int i = 0; i < CHAINS; ++i) r += v[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);                          \
    }
KERNEL_F32_3(k_fma3_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F32_3(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL_F32_3(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL_F32_S(k_mul_f32_s, "v_mul_f32 %0, %1, %0")
KERNEL_F32_S(k_fma_f32_s, "v_fma_f32 %0, %0, %1, %0")
KERNEL_F32(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL_F32(k_cmp_only, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL_U32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_U32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL_U32(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL_U32(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL_U32(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL_U32(k_xor_s, "v_xor_b32 %0, s4, %0")
KERNEL_U32(k_mov, "v_mov_b32 %0, %1")
KERNEL_U32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL_U32(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL_U32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL_U32(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL_F32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL_F32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL_F32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F32(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %0")
KERNEL_F32(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %1")
KERNEL_F32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %1")
KERNEL_F32(k_cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc")
KERNEL_PK(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PK(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL_PK(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
KERNEL_F64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")

__global__ __launch_bounds__(256) void k_cvt_f64_f32(uint32_t *out, uint32_t seed)
{
    float v[CHAINS];
    double d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
    }
    double r = 0;
    for (int i = 0; i < CHAINS; ++i) r += d[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)__double_as_longlong(r);
}

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Entry { const char *name; kern_t k; int insts; };

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint32_t *out;
    const int wavesPerSimd = 8;                       // full occupancy: 32 waves per CU
    const int blocks = cus * wavesPerSimd;            // 256 threads = 4 waves = one per SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    Entry es[] = {
        {"v_fma_f32 (3 regs)", k_fma3_f32, 1}, {"v_fmac_f32", k_fmac_f32, 1}, {"v_min3_f32", k_min3_f32, 1},
        {"v_mul_f32 (sgpr)", k_mul_f32_s, 1}, {"v_fma_f32 (sgpr)", k_fma_f32_s, 1}, {"v_max_f32", k_max_f32, 1},
        {"v_cmp_lt_f32", k_cmp_only, 1}, {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and_b32, 1},
        {"v_max_u32", k_max_u32, 1}, {"v_min_i32", k_min_i32, 1}, {"v_lshlrev_b32", k_lshlrev, 1},
        {"v_xor_b32 (sgpr)", k_xor_s, 1}, {"v_mov_b32", k_mov, 1},
        {"v_xor_b32", k_xor, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_min_u32", k_min_u32, 1}, {"v_lshl_or_b32", k_lshl_or, 1},
        {"v_mul_f32", k_mul_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1},
        {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_div_scale_f32", k_div_scale, 1}, {"v_div_fmas_f32", k_div_fmas, 1},
        {"v_div_fixup_f32", k_div_fixup, 1}, {"v_cmp_lt_f32+v_addc", k_cmp_lt_f32, 2},
        {"v_pk_mul_f32", k_pk_mul_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1},
        {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_cmp_lt_f64", k_cmp_f64, 1},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 1},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-22s %12s %16s %14s\n", "instruction", "ms", "wave-instr/s (T)", "cyc/instr/SIMD@2.4GHz");
    for (const Entry &e : es) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipDeviceSynchronize());
        const int reps = 10;
        CHK(hipEventRecord(e0));
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        double waveInstr = (double)blocks * 4 * ITERS * CHAINS * e.insts;   // per launch
        double rate = waveInstr / (ms * 1e-3);
        double cyc = (double)cus * 4 * 2.4e9 / rate;
        printf("%-22s %12.4f %16.4f %14.2f\n", e.name, ms, rate / 1e12, cyc);
    }
    return 0;
}
