// valu_mix.hip -- do two VALU instruction kinds overlap on a gfx950 SIMD?
// For every (A, B) pair three kernels run with the same total instruction count per wave:
//   all-A, all-B                 : the single-kind rates (as in valu_rates.hip)
//   intra-wave mix               : every wave alternates A and B (independent chains)
//   inter-wave mix               : even work-groups run only A, odd work-groups only B (waves of both kinds
//                                  are resident on every SIMD)
// If the kinds serialise, t(mix) = (t(A) + t(B)) / 2; if they overlap, t(mix) approaches max(t(A), t(B)) / 2.
//   hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 4; // chains per kind

// which: 0 = all A, 1 = all B, 2 = intra-wave alternate, 3 = by work-group parity
#define MIX_KERNEL(NAME, ASM_A, ASM_B)                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, int which)                 \
    {                                                                                                    \
        uint32_t a[2 * CH];                                                                              \
        float b[2 * CH];                                                                                 \
        for (int i = 0; i < 2 * CH; ++i) {                                                               \
            a[i] = seed + threadIdx.x * 7 + i;                                                           \
            b[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3));                                 \
        }                                                                                                \
        uint32_t sa = seed | 1;                                                                          \
        float sb = 1.0000001f;                                                                           \
        int mode = which == 3 ? (int)(blockIdx.x & 1) : which;                                           \
        if (mode == 0) {                                                                                 \
            for (int it = 0; it < ITERS; ++it) {                                                         \
                _Pragma("unroll") for (int i = 0; i < 2 * CH; ++i) asm volatile(ASM_A : "+v"(a[i]) : "v"(sa)); \
            }                                                                                            \
        } else if (mode == 1) {                                                                          \
            for (int it = 0; it < ITERS; ++it) {                                                         \
                _Pragma("unroll") for (int i = 0; i < 2 * CH; ++i) asm volatile(ASM_B : "+v"(b[i]) : "v"(sb)); \
            }                                                                                            \
        } else {                                                                                         \
            for (int it = 0; it < ITERS; ++it) {                                                         \
                _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                         \
                    asm volatile(ASM_A : "+v"(a[i]) : "v"(sa));                                          \
                    asm volatile(ASM_B : "+v"(b[i]) : "v"(sb));                                          \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        uint32_t r = 0;                                                                                  \
        for (int i = 0; i < 2 * CH; ++i) r ^= a[i] ^ __float_as_uint(b[i]);                              \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                         \
    }

// A operates on a uint32 register, B on a float register
MIX_KERNEL(m_bcnt_xor, "v_bcnt_u32_b32 %0, %1, %0", "v_xor_b32 %0, %0, %1")
MIX_KERNEL(m_bcnt_xors, "v_bcnt_u32_b32 %0, %1, %0", "v_xor_b32 %0, s4, %0")
MIX_KERNEL(m_bcnt_mul, "v_bcnt_u32_b32 %0, %1, %0", "v_mul_f32 %0, %0, %1")
MIX_KERNEL(m_bcnt_fma, "v_bcnt_u32_b32 %0, %1, %0", "v_fma_f32 %0, %0, %1, %1")
MIX_KERNEL(m_bcnt_pk, "v_bcnt_u32_b32 %0, %1, %0", "v_max_f32 %0, %0, %1")
MIX_KERNEL(m_min_mul, "v_min_u32 %0, %0, %1", "v_mul_f32 %0, %0, %1")
MIX_KERNEL(m_add_mul, "v_add_u32 %0, %0, %1", "v_mul_f32 %0, %0, %1")
MIX_KERNEL(m_xor_mul, "v_xor_b32 %0, %0, %1", "v_mul_f32 %0, %0, %1")
MIX_KERNEL(m_bcnt_rcp, "v_bcnt_u32_b32 %0, %1, %0", "v_rcp_f32 %0, %0")
MIX_KERNEL(m_add_rcp, "v_add_u32 %0, %0, %1", "v_rcp_f32 %0, %0")
MIX_KERNEL(m_cmp_mul, "v_cmp_lt_u32 vcc, %0, %1", "v_mul_f32 %0, %0, %1")
MIX_KERNEL(m_bcnt_cmp, "v_bcnt_u32_b32 %0, %1, %0", "v_cmp_lt_f32 vcc, %0, %1")
MIX_KERNEL(m_bcnt_max, "v_bcnt_u32_b32 %0, %1, %0", "v_max_f32 %0, %0, %1")
MIX_KERNEL(m_lshl_mul, "v_lshlrev_b32 %0, 1, %0", "v_mul_f32 %0, %0, %1")

typedef void (*kern_t)(uint32_t *, uint32_t, int);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = cus * wavesPerSimd;
    printf("device %s, %d CUs, %d waves per SIMD\n", prop.gcnArchName, cus, wavesPerSimd);
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    Entry es[] = {
        {"bcnt | xor", m_bcnt_xor}, {"bcnt | xor(sgpr)", m_bcnt_xors}, {"bcnt | mul_f32", m_bcnt_mul},
        {"bcnt | fma_f32", m_bcnt_fma}, {"min_u32 | mul_f32", m_min_mul}, {"add_u32 | mul_f32", m_add_mul},
        {"xor | mul_f32", m_xor_mul}, {"bcnt | rcp_f32", m_bcnt_rcp}, {"add_u32 | rcp_f32", m_add_rcp},
        {"cmp_u32 | mul_f32", m_cmp_mul}, {"bcnt | cmp_f32", m_bcnt_cmp}, {"bcnt | max_f32", m_bcnt_max},
        {"lshl | mul_f32", m_lshl_mul},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-20s %9s %9s %9s %9s   %s\n", "A | B", "all-A ms", "all-B ms", "intra ms", "inter ms", "serial (A+B)/2 ms");
    for (const Entry &e : es) {
        float t[4];
        for (int which = 0; which < 4; ++which) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u, which);
            CHK(hipDeviceSynchronize());
            const int reps = 8;
            CHK(hipEventRecord(e0));
            for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u, which);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&t[which], e0, e1));
            t[which] /= reps;
        }
        printf("%-20s %9.4f %9.4f %9.4f %9.4f   %9.4f\n", e.name, t[0], t[1], t[2], t[3], 0.5f * (t[0] + t[1]));
    }
    return 0;
}
