// valu_seq.hip -- cycles per wave for short VALU instruction sequences on gfx950: which instruction kinds
// overlap inside one SIMD (separate execution units) and which serialise.  Every sequence is a loop body of
// independent chains; the figure printed is SIMD cycles per loop iteration per resident wave, i.e. what the
// sequence costs when the SIMD is kept full (normalised to 2.4 GHz like valu_rates.hip).
//   hipcc --offload-arch=gfx950 -O3 -o valu_seq valu_seq.hip && ./valu_seq [wavesPerSimd]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));

#define BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(sa));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(sa));
#define XORS(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(ss));
#define XORS2(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "s"(ss), "v"(a[(i + 4) & 7]));
#define ANDS(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(ss));
#define ADDS(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(ss));
#define MULS(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(b[i]) : "s"(sfs));
#define BCNTS(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(ss));
#define MOVS(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(ss));
#define XAD(i) asm volatile("v_xad_u32 %0, %1, %0, 0" : "+v"(a[i]) : "s"(ss));
#define XORS64(i) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[i]) : "s"(ss));
#define XORSD(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sv[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sa));
#define MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sa));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(sa));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(sa) : "vcc");
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(sb));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(sb));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b[i]) : "v"(sb), "v"(sc));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(sb), "v"(sc));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(b[i]) : "v"(sb));
#define MIN3F(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(sb), "v"(sc));
#define CMPF(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(b[i]), "v"(sb) : "vcc");
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(b[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(sp));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(sd));
#define CVT64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(b[i]));

#define SEQ_KERNEL(NAME, BODY)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)             \
    {                                                                                     \
        uint32_t a[8];                                                                    \
        float b[8];                                                                       \
        f2 p[8];                                                                          \
        double d[4];                                                                      \
        for (int i = 0; i < 8; ++i) {                                                     \
            a[i] = seed + threadIdx.x * 7 + i;                                            \
            b[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i + (seed & 3));                  \
            p[i] = f2{b[i], 1.5f + (float)(seed & 3)};                                    \
        }                                                                                 \
        for (int i = 0; i < 4; ++i) d[i] = 1.0 + 1e-3 * (double)(threadIdx.x + i);        \
        uint32_t sa = seed | 1;                                                           \
        uint32_t ss = seed * 2654435761u + 12345u;                                        \
        uint32_t sv[8];                                                                   \
        for (int i = 0; i < 8; ++i) sv[i] = __builtin_amdgcn_readfirstlane(seed * (2654435761u + 2 * i) + i); \
        float sfs = 1.0f + (float)(seed & 7) * 1e-7f;                                     \
        float sb = 1.0000001f, sc = 0.9999999f + (float)(seed & 1);                       \
        f2 sp = f2{1.0000001f, 0.9999999f};                                               \
        double sd = 1.0000001;                                                            \
        for (int it = 0; it < ITERS; ++it) { BODY }                                       \
        uint32_t r = 0;                                                                   \
        for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(b[i] + p[i].x + p[i].y);  \
        for (int i = 0; i < 4; ++i) r ^= (uint32_t)__double_as_longlong(d[i]);            \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                          \
    }

SEQ_KERNEL(s_xor8, XOR(0) XOR(1) XOR(2) XOR(3) XOR(4) XOR(5) XOR(6) XOR(7))
SEQ_KERNEL(s_xors8, XORS(0) XORS(1) XORS(2) XORS(3) XORS(4) XORS(5) XORS(6) XORS(7))
SEQ_KERNEL(s_xors2_8, XORS2(0) XORS2(1) XORS2(2) XORS2(3) XORS2(4) XORS2(5) XORS2(6) XORS2(7))
SEQ_KERNEL(s_ands8, ANDS(0) ANDS(1) ANDS(2) ANDS(3) ANDS(4) ANDS(5) ANDS(6) ANDS(7))
SEQ_KERNEL(s_adds8, ADDS(0) ADDS(1) ADDS(2) ADDS(3) ADDS(4) ADDS(5) ADDS(6) ADDS(7))
SEQ_KERNEL(s_muls8, MULS(0) MULS(1) MULS(2) MULS(3) MULS(4) MULS(5) MULS(6) MULS(7))
SEQ_KERNEL(s_bcnts8, BCNTS(0) BCNTS(1) BCNTS(2) BCNTS(3) BCNTS(4) BCNTS(5) BCNTS(6) BCNTS(7))
SEQ_KERNEL(s_xors_bcnt, XORS(0) BCNT(4) XORS(1) BCNT(5) XORS(2) BCNT(6) XORS(3) BCNT(7))
SEQ_KERNEL(s_xor_bcnt_dep, XOR(0) BCNT(0) XOR(1) BCNT(1) XOR(2) BCNT(2) XOR(3) BCNT(3))
SEQ_KERNEL(s_movs8, MOVS(0) MOVS(1) MOVS(2) MOVS(3) MOVS(4) MOVS(5) MOVS(6) MOVS(7))
SEQ_KERNEL(s_xad8, XAD(0) XAD(1) XAD(2) XAD(3) XAD(4) XAD(5) XAD(6) XAD(7))
SEQ_KERNEL(s_xors64_8, XORS64(0) XORS64(1) XORS64(2) XORS64(3) XORS64(4) XORS64(5) XORS64(6) XORS64(7))
SEQ_KERNEL(s_ham_sgpr, XORS(0) BCNT(0) XORS(1) BCNT(1) XORS(2) BCNT(2) XORS(3) BCNT(3) XORS(4) BCNT(4) XORS(5) BCNT(5) XORS(6) BCNT(6) XORS(7) BCNT(7))
SEQ_KERNEL(s_ham_vgpr, XOR(0) BCNT(0) XOR(1) BCNT(1) XOR(2) BCNT(2) XOR(3) BCNT(3) XOR(4) BCNT(4) XOR(5) BCNT(5) XOR(6) BCNT(6) XOR(7) BCNT(7))
SEQ_KERNEL(s_ham_mov4, MOVS(0) MOVS(1) XOR(0) BCNT(0) XOR(1) BCNT(1) XOR(2) BCNT(2) XOR(3) BCNT(3) XOR(4) BCNT(4) XOR(5) BCNT(5) XOR(6) BCNT(6) XOR(7) BCNT(7))
SEQ_KERNEL(s_ham_sgpr_d, XORSD(0) BCNT(0) XORSD(1) BCNT(1) XORSD(2) BCNT(2) XORSD(3) BCNT(3) XORSD(4) BCNT(4) XORSD(5) BCNT(5) XORSD(6) BCNT(6) XORSD(7) BCNT(7))
SEQ_KERNEL(s_xorsd8, XORSD(0) XORSD(1) XORSD(2) XORSD(3) XORSD(4) XORSD(5) XORSD(6) XORSD(7))
SEQ_KERNEL(s_mul8, MUL(0) MUL(1) MUL(2) MUL(3) MUL(4) MUL(5) MUL(6) MUL(7))
SEQ_KERNEL(s_bcnt8, BCNT(0) BCNT(1) BCNT(2) BCNT(3) BCNT(4) BCNT(5) BCNT(6) BCNT(7))
SEQ_KERNEL(s_bcnt_mul, BCNT(0) MUL(0) BCNT(1) MUL(1) BCNT(2) MUL(2) BCNT(3) MUL(3))
SEQ_KERNEL(s_bcnt4_mul4, BCNT(0) BCNT(1) BCNT(2) BCNT(3) MUL(0) MUL(1) MUL(2) MUL(3))
SEQ_KERNEL(s_pk8, PKMUL(0) PKMUL(1) PKMUL(2) PKMUL(3) PKMUL(4) PKMUL(5) PKMUL(6) PKMUL(7))
SEQ_KERNEL(s_pk_mul, PKMUL(0) MUL(0) PKMUL(1) MUL(1) PKMUL(2) MUL(2) PKMUL(3) MUL(3))
SEQ_KERNEL(s_pk_bcnt, PKMUL(0) BCNT(0) PKMUL(1) BCNT(1) PKMUL(2) BCNT(2) PKMUL(3) BCNT(3))
SEQ_KERNEL(s_pkfma_bcnt, PKFMA(0) BCNT(0) PKFMA(1) BCNT(1) PKFMA(2) BCNT(2) PKFMA(3) BCNT(3))
SEQ_KERNEL(s_pkfma_mul, PKFMA(0) MUL(0) PKFMA(1) MUL(1) PKFMA(2) MUL(2) PKFMA(3) MUL(3))
SEQ_KERNEL(s_pk_mul_mul, PKMUL(0) MUL(0) MUL(4) PKMUL(1) MUL(1) MUL(5) PKMUL(2) MUL(2) MUL(6) PKMUL(3) MUL(3) MUL(7))
SEQ_KERNEL(s_bcnt_mul_mul, BCNT(0) MUL(0) MUL(4) BCNT(1) MUL(1) MUL(5) BCNT(2) MUL(2) MUL(6) BCNT(3) MUL(3) MUL(7))
SEQ_KERNEL(s_bcnt_mul_xor, BCNT(0) MUL(0) XOR(4) BCNT(1) MUL(1) XOR(5) BCNT(2) MUL(2) XOR(6) BCNT(3) MUL(3) XOR(7))
SEQ_KERNEL(s_bcnt_xor, BCNT(0) XOR(4) BCNT(1) XOR(5) BCNT(2) XOR(6) BCNT(3) XOR(7))
SEQ_KERNEL(s_cmpf_mul, CMPF(4) MUL(0) CMPF(5) MUL(1) CMPF(6) MUL(2) CMPF(7) MUL(3))
SEQ_KERNEL(s_maxf_mul, MAXF(4) MUL(0) MAXF(5) MUL(1) MAXF(6) MUL(2) MAXF(7) MUL(3))
SEQ_KERNEL(s_min3f_mul, MIN3F(4) MUL(0) MIN3F(5) MUL(1) MIN3F(6) MUL(2) MIN3F(7) MUL(3))
SEQ_KERNEL(s_cndm_mul, CNDM(0) MUL(0) CNDM(1) MUL(1) CNDM(2) MUL(2) CNDM(3) MUL(3))
SEQ_KERNEL(s_f64_mul, FMA64(0) MUL(0) FMA64(1) MUL(1) FMA64(2) MUL(2) FMA64(3) MUL(3))
SEQ_KERNEL(s_f64_pk, FMA64(0) PKMUL(0) FMA64(1) PKMUL(1) FMA64(2) PKMUL(2) FMA64(3) PKMUL(3))
SEQ_KERNEL(s_rcp_mul3, RCP(0) MUL(1) MUL(2) MUL(3) RCP(4) MUL(5) MUL(6) MUL(7))
SEQ_KERNEL(s_rcp_bcnt, RCP(0) BCNT(1) RCP(4) BCNT(5))
SEQ_KERNEL(s_rcp_pk, RCP(0) PKMUL(1) RCP(4) PKMUL(5))
SEQ_KERNEL(s_addf_mul, ADDF(4) MUL(0) ADDF(5) MUL(1) ADDF(6) MUL(2) ADDF(7) MUL(3))
SEQ_KERNEL(s_addu_xor, ADDU(0) XOR(4) ADDU(1) XOR(5) ADDU(2) XOR(6) ADDU(3) XOR(7))
SEQ_KERNEL(s_addu_mul, ADDU(0) MUL(0) ADDU(1) MUL(1) ADDU(2) MUL(2) ADDU(3) MUL(3))
SEQ_KERNEL(s_mov_mul, MOV(0) MUL(0) MOV(1) MUL(1) MOV(2) MUL(2) MOV(3) MUL(3))
SEQ_KERNEL(s_bcnt_fmac, BCNT(0) FMAC(0) BCNT(1) FMAC(1) BCNT(2) FMAC(2) BCNT(3) FMAC(3))
SEQ_KERNEL(s_bcnt_fma, BCNT(0) FMA(0) BCNT(1) FMA(1) BCNT(2) FMA(2) BCNT(3) FMA(3))
SEQ_KERNEL(s_bcnt_addf, BCNT(0) ADDF(0) BCNT(1) ADDF(1) BCNT(2) ADDF(2) BCNT(3) ADDF(3))
SEQ_KERNEL(s_minu_pk, MINU(0) PKMUL(0) MINU(1) PKMUL(1) MINU(2) PKMUL(2) MINU(3) PKMUL(3))
SEQ_KERNEL(s_cmpf_pk, CMPF(4) PKMUL(0) CMPF(5) PKMUL(1) CMPF(6) PKMUL(2) CMPF(7) PKMUL(3))
SEQ_KERNEL(s_pkadd_pkmul, PKADD(0) PKMUL(4) PKADD(1) PKMUL(5) PKADD(2) PKMUL(6) PKADD(3) PKMUL(7))
SEQ_KERNEL(s_cvt64_mul, CVT64(0) MUL(4) CVT64(1) MUL(5) CVT64(2) MUL(6) CVT64(3) MUL(7))
SEQ_KERNEL(s_fma8, FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7))
SEQ_KERNEL(s_pk_fma_pair, PKMUL(0) FMA(0) FMA(4) PKMUL(1) FMA(1) FMA(5) PKMUL(2) FMA(2) FMA(6) PKMUL(3) FMA(3) FMA(7))

typedef void (*kern_t)(uint32_t *, uint32_t);
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
        {"8 xor (vgpr)", s_xor8}, {"8 xor (sgpr src0)", s_xors8}, {"8 xor d=s^v(other)", s_xors2_8}, {"8 and (sgpr)", s_ands8},
        {"8 add_u32 (sgpr)", s_adds8}, {"8 mul_f32 (sgpr)", s_muls8}, {"8 bcnt (sgpr)", s_bcnts8},
        {"4 (xor sgpr, bcnt)", s_xors_bcnt}, {"4 (xor, bcnt) same reg", s_xor_bcnt_dep},
        {"8 mov (sgpr)", s_movs8}, {"8 xad (sgpr)", s_xad8}, {"8 xor_e64 (sgpr src1)", s_xors64_8},
        {"8 (xor sgpr, bcnt)", s_ham_sgpr}, {"8 (xor vgpr, bcnt)", s_ham_vgpr}, {"2 mov + 8 (xor vgpr, bcnt)", s_ham_mov4},
        {"8 (xor sgpr_i, bcnt) 8 distinct sgprs", s_ham_sgpr_d}, {"8 xor (8 distinct sgprs)", s_xorsd8},
        {"8 mul_f32", s_mul8}, {"8 bcnt", s_bcnt8}, {"8 pk_mul", s_pk8}, {"8 fma_f32", s_fma8},
        {"4 (bcnt, mul)", s_bcnt_mul}, {"4 bcnt + 4 mul", s_bcnt4_mul4}, {"4 (bcnt, fmac)", s_bcnt_fmac},
        {"4 (bcnt, fma)", s_bcnt_fma}, {"4 (bcnt, add_f32)", s_bcnt_addf},
        {"4 (bcnt, mul, mul)", s_bcnt_mul_mul}, {"4 (bcnt, mul, xor)", s_bcnt_mul_xor}, {"4 (bcnt, xor)", s_bcnt_xor},
        {"4 (pk_mul, mul)", s_pk_mul}, {"4 (pk_mul, mul, mul)", s_pk_mul_mul}, {"4 (pk_mul, fma, fma)", s_pk_fma_pair},
        {"4 (pk_fma, mul)", s_pkfma_mul},
        {"4 (pk_mul, bcnt)", s_pk_bcnt}, {"4 (pk_fma, bcnt)", s_pkfma_bcnt}, {"4 (min_u32, pk_mul)", s_minu_pk},
        {"4 (cmp_f32, pk_mul)", s_cmpf_pk}, {"4 (pk_add, pk_mul)", s_pkadd_pkmul},
        {"4 (cmp_f32, mul)", s_cmpf_mul}, {"4 (max_f32, mul)", s_maxf_mul}, {"4 (min3_f32, mul)", s_min3f_mul},
        {"4 (cndmask, mul)", s_cndm_mul}, {"4 (fma_f64, mul)", s_f64_mul}, {"4 (fma_f64, pk_mul)", s_f64_pk},
        {"4 (cvt_f64_f32, mul)", s_cvt64_mul},
        {"2 (rcp, mul, mul, mul)", s_rcp_mul3}, {"2 (rcp, bcnt)", s_rcp_bcnt}, {"2 (rcp, pk_mul)", s_rcp_pk},
        {"4 (add_f32, mul)", s_addf_mul}, {"4 (add_u32, xor)", s_addu_xor}, {"4 (add_u32, mul)", s_addu_mul},
        {"4 (mov, mul)", s_mov_mul},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-26s %10s %22s\n", "loop body", "ms", "cycles/iteration/wave");
    for (const Entry &e : es) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipDeviceSynchronize());
        const int reps = 8;
        CHK(hipEventRecord(e0));
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("%-26s %10.4f %22.2f\n", e.name, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * wavesPerSimd));
    }
    return 0;
}
