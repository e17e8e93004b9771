// mfma_f16_acc.hip -- how accurate is the accumulation inside v_mfma_f32_32x32x16_f16 on gfx950?
// The ISA documents the operand types (f16 x f16 products, f32 accumulator), not the order or the internal width of the
// 16-term sum.  A kernel that takes DECISIONS from such sums with a proven error band needs a bound
//       |D - (c + sum_k a_k b_k)|  <=  gamma * (|c| + sum_k |a_k b_k|)
// so this program measures gamma: random tiles from four input families (wide exponent spread, heavy cancellation,
// one large + many small terms, accumulator much larger / smaller than the products), the exact value from 128-bit
// integer arithmetic on the host, two MFMAs chained through the accumulator as the scoring kernel uses them.
// It also verifies the operand / result lane layout the kernel relies on (a wrong layout shows up as errors of order 1).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_acc mfma_f16_acc.hip && ./mfma_f16_acc
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// One wave per tile.  A: [tile][32 rows][32 k] halfs (k 0..15 first MFMA, 16..31 second), B: [tile][32 cols][32 k],
// C, D: [tile][32 rows][32 cols] floats.
__global__ __launch_bounds__(64) void tile_kernel(const _Float16 *A, const _Float16 *B, const float *C, float *D1, float *D2)
{
    const int t = blockIdx.x, l = threadIdx.x, rc = l & 31, kb = l >> 5;
    const _Float16 *a = A + ((size_t)t * 32 + rc) * 32, *b = B + ((size_t)t * 32 + rc) * 32;
    h8 a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) {
        a0[i] = a[8 * kb + i];
        a1[i] = a[16 + 8 * kb + i];
        b0[i] = b[8 * kb + i];
        b1[i] = b[16 + 8 * kb + i];
    }
    f16v c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
        c[r] = C[((size_t)t * 32 + row) * 32 + rc];
    }
    f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
    f16v d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, d1, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
        D1[((size_t)t * 32 + row) * 32 + rc] = d1[r];
        D2[((size_t)t * 32 + row) * 32 + rc] = d2[r];
    }
}

static float h2f(_Float16 h) { return (float)h; }

// value of a half as an integer multiple of 2^-24
static __int128 half_fix(_Float16 h) { return (__int128)llround(std::ldexp((double)h2f(h), 24)); }

int main()
{
    const int T = 512; // tiles per family
    const char *names[] = {"wide exponent spread", "heavy cancellation", "one large + small terms", "accumulator >> products",
                           "accumulator << products", "split operands (hh, hl, lh, ll)"};
    const int F = 6;
    std::mt19937_64 rng(20260210);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<_Float16> A((size_t)T * 32 * 32), B((size_t)T * 32 * 32);
    std::vector<float> C((size_t)T * 32 * 32), D1(C.size()), D2(C.size());
    _Float16 *dA, *dB;
    float *dC, *dD1, *dD2;
    CHK(hipMalloc(&dA, A.size() * 2));
    CHK(hipMalloc(&dB, B.size() * 2));
    CHK(hipMalloc(&dC, C.size() * 4));
    CHK(hipMalloc(&dD1, C.size() * 4));
    CHK(hipMalloc(&dD2, C.size() * 4));
    printf("%-34s %12s %12s %12s %12s\n", "family", "gamma16/u", "gamma32/u", "RNE-exact16", "RNE-exact32");
    double worst = 0.0;
    for (int f = 0; f < F; ++f) {
        for (size_t i = 0; i < A.size(); ++i) {
            const int k = (int)(i & 31);
            double a = U(rng), b = U(rng);
            if (f == 0) {
                a = std::ldexp(a, (int)(rng() % 17) - 8);
                b = std::ldexp(b, (int)(rng() % 17) - 8);
            } else if (f == 1) { // neighbours cancel almost exactly (the B side repeats, the A side alternates in sign)
                a = (k & 1) ? -(1.0 + 1e-3 * U(rng)) : 1.0;
                b = 1.0 + ((k >> 1) * 0.01);
            } else if (f == 2) {
                a = (k == 5) ? 2048.0 * (1.0 + 0.4 * a) : 0.01 * a;
                b = (k == 5) ? 1024.0 : b;
            } else if (f == 5) { // what the scoring kernel feeds: groups of four slots = the split of one product
                const int q = k & 3;
                static double x = 0.0, y = 0.0;
                if (q == 0) {
                    x = std::ldexp(U(rng), (int)(rng() % 9) - 2);
                    y = std::ldexp(U(rng), (int)(rng() % 9) - 2);
                }
                const _Float16 xh = (_Float16)(float)x;
                const _Float16 xl = (_Float16)(float)(x - (double)h2f(xh));
                a = (q < 2) ? (double)h2f(xh) : (double)h2f(xl);
                const _Float16 yh = (_Float16)(float)y;
                const _Float16 yl = (_Float16)(float)(y - (double)h2f(yh));
                b = (q & 1) ? (double)h2f(yl) : (double)h2f(yh);
            }
            A[i] = (_Float16)(float)a;
            B[i] = (_Float16)(float)b;
        }
        for (size_t i = 0; i < C.size(); ++i) {
            double c = U(rng);
            if (f == 3) c *= 1.0e6;
            if (f == 4) c *= 1.0e-6;
            if (f == 1) c = 0.0;
            // multiples of 2^-48 below 2^30 so that the exact value fits the host's 128-bit fixed point
            C[i] = (float)c;
            if (std::fabs(C[i]) < std::ldexp(1.0, -24)) C[i] = 0.0f;
        }
        CHK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        tile_kernel<<<T, 64>>>(dA, dB, dC, dD1, dD2);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(D1.data(), dD1, C.size() * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(D2.data(), dD2, C.size() * 4, hipMemcpyDeviceToHost));
        double g1 = 0.0, g2 = 0.0;
        size_t rne1 = 0, rne2 = 0, n = 0;
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    const _Float16 *a = &A[((size_t)t * 32 + r) * 32], *b = &B[((size_t)t * 32 + c) * 32];
                    const size_t o = ((size_t)t * 32 + r) * 32 + c;
                    __int128 s = (__int128)std::ldexp((long double)C[o], 48); // exact: C is a multiple of 2^-48
                    long double mag = std::fabs((long double)C[o]);
                    long double e1 = 0, e2 = 0;
                    for (int k = 0; k < 32; ++k) {
                        const __int128 pr = half_fix(a[k]) * half_fix(b[k]);
                        s += pr;
                        mag += std::fabs((long double)h2f(a[k]) * (long double)h2f(b[k]));
                        if (k == 15 || k == 31) {
                            const long double exact = std::ldexp((long double)s, -48);
                            const float got = (k == 15) ? D1[o] : D2[o];
                            const long double err = std::fabs((long double)got - exact);
                            const long double g = mag > 0 ? err / (mag * std::ldexp(1.0L, -24)) : 0;
                            if (k == 15) {
                                e1 = g;
                                rne1 += ((float)exact == got);
                                // the second MFMA starts from the ROUNDED first result: measure it against the exact sum anyway
                            } else {
                                e2 = g;
                                rne2 += ((float)exact == got);
                            }
                        }
                    }
                    if (e1 > g1) g1 = e1;
                    if (e2 > g2) g2 = e2;
                    ++n;
                }
        printf("%-34s %12.3f %12.3f %11.2f%% %11.2f%%\n", names[f], g1, g2, 100.0 * rne1 / n, 100.0 * rne2 / n);
        if (g2 > worst) worst = g2;
        if (g1 > worst) worst = g1;
    }
    printf("worst gamma over all families: %.3f u  (u = 2^-24; error relative to |c| + sum |a_k b_k|)\n", worst);
    return 0;
}
