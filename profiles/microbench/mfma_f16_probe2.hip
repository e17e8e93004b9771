// mfma_f16_probe2.hip -- second directed probe of v_mfma_f32_32x32x16_f16's adder (gfx950): one large product
// P = a0 * b0 plus n copies of a small product 2^-s, accumulator 0 (or a tiny value).  Shows against which exponent the
// small terms are aligned (the product's true exponent or the sum of the operand exponents) and how many bits below the
// large term's last place survive.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe2 mfma_f16_probe2.hip && ./mfma_f16_probe2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float *av, const float *bv, const float *cv, float *out, int cases)
{
    const int l = threadIdx.x, kb = l >> 5;
    for (int t = 0; t < cases; ++t) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)av[t * 16 + 8 * kb + i];
            b[i] = (_Float16)bv[t * 16 + 8 * kb + i];
        }
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = cv[t];
        f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (l == 0) out[t] = d[0];
    }
}

int main()
{
    std::vector<float> av, bv, cv;
    struct Case { float a0, b0; int s, n, pos; float c; };
    std::vector<Case> cases;
    const float big[][2] = {{1.0f, 1.0f}, {1.5f, 1.5f}, {1.9990234375f, 1.9990234375f}, {1.0f, 1.9990234375f}, {1.25f, 1.5f}};
    for (auto &bg : big)
        for (int s = 22; s <= 29; ++s)
            for (int n : {1, 2, 4, 8, 15})
                for (int pos : {0, 15})
                    cases.push_back({bg[0], bg[1], s, n, pos, 0.0f});
    // the accumulator as the large term, products below it (c = 1, c = 1.5)
    for (float c : {1.0f, 1.5f, 3.0f})
        for (int s = 22; s <= 29; ++s)
            for (int n : {1, 2, 4, 8, 16}) cases.push_back({0.0f, 0.0f, s, n, -1, c});
    for (auto &cs : cases) {
        float a[16], b[16];
        for (int k = 0; k < 16; ++k) a[k] = b[k] = 0.0f;
        int placed = 0;
        for (int k = 0; k < 16 && placed < cs.n; ++k) {
            if (k == cs.pos) continue;
            a[k] = std::ldexp(1.0f, -(cs.s / 2));
            b[k] = std::ldexp(1.0f, -(cs.s - cs.s / 2));
            ++placed;
        }
        if (cs.pos >= 0) {
            a[cs.pos] = cs.a0;
            b[cs.pos] = cs.b0;
        }
        for (int k = 0; k < 16; ++k) {
            av.push_back(a[k]);
            bv.push_back(b[k]);
        }
        cv.push_back(cs.c);
    }
    const int N = (int)cases.size();
    float *da, *db, *dc, *dout;
    hipMalloc(&da, N * 64);
    hipMalloc(&db, N * 64);
    hipMalloc(&dc, N * 4);
    hipMalloc(&dout, N * 4);
    hipMemcpy(da, av.data(), N * 64, hipMemcpyHostToDevice);
    hipMemcpy(db, bv.data(), N * 64, hipMemcpyHostToDevice);
    hipMemcpy(dc, cv.data(), N * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dc, dout, N);
    std::vector<float> out(N);
    hipMemcpy(out.data(), dout, N * 4, hipMemcpyDeviceToHost);
    printf("%10s %10s %8s %3s %3s %4s %14s %14s\n", "a0", "b0", "c", "s", "n", "pos", "exact-P (ulp)", "D-P (ulp)");
    for (int i = 0; i < N; ++i) {
        const Case &cs = cases[i];
        const double P = (double)(float)(_Float16)cs.a0 * (double)(float)(_Float16)cs.b0 + cs.c;
        const double ulp = std::ldexp(1.0, (int)std::floor(std::log2(P)) - 23);
        printf("%10.6f %10.6f %8.3f %3d %3d %4d %14.4f %14.4f\n", cs.a0, cs.b0, cs.c, cs.s, cs.n, cs.pos,
               cs.n * std::ldexp(1.0, -cs.s) / ulp, ((double)out[i] - P) / ulp);
    }
    return 0;
}
