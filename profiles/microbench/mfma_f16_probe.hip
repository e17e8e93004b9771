// mfma_f16_probe.hip -- directed probes of the adder inside v_mfma_f32_32x32x16_f16 (gfx950): an accumulator of 2^24
// (unit in the last place 2) plus n equal products x, for x below, at and above the accumulator's resolution.  The
// printed D - c shows whether small terms are summed among themselves before they meet the accumulator, how many bits
// below the largest term's last place survive, and whether what falls below is truncated or rounded.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe mfma_f16_probe.hip && ./mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// every row/column of the tile carries the same numbers, so every output element is the same dot product
__global__ void probe(const float *xs, const int *ns, const float *cs, float *out, int cases)
{
    const int l = threadIdx.x, kb = l >> 5;
    for (int t = 0; t < cases; ++t) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) {
            const int k = 8 * kb + i;
            a[i] = (_Float16)(k < ns[t] ? 1.0f : 0.0f);
            b[i] = (_Float16)xs[t];
        }
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = cs[t];
        f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (l == 0) out[t] = d[0];
    }
}

int main()
{
    std::vector<float> xs, cs;
    std::vector<int> ns;
    const float xv[] = {1.0f, 0.5f, 0.75f, 0.25f, 0.125f, 0.0625f, 1.5f, 1.75f, 1.9375f, 0.9375f, -0.75f, -1.0f, -0.0625f};
    const int nv[] = {1, 2, 3, 4, 5, 8, 9, 12, 16};
    const float cv[] = {16777216.0f, -16777216.0f, 16777218.0f};
    for (float c : cv)
        for (float x : xv)
            for (int n : nv) {
                xs.push_back(x);
                ns.push_back(n);
                cs.push_back(c);
            }
    const int N = (int)xs.size();
    float *dx, *dc, *dout;
    int *dn;
    hipMalloc(&dx, N * 4);
    hipMalloc(&dc, N * 4);
    hipMalloc(&dout, N * 4);
    hipMalloc(&dn, N * 4);
    hipMemcpy(dx, xs.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, cs.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dn, ns.data(), N * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, dn, dc, dout, N);
    std::vector<float> out(N);
    hipMemcpy(out.data(), dout, N * 4, hipMemcpyDeviceToHost);
    printf("%12s %9s %3s %12s %12s\n", "c", "x", "n", "exact n*x", "D - c");
    for (int i = 0; i < N; ++i)
        printf("%12.0f %9.4f %3d %12.4f %12.1f\n", cs[i], xs[i], ns[i], ns[i] * (double)xs[i], (double)out[i] - (double)cs[i]);
    return 0;
}
