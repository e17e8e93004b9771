// mfma_f16_denorm.hip -- does v_mfma_f32_32x32x16_f16 honour subnormal f16 operands, and does v_cvt_f16_f32 produce them
// (gfx950, default HIP float mode)?   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_denorm mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float *x, float *out, unsigned *bits)
{
    const int l = threadIdx.x, kb = l >> 5;
    for (int t = 0; t < 8; ++t) {
        const _Float16 hx = (_Float16)x[t]; // v_cvt_f16_f32 on the device
        h8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)0.0f;
            b[i] = (_Float16)0.0f;
        }
        if (kb == 0) {
            a[0] = hx;
            b[0] = (_Float16)1024.0f;
        }
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
        f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (l == 0) {
            out[t] = d[0];
            bits[t] = (unsigned)__builtin_bit_cast(unsigned short, hx);
        }
    }
}

int main()
{
    float hx[8] = {std::ldexp(1.0f, -14), std::ldexp(1.0f, -15), std::ldexp(1.5f, -16), std::ldexp(1.0f, -20),
                   std::ldexp(1.0f, -24), std::ldexp(1.0f, -25), std::ldexp(1.75f, -24), -std::ldexp(1.0f, -22)};
    float *dx, *dout, out[8];
    unsigned *db, bits[8];
    hipMalloc(&dx, 32);
    hipMalloc(&dout, 32);
    hipMalloc(&db, 32);
    hipMemcpy(dx, hx, 32, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, dout, db);
    hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost);
    hipMemcpy(bits, db, 32, hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t)
        printf("x = %.10g  half bits 0x%04x  mfma(x * 1024) = %.10g  (exact %.10g)\n", hx[t], bits[t], out[t], hx[t] * 1024.0);
    return 0;
}
