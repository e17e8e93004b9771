// Does hipExtAnyOrderLaunch let a kernel start while its predecessor on the same stream still runs on gfx950?
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.)
// Kernel A spins for ~100 us and stamps its end; kernel B stamps its start.  Ordinary launch: B.start >= A.end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long *out, unsigned long long ticks)
{
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[0] = wall_clock64();
}
__global__ void k_stamp(unsigned long long *out)
{
    if (threadIdx.x == 0) out[1] = wall_clock64();
}
int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 16, s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, 10000ull); // 100 us at 100 MHz
            if (mode == 0)
                hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, d);
            else
                hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            hipError_t e = hipStreamSynchronize(s);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%s rep %d: B.start - A.end = %lld ticks of 10 ns (%s)\n", mode ? "any-order" : "ordinary ", rep,
                   (long long)(h[1] - h[0]), hipGetErrorString(e));
        }
    return 0;
}
