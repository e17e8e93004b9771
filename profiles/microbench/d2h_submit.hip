// d2h_submit.hip -- how long the HOST spends inside hipMemcpyAsync(device -> pinned host) behind a running kernel, by size,
// and what a copy KERNEL writing into mapped pinned memory achieves instead (round 5: the pipelined stream's downloads).
//   hipcc --offload-arch=gfx950 -O2 d2h_submit.hip -o d2h_submit && ./d2h_submit
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void spin(unsigned long long cycles, int *sink)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && cycles == 1) *sink = 1;
}

__global__ void copy_out(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t maxB = (size_t)64 << 20;
    void *d = nullptr, *h = nullptr, *hm = nullptr, *hmDev = nullptr;
    hipMalloc(&d, maxB);
    hipMemset(d, 1, maxB);
    hipHostMalloc(&h, maxB, hipHostMallocDefault);
    hipHostMalloc(&hm, maxB, hipHostMallocMapped);
    hipHostGetDevicePointer(&hmDev, hm, 0);
    const size_t sizes[] = {256, 496, 512, 996, 1024, 4096, 4960, 7936, 16384, 65536 - 64, (size_t)64 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, 4264000, (size_t)6 << 20,
                            (size_t)8 << 20, 8528000, (size_t)16 << 20, (size_t)32 << 20};
    printf("a 300 us kernel is queued first, then the download; times in us\n");
    printf("%10s | %28s | %28s\n", "bytes", "hipMemcpyAsync submit / total", "copy kernel submit / total");
    for (size_t b : sizes) {
        double sub[2] = {0, 0}, tot[2] = {0, 0};
        const int reps = 10;
        for (int v = 0; v < 2; ++v)
            for (int r = 0; r < reps + 2; ++r) {
                hipStreamSynchronize(s);
                const double t0 = now();
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 300ull * 100000ull, (int *)nullptr); // ~300 us at 100 MHz counter
                const double t1 = now();
                if (v == 0)
                    hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, s);
                else
                    hipLaunchKernelGGL(copy_out, dim3(128), dim3(256), 0, s, (const uint4 *)d, (uint4 *)hmDev, b / 16);
                const double t2 = now();
                hipStreamSynchronize(s);
                const double t3 = now();
                (void)t0;
                if (r >= 2) {
                    sub[v] += t2 - t1;
                    tot[v] += t3 - t1;
                }
            }
        printf("%10zu | %12.1f / %12.1f | %12.1f / %12.1f\n", b, 1e6 * sub[0] / reps, 1e6 * tot[0] / reps, 1e6 * sub[1] / reps,
               1e6 * tot[1] / reps);
    }
    // copy kernel bandwidth alone (no kernel in front), by work-group count
    for (int wg : {16, 64, 128, 256, 512}) {
        const size_t b = (size_t)32 << 20;
        hipStreamSynchronize(s);
        const double t0 = now();
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy_out, dim3(wg), dim3(256), 0, s, (const uint4 *)d, (uint4 *)hmDev, b / 16);
        hipStreamSynchronize(s);
        printf("copy kernel, %3d work-groups: %.1f GB/s\n", wg, 10.0 * b / (now() - t0) / 1e9);
    }
    {
        const size_t b = (size_t)32 << 20;
        hipStreamSynchronize(s);
        const double t0 = now();
        for (int r = 0; r < 10; ++r) hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        printf("hipMemcpyAsync D2H 32 MiB: %.1f GB/s\n", 10.0 * b / (now() - t0) / 1e9);
    }
    printf("check: %d\n", (int)((unsigned char *)hm)[12345]);
    return 0;
}
