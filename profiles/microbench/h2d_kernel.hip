// h2d_kernel.hip -- upload by a copy KERNEL reading mapped pinned host memory against hipMemcpyAsync (SDMA), and the
// latency of a cross-stream dependency copy -> kernel either way (round 5: the pipelined stream's uploads).
//   hipcc --offload-arch=gfx950 -O2 h2d_kernel.hip -o h2d_kernel && ./h2d_kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void copy_u4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// four loads in flight per lane
__global__ void copy_u4x4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void touch(int *p) { if (threadIdx.x == 0) p[0] += 1; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t b = (size_t)11 << 20; // one chunk of 125 frames
    void *d = nullptr, *h = nullptr, *hDev = nullptr;
    hipMalloc(&d, b);
    hipHostMalloc(&h, b, hipHostMallocDefault);
    hipHostGetDevicePointer(&hDev, h, 0);
    for (size_t i = 0; i < b; ++i) ((unsigned char *)h)[i] = (unsigned char)i;
    printf("host %p device view %p\n", h, hDev);
    for (int wg : {32, 64, 128, 256, 512, 1024}) {
        for (int v = 0; v < 2; ++v) {
            hipStreamSynchronize(s);
            for (int r = 0; r < 2; ++r)
                if (v) hipLaunchKernelGGL(copy_u4x4, dim3(wg), dim3(256), 0, s, (const uint4 *)hDev, (uint4 *)d, b / 16);
                else hipLaunchKernelGGL(copy_u4, dim3(wg), dim3(256), 0, s, (const uint4 *)hDev, (uint4 *)d, b / 16);
            hipStreamSynchronize(s);
            const double t0 = now();
            for (int r = 0; r < 10; ++r)
                if (v) hipLaunchKernelGGL(copy_u4x4, dim3(wg), dim3(256), 0, s, (const uint4 *)hDev, (uint4 *)d, b / 16);
                else hipLaunchKernelGGL(copy_u4, dim3(wg), dim3(256), 0, s, (const uint4 *)hDev, (uint4 *)d, b / 16);
            hipStreamSynchronize(s);
            printf("upload kernel (%s), %4d work-groups: %.1f GB/s\n", v ? "4 loads in flight" : "1 load in flight", wg, 10.0 * b / (now() - t0) / 1e9);
        }
    }
    {
        hipStreamSynchronize(s);
        const double t0 = now();
        for (int r = 0; r < 10; ++r) hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        printf("hipMemcpyAsync H2D 11 MiB: %.1f GB/s\n", 10.0 * b / (now() - t0) / 1e9);
    }
    // dependency latency: upload on s, event, kernel on s2 waits for it; time from the upload's submission to the kernel's end
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    int *flag;
    hipMalloc((void **)&flag, 4);
    hipMemset(flag, 0, 4);
    for (int v = 0; v < 2; ++v) {
        double tot = 0;
        for (int r = 0; r < 12; ++r) {
            hipDeviceSynchronize();
            const double t0 = now();
            if (v) hipLaunchKernelGGL(copy_u4x4, dim3(256), dim3(256), 0, s, (const uint4 *)hDev, (uint4 *)d, b / 16);
            else hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, s);
            hipEventRecord(ev, s);
            hipStreamWaitEvent(s2, ev, 0);
            hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s2, flag);
            hipStreamSynchronize(s2);
            if (r >= 2) tot += now() - t0;
        }
        printf("%s on stream A -> event -> kernel on stream B: %.1f us in all (11 MiB)\n", v ? "upload kernel " : "hipMemcpyAsync", 1e6 * tot / 10);
    }
    return 0;
}
