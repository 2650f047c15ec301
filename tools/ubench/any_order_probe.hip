// Does hipExtAnyOrderLaunch let a kernel start beside the kernel in front of it on the SAME stream (gfx950, ROCm 7)?
// Two one-workgroup kernels that spin for `us` microseconds each: one after the other takes 2 x us, side by side 1 x us.
// Also: the same on two streams (the reference point), and which of two kernels on two streams starts first.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out) *out = t0;
}
int main()
{
    hipStream_t s, s2;
    hipStreamCreate(&s);
    hipStreamCreate(&s2);
    unsigned long long* d;
    hipMalloc(&d, 16);
    const unsigned long long ticks = 20000; // 200 us at 100 MHz
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d + 1);
            if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d + 1);
            if (mode == 2) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, ticks, d + 1);
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            unsigned long long h[2];
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%s: %.0f us, second started %.1f us after the first\n", mode == 0 ? "same stream" : mode == 1 ? "same stream, any-order" : "two streams", us,
                   ((double) h[1] - (double) h[0]) / 100.0);
        }
    }
    return 0;
}
