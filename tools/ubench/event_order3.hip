// The two-stream pattern of the light operators, mimicked with spin kernels: an occlusion stream that may run up to four
// buffers ahead of the main stream. Does it? Prints every kernel's start / end (us). Diagnostics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long* stamp)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}
int main(int argc, char** argv)
{
    const int ops = 10, nbuf = argc > 1 ? atoi(argv[1]) : 4;
    const int timing_events = argc > 2 ? atoi(argv[2]) : 1; // begin_timed / end_timed around the operator and the frame
    hipStream_t a, b;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least));
    hipEvent_t idle[8], ready[8], tev[4];
    for (auto& e : idle) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ready) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : tev) CK(hipEventCreate(&e));
    unsigned long long* st;
    CK(hipHostMalloc((void**) &st, ops * 8 * sizeof(unsigned long long), hipHostMallocDefault));
    bool used[8] = {};
    for (int n = 0; n < ops; ++n) {
        const int k = n % nbuf;
        if (timing_events) CK(hipEventRecord(tev[0], a));
        if (used[k]) CK(hipStreamWaitEvent(b, idle[k], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 30000ull, st + 8 * n + 0); // occlusion
        CK(hipEventRecord(ready[k], b));
        CK(hipStreamWaitEvent(a, ready[k], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 30000ull, st + 8 * n + 2); // sweep 0
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 30000ull, st + 8 * n + 4); // sweep 1
        CK(hipEventRecord(idle[k], a));
        used[k] = true;
        if (timing_events) { CK(hipEventRecord(tev[1], a)); CK(hipEventRecord(tev[2], a)); }
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 60000ull, st + 8 * n + 6); // frame
        if (timing_events) CK(hipEventRecord(tev[3], a));
    }
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    const unsigned long long t0 = st[0];
    for (int n = 0; n < ops; ++n)
        printf("op %d: occ %7.0f-%7.0f  sweep0 %7.0f-%7.0f  sweep1 %7.0f-%7.0f  frame %7.0f-%7.0f\n", n, (st[8 * n] - t0) * 0.01, (st[8 * n + 1] - t0) * 0.01,
               (st[8 * n + 2] - t0) * 0.01, (st[8 * n + 3] - t0) * 0.01, (st[8 * n + 4] - t0) * 0.01, (st[8 * n + 5] - t0) * 0.01, (st[8 * n + 6] - t0) * 0.01, (st[8 * n + 7] - t0) * 0.01);
    return 0;
}
