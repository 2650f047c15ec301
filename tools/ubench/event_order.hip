// Does a cross-stream hipStreamWaitEvent wait for the event's position in the recording queue, or for whatever has been
// enqueued behind it by the time the wait is issued? Stream A: K1 (1 ms), record E, K2 (1 ms). Stream B: wait E, K3 (0.1 ms).
// Prints when K3 started relative to K1's and K2's ends (device wall clock). Diagnostics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(unsigned long long ticks, unsigned long long* stamp)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
int main(int argc, char** argv)
{
    const unsigned flags = argc > 1 ? (unsigned) atoi(argv[1]) : hipEventDisableTiming;
    const int wait_late = argc > 2 ? atoi(argv[2]) : 1; // 1: the wait is issued after K2 has been enqueued (a host that runs ahead)
    const int grid = argc > 3 ? atoi(argv[3]) : 1;
    hipStream_t a, b;
    int least, greatest;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least);
    hipEvent_t e;
    hipEventCreateWithFlags(&e, flags);
    unsigned long long* st;
    hipHostMalloc((void**) &st, 6 * sizeof(unsigned long long), hipHostMallocDefault);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, a, 100000ull, st + 0);
        hipEventRecord(e, a);
        if (!wait_late) { hipStreamWaitEvent(b, e, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10000ull, st + 4); }
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, a, 100000ull, st + 2);
        if (wait_late) { hipStreamWaitEvent(b, e, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10000ull, st + 4); }
        hipStreamSynchronize(a);
        hipStreamSynchronize(b);
        printf("flags %u wait_late %d grid %d: K1 end %.1f us, K2 start %.1f end %.1f, K3 start %.1f us (after K1 start)\n", flags, wait_late, grid,
               (st[1] - st[0]) * 0.01, (st[2] - st[0]) * 0.01, (st[3] - st[0]) * 0.01, (st[4] - st[0]) * 0.01);
    }
    return 0;
}
