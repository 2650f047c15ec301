// Does a cross-stream wait still refer to the right point of the recording queue when the host has enqueued MANY commands
// behind the recorded event before it issues the wait (a host that runs operators ahead)? Stream A: K1 (1 ms), record E, then
// n filler commands (a tiny kernel + an event record each), then K2 (1 ms). Stream B: wait E, K3. Diagnostics.
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
    const int n = argc > 1 ? atoi(argv[1]) : 100;
    const unsigned flags = argc > 2 ? (unsigned) atoi(argv[2]) : hipEventDisableTiming;
    hipStream_t a, b;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least));
    hipEvent_t e, filler[8];
    CK(hipEventCreateWithFlags(&e, flags));
    for (auto& f : filler) CK(hipEventCreateWithFlags(&f, flags));
    unsigned long long* st;
    CK(hipHostMalloc((void**) &st, 6 * sizeof(unsigned long long), hipHostMallocDefault));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000ull, st + 0);
        CK(hipEventRecord(e, a));
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100ull, nullptr);
            CK(hipEventRecord(filler[i & 7], a));
        }
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000ull, st + 2);
        CK(hipStreamWaitEvent(b, e, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10000ull, st + 4);
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        printf("fillers %d flags %u: K1 end %.1f us, K2 start %.1f end %.1f, K3 start %.1f us (after K1 start)\n", n, flags,
               (st[1] - st[0]) * 0.01, (st[2] - st[0]) * 0.01, (st[3] - st[0]) * 0.01, (st[4] - st[0]) * 0.01);
    }
    return 0;
}
