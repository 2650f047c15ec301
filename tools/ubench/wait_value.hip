// hipStreamWaitValue32 as a "kernel A has started all its workgroups" gate for a kernel B on another stream: is it supported here,
// and how soon after the condition holds does B start? A: 256 workgroups, each bumps a counter in signal memory at its start and
// spins 500 us; B (other stream): waits for counter >= 256, then stamps its start. Diagnostics.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void a_kernel(unsigned int* started, unsigned long long* stamp, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) stamp[0] = t0;
        __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void b_kernel(unsigned long long* stamp) { if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2] = wall_clock64(); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned int* started = nullptr;
    CK(hipExtMallocWithFlags((void**) &started, 8, hipMallocSignalMemory));
    unsigned long long* st;
    CK(hipHostMalloc((void**) &st, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    unsigned int expect = 0;
    CK(hipMemset(started, 0, 8));
    for (int rep = 0; rep < 4; ++rep) {
        expect += 256;
        // B is enqueued FIRST: only the wait may hold it back
        CK(hipStreamWaitValue32(b, started, expect, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(b_kernel, dim3(1), dim3(64), 0, b, st);
        hipLaunchKernelGGL(a_kernel, dim3(256), dim3(704), 0, a, started, st, 50000ull);
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        printf("A ran %.1f us; B started %.1f us after A's first workgroup\n", (st[1] - st[0]) * 0.01, ((long long) st[2] - (long long) st[0]) * 0.01);
    }
    return 0;
}
