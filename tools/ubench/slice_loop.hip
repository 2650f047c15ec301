// Microbenchmark: shader cycles per iteration of a loop shaped like one slice of k_light_chain — R LDS reads, wait, V
// dependent-ish VALU instructions, W LDS writes, S scalar instructions, wait, optional s_barrier — run by 16 waves (one
// 1024-thread workgroup per CU, 256 workgroups). hipcc --offload-arch=gfx950 -O3 -o slice_loop slice_loop.hip && ./slice_loop
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

template <int R, int V, int W, int S, bool BARRIER, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_loop(float* out, long long* cyc, int iters)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += WAVES * 64) lds[i] = i;
    __syncthreads();
    const uint32_t a0 = (threadIdx.x * 4) & 0xfffc, a1 = ((threadIdx.x + 41) * 4) & 0xfffc;
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const float b = 1.0001f, c = 0.5f;
    int s0 = 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // R reads (in groups of 4)
        for (int k = 0; k < R / 4; ++k)
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %4 offset:160\n ds_read_b32 %3, %5 offset:160\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v0 += r0; v1 += r1; v2 += r2; v3 += r3;
        for (int k = 0; k < V / 16; ++k)
            asm volatile(REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(b), "v"(c));
        for (int k = 0; k < S / 8; ++k) asm volatile(REP8("s_add_u32 %0, %0, 1\n") : "+s"(s0) : : "scc");
        for (int k = 0; k < W / 2; ++k)
            asm volatile("ds_write_b32 %0, %2 offset:32768\n ds_write_b32 %1, %3 offset:32768\n" : : "v"(a0), "v"(a1), "v"(v0), "v"(v1) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + (float) s0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int V, int W, int S, bool BARRIER, int WAVES>
static void run(const char* what)
{
    float* out; long long* cyc;
    const int blocks = 256, iters = 2048;
    hipMalloc(&out, blocks * WAVES * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_loop<R, V, W, S, BARRIER, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += v;
    std::printf("%2d waves  reads %2d  valu %3d  writes %d  scalar %2d  barrier %d : %7.0f cycles per iteration   %s\n", WAVES, R, V, W, S, (int) BARRIER,
                mean / 256 / iters, what);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0, 0, 0, 0, true, 16>("barrier alone");
    run<0, 0, 0, 0, true, 8>("barrier alone");
    run<8, 0, 0, 0, false, 16>("8 reads + wait");
    run<16, 0, 0, 0, false, 16>("16 reads + wait");
    run<0, 64, 0, 0, false, 16>("64 VALU");
    run<0, 96, 0, 0, false, 16>("96 VALU");
    run<0, 0, 4, 0, false, 16>("4 writes + wait");
    run<0, 0, 0, 64, false, 16>("64 scalar");
    run<16, 96, 4, 64, false, 16>("everything, no barrier");
    run<16, 96, 4, 64, true, 16>("everything + barrier: the chain's slice");
    run<8, 48, 2, 64, true, 16>("half the per-stream work + barrier");
    run<8, 48, 2, 16, true, 16>("half the work, few scalars");
    run<16, 96, 4, 16, true, 16>("full work, few scalars");
    run<16, 96, 4, 64, true, 8>("8 waves, same per-wave work");
    return 0;
}
