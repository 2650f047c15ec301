// Microbenchmark: how many shader cycles a SIMD needs per instruction for different instruction mixes, with 1, 2 and 4
// waves per SIMD all running the same code (the situation of the chain kernel's slice loop). s_memtime ticks are
// shader cycles. hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum { DEP_FMA, IND_FMA, FMA_SALU, SALU_ONLY, FMA_LDS, FMA_BRANCH_NT, FMA_BRANCH_T, PK_FMA_DEP, FMA_SALU_3to1, FMA_NOP, FMA_WAITCNT };

template <int MIX, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mix(float* out, long long* cyc, int iters)
{
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f, d0 = 0, d1 = 0;
    int s0 = 1, s1 = 2;
    const uint32_t la = (threadIdx.x & 1023) * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MIX == DEP_FMA) {
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (MIX == IND_FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (MIX == FMA_SALU) { // 64 fma + 64 s_add, alternating
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %11, %11, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %11, %11, 1\n"
                              "v_fma_f32 %4, %4, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 %11, %11, 1\n v_fma_f32 %6, %6, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 %11, %11, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s0), "s"(s1) : "scc");
        } else if constexpr (MIX == FMA_SALU_3to1) { // 48 fma + 16 s_add
            asm volatile(REP8("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %6, %7\n v_fma_f32 %4, %4, %6, %7\n v_fma_f32 %5, %5, %6, %7\n s_add_u32 %9, %9, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b), "v"(c), "s"(s0), "s"(s1) : "scc");
        } else if constexpr (MIX == SALU_ONLY) {
            asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s0) : : "scc");
        } else if constexpr (MIX == FMA_LDS) { // 56 fma + 8 ds_read_b32 (waited at the end)
            asm volatile(REP8("ds_read_b32 %8, %9\n v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n"
                              "v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(d0) : "v"(la), "v"(b), "v"(c));
        } else if constexpr (MIX == FMA_BRANCH_NT) { // 56 fma + 8 never-taken branches
            asm volatile(REP8("v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n"
                              "v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n s_cbranch_execz 0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(b), "v"(c));
        } else if constexpr (MIX == FMA_BRANCH_T) { // 56 fma + 8 taken branches (to the next instruction)
            asm volatile(REP8("v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n"
                              "v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n s_branch 0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(b), "v"(c));
        } else if constexpr (MIX == PK_FMA_DEP) { // 32 dependent v_pk_fma (2 chains alternating = 64 lanes-values of work as 32 instrs)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f p0 = {a0, a1}, p1 = {a2, a3}, pb = {b, b}, pc = {c, c};
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")
                         : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if constexpr (MIX == FMA_NOP) { // 56 fma + 8 s_nop 0
            asm volatile(REP8("v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n"
                              "v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n s_nop 0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(b), "v"(c));
        } else if constexpr (MIX == FMA_WAITCNT) { // 56 fma + 8 s_waitcnt (nothing outstanding)
            asm volatile(REP8("v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n"
                              "v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(b), "v"(c));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0 + d1 + (float) (s0 + s1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MIX, int WAVES>
static void run(const char* name, int instr_per_iter)
{
    float* out; long long* cyc;
    const int blocks = 256, iters = 2048;
    hipMalloc(&out, blocks * WAVES * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_mix<MIX, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += v;
    mean /= 256;
    const double per_simd = (double) iters * instr_per_iter * (WAVES / 4.0);
    std::printf("%-44s %d waves/SIMD: %6.2f cycles per instruction per SIMD (%.2f per wave)\n", name, WAVES / 4, mean / per_simd, mean / ((double) iters * instr_per_iter));
    hipFree(out); hipFree(cyc);
}
#define RUN3(MIX, name, n) run<MIX, 4>(name, n); run<MIX, 8>(name, n); run<MIX, 16>(name, n);
int main()
{
    RUN3(DEP_FMA, "64 dependent v_fma", 64)
    RUN3(IND_FMA, "64 independent v_fma (8 chains)", 64)
    RUN3(PK_FMA_DEP, "32 v_pk_fma (2 chains)", 32)
    RUN3(SALU_ONLY, "64 dependent s_add", 64)
    RUN3(FMA_SALU, "64 v_fma + 64 s_add alternating", 128)
    RUN3(FMA_SALU_3to1, "48 v_fma + 16 s_add", 64)
    RUN3(FMA_LDS, "56 v_fma + 8 ds_read_b32 + waitcnt", 65)
    RUN3(FMA_BRANCH_NT, "56 v_fma + 8 branches not taken", 64)
    RUN3(FMA_BRANCH_T, "56 v_fma + 8 branches taken", 64)
    RUN3(FMA_NOP, "56 v_fma + 8 s_nop", 64)
    RUN3(FMA_WAITCNT, "56 v_fma + 8 s_waitcnt", 64)
    return 0;
}
