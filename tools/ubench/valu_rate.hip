// Microbenchmark: cycles per wave64 VALU instruction on one SIMD, scalar fp32 FMA vs packed fp32 FMA.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_scalar(float* out, long long* cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_packed(float* out, long long* cyc, int iters)
{
    v2f a0 = {(float) threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const v2f b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_elementwise_fma(a0, b, c); a1 = __builtin_elementwise_fma(a1, b, c);
            a2 = __builtin_elementwise_fma(a2, b, c); a3 = __builtin_elementwise_fma(a3, b, c);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const v2f s = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K>
static void run(const char* name, K kernel, int waves_per_block, int instr_per_iter)
{
    float* out; long long* cyc;
    const int blocks = 256, iters = 4096;
    hipMalloc(&out, blocks * waves_per_block * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += v;
    mean /= 256;
    // s_memtime ticks at a fixed 100 MHz; report relative numbers and per-instruction time via the event clock instead
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, iters); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double) iters * instr_per_iter * (waves_per_block / 4.0); // waves of a block spread over 4 SIMDs
    std::printf("%-28s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (memtime ticks %.0f)\n", name, waves_per_block / 4, ms,
                ms * 1e6 / instr_per_simd, mean);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run("v_fma_f32 (64 per iter)", k_scalar<4>, 4, 64);
    run("v_fma_f32 (64 per iter)", k_scalar<8>, 8, 64);
    run("v_fma_f32 (64 per iter)", k_scalar<16>, 16, 64);
    run("v_pk_fma_f32 (32 per iter)", k_packed<4>, 4, 32);
    run("v_pk_fma_f32 (32 per iter)", k_packed<8>, 8, 32);
    run("v_pk_fma_f32 (32 per iter)", k_packed<16>, 16, 32);
    return 0;
}
