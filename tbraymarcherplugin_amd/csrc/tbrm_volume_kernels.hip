// tbrm_volume_kernels.hip — the gfx950 kernels that prepare a volume rather than render it: the empty-space-skipping metadata
// (k_brick_minmax, k_brick_empty, k_shell_transparent, k_brick_dist: no reference counterpart — skipped samples are exactly the
// ones whose corrected opacity is 0, so results are unchanged), dense <-> bricked relayout (k_relayout) and the device-side
// self-tests of the UNORM conversions. Compiled with -ffp-contract=off; see tbrm_device_math.h for the arithmetic contract.
#include "tbrm_device_sampling.h"

#include <algorithm>
#include <type_traits>

namespace tbrm {

// ------------------------------------------------------------------------------------------------------------
// empty-space-skipping metadata

// Per brick b: min/max of every voxel a sample whose base tap lies in b can touch: [8b, 8b+8] per axis,
// addressed like the raymarch sampler. NaN voxels poison the range to [-inf, +inf] (never skipped).
template <int FMT, int MODE>
__global__ __launch_bounds__(64) void k_brick_minmax(const BrickParams p)
{
    const int b = blockIdx.x;
    const int bx = b % p.bnx, by = (b / p.bnx) % p.bny, bz = b / (p.bnx * p.bny);
    if (bz < p.bz0 || bz >= p.bz1) { // slab-resident volumes: not held here
        if (threadIdx.x == 0) p.minmax[b] = make_float2(-__builtin_inff(), __builtin_inff());
        return;
    }
    float mn = __builtin_inff(), mx = -__builtin_inff();
    bool nan = false;
    for (int t = threadIdx.x; t < 9 * 9 * 9; t += 64) {
        const int dx = t % 9, dy = (t / 9) % 9, dz = t / 81;
        int x = bx * kBrick + dx, y = by * kBrick + dy, z = bz * kBrick + dz;
        // the +8 tap only exists as the "+1" neighbour of an in-range base tap
        if (x > p.data.nx || y > p.data.ny || z > p.data.nz) continue;
        x = address<MODE>(x, p.data.nx);
        y = address<MODE>(y, p.data.ny);
        z = address<MODE>(z, p.data.nz);
        const float v = load_voxel<FMT>(p.data.data, brick_off(x, y, z, p.data.bnx, p.data.bnxy));
        if (v != v) nan = true;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o, 64));
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
        nan = nan || __shfl_down((int) nan, o, 64);
    }
    if (threadIdx.x == 0) p.minmax[b] = nan ? make_float2(-__builtin_inff(), __builtin_inff()) : make_float2(mn, mx);
}

hipError_t launch_brick_minmax(const BrickParams& p, hipStream_t s)
{
    const int n = p.bnx * p.bny * p.bnz;
    if (n == 0) return hipSuccess;
#define TBRM_BM(F, M) hipLaunchKernelGGL((k_brick_minmax<F, M>), dim3(n), dim3(64), 0, s, p)
    const bool clamp = p.addr_mode == ADDR_CLAMP;
    switch (p.data.fmt) {
        case FMT_U8: if (clamp) TBRM_BM(FMT_U8, ADDR_CLAMP); else TBRM_BM(FMT_U8, ADDR_WRAP); break;
        case FMT_U16: if (clamp) TBRM_BM(FMT_U16, ADDR_CLAMP); else TBRM_BM(FMT_U16, ADDR_WRAP); break;
        default: if (clamp) TBRM_BM(FMT_F32, ADDR_CLAMP); else TBRM_BM(FMT_F32, ADDR_WRAP); break;
    }
#undef TBRM_BM
    return hipGetLastError();
}

// A brick is empty when every value in [min,max] maps to corrected opacity 0: the TF position is monotone in
// the value (width > 0), so it suffices that the part of [pos(min), pos(max)] that survives the cutoffs only
// touches TF texels with alpha <= 0.
__device__ __forceinline__ bool range_maps_to_zero_opacity(float vmin, float vmax, const WindowDev& win, const int* alpha_prefix)
{
    if (!(win.width > 0.0f && vmin <= vmax && vmin > -__builtin_inff() && vmax < __builtin_inff())) return false;
    float lo = tf_position(vmin, win.center, win.width);
    float hi = tf_position(vmax, win.center, win.width);
    if (!(lo == lo && hi == hi)) return false;
    bool all_cut = false;
    if (win.low_cutoff > 0.0f) {
        if (hi < 0.0f) all_cut = true;
        lo = fmaxf(lo, 0.0f);
    }
    if (win.high_cutoff > 0.0f) {
        if (lo > 1.0f) all_cut = true;
        hi = fminf(hi, 1.0f);
    }
    if (all_cut) return true;
    int i_lo, i_hi;
    float f;
    texel_split(lo, 256.0f, i_lo, f);
    texel_split(hi, 256.0f, i_hi, f);
    i_lo = min(max(i_lo, 0), 255);
    i_hi = min(max(i_hi + 1, 0), 255);
    return (alpha_prefix[i_hi + 1] - alpha_prefix[i_lo]) == 0;
}

__global__ __launch_bounds__(256) void k_brick_empty(const EmptyParams p)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    bool empty = false;
    if (b < p.n_bricks) {
        const float2 mm = p.minmax[b];
        empty = range_maps_to_zero_opacity(mm.x, mm.y, p.win, p.alpha_prefix);
    }
    const unsigned long long m = __ballot(empty);
    const int lane = threadIdx.x & 63;
    if (b < p.n_bricks || true) {
        if (lane == 0) p.bits[(blockIdx.x * 256 + threadIdx.x) >> 5] = (uint32_t) m;
        if (lane == 32) p.bits[(blockIdx.x * 256 + threadIdx.x) >> 5] = (uint32_t) (m >> 32);
    }
}

hipError_t launch_brick_empty(const EmptyParams& p, hipStream_t s)
{
    if (p.n_bricks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_empty, dim3((p.n_bricks + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

// Is the volume's shell transparent to the light shaders? *flag (preset to 1) is cleared when some brick of the outer brick
// layer can produce a non-zero opacity from a blend of its values (min/max over the brick and its apron) with the data
// sampler's border colour. When it stays 1, every sample of a light pass whose position lies outside the unit cube — taps
// in that layer and beyond it — has CurrentSample exactly 0 under the Change shader's rules (ChangeDirLightShader.usf
// samples unconditionally), which is what the Add shader's uvw == saturate(uvw) guard makes it (AddDirLightShader.usf:98):
// the two shaders then propagate the same values for a light, and the contribution cache may serve either from the other's.
__global__ __launch_bounds__(256) void k_shell_transparent(const EmptyParams p, int bnx, int bny, int bnz, float border, int* flag)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= p.n_bricks) return;
    const int bx = b % bnx, by = (b / bnx) % bny, bz = b / (bnx * bny);
    if (bx != 0 && bx != bnx - 1 && by != 0 && by != bny - 1 && bz != 0 && bz != bnz - 1) return;
    const float2 mm = p.minmax[b];
    if (!range_maps_to_zero_opacity(fminf(mm.x, border), fmaxf(mm.y, border), p.win, p.alpha_prefix)) *flag = 0;
}

hipError_t launch_shell_transparent(const EmptyParams& p, int bnx, int bny, int bnz, float border, int* flag, hipStream_t s)
{
    if (p.n_bricks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_shell_transparent, dim3((p.n_bricks + 255) / 256), dim3(256), 0, s, p, bnx, bny, bnz, border, flag);
    return hipGetLastError();
}

// ---- empty-space leaping: Chebyshev distance (in bricks) to the nearest non-empty brick -------------------------
// D(b) = the largest t <= kSkipDistCap such that every brick within Chebyshev distance < t of b is empty (0: b itself
// is not). Erosion by a cube is separable, so three 1D passes give the exact value: T_x = distance along x to the
// nearest non-empty brick; T_xy(b) = max{t : T_x(b + dy) >= t for all |dy| < t}; the same along z. Neighbour indices
// follow the raymarch sampler's addressing (wrap: a torus; clamp: the edge brick repeats).
template <int MODE>
__global__ __launch_bounds__(256) void k_brick_dist(const DistParams p)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int nb = p.bn[0] * p.bn[1] * p.bn[2];
    if (b >= nb) return;
    int c[3] = {b % p.bn[0], (b / p.bn[0]) % p.bn[1], b / (p.bn[0] * p.bn[1])};
    const int stride = p.axis == 0 ? 1 : (p.axis == 1 ? p.bn[0] : p.bn[0] * p.bn[1]);
    const int n = p.axis == 0 ? p.bn[0] : (p.axis == 1 ? p.bn[1] : p.bn[2]);
    const int c0 = p.axis == 0 ? c[0] : (p.axis == 1 ? c[1] : c[2]);
    const int row = b - c0 * stride;
    auto value = [&](int ci) -> int { // the previous pass's T at coordinate ci of this row (pass 0: 0 / cap from the bits)
        ci = address<MODE>(ci, n);
        const int q = row + ci * stride;
        if (p.in) return p.in[q];
        return ((p.bits[q >> 5] >> (q & 31)) & 1u) ? kSkipDistCap : 0;
    };
    int t = value(c0);
    for (int d = 1; d < t; ++d) {
        const int m = min(value(c0 - d), value(c0 + d));
        t = min(t, max(m, d));
    }
    p.out[b] = (uint8_t) t;
}

hipError_t launch_brick_dist(const DistParams& p, int addr_mode, hipStream_t s)
{
    const int nb = p.bn[0] * p.bn[1] * p.bn[2];
    const dim3 grid((nb + 255) / 256), block(256);
    if (addr_mode == ADDR_CLAMP) hipLaunchKernelGGL(k_brick_dist<ADDR_CLAMP>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_brick_dist<ADDR_WRAP>, grid, block, 0, s, p);
    return hipGetLastError();
}

__global__ void k_selftest_decode(float* u8, float* u16)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < 256) u8[c] = decode_u8(c);
    if (c < 65536) u16[c] = decode_u16(c);
}
__global__ void k_selftest_roundtrip(const float* in, float* out, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = through_format<FMT_U8>(in[i]);
}
hipError_t launch_selftest_roundtrip(const float* d_in, float* d_out, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_roundtrip, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, d_in, d_out, n);
    return hipGetLastError();
}
hipError_t launch_selftest_decode(float* d_u8, float* d_u16, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_decode, dim3(256), dim3(256), 0, s, d_u8, d_u16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// linear (UVolumeTexture mip, x fastest) <-> bricked. One workgroup moves a run of 16 bricks along x (128 x 8 x 8 voxels)
// through LDS: the linear side is read / written as 64 rows of 128 consecutive voxels (128-512 bytes each, consecutive lanes
// on consecutive voxels), the bricked side as ONE contiguous run of 16 bricks with 16-byte accesses (bricks that follow each
// other along x follow each other in memory). (Round 1 moved one brick per workgroup, i.e. 8-voxel row fragments on the linear
// side: 8x read amplification by the FETCH_SIZE counter.)
constexpr int kRelayoutSeg = 16; // bricks per workgroup
template <typename E>
__global__ __launch_bounds__(256) void k_relayout(const RelayoutParams p)
{
    __shared__ __attribute__((aligned(16))) E s_run[kRelayoutSeg * 512];
    const int segs = (p.bnx + kRelayoutSeg - 1) / kRelayoutSeg;
    const int b = blockIdx.x;
    const int seg = b % segs, by = (b / segs) % (p.bnxy / p.bnx), bz = b / (segs * (p.bnxy / p.bnx));
    const int bx0 = seg * kRelayoutSeg, nb = min(kRelayoutSeg, p.bnx - bx0);
    E* bricked = (E*) (p.to_bricks ? p.dst : const_cast<void*>(p.src)) + ((size_t) bz * p.bnxy + (size_t) by * p.bnx + bx0) * 512;
    E* linear = (E*) (p.to_bricks ? const_cast<void*>(p.src) : p.dst);
    constexpr int V = 16 / (int) sizeof(E); // voxels per 16-byte access
    const int run = nb * 512;
    if (!p.to_bricks) { // bricked -> LDS
        for (int i = threadIdx.x * V; i < run; i += 256 * V) *(uint4*) (s_run + i) = *(const uint4*) (bricked + i);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 64 * 128; i += 256) { // row (y, z) of the run, voxel xl along it
        const int row = i >> 7, xl = i & 127;
        const int x = bx0 * 8 + xl, y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
        if ((xl >> 3) >= nb) continue;
        const bool in = x < p.nx && y < p.ny && z < p.nz;
        const size_t li = ((size_t) z * p.ny + y) * (size_t) p.nx + x;
        const int si = (xl >> 3) * 512 + (row << 3) + (xl & 7);
        if (p.to_bricks) s_run[si] = in ? linear[li] : E(0); // padding voxels are zeroed
        else if (in) linear[li] = s_run[si];
    }
    if (p.to_bricks) { // LDS -> bricked
        __syncthreads();
        for (int i = threadIdx.x * V; i < run; i += 256 * V) *(uint4*) (bricked + i) = *(const uint4*) (s_run + i);
    }
}

hipError_t launch_relayout(const RelayoutParams& p, hipStream_t s)
{
    const int segs = (p.bnx + kRelayoutSeg - 1) / kRelayoutSeg;
    const int n = segs * (p.bnxy / (p.bnx > 0 ? p.bnx : 1)) * p.bnz;
    if (n == 0) return hipSuccess;
    if (p.elem_bytes == 1) hipLaunchKernelGGL(k_relayout<uint8_t>, dim3(n), dim3(256), 0, s, p);
    else if (p.elem_bytes == 2) hipLaunchKernelGGL(k_relayout<uint16_t>, dim3(n), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_relayout<uint32_t>, dim3(n), dim3(256), 0, s, p);
    return hipGetLastError();
}

} // namespace tbrm
