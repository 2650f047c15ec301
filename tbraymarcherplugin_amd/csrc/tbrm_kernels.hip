// tbrm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the raymarch + illumination hot path.
//
//   (the illumination kernels live in tbrm_light_kernels.hip)
//   k_raymarch_lit     : PerformRaymarchCubeSetup + PerformWindowedLitRaymarch
//                        (RaymarchMaterialCommon.usf:23-78, WindowedRaymarchMaterials.usf:21-96).
//   k_fill             : ClearTextureShader.usf:12-16 / ClearVolumeTextureShader.usf:14-20.
//   k_brick_minmax/k_brick_empty : empty-space-skipping metadata (no reference counterpart; skipped samples are
//                        exactly the ones whose corrected opacity is 0, so results are unchanged).
//
// Compiled with -ffp-contract=off; see tbrm_device_math.h for the arithmetic contract.
#include "tbrm_device_sampling.h"

namespace tbrm {

// ------------------------------------------------------------------------------------------------------------
// fill

template <int FMT>
__global__ void k_fill(void* dst, size_t n, float value)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) store_voxel<FMT>(dst, i, value);
}

hipError_t launch_fill(void* dst, int fmt, size_t n, float value, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (fmt == FMT_U8) {
        // every byte gets the same UNORM8 code -> a plain memset (host replicates encode_u8)
        float x = value;
        if (x != x) x = 0.0f;
        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        const int code = (int) (x * 255.0f + 0.5f);
        return hipMemsetAsync(dst, code, n, s);
    }
    const int block = 256;
    const size_t want = (n + block - 1) / block;
    const int grid = (int) (want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_fill<FMT_F32>, dim3(grid), dim3(block), 0, s, dst, n, value);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// raymarch

__device__ __forceinline__ void rand3d_pcg16(int px, int py, int pz, uint32_t& ox)
{
    uint32_t x = (uint32_t) px, y = (uint32_t) py, z = (uint32_t) pz;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x += y * z; y += z * x; z += x * y;
    ox = x >> 16;
}

struct Ray {
    float pos[3];
    float lcv[3];
    float thickness;
};

__device__ __forceinline__ void normalize3(float* v)
{
    const float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}

// PerformRaymarchCubeSetup (RaymarchMaterialCommon.usf:23-69) for framebuffer pixel (px,py).
__device__ __forceinline__ void cube_setup(const RayParams& p, int px, int py, Ray& ray)
{
    const float sx = (((2.0f * ((float) px + 0.5f)) / (float) p.width) - 1.0f) * p.thx;
    const float sy = (1.0f - ((2.0f * ((float) py + 0.5f)) / (float) p.height)) * p.thy;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (p.fwd[c] + p.right[c] * sx) + p.up[c] * sy;
    normalize3(d);
    const float camvec[3] = {-d[0], -d[1], -d[2]};
    float lcp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lcp[c] = ((p.cam_pos[0] * p.m[0 + c] + p.cam_pos[1] * p.m[3 + c]) + p.cam_pos[2] * p.m[6 + c]) + p.m[9 + c];
        ray.lcv[c] = (camvec[0] * p.m[0 + c] + camvec[1] * p.m[3 + c]) + camvec[2] * p.m[6 + c];
    }
    normalize3(ray.lcv);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ray.lcv[c] = -ray.lcv[c];
        lcp[c] = lcp[c] + 0.5f;
    }
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { // RayAABBIntersection (RaymarcherCommon.usf:66-88), box [0,1]^3
        const float inv = 1.0f / ray.lcv[c];
        const float tmin = (0.0f - lcp[c]) * inv, tmax = (1.0f - lcp[c]) * inv;
        const float lo = fminf(tmax, tmin), hi = fmaxf(tmax, tmin);
        if (c == 0) { t0 = lo; t1 = hi; }
        else { t0 = fmaxf(t0, lo); t1 = fminf(t1, hi); }
    }
    t0 = fmaxf(0.0f, t0);
    if (p.depth) {
        float nv[3] = {camvec[0], camvec[1], camvec[2]};
        normalize3(nv);
        const float depth = p.depth[(size_t) py * p.width + px];
        const float wdv[3] = {nv[0] * depth, nv[1] * depth, nv[2] * depth};
        float ldv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) ldv[c] = (wdv[0] * p.m[0 + c] + wdv[1] * p.m[3 + c]) + wdv[2] * p.m[6 + c];
        float lsd = sqrtf((ldv[0] * ldv[0] + ldv[1] * ldv[1]) + ldv[2] * ldv[2]);
        lsd = lsd / fabsf((p.fwd[0] * camvec[0] + p.fwd[1] * camvec[1]) + p.fwd[2] * camvec[2]);
        t1 = fminf(lsd, t1);
    }
    ray.thickness = fmaxf(0.0f, t1 - t0);
#pragma unroll
    for (int c = 0; c < 3; ++c) ray.pos[c] = lcp[c] + (t0 * ray.lcv[c]);
}

__device__ __forceinline__ bool tile_pixel(const RayParams& p, int& i, int& j, int& px, int& py)
{
    // 16x16 pixel block per workgroup, one 8x8 sub-tile per wave64 (coherent rays per wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    j = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    px = p.tile_x0 + i;
    py = p.tile_y0 + (j >> 3) * 8 * p.row_group_step + (j & 7);
    return i < p.tile_w && j < p.tile_h;
}

template <int DFMT, int LFMT, int DMODE>
__global__ __launch_bounds__(256) void k_raymarch_lit(const RayParams p)
{
    __shared__ float4 s_tf[256];
    s_tf[threadIdx.x] = p.tf[threadIdx.x];
    __syncthreads();

    int i, j, px, py;
    const bool valid = tile_pixel(p, i, j, px, py);
    if (!valid) return;

    Ray ray;
    cube_setup(p, px, py, ray);

    // PerformWindowedLitRaymarch (WindowedRaymarchMaterials.usf:36-96)
    const float step_size = 1 / p.steps;
    const float actual = p.steps * ray.thickness;
    const float fl = floorf(actual);
    const int max_steps = (int) fl;
    const float final_step = actual - fl;
    const float sv0 = ray.lcv[0] * step_size, sv1 = ray.lcv[1] * step_size, sv2 = ray.lcv[2] * step_size;
    const float step_world = 100.0f * step_size;
    float pos0 = ray.pos[0], pos1 = ray.pos[1], pos2 = ray.pos[2];
    if (p.jitter_frame >= 0) { // JitterEntryPos (RaymarchMaterialCommon.usf:73-78)
        uint32_t r;
        rand3d_pcg16(px, py, p.jitter_frame & 7, r);
        const float rnd = (float) r / 65535.0f;
        pos0 = pos0 - (sv0 * rnd); pos1 = pos1 - (sv1 * rnd); pos2 = pos2 - (sv2 * rnd);
    }

    const float nx = (float) p.data.nx, ny = (float) p.data.ny, nz = (float) p.data.nz;
    const float lnx = (float) p.lv_dims[0], lny = (float) p.lv_dims[1], lnz = (float) p.lv_dims[2];
    const VolumeDev lightv{p.light, p.lv_dims[0], p.lv_dims[1], p.lv_dims[2], LFMT, p.lv_bnx, p.lv_bnxy};
    float le0 = 0.0f, le1 = 0.0f, le2 = 0.0f, le3 = 0.0f;
    int cached_brick = -1;
    bool cached_empty = false;

    // the light volume shares the data volume's footprint when it has the same size and the position is inside the
    // cube (saturate(CurPos) == CurPos): same texel split, same wrapped indices, same brick offsets
    const bool same_grid = DMODE == ADDR_WRAP && p.share_grid;

    // A sample is split into "issue" (position -> clip / empty-space test -> 16 tap loads) and "shade" (filter, window,
    // transfer function, opacity correction, light, accumulate). Positions do not depend on shading, so the loads of
    // sample k+1 are issued before sample k is shaded: the two dependent memory round trips of the reference's loop body
    // (data fetch, then light fetch) disappear behind the previous sample's arithmetic.
    struct Pending {
        bool valid;        // there is a sample
        bool skip;         // clipped, or inside a brick that maps to opacity 0: exact no-op
        float step;        // StepSize for the opacity correction
        float fx, fy, fz;  // data-volume filter weights
        float gx, gy, gz;  // light-volume filter weights
        RawTaps<DFMT> d;
        RawTaps<LFMT> l;
    };
    auto issue = [&](float step, Pending& n) {
        n.valid = true;
        n.skip = true;
        n.step = step;
        if (p.clip_mode && is_clipped(pos0, pos1, pos2, p.cc, p.cd)) return;
        int ix, iy, iz;
        texel_split(pos0, nx, ix, n.fx);
        texel_split(pos1, ny, iy, n.fy);
        texel_split(pos2, nz, iz, n.fz);
        if (p.empty_bits) {
            const int bx = address<DMODE>(ix, p.data.nx) >> kBrickShift;
            const int by = address<DMODE>(iy, p.data.ny) >> kBrickShift;
            const int bz = address<DMODE>(iz, p.data.nz) >> kBrickShift;
            const int b = (bz * p.bny + by) * p.bnx + bx;
            if (b != cached_brick) {
                cached_brick = b;
                cached_empty = (p.empty_bits[b >> 5] >> (b & 31)) & 1u;
            }
            if (cached_empty) return; // every tap of this sample maps to opacity 0
        }
        n.skip = false;
        const TapOffsets dt = tap_offsets<DMODE>(p.data, ix, iy, iz);
        n.d.issue(p.data.data, dt);
        // LightVolume.SampleLevel(Wrap, saturate(CurPos)) (WindowedRaymarchMaterials.usf:30)
        const float sp0 = saturate_(pos0), sp1 = saturate_(pos1), sp2 = saturate_(pos2);
        if (same_grid && sp0 == pos0 && sp1 == pos1 && sp2 == pos2) {
            n.gx = n.fx; n.gy = n.fy; n.gz = n.fz;
            n.l.issue(p.light, dt);
        } else {
            int lx, ly, lz;
            texel_split(sp0, lnx, lx, n.gx);
            texel_split(sp1, lny, ly, n.gy);
            texel_split(sp2, lnz, lz, n.gz);
            n.l.issue(p.light, tap_offsets<ADDR_WRAP>(lightv, lx, ly, lz));
        }
    };
    // returns true when the early-exit threshold was crossed
    auto shade = [&](const Pending& c) -> bool {
        if (c.skip) return false;
        const float v = c.d.filter(c.fx, c.fy, c.fz);
        // SampleWindowedTransferFunction (WindowedSampling.usf:20-37)
        const float tpos = tf_position(v, p.win.center, p.win.width);
        if ((tpos < 0.0f && p.win.low_cutoff > 0.0f) || (tpos > 1.0f && p.win.high_cutoff > 0.0f)) return false;
        float4 cs = sample_tf(s_tf, tpos);
        const float a_sat = saturate_(cs.w);
        if (a_sat == 0.0f) return false; // 1 - pow(1, s) = 0: the sample contributes exactly nothing
        const float a = 1.0f - pow_(1.0f - a_sat, c.step);
        const float l = c.l.filter(c.gx, c.gy, c.gz);
        cs.x = cs.x * l; cs.y = cs.y * l; cs.z = cs.z * l;
        // AccumulateLightEnergy (RaymarchMaterialCommon.usf:82-88)
        const float om = 1.0f - le3;
        le0 = le0 + ((cs.x * a) * om);
        le1 = le1 + ((cs.y * a) * om);
        le2 = le2 + ((cs.z * a) * om);
        le3 = le3 + (a * om);
        return le3 > 0.95f;
    };
    // sample idx of the ray: the max_steps full steps (CurPos += LocalCamVec before sampling, :67), then the fractional one
    auto advance_and_issue = [&](int idx, Pending& n) {
        if (idx < max_steps) {
            pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2;
            issue(step_world, n);
        } else if (idx == max_steps && final_step > 0.0f) { // :84-93
            pos0 = pos0 + (sv0 * final_step); pos1 = pos1 + (sv1 * final_step); pos2 = pos2 + (sv2 * final_step);
            issue(100.0f * final_step, n);
        } else n.valid = false;
    };

    Pending pa, pb;
    int k = 0;
    advance_and_issue(0, pa);
    for (;;) { // two samples per trip so the pending buffers swap roles without copies
        if (!pa.valid) break;
        advance_and_issue(k + 1, pb);
        if (shade(pa) && k < max_steps) { le3 = 1.0f; break; } // early exit belongs to the full steps only (:75-79)
        ++k;
        if (!pb.valid) break;
        advance_and_issue(k + 1, pa);
        if (shade(pb) && k < max_steps) { le3 = 1.0f; break; }
        ++k;
    }
    reinterpret_cast<float4*>(p.out)[(size_t) j * p.tile_w + i] = make_float4(le0, le1, le2, le3);
}

template <int DFMT, int LFMT>
static hipError_t launch_ray2(const RayParams& p, hipStream_t s)
{
    const dim3 grid((p.tile_w + 15) / 16, (p.tile_h + 15) / 16), block(256);
    if (p.data_addr_mode == ADDR_CLAMP) hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_CLAMP>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_WRAP>), grid, block, 0, s, p);
    return hipGetLastError();
}
template <int DFMT>
static hipError_t launch_ray1(const RayParams& p, hipStream_t s)
{
    return p.lv_fmt == FMT_U8 ? launch_ray2<DFMT, FMT_U8>(p, s) : launch_ray2<DFMT, FMT_F32>(p, s);
}
hipError_t launch_raymarch(const RayParams& p, hipStream_t s)
{
    if (p.tile_w <= 0 || p.tile_h <= 0) return hipSuccess;
    switch (p.data.fmt) {
        case FMT_U8: return launch_ray1<FMT_U8>(p, s);
        case FMT_U16: return launch_ray1<FMT_U16>(p, s);
        default: return launch_ray1<FMT_F32>(p, s);
    }
}

// Nominal samples: sum over rays of floor(Steps*thickness) + [frac > 0].
__global__ __launch_bounds__(256) void k_count_samples(const RayParams p)
{
    int i, j, px, py;
    const bool valid = tile_pixel(p, i, j, px, py);
    unsigned long long n = 0;
    if (valid) {
        Ray ray;
        cube_setup(p, px, py, ray);
        const float actual = p.steps * ray.thickness;
        const float fl = floorf(actual);
        n = (unsigned long long) (int) fl + ((actual - fl) > 0.0f ? 1ull : 0ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(p.sample_counter, n);
}

hipError_t launch_count_samples(const RayParams& p, hipStream_t s)
{
    if (p.tile_w <= 0 || p.tile_h <= 0) return hipSuccess;
    const dim3 grid((p.tile_w + 15) / 16, (p.tile_h + 15) / 16), block(256);
    hipLaunchKernelGGL(k_count_samples, grid, block, 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// empty-space-skipping metadata

// Per brick b: min/max of every voxel a sample whose base tap lies in b can touch: [8b, 8b+8] per axis,
// addressed like the raymarch sampler. NaN voxels poison the range to [-inf, +inf] (never skipped).
template <int FMT, int MODE>
__global__ __launch_bounds__(64) void k_brick_minmax(const BrickParams p)
{
    const int b = blockIdx.x;
    const int bx = b % p.bnx, by = (b / p.bnx) % p.bny, bz = b / (p.bnx * p.bny);
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
__global__ __launch_bounds__(256) void k_brick_empty(const EmptyParams p)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    bool empty = false;
    if (b < p.n_bricks) {
        const float2 mm = p.minmax[b];
        if (p.win.width > 0.0f && mm.x <= mm.y && mm.x > -__builtin_inff() && mm.y < __builtin_inff()) {
            float lo = tf_position(mm.x, p.win.center, p.win.width);
            float hi = tf_position(mm.y, p.win.center, p.win.width);
            if (lo == lo && hi == hi) {
                bool all_cut = false;
                if (p.win.low_cutoff > 0.0f) {
                    if (hi < 0.0f) all_cut = true;
                    lo = fmaxf(lo, 0.0f);
                }
                if (p.win.high_cutoff > 0.0f) {
                    if (lo > 1.0f) all_cut = true;
                    hi = fminf(hi, 1.0f);
                }
                if (all_cut) empty = true;
                else {
                    int i_lo, i_hi;
                    float f;
                    texel_split(lo, 256.0f, i_lo, f);
                    texel_split(hi, 256.0f, i_hi, f);
                    i_lo = min(max(i_lo, 0), 255);
                    i_hi = min(max(i_hi + 1, 0), 255);
                    empty = (p.alpha_prefix[i_hi + 1] - p.alpha_prefix[i_lo]) == 0;
                }
            }
        }
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
// linear (UVolumeTexture mip, x fastest) <-> bricked. One workgroup per brick; the linear side is read/written as
// eight-voxel rows, the bricked side as one contiguous 512-voxel run.
template <typename E>
__global__ __launch_bounds__(256) void k_relayout(const RelayoutParams p)
{
    const int b = blockIdx.x;
    const int bx = b % p.bnx, by = (b / p.bnx) % (p.bnxy / p.bnx), bz = b / p.bnxy;
    E* bricked = (E*) (p.to_bricks ? p.dst : const_cast<void*>(p.src)) + (size_t) b * 512;
    E* linear = (E*) (p.to_bricks ? const_cast<void*>(p.src) : p.dst);
    for (int t = threadIdx.x; t < 512; t += 256) {
        const int x = bx * 8 + (t & 7), y = by * 8 + ((t >> 3) & 7), z = bz * 8 + (t >> 6);
        const bool in = x < p.nx && y < p.ny && z < p.nz;
        const size_t li = ((size_t) z * p.ny + y) * (size_t) p.nx + x;
        if (p.to_bricks) bricked[t] = in ? linear[li] : E(0);
        else if (in) linear[li] = bricked[t];
    }
}

hipError_t launch_relayout(const RelayoutParams& p, hipStream_t s)
{
    const int n = p.bnxy * p.bnz;
    if (n == 0) return hipSuccess;
    if (p.elem_bytes == 1) hipLaunchKernelGGL(k_relayout<uint8_t>, dim3(n), dim3(256), 0, s, p);
    else if (p.elem_bytes == 2) hipLaunchKernelGGL(k_relayout<uint16_t>, dim3(n), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_relayout<uint32_t>, dim3(n), dim3(256), 0, s, p);
    return hipGetLastError();
}

} // namespace tbrm
