// tbrm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the raymarch + illumination hot path.
//
//   (the illumination kernels live in tbrm_light_kernels.hip)
//   k_raymarch_lit     : PerformRaymarchCubeSetup + PerformWindowedLitRaymarch
//                        (RaymarchMaterialCommon.usf:23-78, WindowedRaymarchMaterials.usf:21-96).
//   k_fill             : ClearTextureShader.usf:12-16 / ClearVolumeTextureShader.usf:14-20.
//   (skipping metadata, relayout, self-tests: tbrm_volume_kernels.hip)
//
// Compiled with -ffp-contract=off; see tbrm_device_math.h for the arithmetic contract.
#include "tbrm_device_sampling.h"

#include <algorithm>
#include <type_traits>

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

// Pixel of the tile for ray (i, j) of a launch; rows interleave in groups of 8 across GPUs (tbrm_tile.row_group_step)
__device__ __forceinline__ bool tile_pixel_at(const RayParams& p, int i, int j, int& px, int& py)
{
    px = p.tile_x0 + i;
    py = p.tile_y0 + (j >> 3) * 8 * p.row_group_step + (j & 7);
    return i < p.tile_w && j < p.tile_h;
}

__device__ __forceinline__ bool tile_pixel(const RayParams& p, int& i, int& j, int& px, int& py)
{
    // 16x16 pixel block per workgroup, one 8x8 sub-tile per wave64 (count kernel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    j = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    return tile_pixel_at(p, i, j, px, py);
}

// ---- k_raymarch_lit ---------------------------------------------------------------------------------------------
// A ray is a serial loop in the reference (positions by repeated addition, front-to-back accumulation with early
// exit). One ray per lane makes a frame as slow as its longest ray: a 512-step ray is 512 dependent memory round trips
// while most of the chip has long run out of work (measured: 1.5 resident waves per SIMD on average, 14 % VALU issue).
// Here a wave marches 16 rays with 4 lanes each (or 8 with 8; a 4x4 / 4x2 pixel patch): per trip lane b of a ray evaluates
// sample L*trip + b
// — position, clip / empty-brick test, 16 taps, transfer function, opacity correction, light — so the expensive part of
// L consecutive samples runs side by side, and only the cheap part stays serial: the L lanes exchange their
// (colour*alpha, alpha) through LDS and each replays AccumulateLightEnergy over the L samples in ray order, which also
// decides the early exit exactly where the reference takes it. Arithmetic per sample and per accumulation step is the
// reference's; a lane reaches its sample position by performing every addition of the ray up to it.
//
// SLAB: one stage of a frame marched slab by slab (tbrm_raymarch_lit_slab_device). The handle owns light-volume slices
// [slab_z0, slab_z1); a sample belongs to the slab its (saturated) z position falls in, and along a ray those slabs come
// in order. The stage takes the ray's LightEnergy as the slabs before it left it (p.out, in place), accumulates the
// samples it owns exactly where the unpartitioned loop would, and hands the state on. Positions are still reached by
// performing every addition of the ray, so every sample, and the early exit, are bit for bit those of the whole march.
#ifdef TBRM_RAY_STATS // diagnostics build (tools/ray_stats.sh): how full the waves of the lit march are
__device__ unsigned long long g_ray_stats[6]; // trips of a wave through the loop, lanes not done, lanes sampling, trips in which any lane samples,
                                              // trips in which any lane has something to accumulate (alpha != 0), such lanes
extern "C" __attribute__((visibility("default"))) int tbrm_debug_ray_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_stats), sizeof(g_ray_stats)) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ray_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// TAB: the data volume's texel -> voxel-offset arithmetic (address mode, +1 tap, bricked offset, brick index of the leap-distance
// look-up: ~45 integer instructions per sample, more than the filter itself) comes out of three small tables the workgroup copies
// into LDS, one {voxel offset, brick-index part} pair per texel index -2 .. n + 1 and axis (tbrm_api.cpp build_ray_tables); a
// sample's base and +1 entries of an axis arrive with one 16-byte LDS read. The host picks it when no sample position can lie more than two texels outside the volume
// (a step of at most one texel: positions stay within one step of the unit cube) and the tables are small.
template <int DFMT, int LFMT, int DMODE, int kRayLanes, bool SLAB = false, bool TAB = false>
__global__ __launch_bounds__(256, 6) void k_raymarch_lit(const RayParams p) // 6 waves per SIMD (80 VGPRs): measured 3-8 % faster than 5 or 8
{
    static_assert(kRayLanes == 4 || kRayLanes == 8, "instantiated for 4 and 8 lanes per ray");
    static_assert(!(TAB && SLAB), "slab stages keep the arithmetic path (relocated layers)");
    extern __shared__ __attribute__((aligned(16))) uint2 s_tab[]; // TAB: [x | y | z], n + 4 entries each
    const uint2* const tab_x = s_tab;
    const uint2* const tab_y = tab_x + (TAB ? p.data.nx + 4 : 0);
    const uint2* const tab_z = tab_y + (TAB ? p.data.ny + 4 : 0);
    constexpr int PW = 4, PH = kRayLanes == 4 ? 4 : 2; // rays of a wave: a PW x PH pixel patch
    constexpr int kRayBlockW = 2 * PW, kRayBlockH = 2 * PH;
    constexpr int LSH = kRayLanes == 4 ? 2 : 3;
    __shared__ float4 s_tf[256];
    __shared__ float4 s_x[256]; // per lane: (colour * alpha, alpha) of its sample; alpha < 0: nothing to accumulate
    s_tf[threadIdx.x] = p.tf[threadIdx.x];
    if constexpr (!TAB) __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane & (kRayLanes - 1), r = lane >> LSH; // sample slot, ray within the wave
    // Workgroups are dealt to the 8 XCDs round-robin by linear id: with the plain (x, y) order the eight horizontal neighbours of a
    // pixel block — which march through the same bricks — sit behind eight different L2s. Bands of p.xcd_rows rows of blocks are dealt
    // to the XCDs instead (band 8 m + x to XCD x: every XCD still gets an even share of the silhouette), walked column by column; an
    // affinity for speed only.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_rows > 0 && (int) gridDim.y % (8 * p.xcd_rows) == 0) {
        const int id = by * (int) gridDim.x + bx, k = id >> 3, per_band = p.xcd_rows * (int) gridDim.x;
        const int band = k / per_band, kk = k - band * per_band;
        bx = kk / p.xcd_rows;
        by = (band * 8 + (id & 7)) * p.xcd_rows + (kk - bx * p.xcd_rows);
    }
    const int i = bx * kRayBlockW + (wave & 1) * PW + (r % PW);
    const int j = by * kRayBlockH + (wave >> 1) * PH + (r / PW);
    int px, py;
    const bool valid = tile_pixel_at(p, i, j, px, py);

    Ray ray;
    cube_setup(p, valid ? px : p.tile_x0, valid ? py : p.tile_y0, ray);

    // PerformWindowedLitRaymarch (WindowedRaymarchMaterials.usf:36-96)
    const float step_size = 1 / p.steps;
    const float actual = p.steps * ray.thickness;
    const float fl = floorf(actual);
    const int max_steps = valid ? (int) fl : 0;
    const float final_step = valid ? actual - fl : 0.0f;
    const int n_samples = max_steps + (final_step > 0.0f ? 1 : 0); // the full steps, then the fractional one (:84-93)
    if constexpr (TAB) { // the tables (the handle built them once: tbrm_resources::d_ray_tab), unless no ray of the workgroup meets the volume
        if (__syncthreads_or(n_samples > 0)) {
            const int n16 = (p.data.nx + p.data.ny + p.data.nz + 12 + 1) >> 1;
            for (int i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(s_tab)[i] = reinterpret_cast<const uint4*>(p.tab)[i];
            __syncthreads();
        }
    }
    const float sv0 = ray.lcv[0] * step_size, sv1 = ray.lcv[1] * step_size, sv2 = ray.lcv[2] * step_size;
    const float step_world = 100.0f * step_size;
    float pos0 = ray.pos[0], pos1 = ray.pos[1], pos2 = ray.pos[2];
    if (p.jitter_frame >= 0) { // JitterEntryPos (RaymarchMaterialCommon.usf:73-78)
        uint32_t rr;
        rand3d_pcg16(px, py, p.jitter_frame & 7, rr);
        const float rnd = (float) rr / 65535.0f;
        pos0 = pos0 - (sv0 * rnd); pos1 = pos1 - (sv1 * rnd); pos2 = pos2 - (sv2 * rnd);
    }

    const float nx = (float) p.data.nx, ny = (float) p.data.ny, nz = (float) p.data.nz;
    const float lnx = (float) p.lv_dims[0], lny = (float) p.lv_dims[1], lnz = (float) p.lv_dims[2];
    const VolumeDev lightv{p.light, p.lv_dims[0], p.lv_dims[1], p.lv_dims[2], LFMT, p.lv_bnx, p.lv_bnxy, p.lv_wrap_layer, p.lv_wrap_shift};
    // the light volume shares the data volume's footprint when it has the same size and the position is inside the
    // cube (saturate(CurPos) == CurPos): same texel split, same wrapped indices, same brick offsets
    const bool same_grid = DMODE == ADDR_WRAP && p.share_grid;

    // empty-space leaping: texels the position moves per full step along its fastest axis
    const float inv_texels_per_step = 1.0f / fmaxf(fmaxf(fabsf(sv0) * nx, fabsf(sv1) * ny), fabsf(sv2) * nz);
    int safe_until = -1; // this lane's samples with index <= safe_until are known to be based in empty bricks
    const bool wave_skip = p.wave_skip != 0 && p.skip_dist != nullptr;
    bool eager = false;  // the wave's last trip sampled nothing: this trip's lanes renew their ranges (wave-uniform)
    int renew_wait = 0;  // empty trips to let pass before the next renewal (a renewal that bought no blind trip is not repeated at once)

    float le0 = 0.0f, le1 = 0.0f, le2 = 0.0f, le3 = 0.0f; // LightEnergy, replicated in the 8 lanes of the ray
    bool done = n_samples == 0;
    bool mine = valid; // SLAB: this stage's sweep direction handles the ray (rays with local dz >= 0 go up through the slabs)
    if constexpr (SLAB) {
        mine = valid && (p.slab_dir == 0 || (p.slab_dir > 0) == (ray.lcv[2] >= 0.0f));
        if (mine) {
            const float4 in = reinterpret_cast<const float4*>(p.out)[(size_t) j * p.tile_w + i];
            le0 = in.x; le1 = in.y; le2 = in.z; le3 = in.w;
            done = done || le3 == 1.0f; // the early exit was taken in a slab before this one (it leaves alpha at exactly 1)
        } else done = true;
    }
    int adds = 0; // full-step additions this lane has applied to its position
    float4* const xs = s_x + (threadIdx.x & ~(kRayLanes - 1)); // the ray's 8 exchange slots

    for (int base = 0; __builtin_amdgcn_ballot_w64(!done) != 0; base += kRayLanes) {
        const int idx = base + b; // this lane's sample of the ray
        // CurPos += LocalCamVec before every full sample (:67): sample idx < max_steps sits idx+1 additions in, the
        // fractional sample max_steps additions plus one scaled step
        const int want = min(idx + 1, max_steps);
        if (__builtin_amdgcn_ballot_w64(!done && want - adds != kRayLanes) == 0) { // mid-ray everywhere: no predication
#pragma unroll
            for (int t = 0; t < kRayLanes; ++t) { pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2; }
            adds += kRayLanes;
        } else {
#pragma unroll
            for (int t = 0; t < kRayLanes; ++t)
                if (adds < want) { pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2; ++adds; }
        }
        float q0 = pos0, q1 = pos1, q2 = pos2, step = step_world;
        const bool is_full = idx < max_steps;
        const bool has = !done && idx < n_samples;
        if (!is_full) { q0 = pos0 + (sv0 * final_step); q1 = pos1 + (sv1 * final_step); q2 = pos2 + (sv2 * final_step); step = 100.0f * final_step; }

        // the sample: everything of the loop body up to AccumulateLightEnergy
        float4 x = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        bool live = has && idx > safe_until && !(p.clip_mode && is_clipped(q0, q1, q2, p.cc, p.cd));
        if constexpr (SLAB) { // only the samples of this handle's slab
            const int zi = min((int) (saturate_(q2) * lnz), p.lv_dims[2] - 1);
            live = live && zi >= p.slab_z0 && zi < p.slab_z1;
        }
        int ix = 0, iy = 0, iz = 0;
        float fx = 0.0f, fy = 0.0f, fz = 0.0f;
        // (after a trip in which no lane of the wave sampled, the lanes inside their proven-empty range look their brick up
        // again as well: all ranges then start from here, and the wave can take the trips they share in one go — below)
        const bool renew = eager && has && !live && idx <= safe_until;
        TapOffsets tab_dt{}; // TAB: the data taps' offsets, out of the tables
        if (live || renew) {
            if constexpr (TAB) { // (the host's promise behind the tables: positions within a step of the unit cube)
                texel_split_bounded(q0, nx, ix, fx);
                texel_split_bounded(q1, ny, iy, fy);
                texel_split_bounded(q2, nz, iz, fz);
            } else {
                texel_split(q0, nx, ix, fx);
                texel_split(q1, ny, iy, fy);
                texel_split(q2, nz, iz, fz);
            }
            uint32_t tab_brick = 0;
            if constexpr (TAB) { // (indices -2 .. n: the host's promise; the clamp only keeps a broken promise inside the tables)
                const int tx = min(max(ix + 2, 0), p.data.nx + 2), ty = min(max(iy + 2, 0), p.data.ny + 2), tz = min(max(iz + 2, 0), p.data.nz + 2);
                const uint2 ax = tab_x[tx], bx1 = tab_x[tx + 1], ay = tab_y[ty], by1 = tab_y[ty + 1], az = tab_z[tz], bz1 = tab_z[tz + 1];
                tab_dt.x0 = ax.x; tab_dt.x1 = bx1.x; tab_dt.y0 = ay.x; tab_dt.y1 = by1.x; tab_dt.z0 = az.x; tab_dt.z1 = bz1.x;
                tab_brick = ax.y + ay.y + az.y;
            }
            if (p.skip_dist) { // a sample based in a brick that maps every reachable value to opacity 0 is an exact no-op
                int dist;
                if constexpr (TAB) dist = p.skip_dist[tab_brick];
                else {
                    const int bx = address<DMODE>(ix, p.data.nx) >> kBrickShift;
                    const int by = address<DMODE>(iy, p.data.ny) >> kBrickShift;
                    const int bz = address<DMODE>(iz, p.data.nz) >> kBrickShift;
                    dist = p.skip_dist[(bz * p.bny + by) * p.bnx + bx];
                }
                live = live && dist == 0;
                // Every brick within Chebyshev distance < dist is empty as well. From anywhere inside this brick a base
                // tap has to move more than 8*(dist-1) texels along some axis to leave them, and a base tap moves at
                // most 1 texel more than the position does: the lane's samples up to that many steps ahead need no test.
                if (dist >= 2) safe_until = max(safe_until, idx + (int) fminf(((float) (8 * (dist - 1)) - 1.25f) * inv_texels_per_step, 1.0e6f));
            }
        }
        const bool any_live = !wave_skip || __builtin_amdgcn_ballot_w64(live) != 0;
#ifdef TBRM_RAY_STATS
        {
            const unsigned long long nd = __builtin_popcountll(__builtin_amdgcn_ballot_w64(!done)), nl = __builtin_popcountll(__builtin_amdgcn_ballot_w64(live));
            if (lane == 0) { atomicAdd(&g_ray_stats[0], 1ull); atomicAdd(&g_ray_stats[1], nd); atomicAdd(&g_ray_stats[2], nl); if (nl) atomicAdd(&g_ray_stats[3], 1ull); }
        }
#endif
        if (live) {
            RawTaps<DFMT> dtaps;
            RawTaps<LFMT> ltaps;
            float gx, gy, gz;
            TapOffsets dt;
            if constexpr (TAB) dt = tab_dt;
            else dt = tap_offsets<DMODE, SLAB>(p.data, ix, iy, iz);
            dtaps.issue(p.data.data, dt);
            // LightVolume.SampleLevel(Wrap, saturate(CurPos)) (WindowedRaymarchMaterials.usf:30)
            const float sp0 = saturate_(q0), sp1 = saturate_(q1), sp2 = saturate_(q2);
            if (same_grid && sp0 == q0 && sp1 == q1 && sp2 == q2) {
                gx = fx; gy = fy; gz = fz;
                ltaps.issue(p.light, dt);
            } else {
                int lx, ly, lz; // (saturated coordinates: in [0, 1], or NaN -> 0)
                texel_split_bounded(sp0, lnx, lx, gx);
                texel_split_bounded(sp1, lny, ly, gy);
                texel_split_bounded(sp2, lnz, lz, gz);
                ltaps.issue(p.light, tap_offsets<ADDR_WRAP, SLAB>(lightv, lx, ly, lz));
            }
            const float v = dtaps.filter(fx, fy, fz);
            // SampleWindowedTransferFunction (WindowedSampling.usf:20-37)
            const float tpos = window_position<DFMT != FMT_F32>(v, p.win);
            if (!((tpos < 0.0f && p.win.low_cutoff > 0.0f) || (tpos > 1.0f && p.win.high_cutoff > 0.0f))) {
                const float4 cs = sample_tf(s_tf, tpos);
                const float a_sat = saturate_(cs.w);
                if (a_sat != 0.0f) { // else 1 - pow(1, s) = 0: the sample contributes exactly nothing
                    // (a_sat in (0, 1]; the step is 100 / steps or 100 x a fraction in (0, 1): >= 0 unless the host passed a negative
                    // step count, which build_ray_params rejects — pow01_ is pow_ on that domain, bit for bit)
                    const float a = one_minus_pow01_(1.0f - a_sat, step);
                    const float l = ltaps.filter(gx, gy, gz);
                    x = make_float4((cs.x * l) * a, (cs.y * l) * a, (cs.z * l) * a, a);
                }
            }
        }

        // AccumulateLightEnergy (RaymarchMaterialCommon.usf:82-88) over the ray's 8 samples, in order, in every lane of
        // the ray; the early exit belongs to the full steps only (:75-79). A trip in which no lane of the wave has
        // anything to accumulate (empty space, windowed-out values) needs no exchange.
        const bool any_x = __builtin_amdgcn_ballot_w64(x.w >= 0.0f || x.w != x.w) != 0;
#ifdef TBRM_RAY_STATS
        {
            const unsigned long long nx = __builtin_popcountll(__builtin_amdgcn_ballot_w64(x.w >= 0.0f || x.w != x.w));
            if (lane == 0 && nx) { atomicAdd(&g_ray_stats[4], 1ull); atomicAdd(&g_ray_stats[5], nx); }
        }
#endif
        if (any_x) {
            s_x[threadIdx.x] = x;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < kRayLanes; ++t) {
                const float4 c = xs[t];
                if (!done && !(c.w < 0.0f)) {
                    const float om = 1.0f - le3;
                    le0 = le0 + (c.x * om);
                    le1 = le1 + (c.y * om);
                    le2 = le2 + (c.z * om);
                    le3 = le3 + (c.w * om);
                    if (le3 > 0.95f && base + t < max_steps) { le3 = 1.0f; done = true; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (base + kRayLanes >= n_samples) done = true;
        // Empty space, wave-wide: when no lane of the wave had anything to sample in this trip, the trips that EVERY marching
        // lane would spend the same way — its sample still within its proven-empty range (safe_until), the ray still in its
        // full steps — are taken in one go: their only effect is the positions' additions, performed one by one as before
        // (a position is reached by performing every addition of the ray). 70 % of the benchmark's trips are of this kind.
        const bool renewed = eager;
        eager = false;
        if (!any_live) {
            int k_lane = INT32_MAX; // whole trips this lane can take blind after this one
            if (!done) {
                const int ahead = safe_until - base - b; // its sample of trip t from now is base + kRayLanes t + b
                const int by_safe = ahead >= kRayLanes ? ahead / kRayLanes : 0;
                const int left = max_steps - 1 - (base + kRayLanes); // full steps behind the end of this trip, but for the last
                const int by_length = left >= kRayLanes ? left / kRayLanes : 0;
                k_lane = min(by_safe, by_length);
            }
            int k = 0; // the wave's minimum (small: counted up with ballots)
            while (k < 64 && __builtin_amdgcn_ballot_w64(k_lane <= k) == 0) ++k;
            if (k > 0) {
                for (int t = 0; t < k * kRayLanes; ++t) { pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2; }
                adds += k * kRayLanes;
                base += k * kRayLanes;
            }
            // (next to the volume's content the ranges are a trip or two long: renewing them every empty trip would cost more
            // than the look-ups it aligns — small volumes lost 10 - 20 % of their frame that way)
            if (renewed && k == 0) renew_wait = 4;
            else if (renew_wait > 0) --renew_wait;
            eager = renew_wait == 0;
        }
    }
    if ((SLAB ? mine : valid) && b == 0) reinterpret_cast<float4*>(p.out)[(size_t) j * p.tile_w + i] = make_float4(le0, le1, le2, le3);
}

template <int DFMT, int LFMT, int RL>
static hipError_t launch_ray3(const RayParams& p, hipStream_t s)
{
    constexpr int BW = 8, BH = RL == 4 ? 8 : 4; // 4 waves of 4x4 / 4x2 rays
    const dim3 grid((p.tile_w + BW - 1) / BW, (p.tile_h + BH - 1) / BH), block(256);
    if (p.slab_on) {
        if (p.data_addr_mode == ADDR_CLAMP) hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_CLAMP, RL, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_WRAP, RL, true>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    // the offset tables (k_raymarch_lit TAB): a step of at most one texel along every axis — then no sample's base tap lies below
    // -2 or above n — and tables of at most 16 KiB (six workgroups per CU keep their place)
    const size_t tab_bytes = (size_t) ((p.data.nx + p.data.ny + p.data.nz + 12 + 1) & ~1) * sizeof(uint2);
    const bool tab = p.tab != nullptr && tune(TUNE_RAY_TABLES) != 0 && (float) std::max(p.data.nx, std::max(p.data.ny, p.data.nz)) <= p.steps && tab_bytes <= 16 * 1024;
    if (tab) {
        if (p.data_addr_mode == ADDR_CLAMP) hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_CLAMP, RL, false, true>), grid, block, tab_bytes, s, p);
        else hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_WRAP, RL, false, true>), grid, block, tab_bytes, s, p);
        return hipGetLastError();
    }
    if (p.data_addr_mode == ADDR_CLAMP) hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_CLAMP, RL>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_raymarch_lit<DFMT, LFMT, ADDR_WRAP, RL>), grid, block, 0, s, p);
    return hipGetLastError();
}
template <int DFMT, int LFMT>
static hipError_t launch_ray2(const RayParams& p, hipStream_t s)
{
    // Lanes per ray, measured on MI355X (ms per frame; 2 lanes: 0.85 / 0.64 / 0.27, 16 lanes: 0.80 / 1.50 / 0.15):
    //                     config 3 (1024^2 rays,   config 5 (2048^2,   config 2 (512^2,   config 4 (1024^2,
    //                               512 steps)              512 steps)         256 steps)        1024 steps)
    //   4 lanes per ray          0.57                    0.68                0.18               1.28
    //   8 lanes per ray          0.61                    0.95                0.14               1.17
    // More lanes per ray shorten the serial chain, fewer keep more rays (and their setup) per wave: few rays or long rays
    // want 8. The rule: rays x (512 / steps) <= 700 k.
    const double load = (double) p.tile_w * (double) p.tile_h * 512.0 / (double) (p.steps > 1.0f ? p.steps : 1.0f);
    const int forced = tune(TUNE_RAY_LANES);
    const int rl = forced ? forced : (load <= 700000.0 ? 8 : 4);
    return rl == 8 ? launch_ray3<DFMT, LFMT, 8>(p, s) : launch_ray3<DFMT, LFMT, 4>(p, s);
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

// ---- self-test of the division-free window position (tf_position_fast against the IEEE quotient, every float in [0, 1]) --------------
__global__ __launch_bounds__(256) void k_selftest_window_division(WindowDev w, unsigned long long* mismatches)
{
    const uint32_t last = 0x3f800000u; // 1.0f
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= last; i += (uint64_t) gridDim.x * blockDim.x) {
        const float v = __uint_as_float((uint32_t) i);
        const float a = tf_position(v, w.center, w.width), b = tf_position_fast(v, w.center, w.width, w.inv_width);
        if (__float_as_uint(a) != __float_as_uint(b)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
hipError_t launch_selftest_window_division(const WindowDev& w, unsigned long long* d_mismatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_window_division, dim3(256 * 64), dim3(256), 0, s, w, d_mismatches);
    return hipGetLastError();
}

// ---- self-test of the opacity correction's short form (one_minus_pow01_ against 1 - pow01_, every float in [0, 1]) --------------------
__global__ __launch_bounds__(256) void k_selftest_opacity_correction(float step0, float step1, unsigned long long* mismatches)
{
    const uint32_t last = 0x3f800000u; // 1.0f
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= last; i += (uint64_t) gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t) i);
        float p0, p1, s0, s1;
        pow01_2_(x, step0, step1, p0, p1);
        one_minus_pow01_2_(x, step0, step1, s0, s1);
        const float a = 1.0f - pow01_(x, step0), b = one_minus_pow01_(x, step0);
        if (__float_as_uint(1.0f - p0) != __float_as_uint(s0) || __float_as_uint(1.0f - p1) != __float_as_uint(s1) || __float_as_uint(a) != __float_as_uint(b)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
hipError_t launch_selftest_opacity_correction(float step0, float step1, unsigned long long* d_mismatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_opacity_correction, dim3(256 * 64), dim3(256), 0, s, step0, step1, d_mismatches);
    return hipGetLastError();
}

// ---- k_raymarch_intensity: PerformWindowedIntensityRaymarch (WindowedRaymarchMaterials.usf:187-242) ----------------
// The slice view: the windowed intensity of the first sample the clipping plane leaves. Without a clipping plane that is
// every ray's first sample, so one ray per lane is the right shape here (no long serial loop to split up).
template <int DFMT>
__global__ __launch_bounds__(256) void k_raymarch_intensity(const RayParams p)
{
    int i, j, px, py;
    if (!tile_pixel(p, i, j, px, py)) return;
    Ray ray;
    cube_setup(p, px, py, ray);
    const float step_size = 1 / p.steps;
    const float actual = p.steps * ray.thickness;
    const float fl = floorf(actual);
    const int max_steps = (int) fl;
    const float final_step = actual - fl;
    const float sv0 = ray.lcv[0] * step_size, sv1 = ray.lcv[1] * step_size, sv2 = ray.lcv[2] * step_size;
    float pos0 = ray.pos[0], pos1 = ray.pos[1], pos2 = ray.pos[2];
    if (p.jitter_frame >= 0) {
        uint32_t rr;
        rand3d_pcg16(px, py, p.jitter_frame & 7, rr);
        const float rnd = (float) rr / 65535.0f;
        pos0 = pos0 - (sv0 * rnd); pos1 = pos1 - (sv1 * rnd); pos2 = pos2 - (sv2 * rnd);
    }
    const float nx = (float) p.data.nx, ny = (float) p.data.ny, nz = (float) p.data.nz;
    auto intensity = [&](float u, float v, float w) -> float { // DataVolume.SampleLevel(Clamp, uvw).r -> clamp(TFPos, 0, 1)
        int ix, iy, iz;
        float fx, fy, fz;
        texel_split(u, nx, ix, fx);
        texel_split(v, ny, iy, fy);
        texel_split(w, nz, iz, fz);
        const float val = sample_trilinear_at<DFMT>(p.data.data, tap_offsets<ADDR_CLAMP>(p.data, ix, iy, iz), fx, fy, fz);
        return __builtin_amdgcn_fmed3f(tf_position(val, p.win.center, p.win.width), 0.0f, 1.0f);
    };
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // didn't hit anything (:241)
    bool hit = false;
    for (int k = 0; k < max_steps; ++k) {
        pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2;
        const float s0 = saturate_(pos0), s1 = saturate_(pos1), s2 = saturate_(pos2);
        if (!(p.clip_mode && is_clipped(s0, s1, s2, p.cc, p.cd))) {
            const float t = intensity(s0, s1, s2);
            out = make_float4(t, t, t, 1.0f);
            hit = true;
            break;
        }
    }
    if (!hit && final_step > 0.0f) {
        pos0 = pos0 + (sv0 * final_step); pos1 = pos1 + (sv1 * final_step); pos2 = pos2 + (sv2 * final_step);
        if (!(p.clip_mode && is_clipped(pos0, pos1, pos2, p.cc, p.cd))) {
            const float t = intensity(pos0, pos1, pos2);
            out = make_float4(t, t, t, 1.0f);
        }
    }
    reinterpret_cast<float4*>(p.out)[(size_t) j * p.tile_w + i] = out;
}

hipError_t launch_raymarch_intensity(const RayParams& p, hipStream_t s)
{
    if (p.tile_w <= 0 || p.tile_h <= 0) return hipSuccess;
    const dim3 grid((p.tile_w + 15) / 16, (p.tile_h + 15) / 16), block(256);
    switch (p.data.fmt) {
        case FMT_U8: hipLaunchKernelGGL(k_raymarch_intensity<FMT_U8>, grid, block, 0, s, p); break;
        case FMT_U16: hipLaunchKernelGGL(k_raymarch_intensity<FMT_U16>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(k_raymarch_intensity<FMT_F32>, grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

// ---- Octree render mode (experimental in the reference) -------------------------------------------------------------
// GenerateOctreeShader.usf:28-107: level 0 = the volume (x MinMaxValues.y = 1, OctreeShaders.h:49) as UNORM16, 0 outside the
// volume; level m = max over the 2x2x2 texels of level m-1. One thread per output texel (the reference runs one thread
// per 8^3 leaf, [numthreads(1,1,1)]).
template <int DFMT>
__global__ __launch_bounds__(256) void k_octree_base(const OctreeParams p)
{
    const size_t n = (size_t) p.dims[0] * p.dims[1] * p.dims[2];
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int) (i % p.dims[0]), y = (int) ((i / p.dims[0]) % p.dims[1]), z = (int) (i / ((size_t) p.dims[0] * p.dims[1]));
    float v = 0.0f;
    if (x < p.data.nx && y < p.data.ny && z < p.data.nz) v = load_voxel<DFMT>(p.data.data, brick_off(x, y, z, p.data.bnx, p.data.bnxy)) * 1.0f;
    p.out[i] = (uint16_t) __builtin_floorf(__builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) * 65535.0f + 0.5f); // UNORM16 store, NaN -> 0
}
__global__ __launch_bounds__(256) void k_octree_reduce(const OctreeParams p)
{
    const size_t n = (size_t) p.dims[0] * p.dims[1] * p.dims[2];
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int) (i % p.dims[0]), y = (int) ((i / p.dims[0]) % p.dims[1]), z = (int) (i / ((size_t) p.dims[0] * p.dims[1]));
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sx = 2 * x + (k & 1), sy = 2 * y + ((k >> 1) & 1), sz = 2 * z + (k >> 2);
        if (sx < p.lower_dims[0] && sy < p.lower_dims[1] && sz < p.lower_dims[2])
            mx = max(mx, (uint32_t) p.lower[((size_t) sz * p.lower_dims[1] + sy) * p.lower_dims[0] + sx]);
    }
    p.out[i] = (uint16_t) mx;
}
hipError_t launch_octree_level(const OctreeParams& p, bool base, hipStream_t s)
{
    const size_t n = (size_t) p.dims[0] * p.dims[1] * p.dims[2];
    const dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    if (!base) hipLaunchKernelGGL(k_octree_reduce, grid, block, 0, s, p);
    else if (p.data.fmt == FMT_U8) hipLaunchKernelGGL(k_octree_base<FMT_U8>, grid, block, 0, s, p);
    else if (p.data.fmt == FMT_U16) hipLaunchKernelGGL(k_octree_base<FMT_U16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_octree_base<FMT_F32>, grid, block, 0, s, p);
    return hipGetLastError();
}

// PerformWindowedRaymarchOctree (WindowedRaymarchMaterials.usf:99-183): unlit, point-sampled march over one octree level.
__global__ __launch_bounds__(256) void k_raymarch_octree(const RayParams p)
{
    __shared__ float4 s_tf[256];
    s_tf[threadIdx.x] = p.tf[threadIdx.x];
    __syncthreads();
    int i, j, px, py;
    if (!tile_pixel(p, i, j, px, py)) return;
    Ray ray;
    cube_setup(p, px, py, ray);
    const float step_size = 1 / p.steps;
    const float actual = p.steps * ray.thickness;
    const float fl = floorf(actual);
    const int max_steps = (int) fl;
    const float final_step = actual - fl;
    const float sv0 = ray.lcv[0] * step_size, sv1 = ray.lcv[1] * step_size, sv2 = ray.lcv[2] * step_size;
    const float step_world = 100.0f * step_size;
    float pos0 = ray.pos[0], pos1 = ray.pos[1], pos2 = ray.pos[2];
    if (p.jitter_frame >= 0) {
        uint32_t rr;
        rand3d_pcg16(px, py, p.jitter_frame & 7, rr);
        const float rnd = (float) rr / 65535.0f;
        pos0 = pos0 - (sv0 * rnd); pos1 = pos1 - (sv1 * rnd); pos2 = pos2 - (sv2 * rnd);
    }
    const float ow = (float) p.oct_dims[0], oh = (float) p.oct_dims[1], od = (float) p.oct_dims[2], data_depth = (float) p.data.nz;
    float le0 = 0.0f, le1 = 0.0f, le2 = 0.0f, le3 = 0.0f;
    for (int k = 0; k <= max_steps; ++k) {
        if (k < max_steps) { pos0 = pos0 + sv0; pos1 = pos1 + sv1; pos2 = pos2 + sv2; }
        else {
            if (!(final_step > 0.0f)) break;
            pos0 = pos0 + (sv0 * final_step); pos1 = pos1 + (sv1 * final_step); pos2 = pos2 + (sv2 * final_step);
        }
        if (p.clip_mode && is_clipped(pos0, pos1, pos2, p.cc, p.cd)) continue;
        // int3 VoxelPos = float3(x * W, y * H, (z * DataDepth / OctreeDepth0) * OctreeDepth) (:150): truncation; Load outside -> 0
        const float fx = pos0 * ow, fy = pos1 * oh, fz = ((pos2 * data_depth) / p.oct_depth0) * od;
        float v = 0.0f;
        if (fx == fx && fy == fy && fz == fz && fx > -1.0f && fy > -1.0f && fz > -1.0f && fx < ow && fy < oh && fz < od) {
            const int vx = (int) fx, vy = (int) fy, vz = (int) fz;
            v = decode_u16(p.octree[((size_t) vz * p.oct_dims[1] + vy) * p.oct_dims[0] + vx]);
        }
        // SampleWindowedTransferFunction with StepSizeWorld — also in the fractional step (:176), unlike the lit march
        const float tpos = tf_position(v, p.win.center, p.win.width);
        if ((tpos < 0.0f && p.win.low_cutoff > 0.0f) || (tpos > 1.0f && p.win.high_cutoff > 0.0f)) continue;
        const float4 cs = sample_tf(s_tf, tpos);
        const float a_sat = saturate_(cs.w);
        const float a = 1.0f - pow_(1.0f - a_sat, step_world);
        const float om = 1.0f - le3; // AccumulateLightEnergy
        le0 = le0 + ((cs.x * a) * om);
        le1 = le1 + ((cs.y * a) * om);
        le2 = le2 + ((cs.z * a) * om);
        le3 = le3 + (a * om);
        if (k < max_steps && le3 > 0.95f) { le3 = 1.0f; break; }
    }
    reinterpret_cast<float4*>(p.out)[(size_t) j * p.tile_w + i] = make_float4(le0, le1, le2, le3);
}
hipError_t launch_raymarch_octree(const RayParams& p, hipStream_t s)
{
    if (p.tile_w <= 0 || p.tile_h <= 0) return hipSuccess;
    const dim3 grid((p.tile_w + 15) / 16, (p.tile_h + 15) / 16), block(256);
    hipLaunchKernelGGL(k_raymarch_octree, grid, block, 0, s, p);
    return hipGetLastError();
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

} // namespace tbrm
