// tbrm_light_kernels.hip — gfx950 kernels of the illumination pass (Sunden/Ropinski selective light updates).
//
//   k_light_occlusion + k_light_chain : the production pair. Per chunk of up to 16 consecutive slices, the first
//                       computes the opacity sample of every voxel of the chunk (no halo, fully parallel), the
//                       second advances EVERY tile of the slice plane through the chunk, so an axis pass over a
//                       512-deep volume is 2 x 32 launches instead of the reference's 512 dispatches
//                       (LightingShaders.cpp:132-158).
//   k_propagate_slice : the reference's structure, one slice per launch (AddDirLightShader.usf:68-128,
//                       ChangeDirLightShader.usf:74-156). Fallback for passes the chunk kernel declines
//                       (degenerate offsets) and the A/B baseline (TBRM_FORCE_SLICE_KERNEL=1).
//
// Why chunks work: slice k only needs the previous slice's propagated light inside a small bilinear footprint,
// offset by the constant PrevPixelOffset. A workgroup that owns a 32x32 tile at the END of a chunk therefore only
// needs a (32 + steps*g)^2 window of the plane at the START of the chunk (g = width of the footprint in texels,
// normally 1) and recomputes that shrinking window privately in LDS — no inter-workgroup traffic inside a chunk,
// one kernel boundary between chunks. The window moves with the light (integer shear cx,cy per slice) so strongly
// slanted second-axis passes keep the same small halo. The expensive part of a voxel — the windowed, opacity-
// corrected data sample "CurrentSample" (AddDirLightShader.usf:85-114) — does not depend on the propagated light, so
// it is computed once per voxel, without halo, by k_light_occlusion and handed over through a scratch plane stack
// that k_light_chain stages into LDS with asynchronous global->LDS loads before its first step; the chain itself is
// then a handful of LDS reads and one multiply per voxel, with no memory latency between slices. Arithmetic per voxel is
// exactly the reference's, including the per-slice UNORM8 re-quantisation of the propagated light
// (RaymarchVolume.cpp:857-866).
#include "tbrm_device_sampling.h"

namespace tbrm {

// ------------------------------------------------------------------------------------------------------------
// one slice per launch

template <int LFMT>
__device__ __forceinline__ float sample_buffer_bilinear(const void* buf, int w, int h, float u, float v, float border)
{
    int ix, iy;
    float fx, fy;
    texel_split(u, (float) w, ix, fx);
    texel_split(v, (float) h, iy, fy);
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = ix + (k & 1), y = iy + (k >> 1);
        const bool in = (unsigned) x < (unsigned) w && (unsigned) y < (unsigned) h;
        t[k] = in ? load_voxel<LFMT>(buf, (size_t) y * w + x) : border;
    }
    return lerp_(lerp_(t[0], t[1], fx), lerp_(t[2], t[3], fx), fy);
}

template <int DFMT, int LFMT, bool GUARD>
__device__ __forceinline__ float propagate_stream(const PropParams& p, const PropStream& s, const float* tf_alpha,
                                                  int px, int py, const int* pos)
{
    const float pu = (((float) (uint32_t) px + 0.5f) / (float) p.td[0]) + s.off_u;
    const float pv = (((float) (uint32_t) py + 0.5f) / (float) p.td[1]) + s.off_v;
    const float prev = sample_buffer_bilinear<LFMT>(s.read, p.td[0], p.td[1], pu, pv, s.border_light);

    const float u = (((float) (uint32_t) pos[0] + 0.5f) / (float) (uint32_t) p.lv_dims[0]) + s.uvw_off[0];
    const float v = (((float) (uint32_t) pos[1] + 0.5f) / (float) (uint32_t) p.lv_dims[1]) + s.uvw_off[1];
    const float w = (((float) (uint32_t) pos[2] + 0.5f) / (float) (uint32_t) p.lv_dims[2]) + s.uvw_off[2];

    const float aw = p.clip_mode ? clip_alpha_weight(u, v, w, p.cc, p.cd, p.lv_dims) : 1.0f;
    float cur = 0.0f;
    bool inside = true;
    if constexpr (GUARD) inside = (u == saturate_(u)) && (v == saturate_(v)) && (w == saturate_(w));
    if (aw > 0.0f && inside) {
        const float val = sample_trilinear_border_uvw<DFMT>(p.data, u, v, w, p.data_border);
        cur = windowed_alpha(val, s.step100, tf_alpha, p.win) * aw;
    }
    return prev * (1 - cur);
}

template <int DFMT, int LFMT, bool CHANGE>
__global__ __launch_bounds__(256) void k_propagate_slice(const PropParams p)
{
    __shared__ float s_alpha[256];
    s_alpha[threadIdx.x] = p.tf[threadIdx.x].w;
    __syncthreads();
    const int px = blockIdx.x * 16 + (threadIdx.x & 15);
    const int py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= p.td[0] || py >= p.td[1]) return; // D3D drops the overhanging threads' writes
    int pos[3];
    if (p.axis == 0) { pos[0] = p.loop; pos[1] = px; pos[2] = py; }
    else if (p.axis == 1) { pos[0] = px; pos[1] = p.loop; pos[2] = py; }
    else { pos[0] = px; pos[1] = py; pos[2] = p.loop; }
    const size_t bi = (size_t) py * p.td[0] + px;
    const size_t li = brick_off(pos[0], pos[1], pos[2], p.lv_bnx, p.lv_bnxy);
    if constexpr (!CHANGE) {
        const float l = propagate_stream<DFMT, LFMT, true>(p, p.a, s_alpha, px, py, pos);
        store_voxel<LFMT>(p.a.write, bi, l);
        if (fabsf(l) > 1e-3f) store_voxel<LFMT>(p.light, li, load_voxel<LFMT>(p.light, li) + (l * p.b_added));
    } else {
        const float lr = propagate_stream<DFMT, LFMT, false>(p, p.r, s_alpha, px, py, pos);
        const float la = propagate_stream<DFMT, LFMT, false>(p, p.a, s_alpha, px, py, pos);
        store_voxel<LFMT>(p.r.write, bi, lr);
        store_voxel<LFMT>(p.a.write, bi, la);
        if (fabsf(la - lr) > 1e-3f) store_voxel<LFMT>(p.light, li, load_voxel<LFMT>(p.light, li) + la - lr);
    }
}

template <int DFMT, int LFMT>
static hipError_t launch_prop2(const PropParams& p, bool change, hipStream_t s)
{
    const dim3 grid((p.td[0] + 15) / 16, (p.td[1] + 15) / 16), block(256);
    if (change) hipLaunchKernelGGL((k_propagate_slice<DFMT, LFMT, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_propagate_slice<DFMT, LFMT, false>), grid, block, 0, s, p);
    return hipGetLastError();
}
template <int DFMT>
static hipError_t launch_prop1(const PropParams& p, bool change, hipStream_t s)
{
    return p.lv_fmt == FMT_U8 ? launch_prop2<DFMT, FMT_U8>(p, change, s) : launch_prop2<DFMT, FMT_F32>(p, change, s);
}
hipError_t launch_propagate_slice(const PropParams& p, bool change, hipStream_t s)
{
    switch (p.data.fmt) {
        case FMT_U8: return launch_prop1<FMT_U8>(p, change, s);
        case FMT_U16: return launch_prop1<FMT_U16>(p, change, s);
        default: return launch_prop1<FMT_F32>(p, change, s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// a chunk of slices per launch pair

__device__ __forceinline__ char* carve(char*& cursor, size_t bytes)
{
    char* r = cursor;
    cursor += (bytes + 15) & ~(size_t) 15;
    return r;
}

struct ChunkGeom {
    int n;                  // steps in this chunk
    int lox, hix, loy, hiy; // tap offsets relative to the ownership frame: [lo, hi]
    int gx, gy;             // growth of the window per remaining step
    int HX, HY;             // hull (LDS window) size
    int padx, pady;         // LDS index of ownership-frame coordinate 0
    int tabx0, taby0;       // first plane coordinate covered by the previous-tap tables
    int tablx, tably;       // table lengths
    int occ_total;          // staged occlusion values per stream: sum over steps of the window size
};

__host__ __device__ inline ChunkGeom chunk_geometry(const ChunkParams& p, int tile_x, int tile_y)
{
    ChunkGeom g;
    g.n = p.n_steps;
    g.lox = p.dx_lo - p.cx; g.hix = p.dx_hi - p.cx;
    g.loy = p.dy_lo - p.cy; g.hiy = p.dy_hi - p.cy;
    g.gx = g.hix - g.lox; g.gy = g.hiy - g.loy;
    g.HX = kChunkTile + g.n * g.gx;
    g.HY = kChunkTile + g.n * g.gy;
    g.padx = -g.n * g.lox;
    g.pady = -g.n * g.loy;
    const int nlo_x = g.n * p.dx_lo, nhi_x = g.n * p.dx_hi, nlo_y = g.n * p.dy_lo, nhi_y = g.n * p.dy_hi;
    g.tabx0 = tile_x * kChunkTile + (nlo_x < 0 ? nlo_x : 0);
    g.taby0 = tile_y * kChunkTile + (nlo_y < 0 ? nlo_y : 0);
    g.tablx = kChunkTile + (nhi_x > 0 ? nhi_x : 0) - (nlo_x < 0 ? nlo_x : 0);
    g.tably = kChunkTile + (nhi_y > 0 ? nhi_y : 0) - (nlo_y < 0 ? nlo_y : 0);
    g.occ_total = 0;
    for (int r = 0; r < g.n; ++r) g.occ_total += ((kChunkTile + r * g.gx) * (kChunkTile + r * g.gy) + 63) & ~63;
    return g;
}

size_t chunk_lds_bytes(const ChunkParams& p, bool change)
{
    const ChunkGeom g = chunk_geometry(p, 0, 0);
    const int ns = change ? 2 : 1;
    auto al = [](size_t b) { return (b + 15) & ~(size_t) 15; };
    size_t total = (size_t) ns * 2 * al(((size_t) g.HX * g.HY + 64) * 4);          // double-buffered plane windows (+ DMA slack)
    total += (size_t) ns * 2 * (al((size_t) g.tablx * 4) + al((size_t) g.tably * 4)); // previous-tap tables (frac, delta)
    return total;
}

// ---- k_light_occlusion: CurrentSample (AddDirLightShader.usf:85-114) for every voxel of a chunk ------------------
// grid = (plane tiles of 16x16 pixels, groups of kOccGroup slices); one thread = one pixel x kOccGroup slices, with
// all 8*kOccGroup taps in flight before the first is used. Addressing is separable in the bricked layout: the two
// in-plane axes contribute per-thread constants, the slice axis a per-step value read from a small LDS table (which
// also carries the reference's per-thread (Loop+0.5)/res division out of the kernel).
constexpr int kOccGroup = 4;

struct AxisTaps {       // one axis of a trilinear footprint with border addressing
    uint32_t off0, off1; // brick offsets of the two taps (clamped into range, so always loadable)
    bool ok0, ok1;       // tap inside the volume (else the sampler's border colour applies)
    float f;             // interpolation weight
};

template <int AXIS>
__device__ __forceinline__ AxisTaps axis_taps(float coord, int n, int bnx, int bnxy)
{
    AxisTaps t;
    int i;
    texel_split(coord, (float) n, i, t.f);
    t.ok0 = (unsigned) i < (unsigned) n;
    t.ok1 = (unsigned) (i + 1) < (unsigned) n;
    const int c0 = min(max(i, 0), n - 1), c1 = min(max(i + 1, 0), n - 1);
    if constexpr (AXIS == 0) { t.off0 = brick_off_x(c0); t.off1 = brick_off_x(c1); }
    else if constexpr (AXIS == 1) { t.off0 = brick_off_y(c0, bnx); t.off1 = brick_off_y(c1, bnx); }
    else { t.off0 = brick_off_z(c0, bnxy); t.off1 = brick_off_z(c1, bnxy); }
    return t;
}

__device__ __forceinline__ AxisTaps axis_taps_dyn(int axis, float coord, int n, int bnx, int bnxy)
{
    return axis == 0 ? axis_taps<0>(coord, n, bnx, bnxy) : (axis == 1 ? axis_taps<1>(coord, n, bnx, bnxy) : axis_taps<2>(coord, n, bnx, bnxy));
}

template <int DFMT, bool CHANGE>
__global__ __launch_bounds__(256) void k_light_occlusion(const ChunkParams p)
{
    constexpr int NS = CHANGE ? 2 : 1;
    __shared__ float s_alpha[256];
    __shared__ float s_w[2][kOccGroup], s_f[2][kOccGroup];
    __shared__ uint32_t s_o0[2][kOccGroup], s_o1[2][kOccGroup];
    __shared__ int s_flags[2][kOccGroup]; // bit0: tap0 in range, bit1: tap1 in range, bit2: w == saturate(w)

    const int dim_u = p.axis == 0 ? 1 : 0, dim_v = p.axis == 2 ? 1 : 2, dim_s = p.axis;
    const int data_dims[3] = {p.data.nx, p.data.ny, p.data.nz};
    const int k0 = blockIdx.z * kOccGroup;

    s_alpha[threadIdx.x] = p.tf[threadIdx.x].w;
    if (threadIdx.x < NS * kOccGroup) { // slice-axis taps of each step of this group (wave-uniform values)
        const int si = threadIdx.x / kOccGroup, q = threadIdx.x % kOccGroup;
        const ChunkStream& s = si == 0 ? p.a : p.r;
        const int j = p.j0 + (k0 + q) * p.dir;
        const float w = (((float) (uint32_t) j + 0.5f) / (float) (uint32_t) p.lv_dims[dim_s]) + s.uvw_off[dim_s];
        const AxisTaps t = axis_taps_dyn(dim_s, w, data_dims[dim_s], p.data.bnx, p.data.bnxy);
        s_w[si][q] = w; s_f[si][q] = t.f; s_o0[si][q] = t.off0; s_o1[si][q] = t.off1;
        s_flags[si][q] = (t.ok0 ? 1 : 0) | (t.ok1 ? 2 : 0) | ((w == saturate_(w)) ? 4 : 0);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (px >= p.W || py >= p.H) return;
    const size_t plane_elems = (size_t) p.H * p.W;
    const float border = p.data_border;

#pragma unroll
    for (int si = 0; si < NS; ++si) {
        const ChunkStream& s = si == 0 ? p.a : p.r;
        // GetUVW(pos, res) + UVWOffset (AddDirLightShader.usf:85): the two in-plane components
        const float u = (((float) (uint32_t) px + 0.5f) / (float) (uint32_t) p.lv_dims[dim_u]) + s.uvw_off[dim_u];
        const float v = (((float) (uint32_t) py + 0.5f) / (float) (uint32_t) p.lv_dims[dim_v]) + s.uvw_off[dim_v];
        const AxisTaps tu = axis_taps_dyn(dim_u, u, data_dims[dim_u], p.data.bnx, p.data.bnxy);
        const AxisTaps tv = axis_taps_dyn(dim_v, v, data_dims[dim_v], p.data.bnx, p.data.bnxy);
        const bool guard_uv = (u == saturate_(u)) && (v == saturate_(v));
        // in-plane corner offsets and validity (u index is the faster of the two in the lerp order below only when
        // dim_u < dim_v, which holds for every axis: (1,2), (0,2), (0,1))
        const uint32_t o00 = tu.off0 + tv.off0, o10 = tu.off1 + tv.off0, o01 = tu.off0 + tv.off1, o11 = tu.off1 + tv.off1;
        const bool k00 = tu.ok0 && tv.ok0, k10 = tu.ok1 && tv.ok0, k01 = tu.ok0 && tv.ok1, k11 = tu.ok1 && tv.ok1;

        float taps[kOccGroup][8], aw[kOccGroup];
        bool sample[kOccGroup];
#pragma unroll
        for (int q = 0; q < kOccGroup; ++q) { // phase 1: issue every tap load of the group
            sample[q] = false;
            aw[q] = 0.0f;
            if (k0 + q >= p.n_steps) continue;
            const float w = s_w[si][q];
            const int fl = s_flags[si][q];
            if (p.clip_mode) {
                float c0, c1, c2;
                if (p.axis == 0) { c0 = w; c1 = u; c2 = v; } else if (p.axis == 1) { c0 = u; c1 = w; c2 = v; } else { c0 = u; c1 = v; c2 = w; }
                aw[q] = clip_alpha_weight(c0, c1, c2, p.cc, p.cd, p.lv_dims);
            } else aw[q] = 1.0f;
            bool inside = true;
            if constexpr (!CHANGE) inside = guard_uv && (fl & 4); // all(uvw == saturate(uvw)): Add only
            sample[q] = aw[q] > 0.0f && inside;
            if (sample[q] && !(p.debug & 16)) {
                const uint32_t w0 = s_o0[si][q], w1 = s_o1[si][q];
                const bool a0 = fl & 1, a1 = fl & 2;
                // tap t: bit0 = u tap, bit1 = v tap, bit2 = slice tap (re-ordered into x,y,z order in phase 2)
                taps[q][0] = (k00 && a0) ? load_voxel<DFMT>(p.data.data, o00 + w0) : border;
                taps[q][1] = (k10 && a0) ? load_voxel<DFMT>(p.data.data, o10 + w0) : border;
                taps[q][2] = (k01 && a0) ? load_voxel<DFMT>(p.data.data, o01 + w0) : border;
                taps[q][3] = (k11 && a0) ? load_voxel<DFMT>(p.data.data, o11 + w0) : border;
                taps[q][4] = (k00 && a1) ? load_voxel<DFMT>(p.data.data, o00 + w1) : border;
                taps[q][5] = (k10 && a1) ? load_voxel<DFMT>(p.data.data, o10 + w1) : border;
                taps[q][6] = (k01 && a1) ? load_voxel<DFMT>(p.data.data, o01 + w1) : border;
                taps[q][7] = (k11 && a1) ? load_voxel<DFMT>(p.data.data, o11 + w1) : border;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) taps[q][t] = border;
            }
        }
#pragma unroll
        for (int q = 0; q < kOccGroup; ++q) { // phase 2: filter (x, then y, then z), window, transfer function, opacity correction
            if (k0 + q >= p.n_steps) continue;
            float occ = 0.0f;
            if (sample[q]) {
                const float fs = s_f[si][q];
                const float* t = taps[q];
                float val;
                if (p.axis == 2) { // (u,v,s) = (x,y,z)
                    val = lerp_(lerp_(lerp_(t[0], t[1], tu.f), lerp_(t[2], t[3], tu.f), tv.f),
                                lerp_(lerp_(t[4], t[5], tu.f), lerp_(t[6], t[7], tu.f), tv.f), fs);
                } else if (p.axis == 1) { // (u,s,v) = (x,y,z): x = u, y = slice, z = v
                    val = lerp_(lerp_(lerp_(t[0], t[1], tu.f), lerp_(t[4], t[5], tu.f), fs),
                                lerp_(lerp_(t[2], t[3], tu.f), lerp_(t[6], t[7], tu.f), fs), tv.f);
                } else { // (s,u,v) = (x,y,z): x = slice, y = u, z = v
                    val = lerp_(lerp_(lerp_(t[0], t[4], fs), lerp_(t[1], t[5], fs), tu.f),
                                lerp_(lerp_(t[2], t[6], fs), lerp_(t[3], t[7], fs), tu.f), tv.f);
                }
                occ = (p.debug & 32) ? val : windowed_alpha(val, s.step100, s_alpha, p.win) * aw[q];
            }
            s.occ_next[(size_t) (k0 + q) * plane_elems + (size_t) py * p.W + px] = occ;
        }
    }
}

// ---- k_light_chain: one tile through the slices of the chunk ---------------------------------------------------

// asynchronous global -> LDS copy of one dword per lane: lane l of the wave lands at lds_wave_base + 4*l
__device__ __forceinline__ void dma_dword(const float* src, float* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                     (__attribute__((address_space(3))) void*) lds_wave_base, 4, 0, 0);
}

constexpr int kPrefetch = 3; // slices the occlusion / light-volume operands are fetched ahead of their use

template <int LFMT, bool CHANGE>
__global__ __launch_bounds__(kChunkThreads) void k_light_chain(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = kChunkTile;
    constexpr int NS = CHANGE ? 2 : 1;
    constexpr int KH = (kChunkMaxHull * kChunkMaxHull - T * T + kChunkThreads - 1) / kChunkThreads; // halo slots per thread
    constexpr int KS = 1 + KH;                                                                     // + the owned pixel
    const int tile_x = p.tile_i0 + (int) blockIdx.x, tile_y = p.tile_j0 + (int) blockIdx.y;
    const ChunkGeom g = chunk_geometry(p, tile_x, tile_y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_base = wave * 64;
    const size_t plane_elems = (size_t) p.H * p.W;
    const int base_x = tile_x * T, base_y = tile_y * T;
    const int n_slots = g.HX * g.HY;

    int stamp = 0;
    auto tick = [&]() { if ((p.debug & 64) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && stamp < 64) p.debug_clock[stamp++] = (long long) __builtin_amdgcn_s_memtime(); };
    tick();

    char* cursor = smem;
    float* win0[2]; // win0[si] + cur*win_stride = the window holding the state before the step
    float* tabx_f[2]; int* tabx_d[2]; float* taby_f[2]; int* taby_d[2];
    const int win_stride = (int) ((((size_t) n_slots + 64) * 4 + 15) & ~(size_t) 15) / 4;
#pragma unroll
    for (int si = 0; si < NS; ++si) {
        win0[si] = (float*) carve(cursor, ((size_t) n_slots + 64) * 4);
        (void) carve(cursor, ((size_t) n_slots + 64) * 4);
        tabx_f[si] = (float*) carve(cursor, g.tablx * 4); tabx_d[si] = (int*) carve(cursor, g.tablx * 4);
        taby_f[si] = (float*) carve(cursor, g.tably * 4); taby_d[si] = (int*) carve(cursor, g.tably * 4);
    }

    // ---- input window: the plane after the previous chunk (ownership frame of r = n), row-major over the hull -----
    {
        const float inv_hx = 1.0f / (float) g.HX;
        for (int eb = wave_base; eb < n_slots; eb += kChunkThreads) {
            const int e = eb + lane;
            const int ly = (int) (((float) e + 0.5f) * inv_hx), lx = e - ly * g.HX; // exact for e < 2^20
            const int px = base_x + g.n * p.cx + (lx - g.padx), py = base_y + g.n * p.cy + (ly - g.pady);
            const bool inplane = e < n_slots && (unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H;
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const ChunkStream& s = si == 0 ? p.a : p.r;
                if (inplane && !p.first_chunk) dma_dword(s.plane_in + (size_t) py * p.W + px, win0[si] + eb);
                else if (e < n_slots) win0[si][e] = inplane ? s.init_value : s.border_light;
            }
        }
    }
    // previous-slice tap split per plane coordinate: ((c + 0.5)/size + PrevPixelOffset) -> (tap - c, frac)
    // (AddDirLightShader.usf:81-82); depends only on the coordinate, so one table per axis replaces a division per voxel.
#pragma unroll
    for (int si = 0; si < NS; ++si) {
        const ChunkStream& s = si == 0 ? p.a : p.r;
        for (int k = threadIdx.x; k < g.tablx + g.tably; k += kChunkThreads) {
            const bool is_x = k < g.tablx;
            const int kk = is_x ? k : k - g.tablx;
            const int c = (is_x ? g.tabx0 : g.taby0) + kk;
            const int size = is_x ? p.W : p.H;
            int d = 0;
            float f = 0.0f;
            if (c >= 0 && c < size) {
                const float pu = (((float) (uint32_t) c + 0.5f) / (float) size) + (is_x ? s.off_u : s.off_v);
                int i0;
                texel_split(pu, (float) size, i0, f);
                d = i0 - c;
            }
            if (is_x) { tabx_f[si][kk] = f; tabx_d[si][kk] = d; }
            else { taby_f[si][kk] = f; taby_d[si][kk] = d; }
        }
    }
    tick();

    // ---- this thread's slots: slot 0 = its owned pixel (8x8 patch per wave over the 32x32 core), the rest = its share
    // of the halo (hull minus core), in ownership-frame coordinates
    int sqx[KS], sqy[KS];
    sqx[0] = (wave & 3) * 8 + (lane & 7);
    sqy[0] = (wave >> 2) * 8 + (lane >> 3);
    {
        // halo slots enumerated as: full rows above the core, the two side strips of the core rows, full rows below
        const int top = g.pady * g.HX, side = g.HX - T, mid = T * side;
        const int n_halo = n_slots - T * T;
        const float inv_hx = 1.0f / (float) g.HX, inv_side = side > 0 ? 1.0f / (float) side : 0.0f;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const int h = threadIdx.x + k * kChunkThreads;
            int lx = 0, ly = 0;
            if (h < top) { ly = (int) (((float) h + 0.5f) * inv_hx); lx = h - ly * g.HX; }
            else if (h < top + mid) {
                const int m = h - top;
                const int row = (int) (((float) m + 0.5f) * inv_side), col = m - row * side;
                ly = g.pady + row;
                lx = col < g.padx ? col : col + T;
            } else {
                const int m = h - top - mid;
                const int row = (int) (((float) m + 0.5f) * inv_hx);
                ly = g.pady + T + row;
                lx = m - row * g.HX;
            }
            sqx[1 + k] = h < n_halo ? lx - g.padx : INT32_MIN / 2; // sentinel: never inside a window
            sqy[1 + k] = ly - g.pady;
        }
    }

    // ---- per-slot running state: the pixel of slot k at step s is (base + r*c + q) with r = n-1-s, so from one slice
    // to the next every per-slot quantity changes by a constant ---------------------------------------------------
    int pxc[KS], pyc[KS];   // pixel at the step being computed
    int pxf[KS], pyf[KS];   // pixel at the step being fetched (kPrefetch slices ahead)
    int idxf[KS];           // its index into the occlusion plane stack
    int li[KS];             // LDS index of the slot inside a window (constant)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        pxc[k] = pxf[k] = base_x + (g.n - 1) * p.cx + sqx[k];
        pyc[k] = pyf[k] = base_y + (g.n - 1) * p.cy + sqy[k];
        idxf[k] = pyf[k] * p.W + pxf[k];
        li[k] = (sqy[k] + g.pady) * g.HX + sqx[k] + g.padx;
    }
    const int idx_step = (int) plane_elems - p.cy * p.W - p.cx;

    // operands fetched kPrefetch slices ahead: occlusion per slot and stream, the light-volume voxel of the owned pixel
    float occ_q[kPrefetch][NS][KS];
    float lv_q[kPrefetch];
    uint32_t lva_q[kPrefetch]; // voxel offset of that light-volume voxel, reused by the store
    int sf = 0;                // step being fetched
    auto fetch = [&](float (&occ)[NS][KS], float& lv, uint32_t& lva) {
        const int r = g.n - 1 - sf;
        const int x_lo = r * g.lox, x_hi = T + r * g.hix, y_lo = r * g.loy, y_hi = T + r * g.hiy;
        const bool live = sf < g.n && !(p.debug & 4);
        lv = 0.0f;
        lva = 0;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            bool active = live && (unsigned) pxf[k] < (unsigned) p.W && (unsigned) pyf[k] < (unsigned) p.H;
            if (k > 0) active = active && sqx[k] >= x_lo && sqx[k] < x_hi && sqy[k] >= y_lo && sqy[k] < y_hi;
            occ[0][k] = active ? p.a.occ_cur[idxf[k]] : 0.0f;
            if constexpr (CHANGE) occ[NS - 1][k] = active ? p.r.occ_cur[idxf[k]] : 0.0f;
            if (k == 0 && active && !(p.debug & 8)) {
                const int j = p.j0 + sf * p.dir;
                int x, y, z;
                if (p.axis == 0) { x = j; y = pxf[0]; z = pyf[0]; } else if (p.axis == 1) { x = pxf[0]; y = j; z = pyf[0]; } else { x = pxf[0]; y = pyf[0]; z = j; }
                lva = brick_off(x, y, z, p.lv_bnx, p.lv_bnxy);
                lv = load_voxel<LFMT>(p.light, lva);
            }
            pxf[k] -= p.cx;
            pyf[k] -= p.cy;
            idxf[k] += idx_step;
        }
        ++sf;
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's global->LDS copies of the input window have landed
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kPrefetch; ++d) fetch(occ_q[d], lv_q[d], lva_q[d]);
    tick();

    const int stream_stride = (int) (win0[NS - 1] - win0[0]);
    int cur = 0; // window holding the state BEFORE the step
    for (int s = 0; s < g.n; ++s) {
        const int r = g.n - 1 - s; // steps that remain after this one
        const int x_lo = r * g.lox, x_hi = T + r * g.hix, y_lo = r * g.loy, y_hi = T + r * g.hiy;
        const float* wr = win0[0] + cur * win_stride;  // state before the step (stream 0)
        float* ww = win0[0] + (cur ^ 1) * win_stride;  // state after the step (stream 0)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int px = pxc[k], py = pyc[k];
            pxc[k] -= p.cx;
            pyc[k] -= p.cy;
            if (k > 0 && (sqx[k] < x_lo || sqx[k] >= x_hi || sqy[k] < y_lo || sqy[k] >= y_hi || (p.debug & 2))) continue;
            if ((unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H) {
                const int kx = px - g.tabx0, ky = py - g.taby0;
                float lval[NS];
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    // previous slice, bilinear with border colour (AddDirLightShader.usf:81-82)
                    const float* pw = wr + si * stream_stride + li[k] + (taby_d[si][ky] - p.cy) * g.HX + (tabx_d[si][kx] - p.cx);
                    const float fx = tabx_f[si][kx], fy = taby_f[si][ky];
                    const float prev = lerp_(lerp_(pw[0], pw[1], fx), lerp_(pw[g.HX], pw[g.HX + 1], fx), fy);
                    const float l = prev * (1 - occ_q[0][si][k]); // :117
                    lval[si] = l;
                    ww[si * stream_stride + li[k]] = through_format<LFMT>(l); // WriteBuffer[PixelLoc] = L (:120)
                }
                if (k == 0) { // the owned pixel: this workgroup writes its light-volume voxel
                    if (p.debug & 1) {
                    } else if constexpr (!CHANGE) {
                        if (fabsf(lval[0]) > 1e-3f) store_voxel<LFMT>(p.light, lva_q[0], lv_q[0] + (lval[0] * p.b_added)); // :123-126
                    } else {
                        const float la = lval[0], lr = lval[NS - 1];
                        if (fabsf(la - lr) > 1e-3f) store_voxel<LFMT>(p.light, lva_q[0], lv_q[0] + la - lr); // Change :152-154
                    }
                    if (r == 0) {
#pragma unroll
                        for (int si = 0; si < NS; ++si) (si == 0 ? p.a : p.r).plane_out[(size_t) py * p.W + px] = through_format<LFMT>(lval[si]);
                    }
                }
            } else { // outside the buffer: later fetches must see the sampler's border colour here
#pragma unroll
                for (int si = 0; si < NS; ++si) ww[si * stream_stride + li[k]] = (si == 0 ? p.a : p.r).border_light;
            }
        }
        // rotate the prefetch queue and refill its tail
#pragma unroll
        for (int d = 0; d + 1 < kPrefetch; ++d) {
            lv_q[d] = lv_q[d + 1];
            lva_q[d] = lva_q[d + 1];
#pragma unroll
            for (int si = 0; si < NS; ++si)
#pragma unroll
                for (int k = 0; k < KS; ++k) occ_q[d][si][k] = occ_q[d + 1][si][k];
        }
        fetch(occ_q[kPrefetch - 1], lv_q[kPrefetch - 1], lva_q[kPrefetch - 1]);
        __syncthreads();
        cur ^= 1;
        tick();
    }
}

template <int DFMT>
static hipError_t launch_occ1(const ChunkParams& p, bool change, hipStream_t s)
{
    const dim3 grid((p.W + 15) / 16, (p.H + 15) / 16, (p.n_steps + kOccGroup - 1) / kOccGroup), block(256);
    if (change) hipLaunchKernelGGL((k_light_occlusion<DFMT, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_light_occlusion<DFMT, false>), grid, block, 0, s, p);
    return hipGetLastError();
}
// computes the occlusion of the chunk described by (j0, n_steps) into {a,r}.occ_next
hipError_t launch_light_occlusion(const ChunkParams& p, bool change, hipStream_t s)
{
    if (p.n_steps <= 0) return hipSuccess;
    switch (p.data.fmt) {
        case FMT_U8: return launch_occ1<FMT_U8>(p, change, s);
        case FMT_U16: return launch_occ1<FMT_U16>(p, change, s);
        default: return launch_occ1<FMT_F32>(p, change, s);
    }
}

template <int LFMT, bool CHANGE>
static hipError_t launch_chain2(const ChunkParams& p, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void) hipFuncSetAttribute((const void*) k_light_chain<LFMT, CHANGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const size_t lds = chunk_lds_bytes(p, CHANGE);
    hipLaunchKernelGGL((k_light_chain<LFMT, CHANGE>), dim3(p.tiles_x, p.tiles_y), dim3(kChunkThreads), lds, s, p);
    return hipGetLastError();
}
// advances every tile through the chunk (j0, n_steps), reading {a,r}.occ_cur
hipError_t launch_light_chain(const ChunkParams& p, bool change, int lv_fmt, hipStream_t s)
{
    if (p.n_steps <= 0 || p.tiles_x <= 0 || p.tiles_y <= 0) return hipSuccess;
    if (lv_fmt == FMT_U8) return change ? launch_chain2<FMT_U8, true>(p, s) : launch_chain2<FMT_U8, false>(p, s);
    return change ? launch_chain2<FMT_F32, true>(p, s) : launch_chain2<FMT_F32, false>(p, s);
}

} // namespace tbrm
