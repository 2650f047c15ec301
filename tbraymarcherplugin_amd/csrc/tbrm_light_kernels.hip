// tbrm_light_kernels.hip — gfx950 kernels of the illumination pass (Sunden/Ropinski selective light updates).
//
//   k_occ_flags, k_occ_compact        : once per axis pass — which 16x16x8 occlusion blocks can only see empty bricks,
//                                       and per span the ascending list of the blocks that cannot.
//   k_light_occlusion                 : once per span of up to 128 slices — the factor 1 - CurrentSample
//                                       (AddDirLightShader.usf:85-117) of every voxel of the span, no halo, fully
//                                       parallel, live blocks dealt evenly over the CUs from the list.
//   k_light_sweep (tbrm_light_sweep.hip): once per axis pass — every 32x32 tile of the slice plane walks ALL slices of the
//                                       pass, the tiles a pipeline with dword hand-offs through memory (UNORM8 light
//                                       volumes, passes of whole brick layers: what a loaded scan produces).
//   k_light_chain (tbrm_light_chain.hip): once per chunk of 16 / 8 / 4 / 2 slices — advances EVERY 32x32 tile of the
//                                       slice plane through the chunk, recomputing a halo instead of waiting for its
//                                       neighbours: the passes the sweep declines, and slab-partitioned passes.
//   k_propagate_slice                 : the reference's structure, one slice per launch (AddDirLightShader.usf:68-128,
//                                       ChangeDirLightShader.usf:74-156). Fallback for passes the chunk kernels
//                                       decline (degenerate offsets) and the A/B baseline (tunable force_slice_kernel).
//
// Why chunks work: slice k only needs the previous slice's propagated light inside a small bilinear footprint,
// offset by the constant PrevPixelOffset. A workgroup that owns a 32x32 tile at the END of a chunk therefore only
// needs a (32 + steps*g)^2 window of the plane at the START of the chunk (g = width of the footprint in texels,
// normally 1) and recomputes that shrinking window privately in LDS — no inter-workgroup traffic inside a chunk,
// one kernel boundary between chunks. The expensive part of a voxel — the windowed, opacity-corrected data sample
// "CurrentSample" (AddDirLightShader.usf:85-114) — does not depend on the propagated light, so it is computed once per
// voxel, without halo, by k_light_occlusion and handed over through a scratch plane stack that k_light_chain stages
// into LDS with asynchronous global->LDS copies two slices ahead of their use; the chain itself is then a handful of
// LDS reads and one multiply per voxel, with no memory latency between slices. Arithmetic per voxel is exactly the
// reference's, including the per-slice UNORM8 re-quantisation of the propagated light (RaymarchVolume.cpp:857-866).
#include "tbrm_device_sampling.h"
#include "tbrm_light_chain.h"

#include <type_traits>

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
        cur = windowed_alpha<DFMT != FMT_F32>(val, s.step100, tf_alpha, p.win) * aw;
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
    const int py = (blockIdx.y + p.row_block0) * 16 + (threadIdx.x >> 4);
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
    const dim3 grid((p.td[0] + 15) / 16, p.row_blocks > 0 ? p.row_blocks : (p.td[1] + 15) / 16), block(256);
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

// LDS bytes of a chain workgroup; 1 GiB when no instantiated kernel shape holds the chunk's hull
// (tbrm_light_chain.hip launch_chain3 lists the shapes: square planes of RS 40 / 56 for everything, 72 for a plain Add —
// ten 72 x 72 planes exceed the LDS — and the rectangular 72 x 48 / 56 x 64 for an Add, UNORM8)
size_t chunk_lds_bytes(const ChunkParams& p, int mode, int lv_fmt)
{
    const ChunkGeom g = chunk_geometry(p);
    const int ns = mode == PASS_ADD ? 1 : 2; // streams propagated = planes staged per slice
    const bool rect = g.RR != g.RS; // 72 x 48 / 56 x 64: instantiated for an Add over a UNORM8 light volume
    if (rect && !(mode == PASS_ADD && lv_fmt == FMT_U8)) return (size_t) 1 << 30;
    if (rect && g.HX * g.HY - kChunkTile * kChunkTile > 2 * kChunkThreads) return (size_t) 1 << 30; // (two halo slots per thread)
    if (g.RS == 0 || (mode != PASS_ADD && !rect && g.RS > 56)) return (size_t) 1 << 30;
    size_t total = (size_t) (2 * ns + kOccRing * ns) * chain_plane_elems(g.RS, g.RR) * 4; // windows + staged ring
    if (lv_fmt == FMT_U8) total += (size_t) 16 * g.lv_layers * 512;                        // light-volume tile
    return total;
}

// ---- k_light_occlusion: CurrentSample (AddDirLightShader.usf:85-114) for every voxel of a chunk ------------------
// One workgroup = 16x16 plane pixels x 8 slices. It first copies every data brick its samples can touch into LDS
// with coalesced 16-byte global->LDS copies (a few dozen wide vector-memory instructions), then each thread filters
// its 8 samples out of LDS: the 8 narrow gather loads per sample that bounded the first version (the texture-address
// path handles ~one 64-lane gather per 20 cycles) become LDS reads. Addressing is separable in the bricked layout:
// the two in-plane axes contribute per-thread constants, the slice axis a per-step value from a small LDS table
// (which also carries the reference's per-thread (Loop+0.5)/res division out of the loop).
constexpr int kOccTile = 16;  // pixels per side
constexpr int kOccMaxBricks = 192; // staged bricks per workgroup (96 KiB of UNORM8 bricks)
constexpr int kOccDepth = 8;  // slices per workgroup (16 was measured: fewer halo bricks per sample, but no faster)
static_assert(kOccDepth == kOccSlices, "host and kernel disagree on the occlusion workgroup depth");

struct AxisTaps {        // one axis of a trilinear footprint with border addressing
    int i0;              // base tap index (unclamped)
    bool ok0, ok1;       // tap inside the volume (else the sampler's border colour applies)
    float f;             // interpolation weight
};

__device__ __forceinline__ AxisTaps axis_taps(float coord, int n)
{
    AxisTaps t;
    texel_split(coord, (float) n, t.i0, t.f);
    t.ok0 = (unsigned) t.i0 < (unsigned) n;
    t.ok1 = (unsigned) (t.i0 + 1) < (unsigned) n;
    return t;
}

// offset (in voxels) of coordinate c along `axis` inside the staged brick block: local brick index * 512 + in-brick part
template <int AXIS>
__device__ __forceinline__ uint32_t staged_off(int c, const int* b0, const int* nb)
{
    const int lb = (c >> 3) - b0[AXIS];
    const uint32_t stride = AXIS == 0 ? 1u : (AXIS == 1 ? (uint32_t) nb[0] : (uint32_t) (nb[0] * nb[1]));
    const uint32_t inb = (uint32_t) (c & 7) << (3 * AXIS);
    return (uint32_t) lb * stride * 512u + inb;
}

template <int FMT>
__device__ __forceinline__ float lds_voxel(const void* p, uint32_t i)
{
    if constexpr (FMT == FMT_U8) return decode_u8(((const uint8_t*) p)[i]);
    else if constexpr (FMT == FMT_U16) return decode_u16(((const uint16_t*) p)[i]);
    else return ((const float*) p)[i];
}

// LDS bytes for the staged bricks: worst case over workgroups, from the ratio of data to light-volume resolution
size_t occlusion_lds_bytes(const ChunkParams& p)
{
    const int dim_u = p.axis == 0 ? 1 : 0, dim_v = p.axis == 2 ? 1 : 2, dim_s = p.axis;
    const int dd[3] = {p.data.nx, p.data.ny, p.data.nz};
    auto bricks = [&](int pixels, int dim) {
        const double ratio = (double) dd[dim] / (double) p.lv_dims[dim];
        const int texels = (int) ((pixels - 1) * ratio) + 2; // first base tap .. last +1 tap (a workgroup that needs one
        return (texels - 1 + 7) / 8 + 1;                     // more brick than this reads its taps from global memory)
    };
    const size_t n = (size_t) bricks(kOccTile, dim_u) * bricks(kOccTile, dim_v) * bricks(kOccDepth, dim_s);
    const size_t esz = p.data.fmt == FMT_U8 ? 1 : (p.data.fmt == FMT_U16 ? 2 : 4);
    return n * 512 * esz;
}

// ---- k_occ_flags: once per pass, which occlusion workgroups are empty ---------------------------------------------
// One thread per occlusion workgroup (16x16 pixels x 8 slices of one chunk). The workgroup is empty when the brick of
// every sample's base tap has its k_brick_empty bit set: the bit vouches for every value the 8 taps of a sample based
// in that brick can take (the brick plus its +1 apron), so every CurrentSample is exactly 0. Workgroups with taps
// outside the volume are never flagged (a blend with the sampler's border colour can leave both value ranges).
template <int MODE, int AXIS>
__global__ __launch_bounds__(256) void k_occ_flags(const ChunkParams p, int n_chunks)
{
    constexpr int NS = MODE != PASS_ADD ? 2 : 1;
    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS;
    const int per_chunk = p.occ_groups * p.occ_blocks_y * p.occ_blocks_x;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n_chunks * per_chunk) return;
    const int c = id / per_chunk, rem = id % per_chunk;
    const int bx = rem % p.occ_blocks_x, by = (rem / p.occ_blocks_x) % p.occ_blocks_y, zg = rem / (p.occ_blocks_x * p.occ_blocks_y);
    const int n = min(p.chunk_slices, p.pass_slices - c * p.chunk_slices);
    const int k0 = zg * kOccDepth;
    uint8_t flag = 0;
    if (by < p.roi_by0 || by >= p.roi_by1) flag = 1; // outside the slab's reach: never computed, never read
    else if (k0 < n) {
        const int nk = min(kOccDepth, n - k0);
        const int j0 = p.pass_start + (c * p.chunk_slices + k0) * p.dir, j1 = j0 + (nk - 1) * p.dir;
        const int px0 = bx * kOccTile, py0 = by * kOccTile;
        const int pxl = min(px0 + kOccTile, p.W) - 1, pyl = min(py0 + kOccTile, p.H) - 1;
        const int dn[3] = {p.data.nx, p.data.ny, p.data.nz};
        const int bn[3] = {p.data.bnx, p.data.bnxy / p.data.bnx, (p.data.nz + 7) >> 3};
        int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            const int ends[3][2] = {{px0, pxl}, {py0, pyl}, {j0, j1}};
            constexpr int dims[3] = {dim_u, dim_v, dim_s};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float cc = (((float) (uint32_t) ends[a][e] + 0.5f) / (float) (uint32_t) p.lv_dims[dims[a]]) + s.uvw_off[dims[a]];
                    int i0;
                    float f;
                    texel_split(cc, (float) dn[dims[a]], i0, f);
                    lo[dims[a]] = min(lo[dims[a]], i0);
                    hi[dims[a]] = max(hi[dims[a]], i0 + 1);
                }
        }
        bool ok = true;
        int b_lo[3], b_hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ok = ok && lo[a] >= 0 && hi[a] < dn[a]; // every tap inside the volume
            b_lo[a] = max(lo[a] >> 3, 0);
            b_hi[a] = min(hi[a] >> 3, bn[a] - 1);
        }
        if (ok) {
            for (int z = b_lo[2]; z <= b_hi[2] && ok; ++z)
                for (int y = b_lo[1]; y <= b_hi[1] && ok; ++y)
                    for (int x = b_lo[0]; x <= b_hi[0]; ++x) {
                        const int b = z * p.data.bnxy + y * p.data.bnx + x;
                        if (!((p.empty_bits[b >> 5] >> (b & 31)) & 1u)) { ok = false; break; }
                    }
            flag = ok ? 1 : 0;
        }
    }
    p.occ_flags_out[id] = flag;
}

// ---- k_occ_compact: per chunk, the ascending list of workgroups that are NOT flagged -----------------------------
// A workgroup takes a segment of 4096 of the chunk's flags (16 per thread, one 16-byte load). It first counts the live
// flags of every segment before its own — up to 64 KiB of coalesced reads out of the L2, instead of a second kernel or a
// chain of workgroups waiting for each other — then scans its own: every live workgroup gets its list position
// (deterministic, ascending) and, if wanted, every block its rank (-1: flagged). Flags are bytes of 0 / 1.
constexpr int kCompactSeg = 4096;
__global__ __launch_bounds__(256) void k_occ_compact(const ChunkParams p, int segs)
{
    constexpr int NT = 256;
    __shared__ int s_scan[NT];
    __shared__ int s_base;
    const int per_chunk = p.occ_groups * p.occ_blocks_y * p.occ_blocks_x;
    const int c = (int) blockIdx.x / segs, seg = (int) blockIdx.x % segs;
    const uint8_t* flags = p.occ_flags_out + (size_t) c * per_chunk;
    uint32_t* list = p.occ_list_out + (size_t) c * per_chunk;
    int32_t* const slot = p.occ_slot_out ? p.occ_slot_out + (size_t) c * per_chunk : nullptr;
    const int n = min(p.chunk_slices, p.pass_slices - c * p.chunk_slices);
    const int live_groups = (n + kOccDepth - 1) / kOccDepth;       // slice groups past the chunk's last slice have no work
    const int live = live_groups * p.occ_blocks_y * p.occ_blocks_x;
    const bool wide = (((size_t) flags) & 15) == 0;
    // 16 flags from position i on as four words (bytes past the live part read as flagged)
    auto load16 = [&](int i, uint32_t (&w)[4]) {
        if (wide && i + 16 <= live) {
            const uint4 v = *(const uint4*) (flags + i);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) w[k] |= (uint32_t) ((i + 4 * k + b < live) ? (flags[i + 4 * k + b] ? 1u : 0u) : 1u) << (8 * b);
            }
        }
    };
    auto live_of = [](const uint32_t (&w)[4]) { return 16 - (__builtin_popcount(w[0] & 0x01010101u) + __builtin_popcount(w[1] & 0x01010101u) + __builtin_popcount(w[2] & 0x01010101u) + __builtin_popcount(w[3] & 0x01010101u)); };
    // live flags before this segment
    int before = 0;
    for (int i = (int) threadIdx.x * 16; i < seg * kCompactSeg; i += NT * 16) {
        uint32_t w[4];
        load16(i, w);
        before += live_of(w);
    }
    s_scan[threadIdx.x] = before;
    __syncthreads();
    for (int d = NT / 2; d > 0; d >>= 1) {
        if ((int) threadIdx.x < d) s_scan[threadIdx.x] += s_scan[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) s_base = s_scan[0];
    __syncthreads();
    const int base = s_base;
    __syncthreads();
    // this segment
    const int i0 = seg * kCompactSeg + (int) threadIdx.x * 16;
    uint32_t w[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
    if (i0 < per_chunk) load16(i0, w);
    const int mine = i0 < per_chunk ? live_of(w) : 0;
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) { // inclusive Hillis-Steele scan
        const int v = (int) threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = base + s_scan[threadIdx.x] - mine;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = i0 + k;
        if (i >= per_chunk) break;
        const bool flagged = ((w[k >> 2] >> (8 * (k & 3))) & 1u) != 0;
        if (slot) slot[i] = flagged ? -1 : pos;
        if (!flagged) list[pos++] = (uint32_t) i;
    }
    if (seg == segs - 1 && threadIdx.x == NT - 1) {
        p.occ_count_out[c] = base + s_scan[NT - 1];
        if (p.occ_count_host && c == 0) *p.occ_count_host = base + s_scan[NT - 1]; // (visible to the host once an event behind the launch has completed)
    }
}

template <int MODE>
static hipError_t launch_flags2(const ChunkParams& p, int n_chunks, hipStream_t s)
{
    const int total = n_chunks * p.occ_groups * p.occ_blocks_y * p.occ_blocks_x;
    const dim3 grid((total + 255) / 256), block(256);
    if (p.axis == 0) hipLaunchKernelGGL((k_occ_flags<MODE, 0>), grid, block, 0, s, p, n_chunks);
    else if (p.axis == 1) hipLaunchKernelGGL((k_occ_flags<MODE, 1>), grid, block, 0, s, p, n_chunks);
    else hipLaunchKernelGGL((k_occ_flags<MODE, 2>), grid, block, 0, s, p, n_chunks);
    if (p.occ_list_out) {
        const int per_chunk = p.occ_groups * p.occ_blocks_y * p.occ_blocks_x, segs = (per_chunk + kCompactSeg - 1) / kCompactSeg;
        hipLaunchKernelGGL(k_occ_compact, dim3(n_chunks * segs), dim3(256), 0, s, p, segs);
    }
    return hipGetLastError();
}
hipError_t launch_occ_flags(const ChunkParams& p, int mode, int n_chunks, hipStream_t s)
{
    const bool one = mode == PASS_ADD || mode == PASS_CHANGE_ONE;
    return one ? launch_flags2<PASS_ADD>(p, n_chunks, s) : launch_flags2<PASS_CHANGE>(p, n_chunks, s); // flags only depend on the stream count
}

// DUAL (tbrm_internal.h DualOcc): the launch walks the light volume as a virtual pass along AXIS (z: see DualOcc) and every
// voxel's sample serves both real passes — one filter, one transfer-function look-up, one logarithm, two exponents.
template <int DFMT, int MODE, int AXIS, bool DUAL = false>
#ifndef TBRM_OCC_WAVES_PER_EU
#define TBRM_OCC_WAVES_PER_EU 5
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TBRM_OCC_WAVES_PER_EU, 8))) void k_light_occlusion(const ChunkParams p, int lds_budget_bytes, const DualOcc d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = (MODE == PASS_ADD || MODE == PASS_CHANGE_ONE) ? 1 : 2; // PASS_CHANGE_ONE: one stream with the Change shader's rules
    constexpr bool GUARD = MODE == PASS_ADD || MODE == PASS_ADD2; // all(uvw == saturate(uvw)): the Add shader only (AddDirLightShader.usf:98)
    constexpr int ESZ = DFMT == FMT_U8 ? 1 : (DFMT == FMT_U16 ? 2 : 4);
    __shared__ float2 s_alpha[257]; // pairs (alpha[clamp(i)], alpha[clamp(i + 1)]) at i + 1: sample_tf_alpha
    __shared__ float s_w[2][kOccDepth], s_f[2][kOccDepth];
    __shared__ int s_i[2][kOccDepth], s_flags[2][kOccDepth]; // flags: bit0 tap0 in range, bit1 tap1 in range, bit2 w == saturate(w)
    __shared__ int s_b0[3], s_nb[3], s_staged, s_interior;
    __shared__ uint32_t s_o0[2][kOccDepth], s_o1[2][kOccDepth];
    __shared__ int s_ends[12], s_end_dim[12];
    __shared__ uint32_t s_brick[kOccMaxBricks];

    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS; // plane axes -> volume axes
    const int data_dims[3] = {p.data.nx, p.data.ny, p.data.nz};
    // The launch is a 1-D grid rounded up to a multiple of 8 workgroups. Workgroups are dealt to the 8 XCDs round-robin by
    // id (an affinity used for speed only): XCD x takes the x-th eighth of the work, so blocks that are neighbours in the
    // volume — and stage the same halo bricks — run behind the same L2.
    //
    // The grid may be smaller than the work (a launch that runs beside the chain of the previous span on the occlusion
    // stream, tbrm_light_enqueue.cpp, holds only as many workgroups as fit next to the chain's on every CU, all resident from
    // the start, so none is ever waiting to take a slot the next chain launch needs): a workgroup then walks its XCD's
    // eighth with the stride of the grid.
    const int groups = (p.n_steps + kOccDepth - 1) / kOccDepth;
    const int total = p.occ_list ? *p.occ_count : groups * p.occ_blocks_y * p.occ_blocks_x;
    const int per_xcd = (total + 7) >> 3;
    const int xcd_stride = (int) gridDim.x >> 3;
  for (int slot = (int) blockIdx.x >> 3; slot < per_xcd; slot += xcd_stride) {
    const int entry = ((int) blockIdx.x & 7) * per_xcd + slot;
    if (entry >= total) break;
    if (slot != ((int) blockIdx.x >> 3)) __syncthreads(); // the previous block's tables and staged bricks are done with
    // which block of the span: the entry itself, or (sparse spans) that entry of the span's work list — workgroups flagged
    // empty by k_occ_flags (every CurrentSample exactly 0, and the chain knows it) are not on the list
    const int id = p.occ_list ? (int) p.occ_list[entry] : entry;
    const int gx = id % p.occ_blocks_x, gy = (id / p.occ_blocks_x) % p.occ_blocks_y, gz = id / (p.occ_blocks_x * p.occ_blocks_y);
    if (!p.occ_list && p.occ_flags && p.occ_flags[id]) continue; // list off (A/B runs)
    if (gy < p.roi_by0 || gy >= p.roi_by1) continue;             // dense span of a slab-partitioned pass
    const int px0 = gx * kOccTile, py0 = gy * kOccTile, k0 = gz * kOccDepth;
    const int nk = min(kOccDepth, p.n_steps - k0);

    {
        const float a_t = p.tf[threadIdx.x].w, a_n = p.tf[min((int) threadIdx.x + 1, 255)].w;
        s_alpha[threadIdx.x + 1] = make_float2(a_t, a_n);
        if (threadIdx.x == 0) s_alpha[0] = make_float2(a_t, a_t);
    }
    if (threadIdx.x < NS * kOccDepth) { // slice-axis taps of each step of this workgroup (wave-uniform values)
        const int si = threadIdx.x / kOccDepth, q = threadIdx.x % kOccDepth;
        const ChunkStream& s = si == 0 ? p.a : p.r;
        const int j = p.j0 + min(k0 + q, p.n_steps - 1) * p.dir;
        const float w = (((float) (uint32_t) j + 0.5f) / (float) (uint32_t) p.lv_dims[dim_s]) + s.uvw_off[dim_s];
        const AxisTaps t = axis_taps(w, data_dims[dim_s]);
        s_w[si][q] = w; s_f[si][q] = t.f; s_i[si][q] = t.i0;
        s_flags[si][q] = (t.ok0 ? 1 : 0) | (t.ok1 ? 2 : 0) | ((w == saturate_(w)) ? 4 : 0);
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + NS * 6) { // tap range of the first / last pixel and slice, one lane each
        const int t = threadIdx.x - 64, si = t / 6, a = (t % 6) >> 1, e = t & 1;
        const ChunkStream& s = si == 0 ? p.a : p.r;
        const int pxl = min(px0 + kOccTile, p.W) - 1, pyl = min(py0 + kOccTile, p.H) - 1;
        const int end = a == 0 ? (e ? pxl : px0) : (a == 1 ? (e ? pyl : py0) : p.j0 + (e ? k0 + nk - 1 : k0) * p.dir);
        const int dim = a == 0 ? dim_u : (a == 1 ? dim_v : dim_s);
        const int lvd = dim == 0 ? p.lv_dims[0] : (dim == 1 ? p.lv_dims[1] : p.lv_dims[2]);
        const float off = dim == 0 ? s.uvw_off[0] : (dim == 1 ? s.uvw_off[1] : s.uvw_off[2]);
        const int dd = dim == 0 ? p.data.nx : (dim == 1 ? p.data.ny : p.data.nz);
        const float c = (((float) (uint32_t) end + 0.5f) / (float) (uint32_t) lvd) + off;
        int i0;
        float f;
        texel_split(c, (float) dd, i0, f);
        s_ends[t] = i0;
        s_end_dim[t] = dim;
    }
    __syncthreads();
    if (threadIdx.x == 0) { // brick range the workgroup's samples can touch (union over the streams)
        int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
        for (int t = 0; t < NS * 6; ++t) {
            const int d = s_end_dim[t], i0 = s_ends[t];
            if (d == 0) { lo[0] = min(lo[0], i0); hi[0] = max(hi[0], i0 + 1); }
            else if (d == 1) { lo[1] = min(lo[1], i0); hi[1] = max(hi[1], i0 + 1); }
            else { lo[2] = min(lo[2], i0); hi[2] = max(hi[2], i0 + 1); }
        }
        const int bn[3] = {p.data.bnx, p.data.bnxy / p.data.bnx, (p.data.nz + 7) >> 3};
        const int dn[3] = {p.data.nx, p.data.ny, p.data.nz};
        int count = 1;
        bool touches_border = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int b_lo = max(lo[a] >> 3, 0), b_hi = min(hi[a] >> 3, bn[a] - 1);
            s_b0[a] = b_lo;
            s_nb[a] = max(b_hi - b_lo + 1, 0);
            count *= s_nb[a];
            touches_border = touches_border || lo[a] < 0 || hi[a] >= dn[a];
        }
        s_staged = (count > 0 && count <= kOccMaxBricks && count * 512 * ESZ <= lds_budget_bytes) ? 1 : 0; // else: read taps from global memory
        s_interior = touches_border ? 0 : 1;
    }
    __syncthreads();

    const bool staged = s_staged != 0;
    const int b0[3] = {s_b0[0], s_b0[1], s_b0[2]}, nb[3] = {s_nb[0], s_nb[1], s_nb[2]};
    if (staged) { // copy the bricks: 512*ESZ bytes each, 16 bytes per lane
        constexpr int PIECES = 512 * ESZ / 16; // a power of two
        const int bricks = nb[0] * nb[1] * nb[2];
        // global index of every staged brick, once (the integer divisions stay out of the copy loop)
        if ((int) threadIdx.x < bricks) {
            const int lb = threadIdx.x;
            const int lx = lb % nb[0], ly = (lb / nb[0]) % nb[1], lz = lb / (nb[0] * nb[1]);
            s_brick[lb] = (uint32_t) ((b0[2] + lz) * p.data.bnxy + (b0[1] + ly) * p.data.bnx + (b0[0] + lx));
        }
        __syncthreads();
        const int total = bricks * PIECES;
        const int wave_base = (threadIdx.x >> 6) * 64, lane = threadIdx.x & 63;
        for (int cb = wave_base; cb < total; cb += 256) {
            const int c = cb + lane;
            if (c < total) {
                const uint32_t gb = s_brick[c / PIECES];
                dma_16((const char*) p.data.data + ((size_t) gb * 512 * ESZ + (size_t) (c % PIECES) * 16), smem + (size_t) cb * 16);
            }
        }
    }

    // offset of voxel coordinate c along axis A: in the staged block, or in the bricked global volume
    auto voff = [&](int c, auto axis_c) -> uint32_t {
        constexpr int axis = decltype(axis_c)::value;
        const int n = data_dims[axis];
        c = min(max(c, 0), n - 1); // clamped: out-of-range taps are replaced by the border colour after the load
        if (staged) {
            const int cb = min(max(c >> 3, b0[axis]), b0[axis] + nb[axis] - 1); // stay inside the staged block
            return staged_off<axis>((cb << 3) | (c & 7), b0, nb);
        }
        return axis == 0 ? brick_off_x(c) : (axis == 1 ? brick_off_y(c, p.data.bnx) : brick_off_z(c, p.data.bnxy));
    };
    using AU = std::integral_constant<int, dim_u>; using AV = std::integral_constant<int, dim_v>; using AS = std::integral_constant<int, dim_s>;
    if (threadIdx.x < NS * kOccDepth) { // slice-axis tap offsets of each step (wave-uniform), in the layout just chosen
        const int si = threadIdx.x / kOccDepth, q = threadIdx.x % kOccDepth;
        s_o0[si][q] = voff(s_i[si][q], AS{});
        s_o1[si][q] = voff(s_i[si][q] + 1, AS{});
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = px0 + (wave & 1) * 8 + (lane & 7);
    const int py = py0 + (wave >> 1) * 8 + (lane >> 3);
    const bool pixel_ok = px < p.W && py < p.H;
    const int plane_elems = p.H * p.W;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the sample loop, specialised on workgroup-uniform facts so the common case carries no dead branches:
    //   STAGED   taps come from LDS (else from global memory: a workgroup whose bricks did not fit)
    //   INTERIOR every tap of every sample lies inside the volume (no border-colour selects)
    //   CLIP     the clip plane can change a sample's weight (else AlphaWeight is exactly 1)
    auto run = [&](auto staged_c, auto interior_c, auto clip_c) {
        constexpr bool STAGED = decltype(staged_c)::value, INTERIOR = decltype(interior_c)::value, CLIP = decltype(clip_c)::value;
        const float border = p.data_border;
        auto tap = [&](uint32_t off, bool ok) -> float {
            if constexpr (INTERIOR) return STAGED ? lds_voxel<DFMT>(smem, off) : load_voxel<DFMT>(p.data.data, off);
            else return ok ? (STAGED ? lds_voxel<DFMT>(smem, off) : load_voxel<DFMT>(p.data.data, off)) : border;
        };
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            // GetUVW(pos, res) + UVWOffset (AddDirLightShader.usf:85): the two in-plane components
            const float u = (((float) (uint32_t) px + 0.5f) / (float) (uint32_t) p.lv_dims[dim_u]) + s.uvw_off[dim_u];
            const float v = (((float) (uint32_t) py + 0.5f) / (float) (uint32_t) p.lv_dims[dim_v]) + s.uvw_off[dim_v];
            const AxisTaps tu = axis_taps(u, data_dims[dim_u]), tv = axis_taps(v, data_dims[dim_v]);
            const bool guard_uv = (u == saturate_(u)) && (v == saturate_(v));
            const uint32_t u0 = voff(tu.i0, AU{}), u1 = voff(tu.i0 + 1, AU{}), v0 = voff(tv.i0, AV{}), v1 = voff(tv.i0 + 1, AV{});
            const uint32_t o00 = u0 + v0, o10 = u1 + v0, o01 = u0 + v1, o11 = u1 + v1;
            const bool k00 = tu.ok0 && tv.ok0, k10 = tu.ok1 && tv.ok0, k01 = tu.ok0 && tv.ok1, k11 = tu.ok1 && tv.ok1;
            // where the factors go: the span's plane stack, or (block-compact hand-over) the block's own 8 x 16 x 16 floats —
            // `entry` is the block's rank in the pass's work list
            float* out = nullptr;
            int out_step = 0;
            float* out2[2] = {nullptr, nullptr}; // DUAL: where this thread's voxels go in the two passes' stores (null: block flagged empty)
            int out2_step[2] = {0, 0};
            if constexpr (DUAL) {
                int pos[3];
                pos[dim_u] = px; pos[dim_v] = py; pos[dim_s] = p.j0 + k0 * p.dir;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const DualPass& P = d.pass[k];
                    const int pa = P.axis == 0 ? pos[0] : (P.axis == 1 ? pos[1] : pos[2]);
                    const int pu = P.axis == 0 ? pos[1] : pos[0], pv = P.axis == 2 ? pos[1] : pos[2]; // the pass's plane coordinates
                    const int ks = (pa - P.start) * P.dir;                                              // its slice
                    const int32_t rank = P.fs_slot[((ks >> 3) * P.blocks_y + (pv >> 4)) * P.blocks_x + (pu >> 4)];
                    if (rank >= 0) {
                        float* const blk_base = (uint32_t) rank < P.fs_cap[si] ? P.fs_keep[si] + (size_t) (uint32_t) rank * 2048 : P.fs_spill[si] + (size_t) ((uint32_t) rank - P.fs_cap[si]) * 2048;
                        out2[k] = blk_base + (ks & 7) * 256 + (pv & 15) * 16 + (pu & 15);
                    }
                    // one step along the launch's loop axis, in pass k's block: its own slice axis (a block slice is 256 floats), its
                    // columns, or its rows
                    out2_step[k] = (dim_s == P.axis ? 256 * P.dir : (dim_s == (P.axis == 0 ? 1 : 0) ? 1 : kOccTile)) * p.dir;
                }
            } else {
                out = s.occ_next + k0 * plane_elems + py * p.W + px;
                out_step = plane_elems;
                if (p.compact) {
                    float* const blk_base = (uint32_t) entry < s.fs_cap ? s.fs_keep + (size_t) entry * 2048 : s.fs_spill + (size_t) ((uint32_t) entry - s.fs_cap) * 2048;
                    out = blk_base + (py - py0) * kOccTile + (px - px0);
                    out_step = kOccTile * kOccTile;
                }
            }
            // One texel plane of the footprint (4 taps at one slice-axis coordinate), reduced as far as the filter order
            // (x, then y, then z) allows before the slice-axis weight is applied. Consecutive slices of a pass usually
            // sample consecutive texel planes, so the +1 plane of one step is the base plane of the next: it is kept in
            // registers and only one new plane (4 LDS taps + decode) is fetched per step instead of two.
            constexpr int PV = AXIS == 2 ? 1 : (AXIS == 1 ? 2 : 4);
            struct PlaneVal { float v[PV]; };
            auto fetch_plane = [&](int q, bool upper) -> PlaneVal {
                const uint32_t w = upper ? s_o1[si][q] : s_o0[si][q];
                const bool a = (s_flags[si][q] & (upper ? 2 : 1)) != 0;
                // tap (u, v): bit0 = u tap, bit1 = v tap
                const float t00 = tap(o00 + w, k00 && a), t10 = tap(o10 + w, k10 && a);
                const float t01 = tap(o01 + w, k01 && a), t11 = tap(o11 + w, k11 && a);
                PlaneVal r;
                if constexpr (AXIS == 2) r.v[0] = lerp_(lerp_(t00, t10, tu.f), lerp_(t01, t11, tu.f), tv.f); // (u,v,s) = (x,y,z)
                else if constexpr (AXIS == 1) { r.v[0] = lerp_(t00, t10, tu.f); r.v[1] = lerp_(t01, t11, tu.f); } // x = u, y = slice, z = v
                else { r.v[0] = t00; r.v[1] = t10; r.v[2] = t01; r.v[3] = t11; }                               // x = slice, y = u, z = v
                return r;
            };
            auto combine = [&](const PlaneVal& lo, const PlaneVal& hi, float fs) -> float {
                if constexpr (AXIS == 2) return lerp_(lo.v[0], hi.v[0], fs);
                else if constexpr (AXIS == 1) return lerp_(lerp_(lo.v[0], hi.v[0], fs), lerp_(lo.v[1], hi.v[1], fs), tv.f);
                else return lerp_(lerp_(lerp_(lo.v[0], hi.v[0], fs), lerp_(lo.v[1], hi.v[1], fs), tu.f),
                                  lerp_(lerp_(lo.v[2], hi.v[2], fs), lerp_(lo.v[3], hi.v[3], fs), tu.f), tv.f);
            };
            PlaneVal lo{}, hi{};
            int held = INT32_MIN / 2; // slice-axis texel index of `lo`; `hi` is the plane after it

            // (the opacity correction's step sizes are >= 0 unless the host was handed garbage: decided here, outside the loop)
            auto slices = [&](auto nonneg_c) {
            constexpr bool NONNEG = decltype(nonneg_c)::value;
            // the step sizes in vector registers (as scalars they are spilled around this loop and fetched back every trip)
            float step_v0 = DUAL ? d.pass[0].step100[si] : s.step100, step_v1 = DUAL ? d.pass[1].step100[si] : 0.0f;
            asm volatile("" : "+v"(step_v0), "+v"(step_v1));
            for (int q = 0; q < nk; ++q) {
                const float fs = s_f[si][q];
                const int fl = s_flags[si][q];
                const int i = __builtin_amdgcn_readfirstlane(s_i[si][q]); // wave-uniform: scalar branches below
                if (i == held + 1) { lo = hi; hi = fetch_plane(q, true); }
                else if (i == held - 1) { hi = lo; lo = fetch_plane(q, false); }
                else if (i != held) { lo = fetch_plane(q, false); hi = fetch_plane(q, true); }
                held = i;
                float aw = 1.0f;
                if constexpr (CLIP) {
                    const float w = s_w[si][q];
                    float c0, c1, c2;
                    if (AXIS == 0) { c0 = w; c1 = u; c2 = v; } else if (AXIS == 1) { c0 = u; c1 = w; c2 = v; } else { c0 = u; c1 = v; c2 = w; }
                    aw = clip_alpha_weight(c0, c1, c2, p.cc, p.cd, p.lv_dims);
                }
                bool inside = true;
                if constexpr (GUARD) inside = guard_uv && (fl & 4);
                if constexpr (DUAL) {
                    float occ0 = 0.0f, occ1 = 0.0f;
                    if (aw > 0.0f && inside) {
                        windowed_alpha2<DFMT != FMT_F32, NONNEG>(combine(lo, hi, fs), step_v0, step_v1, s_alpha, p.win, occ0, occ1);
                        occ0 = occ0 * aw;
                        occ1 = occ1 * aw;
                    }
                    if (out2[0]) out2[0][q * out2_step[0]] = 1 - occ0;
                    if (out2[1]) out2[1][q * out2_step[1]] = 1 - occ1;
                    continue;
                }
                float occ = 0.0f;
                if (aw > 0.0f && inside) occ = windowed_alpha<DFMT != FMT_F32, NONNEG>(combine(lo, hi, fs), step_v0, s_alpha, p.win) * aw;
                out[q * out_step] = 1 - occ; // handed over as the factor of AddDirLightShader.usf:117
            }
            };
            const bool steps_nonneg = DUAL ? (d.pass[0].step100[si] >= 0.0f && d.pass[1].step100[si] >= 0.0f) : s.step100 >= 0.0f;
            if (steps_nonneg) slices(std::true_type{});
            else slices(std::false_type{});
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const bool interior = s_interior != 0;
    if (!pixel_ok) continue;
    if (staged && interior && !p.clip_mode) run(T_{}, T_{}, F_{});       // the common case
    else if (staged && !p.clip_mode) run(T_{}, F_{}, F_{});               // shell of the volume
    else if (staged) run(T_{}, F_{}, T_{});                               // clip plane active
    else run(F_{}, F_{}, T_{});                                           // bricks did not fit in LDS
  }
}

template <int DFMT, int MODE, int AXIS>
static hipError_t launch_occ3(const ChunkParams& p, hipStream_t s, const DualOcc* dual)
{
    const int blocks = ((p.W + kOccTile - 1) / kOccTile) * ((p.H + kOccTile - 1) / kOccTile) * ((p.n_steps + kOccDepth - 1) / kOccDepth);
    int wgs = 8 * ((blocks + 7) / 8);
    if (p.occ_grid_cap > 0) wgs = std::min(wgs, 8 * ((p.occ_grid_cap + 7) / 8));
    const dim3 grid(wgs), block(256);
    size_t lds = occlusion_lds_bytes(p);
    if (lds > 96 * 1024) lds = 96 * 1024; // workgroups whose bricks do not fit read their taps from global memory
    if (dual) {
        if constexpr (MODE == PASS_ADD2 || AXIS != 2) return hipErrorInvalidConfiguration; // (dual launches loop along z: DualOcc)
        else {
            if (!p.occ_list || !p.occ_count || p.dir != 1 || p.j0 != 0) return hipErrorInvalidConfiguration; // (the units come from their work list)
            static std::atomic<uint64_t> attr_done2{0};
            if (const hipError_t e = allow_big_lds(k_light_occlusion<DFMT, MODE, AXIS, true>, attr_done2, 128 * 1024); e != hipSuccess) return e;
            hipLaunchKernelGGL((k_light_occlusion<DFMT, MODE, AXIS, true>), grid, block, lds, s, p, (int) lds, *dual);
            return hipGetLastError();
        }
    }
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_occlusion<DFMT, MODE, AXIS>, attr_done, 128 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_light_occlusion<DFMT, MODE, AXIS>), grid, block, lds, s, p, (int) lds, DualOcc{});
    return hipGetLastError();
}
template <int DFMT, int MODE>
static hipError_t launch_occ2(const ChunkParams& p, hipStream_t s, const DualOcc* dual)
{
    return p.axis == 0 ? launch_occ3<DFMT, MODE, 0>(p, s, dual) : (p.axis == 1 ? launch_occ3<DFMT, MODE, 1>(p, s, dual) : launch_occ3<DFMT, MODE, 2>(p, s, dual));
}
template <int DFMT>
static hipError_t launch_occ1(const ChunkParams& p, int mode, hipStream_t s, const DualOcc* dual)
{
    return mode == PASS_ADD ? launch_occ2<DFMT, PASS_ADD>(p, s, dual)
           : (mode == PASS_CHANGE ? launch_occ2<DFMT, PASS_CHANGE>(p, s, dual) : (mode == PASS_ADD2 ? launch_occ2<DFMT, PASS_ADD2>(p, s, dual) : launch_occ2<DFMT, PASS_CHANGE_ONE>(p, s, dual)));
}
// computes the occlusion of the chunk described by (j0, n_steps) into {a,r}.occ_next; dual: of BOTH passes of a light, p
// describing the virtual pass along z (tbrm_internal.h DualOcc)
hipError_t launch_light_occlusion(const ChunkParams& p, int mode, hipStream_t s, const DualOcc* dual)
{
    if (p.n_steps <= 0) return hipSuccess;
    switch (p.data.fmt) {
        case FMT_U8: return launch_occ1<FMT_U8>(p, mode, s, dual);
        case FMT_U16: return launch_occ1<FMT_U16>(p, mode, s, dual);
        default: return launch_occ1<FMT_F32>(p, mode, s, dual);
    }
}

// ---- k_unit_flags: a dual launch's work units that have nothing to do ---------------------------------------------------------
// A unit (16 x 16 voxels in x and y, 8 in z) is part of up to two 16 x 16 x 8 blocks of each pass (one, when the pass runs along
// z: then the unit IS the block); it is flagged when all of them are (then none has a rank, and nothing would be stored). One
// thread per unit.
__global__ __launch_bounds__(256) void k_unit_flags(const ChunkParams pc, const DualOcc d, int n_units)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n_units) return;
    const int gx = id % pc.occ_blocks_x, gy = (id / pc.occ_blocks_x) % pc.occ_blocks_y, gz = id / (pc.occ_blocks_x * pc.occ_blocks_y);
    const int dim_u = pc.axis == 0 ? 1 : 0, dim_v = pc.axis == 2 ? 1 : 2;
    int pos0[3];
    pos0[dim_u] = gx * kOccTile; pos0[dim_v] = gy * kOccTile; pos0[pc.axis] = gz * kOccDepth;
    bool all_flagged = true;
    for (int k = 0; k < 2; ++k) {
        const DualPass& P = d.pass[k];
        for (int half = 0; half < (P.axis == pc.axis ? 1 : 2); ++half) { // (16 voxels of the pass's axis unless it is the loop axis: two slice groups)
            int pos[3] = {pos0[0], pos0[1], pos0[2]};
            pos[P.axis] += 8 * half;
            if (pos[P.axis] >= pc.lv_dims[P.axis]) continue;
            const int pu = P.axis == 0 ? pos[1] : pos[0], pv = P.axis == 2 ? pos[1] : pos[2];
            const int ks = (pos[P.axis] - P.start) * P.dir;
            all_flagged = all_flagged && P.flags[((ks >> 3) * P.blocks_y + (pv >> 4)) * P.blocks_x + (pu >> 4)] != 0;
        }
    }
    pc.occ_flags_out[id] = all_flagged ? 1 : 0;
}

hipError_t launch_unit_flags(const ChunkParams& pc, const DualOcc& d, hipStream_t s)
{
    const int per_chunk = pc.occ_groups * pc.occ_blocks_y * pc.occ_blocks_x;
    hipLaunchKernelGGL(k_unit_flags, dim3((per_chunk + 255) / 256), dim3(256), 0, s, pc, d, per_chunk);
    const int segs = (per_chunk + kCompactSeg - 1) / kCompactSeg;
    hipLaunchKernelGGL(k_occ_compact, dim3(segs), dim3(256), 0, s, pc, segs);
    return hipGetLastError();
}

hipError_t launch_light_chain(const ChunkParams& p, int mode, int lv_fmt, hipStream_t s)
{
    if (p.n_steps <= 0 || p.tiles_x <= 0 || p.tiles_y <= 0) return hipSuccess;
    return lv_fmt == FMT_U8 ? launch_light_chain_u8(p, mode, s) : launch_light_chain_f32(p, mode, s);
}

} // namespace tbrm
