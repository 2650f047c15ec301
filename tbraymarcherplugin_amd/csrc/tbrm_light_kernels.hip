// tbrm_light_kernels.hip — gfx950 kernels of the illumination pass (Sunden/Ropinski selective light updates).
//
//   k_propagate_chunk : the production kernel. One launch advances EVERY tile of the slice plane through a chunk
//                       of up to 16 consecutive slices, so an axis pass over a 512-deep volume is 32 launches
//                       instead of the reference's 512 dispatches (LightingShaders.cpp:132-158).
//   k_propagate_slice : the reference's structure, one slice per launch (AddDirLightShader.usf:68-128,
//                       ChangeDirLightShader.usf:74-156). Fallback for passes the chunk kernel declines
//                       (degenerate offsets) and the A/B baseline (TBRM_FORCE_SLICE_KERNEL=1).
//
// Why chunks work: slice k only needs the previous slice's propagated light inside a small bilinear footprint,
// offset by the constant PrevPixelOffset. A workgroup that owns a 32x32 tile at the END of a chunk therefore only
// needs a (32 + steps*g)^2 window of the plane at the START of the chunk (g = width of the footprint in texels,
// normally 1) and recomputes that shrinking window privately in LDS — no inter-workgroup traffic inside a chunk,
// one kernel boundary between chunks. The window moves with the light (integer shear cx,cy per slice) so strongly
// slanted second-axis passes keep the same small halo. Arithmetic per voxel is exactly the reference's, including
// the per-slice UNORM8 re-quantisation of the propagated light (RaymarchVolume.cpp:857-866).
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
// a chunk of slices per launch

// Per-coordinate lookup tables a workgroup builds once per chunk (they only depend on the pixel coordinate, not on
// the slice): the bilinear split of the previous-slice fetch and the trilinear split of the data-volume fetch. They
// carry the reference's per-thread divisions ((px+0.5)/size + offset, AddDirLightShader.usf:81,:85) out of the loop.
struct AxisTable {
    float* prev_f;  // frac of the previous-slice tap
    int* prev_d;    // tap index - pixel index
    float* uvw;     // SampleUVW component
    float* tex_f;   // frac of the data-volume tap
    int* tex_i;     // base data-volume tap index
    int* guard;     // uvw == saturate(uvw)
};

__device__ __forceinline__ char* carve(char*& cursor, size_t bytes)
{
    char* r = cursor;
    cursor += (bytes + 15) & ~(size_t) 15;
    return r;
}

__device__ __forceinline__ AxisTable carve_table(char*& cursor, int len)
{
    AxisTable t;
    t.prev_f = (float*) carve(cursor, len * 4);
    t.prev_d = (int*) carve(cursor, len * 4);
    t.uvw = (float*) carve(cursor, len * 4);
    t.tex_f = (float*) carve(cursor, len * 4);
    t.tex_i = (int*) carve(cursor, len * 4);
    t.guard = (int*) carve(cursor, len * 4);
    return t;
}

// entries for in-plane coordinates c0 .. c0+len-1 (out-of-plane entries are never read)
__device__ __forceinline__ void fill_table(const AxisTable& t, int len, int c0, int plane_size, float prev_off, int lv_dim,
                                           float uvw_off, int data_dim)
{
    for (int k = threadIdx.x; k < len; k += blockDim.x) {
        const int c = c0 + k;
        int pd = 0, ti = 0, g = 0;
        float pf = 0.0f, u = 0.0f, tf = 0.0f;
        if (c >= 0 && c < plane_size) {
            const float pu = (((float) (uint32_t) c + 0.5f) / (float) plane_size) + prev_off;
            int i0;
            texel_split(pu, (float) plane_size, i0, pf);
            pd = i0 - c;
            u = (((float) (uint32_t) c + 0.5f) / (float) (uint32_t) lv_dim) + uvw_off;
            texel_split(u, (float) data_dim, ti, tf);
            g = (u == saturate_(u)) ? 1 : 0;
        }
        t.prev_f[k] = pf; t.prev_d[k] = pd; t.uvw[k] = u; t.tex_f[k] = tf; t.tex_i[k] = ti; t.guard[k] = g;
    }
}

struct ChunkGeom {
    int n;              // steps in this chunk
    int lox, hix, loy, hiy; // tap offsets relative to the ownership frame: [lo, hi]
    int HX, HY;         // hull (LDS window) size, multiples of 8
    int padx, pady;     // LDS index of ownership-frame coordinate 0
    int tabx0, taby0;   // first in-plane coordinate of the tables
    int tablx, tably;   // table lengths
};

__host__ __device__ inline int round_up8(int v) { return (v + 7) & ~7; }

__host__ __device__ inline ChunkGeom chunk_geometry(const ChunkParams& p, int tile_x, int tile_y)
{
    ChunkGeom g;
    g.n = p.n_steps;
    g.lox = p.dx_lo - p.cx; g.hix = p.dx_hi - p.cx;
    g.loy = p.dy_lo - p.cy; g.hiy = p.dy_hi - p.cy;
    g.HX = round_up8(kChunkTile + g.n * (g.hix - g.lox));
    g.HY = round_up8(kChunkTile + g.n * (g.hiy - g.loy));
    g.padx = -g.n * g.lox;
    g.pady = -g.n * g.loy;
    const int nlo_x = g.n * p.dx_lo, nhi_x = g.n * p.dx_hi, nlo_y = g.n * p.dy_lo, nhi_y = g.n * p.dy_hi;
    g.tabx0 = tile_x * kChunkTile + (nlo_x < 0 ? nlo_x : 0);
    g.taby0 = tile_y * kChunkTile + (nlo_y < 0 ? nlo_y : 0);
    g.tablx = kChunkTile + (nhi_x > 0 ? nhi_x : 0) - (nlo_x < 0 ? nlo_x : 0);
    g.tably = kChunkTile + (nhi_y > 0 ? nhi_y : 0) - (nlo_y < 0 ? nlo_y : 0);
    return g;
}

size_t chunk_lds_bytes(const ChunkParams& p, bool change)
{
    const ChunkGeom g = chunk_geometry(p, 0, 0);
    const int ns = change ? 2 : 1;
    auto al = [](size_t b) { return (b + 15) & ~(size_t) 15; };
    size_t total = al(256 * 4);                                   // TF alpha
    total += (size_t) ns * 2 * al((size_t) g.HX * g.HY * 4);       // double-buffered plane windows
    total += (size_t) ns * 6 * (al((size_t) g.tablx * 4) + al((size_t) g.tably * 4)); // axis tables
    total += (size_t) ns * 6 * al((size_t) g.n * 4);               // slice-axis tables
    return total;
}

template <int DFMT, int LFMT, bool CHANGE>
__global__ __launch_bounds__(kChunkThreads) void k_propagate_chunk(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = kChunkTile;
    constexpr int NS = CHANGE ? 2 : 1;
    const int tile_x = p.tile_i0 + (int) blockIdx.x, tile_y = p.tile_j0 + (int) blockIdx.y;
    const ChunkGeom g = chunk_geometry(p, tile_x, tile_y);
    const ChunkStream* streams[2] = {&p.a, &p.r};

    // ---- carve LDS --------------------------------------------------------------------------------------
    char* cursor = smem;
    float* s_alpha = (float*) carve(cursor, 256 * 4);
    float* win[2][2];
    AxisTable tx[2], ty[2], ts[2];
    for (int si = 0; si < NS; ++si) {
        win[si][0] = (float*) carve(cursor, (size_t) g.HX * g.HY * 4);
        win[si][1] = (float*) carve(cursor, (size_t) g.HX * g.HY * 4);
        tx[si] = carve_table(cursor, g.tablx);
        ty[si] = carve_table(cursor, g.tably);
        ts[si] = carve_table(cursor, g.n);
    }

    // plane axes -> volume axes (GetPermutationMatrix, LightingShaderUtils.cpp:227-249)
    const int dim_u = p.axis == 0 ? 1 : 0, dim_v = p.axis == 2 ? 1 : 2, dim_s = p.axis;
    const int data_dims[3] = {p.data.nx, p.data.ny, p.data.nz};

    for (int k = threadIdx.x; k < 256; k += blockDim.x) s_alpha[k] = p.tf[k].w;
    for (int si = 0; si < NS; ++si) {
        const ChunkStream& s = *streams[si];
        fill_table(tx[si], g.tablx, g.tabx0, p.W, s.off_u, p.lv_dims[dim_u], s.uvw_off[dim_u], data_dims[dim_u]);
        fill_table(ty[si], g.tably, g.taby0, p.H, s.off_v, p.lv_dims[dim_v], s.uvw_off[dim_v], data_dims[dim_v]);
        // slice axis: entry k = step k of this chunk (only the uvw / texel-split / guard fields are used)
        for (int k = threadIdx.x; k < g.n; k += blockDim.x) {
            const int j = p.j0 + k * p.dir;
            const float w = (((float) (uint32_t) j + 0.5f) / (float) (uint32_t) p.lv_dims[dim_s]) + s.uvw_off[dim_s];
            int ti;
            float tf;
            texel_split(w, (float) data_dims[dim_s], ti, tf);
            ts[si].uvw[k] = w; ts[si].tex_i[k] = ti; ts[si].tex_f[k] = tf; ts[si].guard[k] = (w == saturate_(w)) ? 1 : 0;
        }
    }

    // ---- slot mapping: 8x8 patches of the hull per wave ---------------------------------------------------
    const int n_slots = g.HX * g.HY;
    const int patches_x = g.HX >> 3;
    const int base_x = tile_x * T, base_y = tile_y * T;

    // ---- input window: the plane after the previous chunk (ownership frame of r = n) ------------------------
    for (int e = threadIdx.x; e < n_slots; e += blockDim.x) {
        const int patch = e >> 6, lane = e & 63;
        const int lx = (patch % patches_x) * 8 + (lane & 7), ly = (patch / patches_x) * 8 + (lane >> 3);
        const int px = base_x + g.n * p.cx + (lx - g.padx), py = base_y + g.n * p.cy + (ly - g.pady);
        const bool inplane = (unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H;
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = *streams[si];
            float v = s.border_light;
            if (inplane) v = p.first_chunk ? s.init_value : s.plane_in[(size_t) py * p.W + px];
            win[si][0][ly * g.HX + lx] = v;
        }
    }
    __syncthreads();

    int cur = 0; // window holding the state BEFORE the step
    for (int step = 0; step < g.n; ++step) {
        const int r = g.n - 1 - step; // steps that remain after this one
        const int j = p.j0 + step * p.dir;
        const int x_lo = r * g.lox, x_hi = T + r * g.hix, y_lo = r * g.loy, y_hi = T + r * g.hiy;
        for (int e = threadIdx.x; e < n_slots; e += blockDim.x) {
            const int patch = e >> 6, lane = e & 63;
            const int lx = (patch % patches_x) * 8 + (lane & 7), ly = (patch / patches_x) * 8 + (lane >> 3);
            const int qx = lx - g.padx, qy = ly - g.pady;
            if (qx < x_lo || qx >= x_hi || qy < y_lo || qy >= y_hi) continue;
            const int px = base_x + r * p.cx + qx, py = base_y + r * p.cy + qy;
            const bool inplane = (unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H;
            if (!inplane) { // outside the buffer: later fetches must see the sampler's border colour here
                for (int si = 0; si < NS; ++si) win[si][cur ^ 1][ly * g.HX + lx] = streams[si]->border_light;
                continue;
            }
            int pos[3];
            if (p.axis == 0) { pos[0] = j; pos[1] = px; pos[2] = py; }
            else if (p.axis == 1) { pos[0] = px; pos[1] = j; pos[2] = py; }
            else { pos[0] = px; pos[1] = py; pos[2] = j; }
            float lval[2] = {0.0f, 0.0f};
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const ChunkStream& s = *streams[si];
                const int kx = px - g.tabx0, ky = py - g.taby0;
                // previous slice, bilinear with border (AddDirLightShader.usf:81-82)
                const float* pw = win[si][cur] + (ly + ty[si].prev_d[ky] - p.cy) * g.HX + (lx + tx[si].prev_d[kx] - p.cx);
                const float fx = tx[si].prev_f[kx], fy = ty[si].prev_f[ky];
                const float prev = lerp_(lerp_(pw[0], pw[1], fx), lerp_(pw[g.HX], pw[g.HX + 1], fx), fy);
                // occlusion sample (AddDirLightShader.usf:85-114)
                float uvw[3], tf3[3];
                int ti3[3];
                uvw[dim_u] = tx[si].uvw[kx]; uvw[dim_v] = ty[si].uvw[ky]; uvw[dim_s] = ts[si].uvw[step];
                ti3[dim_u] = tx[si].tex_i[kx]; ti3[dim_v] = ty[si].tex_i[ky]; ti3[dim_s] = ts[si].tex_i[step];
                tf3[dim_u] = tx[si].tex_f[kx]; tf3[dim_v] = ty[si].tex_f[ky]; tf3[dim_s] = ts[si].tex_f[step];
                const float aw = p.clip_mode ? clip_alpha_weight(uvw[0], uvw[1], uvw[2], p.cc, p.cd, p.lv_dims) : 1.0f;
                bool inside = true;
                if constexpr (!CHANGE) inside = tx[si].guard[kx] && ty[si].guard[ky] && ts[si].guard[step];
                float occ = 0.0f;
                if (aw > 0.0f && inside) {
                    const float val = sample_trilinear_border<DFMT>(p.data, ti3[0], ti3[1], ti3[2], tf3[0], tf3[1], tf3[2], p.data_border);
                    occ = windowed_alpha(val, s.step100, s_alpha, p.win) * aw;
                }
                const float l = prev * (1 - occ);
                lval[si] = l;
                win[si][cur ^ 1][ly * g.HX + lx] = through_format<LFMT>(l); // WriteBuffer[PixelLoc] = L
            }
            const bool owner = qx >= 0 && qx < T && qy >= 0 && qy < T;
            if (owner) {
                const size_t li = brick_off(pos[0], pos[1], pos[2], p.lv_bnx, p.lv_bnxy);
                if constexpr (!CHANGE) {
                    if (fabsf(lval[0]) > 1e-3f) store_voxel<LFMT>(p.light, li, load_voxel<LFMT>(p.light, li) + (lval[0] * p.b_added));
                } else {
                    const float la = lval[0], lr = lval[1];
                    if (fabsf(la - lr) > 1e-3f) store_voxel<LFMT>(p.light, li, load_voxel<LFMT>(p.light, li) + la - lr);
                }
                if (r == 0)
                    for (int si = 0; si < NS; ++si) streams[si]->plane_out[(size_t) py * p.W + px] = through_format<LFMT>(lval[si]);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

template <int DFMT, int LFMT>
static hipError_t launch_chunk2(const ChunkParams& p, bool change, int tiles_x, int tiles_y, hipStream_t s)
{
    const size_t lds = chunk_lds_bytes(p, change);
    const dim3 grid(tiles_x, tiles_y), block(kChunkThreads);
    if (change) {
        static bool attr_c = false;
        if (!attr_c) { (void) hipFuncSetAttribute((const void*) k_propagate_chunk<DFMT, LFMT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_c = true; }
        hipLaunchKernelGGL((k_propagate_chunk<DFMT, LFMT, true>), grid, block, lds, s, p);
    } else {
        static bool attr_a = false;
        if (!attr_a) { (void) hipFuncSetAttribute((const void*) k_propagate_chunk<DFMT, LFMT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_a = true; }
        hipLaunchKernelGGL((k_propagate_chunk<DFMT, LFMT, false>), grid, block, lds, s, p);
    }
    return hipGetLastError();
}
template <int DFMT>
static hipError_t launch_chunk1(const ChunkParams& p, bool change, int lv_fmt, int tx, int ty, hipStream_t s)
{
    return lv_fmt == FMT_U8 ? launch_chunk2<DFMT, FMT_U8>(p, change, tx, ty, s) : launch_chunk2<DFMT, FMT_F32>(p, change, tx, ty, s);
}
hipError_t launch_propagate_chunk(const ChunkParams& p, bool change, int lv_fmt, int tiles_x, int tiles_y, hipStream_t s)
{
    if (tiles_x <= 0 || tiles_y <= 0 || p.n_steps <= 0) return hipSuccess;
    switch (p.data.fmt) {
        case FMT_U8: return launch_chunk1<FMT_U8>(p, change, lv_fmt, tiles_x, tiles_y, s);
        case FMT_U16: return launch_chunk1<FMT_U16>(p, change, lv_fmt, tiles_x, tiles_y, s);
        default: return launch_chunk1<FMT_F32>(p, change, lv_fmt, tiles_x, tiles_y, s);
    }
}

} // namespace tbrm
