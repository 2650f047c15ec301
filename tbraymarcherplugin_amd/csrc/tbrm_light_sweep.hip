// tbrm_light_sweep.hip — k_light_sweep, the serial half of an axis pass of the illumination operators
// (AddDirLightShader.usf:68-128, ChangeDirLightShader.usf:74-156) as ONE launch per span of slices: every 32x32 tile of
// the slice plane is a workgroup that walks all slices of the span, and the tiles form a pipeline.
//
// Why a pipeline works. Slice k reads slice k-1 through a bilinear fetch at the constant offset PrevPixelOffset
// (AddDirLightShader.usf:81-82): per plane axis every tap of every pixel lies on the SAME side of its pixel — towards the
// light. A tile therefore needs, of slice k-1, its own pixels plus hx columns / hy rows of at most three neighbours, all on
// the light's side, and nothing of the tiles behind it: tile t can run slice k as soon as its upstream neighbours have
// finished slice k-1. The chunked chain (tbrm_light_chain.hip) breaks that dependency by recomputing a halo that grows by
// the tap range per slice and paying a kernel boundary every 16 / 8 / 4 / 2 slices; here nothing is recomputed and there is
// no boundary: a tile publishes the few columns / rows its downstream neighbours need after every slice and runs a few
// slices behind its upstream neighbours, whose values it has requested `prefetch` slices ahead.
//
// Hand-off. A plane value is a UNORM8 code (the read / write buffers are re-quantised every slice), so a record word is ONE
// dword, {24-bit tag of this launch, code}, written with one relaxed agent-scope store and polled with relaxed agent-scope
// loads (global_store / global_load ... sc1, MI355X_MICROARCH.md "handoff-1to1"): data and flag travel together, so there is
// no fence, no write-back of the L2, no second round trip and nothing that could tear. Records are addressed by
// [slice][tile][word] and never reused within a launch, so a producer never waits for a consumer; the tag makes clearing
// them between launches unnecessary.
//
// Forward progress. Tiles are dealt by an atomic ticket in upstream-first order: a tile only ever waits for tiles with a
// smaller ticket, which have started — whatever the number of workgroups the device keeps resident. A poll gives up after
// ~2^20 tries and raises SweepParams::error instead of hanging the device.
//
// Inside a tile. A lane owns one column of R consecutive rows; per slice and pixel it reads the four taps from an LDS plane
// (tile + halo + one guard ring for taps of weight 0), multiplies by the occlusion factor 1 - CurrentSample that
// k_light_occlusion left in the span's plane stack (requested 6 slices ahead, straight into registers), re-quantises
// (RaymarchVolume.cpp:857-866), updates its voxel of the light-volume bricks staged in LDS (a brick layer per 8 slices,
// double-buffered) and stores the new plane value; ONE LDS-only barrier per slice. No thread owns halo pixels, so a slice
// costs what its 1024 pixels cost. Arithmetic per voxel is the chain's and the reference's, bit for bit.
#include "tbrm_device_sampling.h"
#include "tbrm_light_chain.h"

#include <type_traits>
#include <utility>

namespace tbrm {

constexpr int kSweepTile = 32;
constexpr int kSweepRS = 48, kSweepRows = 48;          // LDS plane: row stride / rows (tile + halo <= 14 + guard ring)
constexpr int kSweepPlane = kSweepRS * kSweepRows;
constexpr int kSweepLvBrick = 528;                     // bytes per staged light-volume brick: 512 + 16, so that the four bricks
                                                       // under a tile row start 4 banks apart
constexpr int kSweepRing = 8;                          // register ring of requested factors / hand-off words (slices)
constexpr int kSweepFactorAhead = 6;                   // slices ahead that the occlusion factors are requested

size_t sweep_lds_bytes(int mode)
{
    const int ns = mode == PASS_ADD ? 1 : 2;
    return (size_t) 2 * ns * kSweepPlane * 4 + 2 * 16 * kSweepLvBrick + 2 * ns * 4 * 64; // planes, two brick layers, block flags
}

template <class F, int... S>
__device__ __forceinline__ void sweep_each_const(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

__device__ __forceinline__ uint32_t sweep_load_word(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sweep_store_word(uint32_t* p, uint32_t w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The slow path of a hand-off: the neighbour has not published the word yet. Its load is inline assembly so that the
// compiler's wait-count bookkeeping of the caller never sees a loop with a memory operation in it (it would answer with
// s_waitcnt vmcnt(0) at every later use of a request that is still in flight).
__device__ __noinline__ uint32_t sweep_poll(const uint32_t* src, uint32_t epoch, int* error)
{
    uint32_t w = 0;
    for (int tries = 0; tries < (1 << 20); ++tries) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
        if ((w >> 8) == epoch) return w;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicOr(error, 1);
    return w;
}

// MODE: PASS_ADD (stream a) or PASS_CHANGE (a added, r removed). R: rows per lane; the workgroup has 16 / R waves.
// PF: slices ahead of their use that the neighbours' hand-off words are requested (a tile settles PF slices + one memory
// round trip behind its upstream neighbours). ALIGNED: the span starts on a brick layer of the light volume and is whole
// layers long — every slice of the unrolled 8-slice body runs, so the body has no skip branches (whose joins would cost the
// requests in flight a s_waitcnt vmcnt(0) per slice).
template <int MODE, int AXIS, int R, int PF, bool ALIGNED>
__global__ __launch_bounds__(1024 / R) void k_light_sweep(const ChunkParams p, const SweepParams q)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_ticket;
    constexpr int T = kSweepTile, RS = kSweepRS, PLANE = kSweepPlane, LVB = kSweepLvBrick;
    constexpr int NW = 16 / R, NT = NW * 64;
    constexpr int NS = MODE == PASS_ADD ? 1 : 2;
    constexpr int RING = kSweepRing, FA = kSweepFactorAhead;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);

    // ---- which tile: tickets in upstream-first order -------------------------------------------------------------------
    if (threadIdx.x == 0) s_ticket = atomicAdd(q.ticket, 1);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int n_tiles = p.tiles_x * p.tiles_y;
    const int ui = ticket % p.tiles_x, uj = ticket / p.tiles_x;
    const int tile_x = q.sx > 0 ? p.tiles_x - 1 - ui : ui, tile_y = q.sy > 0 ? p.tiles_y - 1 - uj : uj;
    const int tile_lin = tile_y * p.tiles_x + tile_x;
    const int base_x = tile_x * T, base_y = tile_y * T;
    const int plane_elems = p.W * p.H;
    const int n = p.n_steps;
    const int hx = q.hx, hy = q.hy;
    const int ox = 1 + (q.sx < 0 ? hx : 0), oy = 1 + (q.sy < 0 ? hy : 0); // LDS plane coordinates of tile pixel (0, 0)

    // LDS map: plane(buf, si) = the propagated light of stream si before (buf = parity of the slice's iteration) / after a
    // slice; two light-volume brick layers; the tile's empty-block flags
    float* const lds = (float*) smem;
    auto plane = [&](int buf, int si) -> float* { return lds + (buf * NS + si) * PLANE; };
    uint8_t* const lvt = (uint8_t*) (lds + 2 * NS * PLANE);
    uint8_t* const sflag = lvt + 2 * 16 * LVB; // [si][2 x 2 blocks][64 slice groups]
    auto stream = [&](int si) -> const ChunkStream& { return si == 0 ? p.a : p.r; };

    // Iterations: `it` counts slices from the start of the first brick layer of the light volume the span touches, so that
    // it & 7 is a slice's row inside its layer and it >> 3 its layer; the span's slices are it = i0 .. i0 + n - 1.
    const bool down = p.dir < 0;
    const int i0 = down ? 7 - (p.j0 & 7) : (p.j0 & 7);
    const int IT = i0 + n, G = (IT + 7) >> 3;
    const int layer0 = p.j0 >> 3;

    // ---- the planes before the span's first slice ------------------------------------------------------------------------
    for (int i = threadIdx.x; i < PLANE; i += NT) {
        const int cy = i / RS, cx = i - cy * RS;
        const int gx = base_x + cx - ox, gy = base_y + cy - oy;
        const bool in = (unsigned) gx < (unsigned) p.W && (unsigned) gy < (unsigned) p.H;
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = stream(si);
            // outside the buffer: the read sampler's border colour, for the whole pass (AddDirLightShader.usf:22-25)
            float v = s.border_light;
            if (in) v = p.first_chunk ? s.init_value : s.plane_in[gy * p.W + gx];
            plane(i0 & 1, si)[i] = v;
            plane((i0 & 1) ^ 1, si)[i] = s.border_light;
        }
    }
    // empty-block flags of the tile's 2 x 2 occlusion blocks, every slice group of the span
    {
        const int groups = (p.occ_phase + n + 7) >> 3;
        for (int i = threadIdx.x; i < NS * 4 * 64; i += NT) {
            const int si = i >> 8, blk = (i >> 6) & 3, zg = i & 63;
            const uint8_t* flags = stream(si).occ_flags;
            const int bx = (base_x >> 4) + (blk & 1), by = (base_y >> 4) + (blk >> 1);
            uint8_t f = 0;
            if (flags && zg < groups && bx < p.occ_blocks_x && by < p.occ_blocks_y) f = flags[((size_t) zg * p.occ_blocks_y + by) * p.occ_blocks_x + bx];
            sflag[i] = f;
        }
    }

    // ---- light-volume bricks: a layer = the 4 x 4 bricks under the tile, 512 pieces of 16 bytes ---------------------------
    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS; // plane axes -> volume axes
    const int lbn[3] = {p.lv_bnx, p.lv_bnxy / p.lv_bnx, (p.lv_dims[2] + 7) >> 3};
    constexpr int PPT = (512 + NT - 1) / NT; // pieces per thread
    auto piece_global = [&](int piece, int layer, bool& exists) -> uint32_t {
        const int lb = piece >> 5;
        const int bu = (base_x >> 3) + (lb & 3), bv = (base_y >> 3) + (lb >> 2);
        exists = piece < 512 && bu < lbn[dim_u] && bv < lbn[dim_v] && (unsigned) layer < (unsigned) lbn[dim_s];
        int b3[3];
        b3[dim_u] = bu; b3[dim_v] = bv; b3[dim_s] = layer;
        return (uint32_t) ((b3[2] * lbn[1] + b3[1]) * lbn[0] + b3[0]) * 512u + (uint32_t) (piece & 31) * 16u;
    };
    auto piece_lds = [&](int piece, int buf) -> uint4* { return (uint4*) (lvt + (buf * 16 + (piece >> 5)) * LVB + (piece & 31) * 16); };
    auto layer_of = [&](int g) -> int { return down ? layer0 - g : layer0 + g; };
    uint4 lv_next[PPT];
    auto load_layer = [&](int g) { // into registers
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            bool exists;
            const uint32_t go = piece_global((int) threadIdx.x + k * NT, layer_of(g), exists);
            // (bricks that do not exist read the volume's first bytes instead and are never written back: no branch around the load)
            lv_next[k] = *(const uint4*) ((const uint8_t*) p.light + (exists && g < G ? go : 0u));
        }
    };
    auto install_layer = [&](int g) { // registers -> LDS
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int piece = (int) threadIdx.x + k * NT;
            if (piece < 512) *piece_lds(piece, g & 1) = lv_next[k];
        }
    };
    auto write_back_layer = [&](int g) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int piece = (int) threadIdx.x + k * NT;
            bool exists;
            const uint32_t go = piece_global(piece, layer_of(g), exists);
            if (exists) *(uint4*) ((uint8_t*) p.light + go) = *piece_lds(piece, g & 1);
        }
    };
    load_layer(0);
    install_layer(0);
    load_layer(1);

    // ---- this lane's pixels: column c, rows r0 .. r0 + R - 1 of the tile -----------------------------------------------------
    const int c = lane & 31, r0 = (wave * 2 + (lane >> 5)) * R;
    const int px = base_x + c;
    const bool in_x = px < p.W;
    bool in_pl[R];            // inside the buffer (D3D drops the overhanging threads' writes)
    int own[R];               // LDS index of the pixel inside a plane
    int tap[NS][R];           // LDS index of its first previous-slice tap
    float wfx[NS], wfy[NS][R];
    uint32_t lv_at[R];        // byte of its voxel in a staged layer, slice row 0
    uint32_t own_idx[R];      // pixel inside a buffer plane
    bool bad = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = r0 + k, py = base_y + r;
        in_pl[k] = in_x && py < p.H;
        own[k] = (oy + r) * RS + ox + c;
        own_idx[k] = in_pl[k] ? (uint32_t) (py * p.W + px) : 0u;
        const uint32_t lb = (uint32_t) ((r >> 3) * 4 + (c >> 3)) * (uint32_t) LVB;
        if (AXIS == 0) lv_at[k] = lb + (uint32_t) (r & 7) * 64u + (uint32_t) (c & 7) * 8u;
        else if (AXIS == 1) lv_at[k] = lb + (uint32_t) (r & 7) * 64u + (uint32_t) (c & 7);
        else lv_at[k] = lb + (uint32_t) (r & 7) * 8u + (uint32_t) (c & 7);
    }
    constexpr uint32_t kLvStep = AXIS == 0 ? 1u : (AXIS == 1 ? 8u : 64u);
#pragma unroll
    for (int si = 0; si < NS; ++si) {
        const ChunkStream& s = stream(si);
        // previous-slice tap split: ((c + 0.5)/size + PrevPixelOffset) -> (tap - c, frac) (AddDirLightShader.usf:81-82)
        int ix = 0;
        float fx = 0.0f;
        if (in_x) {
            const float pu = ((float) (uint32_t) px + 0.5f) / (float) p.W;
            texel_split(pu + s.off_u, (float) p.W, ix, fx);
            ix -= px;
            // the host promised: taps of non-zero weight within hx columns on side sx (SweepParams)
            const int lo = ix, hi = fx != 0.0f ? ix + 1 : ix;
            bad = bad || (q.sx >= 0 ? (lo < 0 || hi > hx) : (lo < -hx || hi > 0)) || ix < -hx - 1 || ix > hx;
        }
        wfx[si] = fx;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int py = base_y + r0 + k;
            int iy = 0;
            float fy = 0.0f;
            if (in_pl[k]) {
                const float pv = ((float) (uint32_t) py + 0.5f) / (float) p.H;
                texel_split(pv + s.off_v, (float) p.H, iy, fy);
                iy -= py;
                const int lo = iy, hi = fy != 0.0f ? iy + 1 : iy;
                bad = bad || (q.sy >= 0 ? (lo < 0 || hi > hy) : (lo < -hy || hi > 0)) || iy < -hy - 1 || iy > hy;
            }
            wfy[si][k] = fy;
            tap[si][k] = own[k] + (in_pl[k] ? iy * RS + ix : 0);
        }
    }
    if (bad) { atomicOr(q.error, 2); }
    // (a lane whose taps are not where the host said reads its own cell instead: wrong values, flagged, but in bounds)
    if (bad) {
#pragma unroll
        for (int si = 0; si < NS; ++si)
#pragma unroll
            for (int k = 0; k < R; ++k) tap[si][k] = own[k];
    }

    // ---- hand-off geometry ---------------------------------------------------------------------------------------------------
    // A tile publishes E_x = its hx columns and E_y = its hy rows on the side AWAY from the light, words [row][column of E_x]
    // then [row of E_y][column]; its halo is the same cells of the three upstream neighbours.
    const int RW = T * (hx + hy);
    const int e0x = q.sx > 0 ? 0 : T - hx, e0y = q.sy > 0 ? 0 : T - hy;
    const bool pub_x = hx > 0 && c >= e0x && c < e0x + hx;
    uint32_t pubx_w[R], puby_w[R];
    bool pub_y[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = r0 + k;
        pubx_w[k] = (uint32_t) (r * hx + (c - e0x));
        pub_y[k] = hy > 0 && r >= e0y && r < e0y + hy;
        puby_w[k] = (uint32_t) (T * hx + (r - e0y) * T + c);
    }
    const bool any_pub_y = __builtin_amdgcn_ballot_w64(pub_y[0] || pub_y[R - 1]) != 0; // (a lane's rows are consecutive)
    // halo word of this thread: h < 32 hx: column strip, < 32 hx + 32 hy: row strip, then the corner
    bool hal_on = false;
    int hal_dst = 0;
    uint32_t hal_src = 0; // (neighbour tile * RW + word)
    {
        const int h = (int) threadIdx.x;
        const int nx = T * hx, ny = T * hy, nc = hx * hy;
        int ntx = tile_x, nty = tile_y, word = 0, cxh = 0, cyh = 0; // neighbour tile, its word, the halo cell in tile coordinates
        if (h < nx) { const int row = h / max(hx, 1), kx = h - row * hx; ntx += q.sx; word = row * hx + kx; cxh = (q.sx > 0 ? T : -hx) + kx; cyh = row; hal_on = true; }
        else if (h < nx + ny) { const int m = h - nx, ky = m / T, col = m - ky * T; nty += q.sy; word = nx + ky * T + col; cxh = col; cyh = (q.sy > 0 ? T : -hy) + ky; hal_on = true; }
        else if (h < nx + ny + nc) {
            const int m = h - nx - ny, ky = m / max(hx, 1), kx = m - ky * hx;
            ntx += q.sx; nty += q.sy;
            word = (e0y + ky) * hx + kx;
            cxh = (q.sx > 0 ? T : -hx) + kx; cyh = (q.sy > 0 ? T : -hy) + ky;
            hal_on = true;
        }
        // (a neighbour's pixels beyond the buffer are not handed over: their cells hold the border colour from the start)
        hal_on = hal_on && (unsigned) ntx < (unsigned) p.tiles_x && (unsigned) nty < (unsigned) p.tiles_y &&
                 (unsigned) (base_x + cxh) < (unsigned) p.W && (unsigned) (base_y + cyh) < (unsigned) p.H;
        hal_dst = (oy + cyh) * RS + ox + cxh;
        hal_src = hal_on ? (uint32_t) ((nty * p.tiles_x + ntx) * RW + word) : 0u;
    }
    const uint32_t rec_slice = (uint32_t) (n_tiles * RW); // words per slice
    const uint32_t epoch = q.epoch & 0xffffffu, tag = epoch << 8;
    static_assert(PF >= 1 && PF < kSweepRing, "the request ring holds 8 slices");

    // ---- occlusion factors: requested kSweepFactorAhead slices ahead, one register per pixel and slice ---------------------
    // A flagged-empty block (k_occ_flags) was never computed: its factor 1 - 0 comes from the page of ones at the head of the
    // stream's allocation (ChunkStream::occ_base), as in the chain.
    const int blk = ((r0 >> 4) << 1) | (c >> 4);
    float freg[RING][NS][R];
    bool f_one[NS]; // the block of this lane's pixels is flagged empty in the slice group being requested
#pragma unroll
    for (int si = 0; si < NS; ++si) f_one[si] = false;
    auto refresh_flags = [&](int s) { // for the slice group of slice s
        const int zg = (p.occ_phase + s) >> 3;
#pragma unroll
        for (int si = 0; si < NS; ++si) f_one[si] = sflag[(si * 4 + blk) * 64 + (zg & 63)] != 0;
    };
    auto request_factors = [&](int s, auto slot_c) { // slice s of the span into ring slot SLOT
        constexpr int SLOT = decltype(slot_c)::value;
        // ALIGNED spans request whole slice groups from slot (8 - FA) & 7 on: the flags are refreshed there (the main loop)
        if constexpr (!ALIGNED) refresh_flags(s);
        const uint32_t plane_off = (uint32_t) s * (uint32_t) plane_elems;
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& st = stream(si);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const uint32_t off = (f_one[si] || !in_pl[k]) ? (uint32_t) lane : st.occ_off + plane_off + own_idx[k];
                freg[SLOT][si][k] = st.occ_base[off];
            }
        }
    };
    uint32_t hreg[RING][NS];
    // (every lane loads: the ones without a halo word read word 0 of the slice, one request per wave — a branch around a
    // load whose result is consumed slices later would make the compiler drain every request in flight at the join)
    auto request_halo = [&](int s, auto slot_c) { // the neighbours' slice s
        constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int si = 0; si < NS; ++si) hreg[SLOT][si] = sweep_load_word((const uint32_t*) q.rec[si] + ((uint32_t) s * rec_slice + hal_src));
    };

    __syncthreads(); // planes, flags and the first brick layer are in LDS
    // prologue: what the first slices will find in their ring slots
    sweep_each_const([&](auto sl) {
        constexpr int SL = decltype(sl)::value;
        const int it = i0 + ((SL - i0) & 7), s = it - i0; // the first iteration that uses slot SL
        if (s < FA) {
            if constexpr (ALIGNED) refresh_flags(0);
            request_factors(min(s, n - 1), sl);
        }
        if (s < PF) request_halo(max(min(s, n - 2), 0), sl);
        (void) it;
    }, std::make_integer_sequence<int, RING>{});

    const float thresh = 1e-3f;
    for (int g = 0; g < G; ++g) {
        if (g > 0) write_back_layer(g - 1);
        sweep_each_const([&](auto kc) {
            constexpr int K8 = decltype(kc)::value, CUR = K8 & 1;
            const int s = g * 8 + K8 - i0;
            if (K8 == 3 && g + 1 < G) { // the next layer: loaded eight slices ago, first used four barriers from now
                install_layer(g + 1);
                load_layer(g + 2);
            }
            if constexpr (!ALIGNED)
                if (s < 0 || s >= n) return;
            // requests for the slices ahead (past the span's end the last slice is requested again: no branch around the loads)
            if constexpr (ALIGNED && ((K8 + FA) & 7) == 0) refresh_flags(min(s + FA, n - 1)); // a new slice group starts
            request_factors(min(s + FA, n - 1), std::integral_constant<int, (K8 + FA) & 7>{});
            // this slice
            const uint32_t jrow = (uint32_t) (down ? 7 - K8 : K8) * kLvStep;
            uint8_t* const lv_layer = lvt + (g & 1) * 16 * LVB;
            float t00[NS][R], t01[NS][R], t10[NS][R], t11[NS][R];
            uint32_t code_old[R];
#pragma unroll
            for (int si = 0; si < NS; ++si)
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const float* pw = plane(CUR, si) + tap[si][k];
                    t00[si][k] = pw[0]; t01[si][k] = pw[1]; t10[si][k] = pw[RS]; t11[si][k] = pw[RS + 1];
                }
#pragma unroll
            for (int k = 0; k < R; ++k) code_old[k] = lv_layer[lv_at[k] + jrow];
            float lval[NS][R], qval[NS][R], pval[NS][R];
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const ChunkStream& st = stream(si);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    // previous slice, bilinear with border colour (AddDirLightShader.usf:81-82), times 1 - CurrentSample (:117)
                    const float prev = lerp_(lerp_(t00[si][k], t01[si][k], wfx[si]), lerp_(t10[si][k], t11[si][k], wfx[si]), wfy[si][k]);
                    const float l = prev * freg[K8][si][k];
                    lval[si][k] = l;
                    qval[si][k] = quantize_u8(l);                                           // WriteBuffer[PixelLoc] = L (:120): the code
                    pval[si][k] = in_pl[k] ? decode_u8f(qval[si][k]) : st.border_light;    // ... and what a read of it returns
                    plane(CUR ^ 1, si)[own[k]] = pval[si][k];
                }
            }
            // the owned voxels (:123-126 / ChangeDirLightShader.usf:152-154)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float lv_old = decode_u8(code_old[k]);
                float nv;
                bool write;
                if constexpr (MODE == PASS_ADD) { nv = lv_old + lval[0][k] * p.b_added; write = fabsf(lval[0][k]) > thresh; }
                else { nv = lv_old + lval[0][k] - lval[NS - 1][k]; write = fabsf(lval[0][k] - lval[NS - 1][k]) > thresh; }
                lv_layer[lv_at[k] + jrow] = (uint8_t) ((write && in_pl[k]) ? encode_u8(nv) : code_old[k]);
            }
            bool last = s == n - 1;
            if constexpr (ALIGNED && K8 != 7) last = false;
            if (last) { // the state the next span starts from
#pragma unroll
                for (int si = 0; si < NS; ++si)
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (in_pl[k]) stream(si).plane_out[own_idx[k]] = pval[si][k];
            } else {
                // publish what the downstream neighbours need of this slice, take what the upstream ones published
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    uint32_t* const rec = (uint32_t*) q.rec[si] + ((uint32_t) s * rec_slice + (uint32_t) (tile_lin * RW));
                    if (pub_x) {
#pragma unroll
                        for (int k = 0; k < R; ++k) sweep_store_word(rec + pubx_w[k], tag | (uint32_t) qval[si][k]);
                    }
                    if (any_pub_y) {
#pragma unroll
                        for (int k = 0; k < R; ++k)
                            if (pub_y[k]) sweep_store_word(rec + puby_w[k], tag | (uint32_t) qval[si][k]);
                    }
                    uint32_t w = hreg[K8][si];
                    if (hal_on && (w >> 8) != epoch) w = sweep_poll((const uint32_t*) q.rec[si] + ((uint32_t) s * rec_slice + hal_src), epoch, q.error);
                    if (hal_on) plane(CUR ^ 1, si)[hal_dst] = decode_u8(w & 255u);
                }
            }
            // (near the span's end a slice that nobody will use is requested: no branch around the loads)
            request_halo(s + max(min(PF, n - 2 - s), 0), std::integral_constant<int, (K8 + PF) & 7>{});
            lds_barrier();
        }, std::make_integer_sequence<int, 8>{});
    }
    write_back_layer(G - 1);

    // ---- the last tile to finish re-arms the tickets for the next launch -----------------------------------------------------
    if (threadIdx.x == 0) {
        const int done = atomicAdd(q.ticket + 1, 1);
        if (done == n_tiles - 1) {
            atomicExch(q.ticket + 1, 0);
            atomicExch(q.ticket, 0);
        }
    }
}

template <int MODE, int AXIS, int R, int PF, bool ALIGNED>
static hipError_t launch_sweep6(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_sweep<MODE, AXIS, R, PF, ALIGNED>, attr_done, 96 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_light_sweep<MODE, AXIS, R, PF, ALIGNED>), dim3(p.tiles_x * p.tiles_y), dim3(1024 / R), sweep_lds_bytes(MODE), s, p, q);
    return hipGetLastError();
}
template <int MODE, int AXIS, int R, int PF>
static hipError_t launch_sweep5(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    const bool aligned = (p.n_steps & 7) == 0 && (p.j0 & 7) == (p.dir > 0 ? 0 : 7) && p.occ_phase == 0;
    return aligned ? launch_sweep6<MODE, AXIS, R, PF, true>(p, q, s) : launch_sweep6<MODE, AXIS, R, PF, false>(p, q, s);
}
template <int MODE, int AXIS, int R>
static hipError_t launch_sweep4(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    return q.prefetch <= 2 ? launch_sweep5<MODE, AXIS, R, 2>(p, q, s) : (q.prefetch <= 4 ? launch_sweep5<MODE, AXIS, R, 4>(p, q, s) : launch_sweep5<MODE, AXIS, R, 6>(p, q, s));
}
template <int MODE, int AXIS>
static hipError_t launch_sweep3(const ChunkParams& p, const SweepParams& q, int rows, hipStream_t s)
{
    return rows == 1 ? launch_sweep4<MODE, AXIS, 1>(p, q, s) : launch_sweep4<MODE, AXIS, 2>(p, q, s);
}
template <int MODE>
static hipError_t launch_sweep2(const ChunkParams& p, const SweepParams& q, int rows, hipStream_t s)
{
    return p.axis == 0 ? launch_sweep3<MODE, 0>(p, q, rows, s) : (p.axis == 1 ? launch_sweep3<MODE, 1>(p, q, rows, s) : launch_sweep3<MODE, 2>(p, q, rows, s));
}
// advances every tile through the span (j0, n_steps) in one launch; mode PASS_ADD or PASS_CHANGE, UNORM8 light volume
hipError_t launch_light_sweep(const ChunkParams& p, const SweepParams& q, int mode, int rows, hipStream_t s)
{
    if (p.n_steps <= 0 || p.tiles_x <= 0 || p.tiles_y <= 0) return hipSuccess;
    if (mode == PASS_ADD) return launch_sweep2<PASS_ADD>(p, q, rows, s);
    if (mode == PASS_CHANGE) return launch_sweep2<PASS_CHANGE>(p, q, rows, s);
    return hipErrorInvalidConfiguration;
}

} // namespace tbrm
