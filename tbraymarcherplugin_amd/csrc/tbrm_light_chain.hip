// tbrm_light_chain.hip — k_light_chain, the serial half of a chunked axis pass of the illumination operators
// (AddDirLightShader.usf:68-128, ChangeDirLightShader.usf:74-156): one workgroup advances one tile of the slice plane
// through the 16 / 8 / 4 / 2 slices of a chunk. Compiled once per light-volume format (TBRM_CHAIN_LFMT = 0: UNORM8,
// 2: R32F) so that the two halves build in parallel; tbrm_light_kernels.hip holds the rest of the pass (occlusion,
// flags, the one-slice-per-launch kernel) and the story of why the pass is cut this way.
#include "tbrm_device_sampling.h"
#include "tbrm_light_chain.h"

#include <type_traits>
#include <utility>

#ifndef TBRM_CHAIN_LFMT
#error "compile with -DTBRM_CHAIN_LFMT=0 (UNORM8 light volume) or 2 (R32F)"
#endif

namespace tbrm {

// ---- k_light_chain: one tile through the slices of the chunk ---------------------------------------------------
// Everything a slice needs is in LDS before the slice starts: the propagated-light windows, the occlusion factors of
// the window (staged two slices ahead with asynchronous 16-byte global->LDS copies) and, for UNORM8 light volumes, the
// tile's light-volume bricks (loaded once, read-modify-written in LDS, stored once).
//
// LDS planes are RS x RR floats with RS a compile-time odd multiple of 8: every window/ring/stream offset is an
// immediate of the ds instruction (one address register per slot and stream instead of five), and the eight rows of a
// wave's 8x8 patch fall on disjoint groups of eight banks. Per slice: refill the ring slot read in the previous slice
// with the slice two ahead, issue every LDS read of the slice, compute, write, and meet ONCE at a barrier.

// compile-time loop over 0 .. N-1 (the fast slice loop: slice number, window parity and ring slot are immediates)
template <class F, int... S>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

// KH = halo pixels per thread: ceil((hull area - tile area) / threads).
// M > 0: the chunk has exactly M slices (8 or 16), starts on a brick layer of the light volume (occ_phase 0) and its planes
// are staged in one round (RS <= 56), UNORM8 light volume: the slice loop is fully unrolled with every slice-dependent
// quantity an immediate and no per-lane branches — see "fast slice loop" below. M == 0: any chunk.
template <int LFMT, int MODE, int AXIS, int KH, int RS, int M = 0, int RR = RS>
__global__ __launch_bounds__(kChunkThreads) void k_light_chain(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TX = kChunkTile, TY = kChunkTile;
    constexpr int NT = kChunkThreads;                                    // one thread per tile pixel
    constexpr int BY = TY / 8;                                           // light-volume bricks under the tile along v (4 along u)
    constexpr int NS = MODE == PASS_ADD ? 1 : 2;                         // streams propagated (windows)
    constexpr int NR = NS;                                               // copies a wave issues per staged slice and round: the occlusion factors of each stream
    constexpr int NRP = NR;
    constexpr bool LV_LDS = LFMT == FMT_U8;
    constexpr int KS = 1 + KH; // + the owned pixel
    constexpr int PLANE = chain_plane_elems(RS, RR);
    constexpr int GPR = RS / 4;                                         // 16-byte copy groups per plane row
    constexpr int GROUPS = RR * GPR;
    constexpr int ROUNDS = (GROUPS + NT - 1) / NT;                       // copy groups per thread
    static_assert(RS % 16 == 8 && ROUNDS <= 2, "row stride must be an odd multiple of 8");
    const ChunkGeom g = chunk_geometry(p);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int plane_elems = p.H * p.W;
    // Tile of this workgroup. Workgroups go to the 8 XCDs round-robin by linear id (an affinity used for speed only): XCD x
    // takes the x-th eighth of the row-major tile list — a band of neighbouring tiles whose overlapping halo reads of the
    // occlusion planes then meet in one L2.
    const int n_tiles = p.tiles_x * p.tiles_y, per_xcd = (n_tiles + 7) >> 3;
    const int tile_id = ((int) blockIdx.x & 7) * per_xcd + ((int) blockIdx.x >> 3);
    if (((int) blockIdx.x >> 3) >= per_xcd || tile_id >= n_tiles) return;
    const int tile_y = tile_id / p.tiles_x, tile_x = tile_id - tile_y * p.tiles_x;
    const int base_x = tile_x * TX, base_y = (p.tile_row0 + tile_y) * TY;

    // LDS map (floats): window w of stream si at (w*NS + si)*PLANE, staged plane si of ring slot q at (2*NS + q*NRP + si)*PLANE;
    // then the light-volume tile (bytes)
    float* const lds = (float*) smem;
    uint8_t* const lv_tile = (uint8_t*) (lds + (2 * NS + kOccRing * NRP) * PLANE);
    auto window = [&](int w, int si) -> float* { return lds + (w * NS + si) * PLANE; };
    auto ring = [&](int q, int si) -> float* { return lds + (2 * NS + q * NRP + si) * PLANE; };

    // ---- 16-byte staging pattern: copy group i = floats [4i, 4i+4) of an LDS plane = 4 pixels of one hull row ------
    int st_src[ROUNDS];   // pixel index of the group's first pixel inside a plane (may run off the row ends: guard bands)
    bool st_ok[ROUNDS];
    int st_dst[ROUNDS];   // this wave's 64 x 4 floats
    bool st_one[NRP][ROUNDS][2] = {}; // the group's 4 pixels lie in blocks of slice group 0 / 1 of the chunk that are empty for the stream
    int ndma = 0;         // copies this WAVE issues per staged slice (wave-uniform)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int gi = (int) threadIdx.x + rd * NT;
        const int row = gi / GPR, col = (gi - row * GPR) * 4;
        const int py = base_y - g.pady + row;
        st_src[rd] = py * p.W + base_x - g.padx + col;
        st_ok[rd] = gi < GROUPS && row < g.HY && col < g.HX && (unsigned) py < (unsigned) p.H;
        st_dst[rd] = (wave * 64 + rd * NT) * 4;
        if (__builtin_amdgcn_ballot_w64(st_ok[rd]) != 0) ndma += NRP;
        // empty occlusion blocks (16x16 pixels x 8 slices) are handed over as one flag: their factor 1 - 0 is staged from
        // a page of ones
        if (st_ok[rd]) {
            const int x_first = base_x - g.padx + col, x_last = x_first + 3;
            const int bx0 = max(x_first, 0) >> 4, bx1 = min(x_last, p.W - 1) >> 4, by = py >> 4;
#pragma unroll
            for (int si = 0; si < NRP; ++si) {
                const uint8_t* flags = si == 0 ? p.a.occ_flags : p.r.occ_flags;
                if (!flags) continue;
                if constexpr (NRP == 2)
                    if (si == 1 && flags == p.a.occ_flags) { st_one[NRP - 1][rd][0] = st_one[0][rd][0]; st_one[NRP - 1][rd][1] = st_one[0][rd][1]; continue; } // computed jointly
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    bool one = x_last >= 0 && x_first < p.W && z * kOccSlices < p.occ_phase + g.n;
                    if (one) {
                        const uint8_t* frow = flags + (z * p.occ_blocks_y + by) * p.occ_blocks_x;
                        one = frow[bx0] != 0 && frow[bx1] != 0;
                    }
                    st_one[si][rd][z] = one;
                }
            }
        }
    }
    // A stream's occlusion planes and a page of ones live in one allocation: a copy's source is the stream's uniform base
    // plus a 32-bit offset, and flagged-empty lanes only swap the offset (the same number of copy instructions per wave and
    // slice either way, which the vmcnt bookkeeping of the slice loop relies on).
    auto stage_occ = [&](int sf, int q) {
        if (sf >= g.n) return;
        const int group = (p.occ_phase + sf) / kOccSlices;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (!st_ok[rd]) continue;
            const uint32_t px = (uint32_t) (sf * plane_elems + st_src[rd]);
#pragma unroll
            for (int si = 0; si < NRP; ++si) {
                const ChunkStream& s = si == 0 ? p.a : p.r;
                const bool one = group == 0 ? st_one[si][rd][0] : st_one[si][rd][1];
                const uint32_t off = one ? (uint32_t) lane * 4u : s.occ_off + px;
                dma_16(s.occ_base + off, ring(q, si) + st_dst[rd]);
            }
        }
    };

    // ---- input state: the plane after the previous chunk ------------------------------------------------------------
    if (!p.first_chunk) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (!st_ok[rd]) continue;
            dma_16(p.a.plane_in + st_src[rd], window(0, 0) + st_dst[rd]);
            if constexpr (NS == 2) dma_16(p.r.plane_in + st_src[rd], window(0, 1) + st_dst[rd]);
        }
    }

    // ---- light-volume tile: the 4x4 brick columns under the tile, every brick layer the chunk touches --------------
    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS; // plane axes -> volume axes
    const int lbn[3] = {p.lv_bnx, p.lv_bnxy / p.lv_bnx, (p.lv_dims[2] + 7) >> 3};
    auto tile_brick_global = [&](int lb, bool& exists) -> uint32_t { // lb = (layer*BY + bv)*4 + bu
        const int bu = (base_x >> 3) + (lb & 3), bv = (base_y >> 3) + ((lb >> 2) & (BY - 1)), bl = g.lv_layer0 + lb / (4 * BY);
        exists = bu < lbn[dim_u] && bv < lbn[dim_v] && bl < lbn[dim_s];
        int b3[3];
        b3[dim_u] = bu; b3[dim_v] = bv; b3[dim_s] = bl;
        return (uint32_t) ((b3[2] * lbn[1] + b3[1]) * lbn[0] + b3[0]) * 512u;
    };
    if constexpr (LV_LDS) {
        const int chunks = 4 * BY * g.lv_layers * 32; // 16-byte pieces
        for (int cb = wave * 64; cb < chunks; cb += NT) {
            const int c = cb + lane;
            bool exists;
            const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
            if (c < chunks && exists) dma_16((const uint8_t*) p.light + gofs, lv_tile + cb * 16);
        }
    }

    // ---- this thread's slots: slot 0 = its owned pixel (8x8 patch per wave over the 32x32 tile), the rest = its share of
    // the halo (hull minus tile). A slot's pixel, window index, tap offsets and weights never change during the chunk.
    int sqx[KS], sqy[KS];
    sqx[0] = (wave & 3) * 8 + (lane & 7);
    sqy[0] = (wave >> 2) * 8 + (lane >> 3);
    {
        // halo slots enumerated as: full rows above the tile, the two side strips of the tile rows, full rows below
        const int top = g.pady * g.HX, side = g.HX - TX, mid = TY * side;
        const int n_halo = g.HX * g.HY - TX * TY;
        const float inv_hx = 1.0f / (float) g.HX, inv_side = side > 0 ? 1.0f / (float) side : 0.0f;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const int h = threadIdx.x + k * NT;
            int lx = 0, ly = 0;
            if (h < top) { ly = (int) (((float) h + 0.5f) * inv_hx); lx = h - ly * g.HX; }
            else if (h < top + mid) {
                const int m = h - top;
                const int row = (int) (((float) m + 0.5f) * inv_side), col = m - row * side;
                ly = g.pady + row;
                lx = col < g.padx ? col : col + TX;
            } else {
                const int m = h - top - mid;
                const int row = (int) (((float) m + 0.5f) * inv_hx);
                ly = g.pady + TY + row;
                lx = m - row * g.HX;
            }
            sqx[1 + k] = h < n_halo ? lx - g.padx : INT32_MIN / 2; // sentinel: never inside a window
            sqy[1 + k] = ly - g.pady;
        }
    }
    // floor(a / d) for 0 <= a < 4096 and a small positive integer d, as a multiplication: (a + 0.5) / d is never within
    // 0.5 / d of an integer, far more than the rounding error of the product
    auto small_div = [](int a, float inv_d) -> int { return (int) (((float) a + 0.5f) * inv_d); };
    const float inv_lox = g.lox < 0 ? 1.0f / (float) -g.lox : 0.0f, inv_hix = g.hix > 0 ? 1.0f / (float) g.hix : 0.0f;
    const float inv_loy = g.loy < 0 ? 1.0f / (float) -g.loy : 0.0f, inv_hiy = g.hiy > 0 ? 1.0f / (float) g.hiy : 0.0f;
    int rmin[KS];          // the slot is inside the window while r >= rmin (huge: never / outside the buffer)
    int li[KS];            // LDS index of the slot inside a plane
    int ti[NS][KS];        // LDS index of the first previous-slice tap inside a plane
    float wfx[NS][KS], wfy[NS][KS]; // bilinear weights of the previous-slice fetch
    bool off_plane[KS];    // inside the hull but outside the buffer: holds the border colour
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int qx = sqx[k], qy = sqy[k];
        const int px = base_x + qx, py = base_y + qy;
        const bool valid = qx > INT32_MIN / 4;
        const bool inplane = valid && (unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H;
        int need = 0; // smallest r whose window contains the slot: ceil(distance to the tile / growth per slice)
        if (qx < 0) need = max(need, g.lox < 0 ? small_div(-qx + (-g.lox) - 1, inv_lox) : INT32_MAX / 2);
        if (qx >= TX) need = max(need, g.hix > 0 ? small_div(qx - TX + g.hix, inv_hix) : INT32_MAX / 2);
        if (qy < 0) need = max(need, g.loy < 0 ? small_div(-qy + (-g.loy) - 1, inv_loy) : INT32_MAX / 2);
        if (qy >= TY) need = max(need, g.hiy > 0 ? small_div(qy - TY + g.hiy, inv_hiy) : INT32_MAX / 2);
        rmin[k] = inplane ? need : INT32_MAX / 2;
        off_plane[k] = valid && !inplane;
        li[k] = valid ? (qy + g.pady) * RS + qx + g.padx : 0;
        // (PixelLoc + 0.5) / BufferSize: the same for both streams
        const float pu = ((float) (uint32_t) px + 0.5f) / (float) p.W, pv = ((float) (uint32_t) py + 0.5f) / (float) p.H;
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            // previous-slice tap split: ((c + 0.5)/size + PrevPixelOffset) -> (tap - c, frac) (AddDirLightShader.usf:81-82)
            const ChunkStream& s = si == 0 ? p.a : p.r;
            int ix = 0, iy = 0;
            float fx = 0.0f, fy = 0.0f;
            if (inplane) {
                texel_split(pu + s.off_u, (float) p.W, ix, fx);
                texel_split(pv + s.off_v, (float) p.H, iy, fy);
                ix -= px;
                iy -= py;
            }
            ti[si][k] = li[k] + iy * RS + ix;
            wfx[si][k] = fx;
            wfy[si][k] = fy;
        }
    }
    const int own_idx = (base_y + sqy[0]) * p.W + base_x + sqx[0]; // owned pixel inside a plane

    // the owned pixel's voxel: constant in-plane part + per-slice part, as an offset into the bricked global volume
    // (float light volumes) or into the LDS tile (UNORM8)
    uint32_t lv_const = 0;
    {
        const int px = base_x + sqx[0], py = base_y + sqy[0];
        if constexpr (LV_LDS) {
            const uint32_t lb = (uint32_t) ((sqy[0] >> 3) * 4 + (sqx[0] >> 3)) * 512u;
            if (AXIS == 0) lv_const = lb + (uint32_t) (py & 7) * 64u + (uint32_t) (px & 7) * 8u;
            else if (AXIS == 1) lv_const = lb + (uint32_t) (py & 7) * 64u + (uint32_t) (px & 7);
            else lv_const = lb + (uint32_t) (py & 7) * 8u + (uint32_t) (px & 7);
        } else {
            if (AXIS == 0) lv_const = brick_off_y(px, p.lv_bnx) + brick_off_z(py, p.lv_bnxy);
            else if (AXIS == 1) lv_const = brick_off_x(px) + brick_off_z(py, p.lv_bnxy);
            else lv_const = brick_off_x(px) + brick_off_y(py, p.lv_bnx);
        }
    }
    auto lv_slice_off = [&](int j) -> uint32_t {
        if constexpr (LV_LDS) {
            const uint32_t layer = (uint32_t) ((j >> 3) - g.lv_layer0) * (uint32_t) (4 * BY) * 512u;
            return layer + (uint32_t) (j & 7) * (AXIS == 0 ? 1u : (AXIS == 1 ? 8u : 64u));
        } else {
            return AXIS == 0 ? brick_off_x(j) : (AXIS == 1 ? brick_off_y(j, p.lv_bnx) : brick_off_z(j, p.lv_bnxy));
        }
    };

    // the first two slices' occlusion planes, last: the flag loads they select their source with have had the slot set-up
    // above to arrive in
    stage_occ(0, 0);
    stage_occ(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's copies (input window, first occlusion planes, tile) have landed
    __syncthreads();
    // slots outside the buffer hold the read sampler's border colour in BOTH windows for the whole chunk
    // (AddDirLightShader.usf:22-25); in the first chunk the buffers were just cleared to the initial light
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            if (off_plane[k]) { window(0, si)[li[k]] = s.border_light; window(1, si)[li[k]] = s.border_light; }
            else if (p.first_chunk && rmin[k] < INT32_MAX / 2) window(0, si)[li[k]] = s.init_value;
        }
    }
    __syncthreads();
    // one slice: window `cur` holds the state before it, ring slot `q` the slice's occlusion factors
    auto step = [&](int s, int cur, int q) {
        const int r = g.n - 1 - s; // slices that remain after this one
        const uint32_t vi = lv_const + lv_slice_off(p.j0 + s * p.dir);
        // every LDS read of the slice first (the writes below may alias them as far as the compiler can tell, so reads
        // issued after a write would wait for it: issued up front, their latencies overlap instead of adding up)
        bool act[KS];
        float t00[KS][NS], t01[KS][NS], t10[KS][NS], t11[KS][NS], fac[KS][NS];
        float lv_old = 0.0f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            act[k] = r >= rmin[k];
            if (!act[k]) continue;
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const float* pw = window(cur, si) + ti[si][k];
                t00[k][si] = pw[0]; t01[k][si] = pw[1]; t10[k][si] = pw[RS]; t11[k][si] = pw[RS + 1];
                fac[k][si] = ring(q, si)[li[k]];
            }
            if (k == 0) {
                if constexpr (LV_LDS) lv_old = decode_u8(lv_tile[vi]);
                else lv_old = load_voxel<LFMT>(p.light, vi);
            }
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (!act[k]) continue;
            float lval[NS];
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                // previous slice, bilinear with border colour (AddDirLightShader.usf:81-82)
                const float prev = lerp_(lerp_(t00[k][si], t01[k][si], wfx[si][k]), lerp_(t10[k][si], t11[k][si], wfx[si][k]), wfy[si][k]);
                const float l = prev * fac[k][si]; // :117 (the occlusion kernel stored 1 - CurrentSample)
                lval[si] = l;
                window(cur ^ 1, si)[li[k]] = through_format<LFMT>(l); // WriteBuffer[PixelLoc] = L (:120)
            }
            if (k == 0) { // the owned pixel: this workgroup writes its light-volume voxel
                float nv;
                bool write;
                if constexpr (MODE == PASS_ADD) { nv = lv_old + lval[0] * p.b_added; write = fabsf(lval[0]) > 1e-3f; } // :123-126
                else if constexpr (MODE == PASS_CHANGE) { nv = lv_old + lval[0] - lval[NS - 1]; write = fabsf(lval[0] - lval[NS - 1]) > 1e-3f; } // Change :152-154
                else { // two lights added in one pass: light a's read-modify-write, then light r's on its result (:123-126 twice)
                    const bool wa = fabsf(lval[0]) > 1e-3f, wb = fabsf(lval[NS - 1]) > 1e-3f;
                    nv = wa ? through_format<LFMT>(lv_old + lval[0] * p.b_added) : lv_old;
                    if (wb) nv = nv + lval[NS - 1] * p.b_added2;
                    write = wa || wb;
                }
                if (write) {
                    if constexpr (LV_LDS) lv_tile[vi] = (uint8_t) encode_u8(nv);
                    else store_voxel<LFMT>(p.light, vi, nv);
                }
                if (r == 0) {
#pragma unroll
                    for (int si = 0; si < NS; ++si) (si == 0 ? p.a : p.r).plane_out[own_idx] = through_format<LFMT>(lval[si]);
                }
            }
        }
    };

    static_assert(kOccRing == 3, "the slice loops below are unrolled for a ring of three");
    if constexpr (M > 0) {
        // ---- fast slice loop ------------------------------------------------------------------------------------------
        // tools/ubench/slice_loop.hip reproduces a slice of the generic loop below (16 reads, 96 VALU, 4 writes, 64 scalar
        // instructions per wave and one barrier: 1950 cycles) and prices its parts: with one barrier per slice all 16 waves
        // of the CU are in the same phase at the same time, so the phases add up, and a SCALAR instruction costs the slice
        // as much as a vector one (~10 cycles each: 64 -> 16 scalars per wave: -470 cycles). The generic loop spends 74
        // scalar instructions per wave and slice on exec masks, loop and ring bookkeeping, runtime vmcnt selection and
        // brick addressing. Here the chunk length is a template parameter and the loop is unrolled, so slice number,
        // remaining slices, window parity, ring slot and the voxel's offset in the light-volume tile are immediates;
        // a pixel outside the plane or outside the current window is still computed but written to the plane's slack
        // word (one v_cndmask instead of an exec-mask region); every wave issues the same number of vector-memory
        // operations per slice, so the copy counter's operand is an immediate too.
        static_assert(ROUNDS == 1 && LV_LDS && (M == 8 || M == 16), "fast slice loop: one staging round, UNORM8 light volume");
        constexpr int DUMMY = RS * RR; // the plane's slack word: nobody reads it
        // staging: every lane with a group inside the plane copies (rows beyond the hull / the buffer copy a clamped row:
        // the pixels they feed are never valid); per stream the source offset of slice 0 and its advance per slice
        const bool st_in = (int) threadIdx.x < GROUPS;
        uint32_t st_off[NRP], st_adv[NRP][2];
        {
            const int gi = (int) threadIdx.x, row = gi / GPR, col = (gi - row * GPR) * 4;
            const int py = min(max(base_y - g.pady + row, 0), p.H - 1);
            const uint32_t src = (uint32_t) (py * p.W + base_x - g.padx + col);
#pragma unroll
            for (int si = 0; si < NRP; ++si) {
                const ChunkStream& st = si == 0 ? p.a : p.r;
#pragma unroll
                for (int z = 0; z < 2; ++z) st_adv[si][z] = (st_ok[0] && st_one[si][0][z]) ? 0u : (uint32_t) plane_elems;
                st_off[si] = st.occ_off + src;
            }
        }
        auto stage_fast = [&](auto sfc, auto qc) { // the planes of slice SF into ring slot Q
            constexpr int SF = decltype(sfc)::value, Q = decltype(qc)::value;
            if constexpr (SF < M) {
                if (st_in) {
#pragma unroll
                    for (int si = 0; si < NRP; ++si) {
                        const ChunkStream& st = si == 0 ? p.a : p.r;
                        const bool one = st_adv[si][SF / kOccSlices] == 0u;
                        const uint32_t off = one ? (uint32_t) lane * 4u : st_off[si] + (uint32_t) SF * (uint32_t) plane_elems;
                        dma_16(st.occ_base + off, ring(Q, si) + st_dst[0]);
                    }
                }
            }
        };
        // slot 0 never branches: a pixel outside the buffer is computed like any other and lands in the slack word
        const bool own_in = rmin[0] < INT32_MAX / 2;
        const int liw0 = own_in ? li[0] : DUMMY;
        // the owned voxel's offset in the light-volume tile: slice S of an aligned chunk is row S & 7 of layer S >> 3 (counted
        // from the chunk's last layer when the pass runs downwards)
        constexpr uint32_t kLvStep = AXIS == 0 ? 1u : (AXIS == 1 ? 8u : 64u);
        const bool down = p.dir < 0;
        const uint32_t lv_first = lv_const + (down ? (uint32_t) (M / 8 - 1) * 16u * 512u + 7u * kLvStep : 0u);
        auto slice_fast = [&](auto sc) {
            constexpr int S = decltype(sc)::value, R = M - 1 - S, CUR = S & 1, Q = S % 3;
            stage_fast(std::integral_constant<int, S + 2>{}, std::integral_constant<int, (S + 2) % 3>{});
            constexpr uint32_t lv_delta = (uint32_t) (S >> 3) * 16u * 512u + (uint32_t) (S & 7) * kLvStep;
            const uint32_t vi = down ? lv_first - lv_delta : lv_first + lv_delta;
            // owned pixel
            float t00[NS], t01[NS], t10[NS], t11[NS], fac0[NS];
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const float* pw = window(CUR, si) + ti[si][0];
                t00[si] = pw[0]; t01[si] = pw[1]; t10[si] = pw[RS]; t11[si] = pw[RS + 1];
                fac0[si] = ring(Q, si)[li[0]];
            }
            const uint32_t code_old = lv_tile[vi];
            // halo pixels: a wave none of whose lanes has a pixel inside the window skips the slot
            bool act[KS];
            float h00[KS][NS], h01[KS][NS], h10[KS][NS], h11[KS][NS], hfac[KS][NS];
#pragma unroll
            for (int k = 1; k < KS; ++k) {
                act[k] = R >= rmin[k];
                if (__builtin_amdgcn_ballot_w64(act[k]) == 0) continue;
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    const float* pw = window(CUR, si) + ti[si][k];
                    h00[k][si] = pw[0]; h01[k][si] = pw[1]; h10[k][si] = pw[RS]; h11[k][si] = pw[RS + 1];
                    hfac[k][si] = ring(Q, si)[li[k]];
                }
            }
            float lval[NS];
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const float prev = lerp_(lerp_(t00[si], t01[si], wfx[si][0]), lerp_(t10[si], t11[si], wfx[si][0]), wfy[si][0]);
                lval[si] = prev * fac0[si];
                window(CUR ^ 1, si)[liw0] = through_format<LFMT>(lval[si]);
            }
            {
                const float lv_old = decode_u8(code_old);
                float nv;
                bool write;
                if constexpr (MODE == PASS_ADD) { nv = lv_old + lval[0] * p.b_added; write = fabsf(lval[0]) > 1e-3f; }
                else if constexpr (MODE == PASS_CHANGE) { nv = lv_old + lval[0] - lval[NS - 1]; write = fabsf(lval[0] - lval[NS - 1]) > 1e-3f; }
                else {
                    const bool wa = fabsf(lval[0]) > 1e-3f, wb = fabsf(lval[NS - 1]) > 1e-3f;
                    nv = wa ? through_format<LFMT>(lv_old + lval[0] * p.b_added) : lv_old;
                    if (wb) nv = nv + lval[NS - 1] * p.b_added2;
                    write = wa || wb;
                }
                lv_tile[vi] = (uint8_t) ((write && own_in) ? encode_u8(nv) : code_old); // always stored: no exec-mask region
            }
            if constexpr (R == 0) {
                if (own_in) {
#pragma unroll
                    for (int si = 0; si < NS; ++si) (si == 0 ? p.a : p.r).plane_out[own_idx] = through_format<LFMT>(lval[si]);
                }
            }
#pragma unroll
            for (int k = 1; k < KS; ++k) {
                if (__builtin_amdgcn_ballot_w64(act[k]) == 0) continue;
                const int liw = act[k] ? li[k] : DUMMY;
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    const float prev = lerp_(lerp_(h00[k][si], h01[k][si], wfx[si][k]), lerp_(h10[k][si], h11[k][si], wfx[si][k]), wfy[si][k]);
                    window(CUR ^ 1, si)[liw] = through_format<LFMT>(prev * hfac[k][si]);
                }
            }
            // copies of slice S+1 (issued a slice ago) have to have landed; this slice's own (slice S+2's planes, the last
            // slice's planes) may stay in flight
            constexpr int in_flight = (S + 2 < M ? NR : 0) + (R == 0 ? NS : 0);
            if constexpr (in_flight == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (in_flight == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if constexpr (in_flight == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (in_flight == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if constexpr (in_flight == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            lds_barrier();
        };
        for_each_const(slice_fast, std::make_integer_sequence<int, M>{});
    } else
    // Per slice: refill the ring slot the PREVIOUS slice read (every wave left that slice at the barrier) with the slice
    // two ahead, compute, then wait until only that refill may still be in flight — the copies of slice s+1, issued a
    // whole slice ago, have landed — and meet at the barrier that also publishes this slice's window writes. Copies
    // complete in issue order and, with a UNORM8 light volume, are the only vector-memory operations of the loop; a float
    // light volume adds the owned voxel's load and conditional store, so that variant drains everything.
    for (int s0 = 0; s0 < g.n; s0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = s0 + u;
            stage_occ(s + 2, (u + 2) % 3);
            if (s < g.n) step(s, u & 1, u % 3);
            const int pending = (LV_LDS && s + 2 < g.n) ? ndma : 0;
            if (pending == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (pending == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (pending == 2 || NS * ROUNDS <= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (pending == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        }
    }

    // ---- write the tile's light-volume bricks back ----------------------------------------------------------------
    if constexpr (LV_LDS) {
        const int chunks = 4 * BY * g.lv_layers * 32;
        for (int c = threadIdx.x; c < chunks; c += NT) {
            bool exists;
            const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
            if (exists) *(uint4*) ((uint8_t*) p.light + gofs) = *(const uint4*) (lv_tile + c * 16);
        }
    }
}


template <int MODE, int AXIS, int KH, int RS, int M = 0, int RR = RS>
static hipError_t launch_chain4(const ChunkParams& p, hipStream_t s)
{
    constexpr int LFMT = TBRM_CHAIN_LFMT;
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_chain<LFMT, MODE, AXIS, KH, RS, M, RR>, attr_done, 160 * 1024); e != hipSuccess) return e;
    const size_t lds = chunk_lds_bytes(p, MODE, LFMT);
    hipLaunchKernelGGL((k_light_chain<LFMT, MODE, AXIS, KH, RS, M, RR>), dim3(8 * ((p.tiles_x * p.tiles_y + 7) / 8)), dim3(kChunkThreads), lds, s, p);
    return hipGetLastError();
}

// The fast slice loop (k_light_chain, M > 0) is instantiated for the shapes the planner produces for full chunks of light
// passes with taps one or two texels wide: 16 slices in 56 x 56 planes with 2 halo pixels per thread, 8 slices in 40 x 40
// planes (1) or 56 x 56 planes (1, 2); an Add also 16 and 8 slices in the rectangular 72 x 48 and 56 x 64 planes (2) — for a
// UNORM8 light volume. false: not one of them (the generic loop runs).
template <int MODE, int AXIS>
static bool launch_chain_fast(const ChunkParams& p, const ChunkGeom& g, int kh, hipStream_t s, hipError_t& err)
{
#if TBRM_CHAIN_LFMT == 0
    const bool aligned = p.occ_phase == 0 && (p.j0 & 7) == (p.dir > 0 ? 0 : 7) && g.lv_layers == g.n / 8;
    if (!aligned || tune(TUNE_CHAIN_FAST_LOOP) == 0) return false;
    auto go = [&](auto khc, auto rsc, auto mc) {
        constexpr int KH = decltype(khc)::value, RS = decltype(rsc)::value, M = decltype(mc)::value;
        err = launch_chain4<MODE, AXIS, KH, RS, M>(p, s);
        return true;
    };
    using std::integral_constant;
    if constexpr (MODE == PASS_ADD) { // the rectangular planes (ChunkParams::rect_planes)
        if (kh <= 2 && g.RS == 72 && g.RR == 48) {
            if (g.n == 16) { err = launch_chain4<MODE, AXIS, 2, 72, 16, 48>(p, s); return true; }
            if (g.n == 8) { err = launch_chain4<MODE, AXIS, 2, 72, 8, 48>(p, s); return true; }
        }
        if (kh <= 2 && g.RS == 56 && g.RR == 64) {
            if (g.n == 16) { err = launch_chain4<MODE, AXIS, 2, 56, 16, 64>(p, s); return true; }
            if (g.n == 8) { err = launch_chain4<MODE, AXIS, 2, 56, 8, 64>(p, s); return true; }
        }
    }
    if (g.RR != g.RS) return false;
    if (g.n == 16 && g.RS == 56 && kh == 2) return go(integral_constant<int, 2>{}, integral_constant<int, 56>{}, integral_constant<int, 16>{});
    if (g.n == 8 && g.RS == 40 && kh <= 1) return go(integral_constant<int, 1>{}, integral_constant<int, 40>{}, integral_constant<int, 8>{});
    if (g.n == 8 && g.RS == 56 && kh <= 1) return go(integral_constant<int, 1>{}, integral_constant<int, 56>{}, integral_constant<int, 8>{});
    if (g.n == 8 && g.RS == 56 && kh == 2) return go(integral_constant<int, 2>{}, integral_constant<int, 56>{}, integral_constant<int, 8>{});
#endif
    return false;
}

// The instantiated shapes of the generic loop (chunk_lds_bytes tells the planner which hulls have one): two streams RS 40
// (1 halo slot per thread) / 56 (1, 2, 3); one stream RS 40 (1) / 56 (3) / 72 (3) / 72 x 48 and 56 x 64 (2, UNORM8)
template <int MODE, int AXIS>
static hipError_t launch_chain3(const ChunkParams& p, hipStream_t s)
{
    const ChunkGeom g = chunk_geometry(p);
    const int halo = g.HX * g.HY - kChunkTile * kChunkTile;
    const int kh = (halo + kChunkThreads - 1) / kChunkThreads; // <= 3 for hulls up to 64 x 64
    hipError_t err = hipSuccess;
    if (launch_chain_fast<MODE, AXIS>(p, g, kh, s, err)) return err;
#if TBRM_CHAIN_LFMT == 0
    if constexpr (MODE == PASS_ADD) { // (a full chunk off the brick grid)
        if (g.RS == 72 && g.RR == 48 && kh <= 2) return launch_chain4<MODE, AXIS, 2, 72, 0, 48>(p, s);
        if (g.RS == 56 && g.RR == 64 && kh <= 2) return launch_chain4<MODE, AXIS, 2, 56, 0, 64>(p, s);
    }
#endif
    if (g.RR != g.RS) return hipErrorInvalidConfiguration;
    if constexpr (MODE != PASS_ADD) { // two streams double the per-slot state: the exact slot count keeps the kernel out of scratch
        if (g.RS == 40) return launch_chain4<MODE, AXIS, 1, 40>(p, s);
        if (g.RS == 56) {
            if (kh <= 1) return launch_chain4<MODE, AXIS, 1, 56>(p, s);
            if (kh == 2) return launch_chain4<MODE, AXIS, 2, 56>(p, s);
            return launch_chain4<MODE, AXIS, 3, 56>(p, s);
        }
    } else {
        if (g.RS == 40) return launch_chain4<MODE, AXIS, 1, 40>(p, s);
        if (g.RS == 56) return launch_chain4<MODE, AXIS, 3, 56>(p, s);
        if (g.RS == 72 && kh <= 3) return launch_chain4<MODE, AXIS, 3, 72>(p, s);
    }
    return hipErrorInvalidConfiguration; // the host's check (chunk_lds_bytes) rules these shapes out
}
template <int MODE>
static hipError_t launch_chain2(const ChunkParams& p, hipStream_t s)
{
    return p.axis == 0 ? launch_chain3<MODE, 0>(p, s) : (p.axis == 1 ? launch_chain3<MODE, 1>(p, s) : launch_chain3<MODE, 2>(p, s));
}
// advances every tile through the chunk (j0, n_steps), reading the occlusion planes at occ_base + {a,r}.occ_off
#if TBRM_CHAIN_LFMT == 0
hipError_t launch_light_chain_u8(const ChunkParams& p, int mode, hipStream_t s)
#else
hipError_t launch_light_chain_f32(const ChunkParams& p, int mode, hipStream_t s)
#endif
{
    return mode == PASS_ADD ? launch_chain2<PASS_ADD>(p, s) : (mode == PASS_CHANGE ? launch_chain2<PASS_CHANGE>(p, s) : launch_chain2<PASS_ADD2>(p, s));
}

} // namespace tbrm
