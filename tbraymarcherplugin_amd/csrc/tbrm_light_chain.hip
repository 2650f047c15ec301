// tbrm_light_chain.hip — k_light_chain, the serial half of a chunked axis pass of the illumination operators
// (AddDirLightShader.usf:68-128, ChangeDirLightShader.usf:74-156): one workgroup advances one tile of the slice plane
// through the 16 / 8 / 4 / 2 slices of a chunk. Compiled once per light-volume format (TBRM_CHAIN_LFMT = 0: UNORM8,
// 2: R32F) so that the two halves build in parallel; tbrm_light_kernels.hip holds the rest of the pass (occlusion,
// flags, the one-slice-per-launch kernel) and the story of why the pass is cut this way.
#include "tbrm_device_sampling.h"
#include "tbrm_light_chain.h"

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

// KH = halo pixels per thread: ceil((hull area - tile area) / threads); TY = tile height: 32 (1024 threads, one workgroup
// per CU at 512^2) or 16 (512 threads; two or three workgroups share a CU and fill each other's barrier and LDS stalls)
template <int LFMT, int MODE, int AXIS, int KH, int RS, int RR, int TY>
__global__ __launch_bounds__(kChunkTileW * TY, 4) void k_light_chain(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TX = kChunkTileW;
    constexpr int NT = TX * TY;                                          // threads: one per tile pixel
    constexpr int BY = TY / 8;                                           // light-volume bricks under the tile along v (4 along u)
    constexpr int NS = MODE != PASS_ADD ? 2 : 1;
    constexpr bool LV_LDS = LFMT == FMT_U8;
    constexpr int KS = 1 + KH; // + the owned pixel
    constexpr int PLANE = chain_plane_elems(RS, RR);
    constexpr int GPR = RS / 4;                                         // 16-byte copy groups per plane row
    constexpr int GROUPS = RR * GPR;
    constexpr int ROUNDS = (GROUPS + NT - 1) / NT;                       // copy groups per thread
    static_assert(RS % 16 == 8 && ROUNDS <= 2, "row stride must be an odd multiple of 8");
    static_assert(TY == 16 || TY == 32, "tile height");
    const ChunkGeom g = chunk_geometry(p);
    const int dbg = p.stagger < 0 ? -p.stagger : 0; // timing experiments (wrong results): see the chain_stagger tunable
    auto stamp = [&](int i) { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t) blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int plane_elems = p.H * p.W;
    // Tile of this workgroup. Workgroups go to the 8 XCDs round-robin by linear id (an affinity used for speed only): XCD x
    // takes the x-th eighth of the row-major tile list — a band of neighbouring tiles whose overlapping halo reads of the
    // occlusion planes then meet in one L2.
    const int n_tiles = p.tiles_x * p.tiles_y, per_xcd = (n_tiles + 7) >> 3;
    const int tile_id = ((int) blockIdx.x & 7) * per_xcd + ((int) blockIdx.x >> 3);
    if (((int) blockIdx.x >> 3) >= per_xcd || tile_id >= n_tiles) return;
    const int tile_y = tile_id / p.tiles_x, tile_x = tile_id - tile_y * p.tiles_x;
    const int base_x = tile_x * TX, base_y = (p.tile_row0 + tile_y) * TY;

    // LDS map (floats): window w of stream si at (w*NS + si)*PLANE, ring slot q at ((2 + q)*NS + si)*PLANE; then the
    // light-volume tile (bytes)
    float* const lds = (float*) smem;
    uint8_t* const lv_tile = (uint8_t*) (lds + (2 + kOccRing) * NS * PLANE);
    auto window = [&](int w, int si) -> float* { return lds + (w * NS + si) * PLANE; };
    auto ring = [&](int q, int si) -> float* { return lds + ((2 + q) * NS + si) * PLANE; };

    // ---- 16-byte staging pattern: copy group i = floats [4i, 4i+4) of an LDS plane = 4 pixels of one hull row ------
    int st_src[ROUNDS];   // pixel index of the group's first pixel inside a plane (may run off the row ends: guard bands)
    bool st_ok[ROUNDS];
    int st_dst[ROUNDS];   // this wave's 64 x 4 floats
    bool st_one[ROUNDS][2] = {}; // the group's 4 pixels lie in empty blocks of slice group 0 / 1 of the chunk
    int ndma = 0;         // copies this WAVE issues per staged slice (wave-uniform)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int gi = (int) threadIdx.x + rd * NT;
        const int row = gi / GPR, col = (gi - row * GPR) * 4;
        const int py = base_y - g.pady + row;
        st_src[rd] = py * p.W + base_x - g.padx + col;
        st_ok[rd] = gi < GROUPS && row < g.HY && col < g.HX && (unsigned) py < (unsigned) p.H;
        st_dst[rd] = (wave * 64 + rd * NT) * 4;
        if (__builtin_amdgcn_ballot_w64(st_ok[rd]) != 0) ndma += NS;
        // empty occlusion blocks (16x16 pixels x 8 slices) are handed over as one flag: their factor 1 - 0 is staged from
        // a page of ones
        if (p.occ_flags && st_ok[rd] && !(dbg & 64)) {
            const int x_first = base_x - g.padx + col, x_last = x_first + 3;
            const int bx0 = max(x_first, 0) >> 4, bx1 = min(x_last, p.W - 1) >> 4, by = py >> 4;
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                bool one = x_last >= 0 && x_first < p.W && z * kOccSlices < p.occ_phase + g.n;
                if (one) {
                    const uint8_t* frow = p.occ_flags + (z * p.occ_blocks_y + by) * p.occ_blocks_x;
                    one = frow[bx0] != 0 && frow[bx1] != 0;
                }
                st_one[rd][z] = one;
            }
        }
        if (p.stagger < 0 && ((-p.stagger) & 1)) st_one[rd][0] = st_one[rd][1] = true; // timing experiment: every factor from the L2-hot page of ones
    }
    stamp(1);
    // The occlusion stacks of both streams and the page of ones live in one allocation: a copy's source is the uniform
    // base plus a 32-bit offset, and flagged-empty lanes only swap the offset (the same number of copy instructions per
    // wave and slice either way, which the vmcnt bookkeeping of the slice loop relies on).
    auto stage_occ = [&](int sf, int q) {
        if (sf >= g.n) return;
        if (p.stagger < 0 && ((-p.stagger) & 2) && sf >= 2) return; // timing experiment: no copies inside the slice loop
        const int group = (p.occ_phase + sf) / kOccSlices;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (!st_ok[rd]) continue;
            const bool one = group == 0 ? st_one[rd][0] : st_one[rd][1];
            const uint32_t px = (uint32_t) (sf * plane_elems + st_src[rd]);
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const uint32_t off = one ? (uint32_t) lane * 4u : (si == 0 ? p.a.occ_off : p.r.occ_off) + px;
                dma_16(p.occ_base + off, ring(q, si) + st_dst[rd]);
            }
        }
    };

    // ---- input state: the plane after the previous chunk ------------------------------------------------------------
    if (!p.first_chunk && !(dbg & 256)) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (!st_ok[rd]) continue;
            dma_16(p.a.plane_in + st_src[rd], window(0, 0) + st_dst[rd]);
            if constexpr (NS == 2) dma_16(p.r.plane_in + st_src[rd], window(0, 1) + st_dst[rd]);
        }
    }
    stage_occ(0, 0);
    stage_occ(1, 1);

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
    if (LV_LDS && !(dbg & 128)) {
        const int chunks = 4 * BY * g.lv_layers * 32; // 16-byte pieces
        for (int cb = wave * 64; cb < chunks; cb += NT) {
            const int c = cb + lane;
            bool exists;
            const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
            if (c < chunks && exists) dma_16((const uint8_t*) p.light + gofs, lv_tile + cb * 16);
        }
    }

    stamp(2);
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

    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's copies (input window, first occlusion planes, tile) have landed
    __syncthreads();
    stamp(4);
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
    if (p.stagger > 0) {
        __shared__ int s_delay;
        if (threadIdx.x == 0) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            const uint32_t old = atomicAdd(&p.stagger_counters[((xcc & 7u) << 8) | ((hw >> 8) & 0xffu)], 1u);
            s_delay = (old & 1u) ? p.stagger : 0;
        }
        __syncthreads();
        const int d = s_delay;
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(2);
    }

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
            act[k] = r >= rmin[k] && !(k > 0 && dbg & 16);
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
                if (!(dbg & 8)) window(cur ^ 1, si)[li[k]] = through_format<LFMT>(l); // WriteBuffer[PixelLoc] = L (:120)
            }
            if (k == 0 && !(dbg & 32)) { // the owned pixel: this workgroup writes its light-volume voxel
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

    stamp(5);
    static_assert(kOccRing == 3, "the slice loop below is unrolled for a ring of three");
    // Per slice: refill the ring slot the PREVIOUS slice read (every wave left that slice at the barrier) with the slice
    // two ahead, compute, then wait until only that refill may still be in flight — the copies of slice s+1, issued a
    // whole slice ago, have landed — and meet at the barrier that also publishes this slice's window writes. Copies
    // complete in issue order and, with a UNORM8 light volume, are the only vector-memory operations of the loop; a float
    // light volume adds the owned voxel's load and conditional store, so that variant drains everything.
    for (int s0 = 0; s0 < ((dbg & 512) ? 0 : g.n); s0 += 6) {
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
            if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else lds_barrier();
        }
    }

    stamp(6);
    // ---- write the tile's light-volume bricks back ----------------------------------------------------------------
    if (LV_LDS && !(dbg & 128)) {
        const int chunks = 4 * BY * g.lv_layers * 32;
        for (int c = threadIdx.x; c < chunks; c += NT) {
            bool exists;
            const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
            if (exists) *(uint4*) ((uint8_t*) p.light + gofs) = *(const uint4*) (lv_tile + c * 16);
        }
    }
    stamp(7);
}


// ---- k_light_chain2: the same chunk, cut for the way a CU actually executes it --------------------------------------
// Measured on MI355X (tools/ubench/issue_mix.hip, tools/chain_stamps.py): a wave issues one instruction every ~5 cycles
// whatever its kind, waves on a SIMD overlap each other perfectly, and a slice of the kernel above takes exactly as long
// as ONE wave needs for its ~180 instructions (~2500 cycles) — the SIMDs idle at 0.3 instructions per cycle. So the slice
// time is set by the instruction count per wave, not by the work per CU: this kernel spreads a 32 x 16 tile over 1024
// threads — waves 0-7 own the tile's pixels (and its light-volume voxels), waves 8-15 take the halo pixels of the hull
// (one to three per lane) and all the global->LDS copies — and strips the per-slice instruction stream:
//   * every hull pixel is updated in every slice. The window of valid pixels still shrinks by the tap range per slice,
//     but a pixel that has left it is simply computed from garbage: no valid pixel ever reads it (its taps lie inside
//     the previous, larger window by construction), so the per-slot activity tests, exec masks and branches are gone;
//   * the two x taps of both footprint rows come from two ds_read2_b32 as (row0, row1) pairs, so the x lerp of both
//     rows is one packed multiply-add;
//   * the occlusion ring is as deep as the LDS share of half a CU allows (6 slices for a 40 x 24 hull), because a slice
//     now takes less time than a global->LDS copy needs to land;
//   * all copies of the prologue are issued before the slot set-up arithmetic, which then runs in their shadow.
// Two such workgroups share a CU (<= 64 VGPRs, <= 78 KB of LDS). Per-pixel arithmetic is that of the kernel above.
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kC2Threads = 1024, kC2TileH = 16, kC2Owners = kChunkTileW * kC2TileH;

template <int LFMT, int MODE, int AXIS, int KH, int RS, int RR>
__global__ __launch_bounds__(kC2Threads, 8) void k_light_chain2(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TX = kChunkTileW, TY = kC2TileH;
    constexpr int NS = MODE != PASS_ADD ? 2 : 1;
    constexpr bool LV_LDS = LFMT == FMT_U8;
    constexpr int PLANE = chain_plane_elems(RS, RR);
    constexpr int R = chain2_ring(RS, RR, NS);
    constexpr int GPR = RS / 4, GROUPS = RR * GPR;                      // 16-byte copy groups per plane row / per plane
    constexpr int NH = kC2Threads - kC2Owners;                          // halo lanes (waves 8-15)
    constexpr int ROUNDS = (GROUPS + NH - 1) / NH;                      // copy groups per halo lane
    constexpr int DUMMY = RS * RR;                                      // plane index nobody reads (the plane's slack)
    static_assert(RS % 16 == 8 && ROUNDS <= 2 && R >= 3 && R <= 6, "plane shape");
    const ChunkGeom g = chunk_geometry(p);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const bool owner = wave < kC2Owners / 64;
    const int plane_elems = p.H * p.W;
    const int n = g.n;
    // tile of this workgroup: XCD x takes the x-th eighth of the row-major tile list (see k_light_chain)
    const int n_tiles = p.tiles_x * p.tiles_y, per_xcd = (n_tiles + 7) >> 3;
    const int tile_id = ((int) blockIdx.x & 7) * per_xcd + ((int) blockIdx.x >> 3);
    if (((int) blockIdx.x >> 3) >= per_xcd || tile_id >= n_tiles) return;
    const int tile_y = tile_id / p.tiles_x, tile_x = tile_id - tile_y * p.tiles_x;
    const int base_x = tile_x * TX, base_y = (p.tile_row0 + tile_y) * TY;
    const int dbg = p.stagger < 0 ? -p.stagger : 0; // timing experiments (wrong results): see the chain_stagger tunable
    auto stamp = [&](int i) { if (p.stamps && (threadIdx.x == 0 || threadIdx.x == kC2Owners)) p.stamps[(size_t) blockIdx.x * 32 + (threadIdx.x ? 4 : 0) + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    unsigned long long ts[6] = {}; // timing experiment: phases of slice 3 of this wave (stored at the end)
    const bool stamping = p.stamps != nullptr;

    // LDS map (floats): ring slot q of stream si at (q*NS + si)*PLANE, window w at ((R + w)*NS + si)*PLANE, then the
    // light-volume tile (bytes) and 4 KB of slack. Pixels that have left the window read taps beyond their plane's ends
    // (up to 16 rows): with this order those reads stay inside the allocation.
    float* const lds = (float*) smem;
    uint8_t* const lv_tile = (uint8_t*) (lds + (2 + R) * NS * PLANE);
    auto ring = [&](int q, int si) -> float* { return lds + (q * NS + si) * PLANE; };
    auto window = [&](int w, int si) -> float* { return lds + ((R + w) * NS + si) * PLANE; };

    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS; // plane axes -> volume axes
    const int lbn[3] = {p.lv_bnx, p.lv_bnxy / p.lv_bnx, (p.lv_dims[2] + 7) >> 3};
    auto tile_brick_global = [&](int lb, bool& exists) -> uint32_t { // lb = (layer*2 + bv)*4 + bu
        const int bu = (base_x >> 3) + (lb & 3), bv = (base_y >> 3) + ((lb >> 2) & 1), bl = g.lv_layer0 + (lb >> 3);
        exists = bu < lbn[dim_u] && bv < lbn[dim_v] && bl < lbn[dim_s];
        int b3[3];
        b3[dim_u] = bu; b3[dim_v] = bv; b3[dim_s] = bl;
        return (uint32_t) ((b3[2] * lbn[1] + b3[1]) * lbn[0] + b3[0]) * 512u;
    };
    // previous-slice tap split of one pixel and stream: ((c + 0.5)/size + PrevPixelOffset) -> (tap - c, frac)
    // (AddDirLightShader.usf:81-82); (PixelLoc + 0.5) / BufferSize is the same for both streams
    struct Slot { int li, liw; int ti[NS]; float wfx[NS], wfy[NS]; bool off_plane, inplane; };
    auto make_slot = [&](int qx, int qy, bool valid) -> Slot {
        Slot t;
        const int px = base_x + qx, py = base_y + qy;
        t.inplane = valid && (unsigned) px < (unsigned) p.W && (unsigned) py < (unsigned) p.H;
        t.off_plane = valid && !t.inplane;
        t.li = valid ? (qy + g.pady) * RS + qx + g.padx : DUMMY;
        t.liw = t.inplane ? t.li : DUMMY; // pixels outside the buffer keep the border colour: their updates go nowhere
        const float pu = ((float) (uint32_t) px + 0.5f) / (float) p.W, pv = ((float) (uint32_t) py + 0.5f) / (float) p.H;
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            int ix = 0, iy = 0;
            float fx = 0.0f, fy = 0.0f;
            if (t.inplane) {
                texel_split(pu + s.off_u, (float) p.W, ix, fx);
                texel_split(pv + s.off_v, (float) p.H, iy, fy);
                ix -= px;
                iy -= py;
            }
            t.ti[si] = t.li + iy * RS + ix;
            t.wfx[si] = fx;
            t.wfy[si] = fy;
        }
        return t;
    };
    // one pixel of one slice: L = bilinear(previous slice) * (1 - CurrentSample), WriteBuffer[PixelLoc] = L (:117-120)
    auto advance = [&](const Slot& t, int cur, int ring_off, float (&l)[NS]) {
        float2 top[NS], bot[NS];
        float fac[NS];
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const float* pw = window(cur, si) + t.ti[si];
            top[si] = make_float2(pw[0], pw[RS]);      // (row 0, row 1) at the left tap: one ds_read2_b32
            bot[si] = make_float2(pw[1], pw[RS + 1]);  // ... and at the right tap
            fac[si] = (ring(0, si) + ring_off)[t.li];
        }
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const v2f a = {top[si].x, top[si].y}, b = {bot[si].x, bot[si].y}, f = {t.wfx[si], t.wfx[si]};
            const v2f h = __builtin_elementwise_fma(f, b - a, a); // lerp in x of both rows
            const float prev = lerp_(h.x, h.y, t.wfy[si]);
            l[si] = prev * fac[si];
            if (!(dbg & 8)) window(cur ^ 1, si)[t.liw] = through_format<LFMT>(l[si]);
        }
    };

    if (owner) {
        // ================= waves 0-7: the tile's pixels and light-volume voxels =================
        if constexpr (LV_LDS) { // the 4 x 2 brick columns under the tile, every brick layer the chunk touches
            const int pieces = 8 * g.lv_layers * 32; // 16-byte pieces
            for (int cb = wave * 64; cb < pieces; cb += kC2Owners) {
                const int c = cb + lane;
                bool exists;
                const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
                if (c < pieces && exists) dma_16((const uint8_t*) p.light + gofs, lv_tile + cb * 16);
            }
        }
        const int qx = (wave & 3) * 8 + (lane & 7), qy = (wave >> 2) * 8 + (lane >> 3);
        const Slot t = make_slot(qx, qy, true);
        const int px = base_x + qx, py = base_y + qy;
        const int own_idx = py * p.W + px;
        // the owned pixel's voxel: constant in-plane part + per-slice part, as an offset into the LDS tile (UNORM8) or into
        // the bricked global volume (float light volumes)
        uint32_t lv_const = 0;
        if constexpr (LV_LDS) {
            const uint32_t lb = (uint32_t) ((qy >> 3) * 4 + (qx >> 3)) * 512u;
            if (AXIS == 0) lv_const = lb + (uint32_t) (py & 7) * 64u + (uint32_t) (px & 7) * 8u;
            else if (AXIS == 1) lv_const = lb + (uint32_t) (py & 7) * 64u + (uint32_t) (px & 7);
            else lv_const = lb + (uint32_t) (py & 7) * 8u + (uint32_t) (px & 7);
        } else {
            if (AXIS == 0) lv_const = brick_off_y(px, p.lv_bnx) + brick_off_z(py, p.lv_bnxy);
            else if (AXIS == 1) lv_const = brick_off_x(px) + brick_off_z(py, p.lv_bnxy);
            else lv_const = brick_off_x(px) + brick_off_y(py, p.lv_bnx);
        }
        auto lv_slice_off = [&](int j) -> uint32_t {
            if constexpr (LV_LDS) {
                const uint32_t layer = (uint32_t) ((j >> 3) - g.lv_layer0) * 8u * 512u;
                return layer + (uint32_t) (j & 7) * (AXIS == 0 ? 1u : (AXIS == 1 ? 8u : 64u));
            } else {
                return AXIS == 0 ? brick_off_x(j) : (AXIS == 1 ? brick_off_y(j, p.lv_bnx) : brick_off_z(j, p.lv_bnxy));
            }
        };
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of the light-volume tile have landed
        __syncthreads();                                   // ... and the halo waves' copies of the input windows
        // pixels outside the buffer hold the read sampler's border colour in BOTH windows for the whole chunk
        // (AddDirLightShader.usf:22-25); in the first chunk the buffers were just cleared to the initial light
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            if (t.off_plane) { window(0, si)[t.li] = s.border_light; window(1, si)[t.li] = s.border_light; }
            else if (p.first_chunk) window(0, si)[t.li] = s.init_value;
        }
        __syncthreads();
        stamp(2);
        int ring_off = 0; // float offset of the ring slot of the current slice
        for (int s0 = 0; s0 < n; s0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = s0 + u;
                if (stamping && s == 3) ts[0] = __builtin_amdgcn_s_memtime();
                if (s < n) {
                    const uint32_t vi = lv_const + lv_slice_off(p.j0 + s * p.dir);
                    float lv_old;
                    if constexpr (LV_LDS) lv_old = decode_u8(lv_tile[vi]);
                    else lv_old = load_voxel<LFMT>(p.light, vi);
                    float l[NS];
                    advance(t, u, ring_off, l);
                    float nv;
                    bool write;
                    if constexpr (MODE == PASS_ADD) { nv = lv_old + l[0] * p.b_added; write = fabsf(l[0]) > 1e-3f; } // :123-126
                    else if constexpr (MODE == PASS_CHANGE) { nv = lv_old + l[0] - l[NS - 1]; write = fabsf(l[0] - l[NS - 1]) > 1e-3f; } // Change :152-154
                    else { // two lights added in one pass: light a's read-modify-write, then light r's on its result (:123-126 twice)
                        const bool wa = fabsf(l[0]) > 1e-3f, wb = fabsf(l[NS - 1]) > 1e-3f;
                        nv = wa ? through_format<LFMT>(lv_old + l[0] * p.b_added) : lv_old;
                        if (wb) nv = nv + l[NS - 1] * p.b_added2;
                        write = wa || wb;
                    }
                    if (write && t.inplane) { // (a tile may overhang the plane: D3D drops those threads' writes)
                        if constexpr (LV_LDS) lv_tile[vi] = (uint8_t) encode_u8(nv);
                        else store_voxel<LFMT>(p.light, vi, nv);
                    }
                    if (s == n - 1 && t.inplane) {
#pragma unroll
                        for (int si = 0; si < NS; ++si) (si == 0 ? p.a : p.r).plane_out[own_idx] = through_format<LFMT>(l[si]);
                    }
                    ring_off = ring_off + NS * PLANE == R * NS * PLANE ? 0 : ring_off + NS * PLANE;
                }
                if (stamping && s == 3) ts[3] = __builtin_amdgcn_s_memtime();
                if constexpr (!LV_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (stamping && s == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[4] = __builtin_amdgcn_s_memtime(); }
                if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else lds_barrier();
                if (stamping && s == 3) ts[5] = __builtin_amdgcn_s_memtime();
            }
        }
        stamp(3);
        if (stamping && threadIdx.x == 0) for (int i = 0; i < 6; ++i) p.stamps[(size_t) blockIdx.x * 32 + 8 + i] = ts[i];
        if constexpr (LV_LDS) { // write the tile's light-volume bricks back
            const int pieces = 8 * g.lv_layers * 32;
            for (int c = threadIdx.x; c < pieces; c += kC2Owners) {
                bool exists;
                const uint32_t gofs = tile_brick_global(c >> 5, exists) + (uint32_t) (c & 31) * 16u;
                if (exists) *(uint4*) ((uint8_t*) p.light + gofs) = *(const uint4*) (lv_tile + c * 16);
            }
        }
    } else {
        // ================= waves 8-15: the halo pixels and every global->LDS copy =================
        const int hl = (int) threadIdx.x - kC2Owners; // 0 .. NH-1
        const int hwave = wave - kC2Owners / 64;
        // 16-byte staging pattern: copy group i = floats [4i, 4i+4) of an LDS plane = 4 pixels of one hull row
        int st_src[ROUNDS];   // pixel index of the group's first pixel inside a plane (may run off the row ends: guard bands)
        bool st_ok[ROUNDS];
        int st_dst[ROUNDS];   // this wave's 64 x 4 floats
        int fl_row[ROUNDS], fl_b0[ROUNDS], fl_b1[ROUNDS]; // occlusion-block flags of the group: row offset, first / last block
        bool fl_in[ROUNDS];
        int ndma = 0;         // copies this WAVE issues per staged slice and stream (wave-uniform)
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int gi = hl + rd * NH;
            const int row = gi / GPR, col = (gi - row * GPR) * 4;
            const int py = base_y - g.pady + row;
            st_src[rd] = py * p.W + base_x - g.padx + col;
            st_ok[rd] = gi < GROUPS && row < g.HY && col < g.HX && (unsigned) py < (unsigned) p.H;
            st_dst[rd] = (hwave * 64 + rd * NH) * 4;
            if (__builtin_amdgcn_ballot_w64(st_ok[rd]) != 0) ++ndma;
            const int x_first = base_x - g.padx + col, x_last = x_first + 3;
            fl_in[rd] = st_ok[rd] && x_last >= 0 && x_first < p.W;
            fl_b0[rd] = max(x_first, 0) >> 4;
            fl_b1[rd] = min(x_last, p.W - 1) >> 4;
            fl_row[rd] = (py >> 4) * p.occ_blocks_x;
        }
        // empty occlusion blocks (16x16 pixels x 8 slices) are handed over as one flag: their factor 1 - 0 is staged from
        // a page of ones. The flag bytes are requested first and used after the input windows' copies have been issued.
        uint8_t fb[ROUNDS][2][2] = {};
        if (p.occ_flags) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
                for (int z = 0; z < 2; ++z)
                    if (fl_in[rd] && z * kOccSlices < p.occ_phase + n) {
                        const uint8_t* frow = p.occ_flags + z * p.occ_blocks_y * p.occ_blocks_x + fl_row[rd];
                        fb[rd][z][0] = __builtin_nontemporal_load(frow + fl_b0[rd]);
                        fb[rd][z][1] = __builtin_nontemporal_load(frow + fl_b1[rd]);
                    }
        }
        if (!p.first_chunk) { // input state: the plane after the previous chunk
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (!st_ok[rd]) continue;
                dma_16(p.a.plane_in + st_src[rd], window(0, 0) + st_dst[rd]);
                if constexpr (NS == 2) dma_16(p.r.plane_in + st_src[rd], window(0, 1) + st_dst[rd]);
            }
        }
        // this lane's halo pixels, enumerated as: full rows above the tile, the two side strips of the tile rows, full rows below
        Slot t[KH];
        {
            const int top = g.pady * g.HX, side = g.HX - TX, mid = TY * side;
            const int n_halo = g.HX * g.HY - TX * TY;
            const float inv_hx = 1.0f / (float) g.HX, inv_side = side > 0 ? 1.0f / (float) side : 0.0f;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int h = hl + k * NH;
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
                t[k] = make_slot(lx - g.padx, ly - g.pady, h < n_halo);
            }
        }
        // The occlusion stacks of both streams and the page of ones live in one allocation: a copy's source is the uniform
        // base plus a 32-bit offset, and flagged-empty lanes only swap the offset (the same number of copy instructions per
        // wave and slice either way, which the vmcnt bookkeeping of the slice loop relies on).
        bool st_one[ROUNDS][2];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int z = 0; z < 2; ++z) st_one[rd][z] = (dbg & 1) || (fl_in[rd] && p.occ_flags && z * kOccSlices < p.occ_phase + n && fb[rd][z][0] != 0 && fb[rd][z][1] != 0);
        auto stage_occ = [&](int sf, int q) {
            if (sf >= n) return;
            if ((dbg & 2) && sf >= R - 1) return;
            const int group = (p.occ_phase + sf) / kOccSlices;
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (!st_ok[rd]) continue;
                const bool one = group == 0 ? st_one[rd][0] : st_one[rd][1];
                const uint32_t px = (uint32_t) (sf * plane_elems + st_src[rd]);
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    const uint32_t off = one ? (uint32_t) lane * 4u : (si == 0 ? p.a.occ_off : p.r.occ_off) + px;
                    dma_16(p.occ_base + off, ring(q, si) + st_dst[rd]);
                }
            }
        };
#pragma unroll
        for (int q = 0; q < R - 1; ++q) stage_occ(q, q);
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's copies (input windows, first occlusion planes) have landed
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KH; ++k)
#pragma unroll
            for (int si = 0; si < NS; ++si) {
                const ChunkStream& s = si == 0 ? p.a : p.r;
                if (t[k].off_plane) { window(0, si)[t[k].li] = s.border_light; window(1, si)[t[k].li] = s.border_light; }
                else if (p.first_chunk && t[k].inplane) window(0, si)[t[k].li] = s.init_value;
            }
        __syncthreads();
        stamp(2);
        // Per slice: refill the ring slot the PREVIOUS slice read (every wave left that slice at the barrier) with the slice
        // R-1 ahead, update the halo pixels, then wait until only the copies of the slices after the next one may still be
        // in flight and meet at the barrier that also publishes this slice's window writes.
        int ring_off = 0, refill = R - 1; // float offset of the current slice's ring slot; ring slot the next copies go to
        for (int s0 = 0; s0 < n; s0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = s0 + u;
                if (stamping && s == 3) ts[0] = __builtin_amdgcn_s_memtime();
                if (s < n) {
                    stage_occ(s + R - 1, refill);
                    if (stamping && s == 3) ts[1] = __builtin_amdgcn_s_memtime();
                    refill = refill + 1 == R ? 0 : refill + 1;
#pragma unroll
                    for (int k = 0; k < KH; ++k) {
                        float l[NS];
                        if (!(dbg & 16)) advance(t[k], u, ring_off, l);
                    }
                    ring_off = ring_off + NS * PLANE == R * NS * PLANE ? 0 : ring_off + NS * PLANE;
                }
                if (stamping && s == 3) ts[3] = __builtin_amdgcn_s_memtime();
                // copies that may stay in flight: those of slices s+2 .. min(s+R-1, n-1)
                const int ahead = min(s + R - 1, n - 1) - (s + 1);
                const int pending = ahead > 0 ? ahead * ndma * NS : 0;
                // (the counter's operand is an immediate: the largest listed value not above `pending` is the safe choice)
                if (pending >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (pending >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (pending >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (pending >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (pending >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (pending == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (pending == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (pending == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (stamping && s == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[4] = __builtin_amdgcn_s_memtime(); }
                if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else lds_barrier();
                if (stamping && s == 3) ts[5] = __builtin_amdgcn_s_memtime();
            }
        }
        stamp(3);
        if (stamping && threadIdx.x == kC2Owners) for (int i = 0; i < 6; ++i) p.stamps[(size_t) blockIdx.x * 32 + 16 + i] = ts[i];
    }
}

template <int MODE, int AXIS, int KH, int RS, int RR, int TY>
static hipError_t launch_chain4(const ChunkParams& p, hipStream_t s)
{
    constexpr int LFMT = TBRM_CHAIN_LFMT;
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_chain<LFMT, MODE, AXIS, KH, RS, RR, TY>, attr_done, 159 * 1024); e != hipSuccess) return e;
    const size_t lds = chunk_lds_bytes(p, MODE != PASS_ADD, LFMT);
    hipLaunchKernelGGL((k_light_chain<LFMT, MODE, AXIS, KH, RS, RR, TY>), dim3(8 * ((p.tiles_x * p.tiles_y + 7) / 8)), dim3(kChunkTileW * TY), lds, s, p);
    return hipGetLastError();
}

template <int MODE, int AXIS, int KH, int RS, int RR>
static hipError_t launch_chain2_4(const ChunkParams& p, hipStream_t s)
{
    constexpr int LFMT = TBRM_CHAIN_LFMT;
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_chain2<LFMT, MODE, AXIS, KH, RS, RR>, attr_done, 159 * 1024); e != hipSuccess) return e;
    const size_t lds = chunk_lds_bytes(p, MODE != PASS_ADD, LFMT);
    hipLaunchKernelGGL((k_light_chain2<LFMT, MODE, AXIS, KH, RS, RR>), dim3(8 * ((p.tiles_x * p.tiles_y + 7) / 8)), dim3(kC2Threads), lds, s, p);
    return hipGetLastError();
}

// The instantiated shapes (chunk_lds_bytes tells the planner which hulls have one):
//   32 x 32 tiles: two streams RS 40 (1 halo slot per thread) / 56 (1, 2, 3); one stream RS 40 (1) / 56 (3) / 72 (3)
//   32 x 16 tiles (k_light_chain2): RS x RR = 40 x 24 (1 halo pixel per halo lane), 40 x 32 (2), 56 x 24 (2), 56 x 32 (2, 3)
template <int MODE, int AXIS>
static hipError_t launch_chain3(const ChunkParams& p, hipStream_t s)
{
    const ChunkGeom g = chunk_geometry(p);
    const int threads = kChunkTileW * g.TY;
    const int halo = g.HX * g.HY - threads;
    const int kh = (halo + threads - 1) / threads;
    if (g.TY == 16) { // 32 x 16 tiles: k_light_chain2, 512 halo lanes
        const int kh2 = (g.HX * g.HY - kC2Owners + 511) / 512;
        if (g.RS == 40 && g.RR == 24 && kh2 <= 1) return launch_chain2_4<MODE, AXIS, 1, 40, 24>(p, s);
        if (g.RS == 40 && g.RR == 32 && kh2 <= 2) return launch_chain2_4<MODE, AXIS, 2, 40, 32>(p, s);
        if (g.RS == 56 && g.RR == 24 && kh2 <= 2) return launch_chain2_4<MODE, AXIS, 2, 56, 24>(p, s);
        if (g.RS == 56 && g.RR == 32 && kh2 <= 2) return launch_chain2_4<MODE, AXIS, 2, 56, 32>(p, s);
        if (g.RS == 56 && g.RR == 32 && kh2 == 3) return launch_chain2_4<MODE, AXIS, 3, 56, 32>(p, s);
        return hipErrorInvalidConfiguration;
    }
    if constexpr (MODE != PASS_ADD) { // two streams double the per-slot state: the exact slot count keeps the kernel out of scratch
        if (g.RS == 40) return launch_chain4<MODE, AXIS, 1, 40, 40, 32>(p, s);
        if (g.RS == 56) {
            if (kh <= 1) return launch_chain4<MODE, AXIS, 1, 56, 56, 32>(p, s);
            if (kh == 2) return launch_chain4<MODE, AXIS, 2, 56, 56, 32>(p, s);
            return launch_chain4<MODE, AXIS, 3, 56, 56, 32>(p, s);
        }
    } else {
        if (g.RS == 40) return launch_chain4<MODE, AXIS, 1, 40, 40, 32>(p, s);
        if (g.RS == 56) return launch_chain4<MODE, AXIS, 3, 56, 56, 32>(p, s);
        if (g.RS == 72 && kh <= 3) return launch_chain4<MODE, AXIS, 3, 72, 72, 32>(p, s);
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
