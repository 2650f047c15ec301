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
// no boundary: after every slice a tile publishes the few columns / rows its downstream neighbours need, and it runs a few
// slices behind its upstream neighbours, whose values it has requested PF slices ahead.
//
// Hand-off. A plane value is a UNORM8 code (the read / write buffers are re-quantised every slice), so a record word is ONE
// dword, {16-bit tag of this launch, the removed light's code, the added light's code} — both streams of a Change travel in
// the same word —, written with one relaxed agent-scope store and polled with relaxed agent-scope
// loads (global_store / global_load ... sc1, MI355X_MICROARCH.md "handoff-1to1"): data and flag travel together, so there is
// no fence, no write-back of the L2, no second round trip and nothing that could tear. Records are addressed by
// [slice][tile][word] and never reused within a launch, so a producer never waits for a consumer; the tag makes clearing
// them between launches unnecessary.
//
// Forward progress. Tiles are dealt by an atomic ticket in upstream-first order: a tile only ever waits for tiles with a
// smaller ticket, which have started — whatever the number of workgroups the device keeps resident. A poll gives up after
// SweepParams::give_up_ticks of wall time (2 s by default) and raises SweepParams::error instead of hanging the device.
//
// Inside a tile: eight compute waves and one hand-off wave, ONE LDS-only barrier per slice.
//  - A compute lane owns one column of two consecutive rows. Per slice it reads its pixels' four taps from an LDS plane
//    (tile + halo + one guard ring for taps of weight 0), multiplies by the occlusion factor 1 - CurrentSample that
//    k_light_occlusion left in the span's plane stack (requested 6 slices ahead, straight into registers), re-quantises
//    (RaymarchVolume.cpp:857-866), updates its voxels in the light-volume bricks staged in LDS (a brick layer per 8 slices,
//    three buffers: in use, being written back, being installed) and stores the new plane values. The two rows go through
//    the packed fp32 instructions (v_pk_fma_f32 ...): a slice is bound by how fast ONE wave issues, not by the SIMD, and a
//    packed instruction is one issue slot for both pixels. The loop has no branch and no hand-off code.
//  - The hand-off wave publishes the tile's boundary cells of the slice just finished (read back from the LDS plane), writes
//    the upstream neighbours' cells of the current slice into the halo of the plane being built, and requests the words it
//    will need PF slices from now. Nothing it waits for delays the compute waves unless the neighbours really are late.
// No thread owns halo pixels, so a slice costs what its 1024 pixels cost. Arithmetic per voxel is the chain's and the
// reference's, bit for bit. Spans start on a brick layer of the light volume and are whole layers long: a pass whose length is
// no multiple of 8 is run over the volume padded to whole layers — the slices beyond the volume compute garbage into the light
// volume's padding voxels (which nothing reads), behind the real slices when the pass runs upwards, and in front of them when it
// runs downwards: then the last of them hands the pass's initial plane on (SweepParams::reinit_slice).
#include "tbrm_device_sampling.h"
#include "tbrm_light_chain.h"
#include "tbrm_light_sweep.h"

#include <type_traits>
#include <utility>

namespace tbrm {

template <class F, int... S>
__device__ __forceinline__ void sweep_each_const(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

// (W: uint32_t — a tagged dword of UNORM8 codes — or uint64_t — {float, launch tag}, one naturally aligned 8-byte granule)
template <class W>
__device__ __forceinline__ W sweep_load_word(const W* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class W>
__device__ __forceinline__ void sweep_store_word(W* p, W w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The slow path of a hand-off: the neighbour has not published the word yet. Its load is inline assembly so that the
// compiler's wait-count bookkeeping of the caller never sees a loop with a memory operation in it (it would answer with
// s_waitcnt vmcnt(0) at every later use of a request that is still in flight).
__device__ __noinline__ uint32_t sweep_poll(const uint32_t* src, uint32_t epoch, int* error, unsigned long long give_up_ticks)
{
    uint32_t w = 0;
    const unsigned long long t0 = wall_clock64(); // (100 MHz, constant: a starved or shared device gets wall time, not a poll count)
    for (;;) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
        if ((w >> 16) == epoch) return w;
        if (wall_clock64() - t0 >= give_up_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicOr(error, 1);
    return w;
}
// float light volumes: a record word is {float, launch tag} in one 8-byte granule (one store, one load: nothing can tear)
__device__ __noinline__ uint64_t sweep_poll(const uint64_t* src, uint32_t epoch, int* error, unsigned long long give_up_ticks)
{
    uint64_t w = 0;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
        if ((uint32_t) (w >> 32) == epoch) return w;
        if (wall_clock64() - t0 >= give_up_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicOr(error, 1);
    return w;
}

// CHAIN (k_light_sweep_chain, tbrm_internal.h SweepLink): the slow path of waiting for the pass before — the tile that owns the
// bricks about to be read has not written them back yet. Same form as sweep_poll.
__device__ __noinline__ uint32_t sweep_poll_progress(const uint32_t* src, uint32_t epoch, uint32_t needed, int* error, unsigned long long give_up_ticks)
{
    uint32_t w = 0;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
        if ((w >> 16) == epoch && (w & 0xffffu) >= needed) return w;
        if (wall_clock64() - t0 >= give_up_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicOr(error, 1);
    return w;
}
// 16 bytes of the light volume past the XCD's L2 in both directions (MI355X_MICROARCH.md: "16-B sc1 stores AND sc1 loads")
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sweep_store16_sc1(void* dst, uint4 v)
{
    v4u d;
    d.x = v.x; d.y = v.y; d.z = v.z; d.w = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(d) : "memory");
}
// ... as TWO vector-memory operations whether coherent (sc1) or not: the slice loop counts them (s_waitcnt vmcnt(2) leaves exactly
// these two in flight and nothing older — the layer written back before them has left the wave)
template <bool COHERENT>
__device__ __forceinline__ uint4 sweep_load16_two_ops(const void* src)
{
    constexpr int SCOPE = COHERENT ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    const uint64_t a = __hip_atomic_load((const uint64_t*) src, __ATOMIC_RELAXED, SCOPE);
    const uint64_t b = __hip_atomic_load((const uint64_t*) src + 1, __ATOMIC_RELAXED, SCOPE);
    return make_uint4((uint32_t) a, (uint32_t) (a >> 32), (uint32_t) b, (uint32_t) (b >> 32));
}

// One block slice of occlusion factors (64 lanes x 16 bytes = the 256 floats of a 16 x 16 block) from global memory straight
// into LDS at `lds_dst` (wave-uniform byte address) + lane * 16: no registers, counted by vmcnt like any load — but invisible
// to the compiler's own wait-count bookkeeping, which is the point: the loader waits with sweep_wait_loads<N>() for exactly
// the slices it needs (cdna_hip_programming.md, "LDS-DMA recipe").
__device__ __forceinline__ void sweep_dma_block(const void* src, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void sweep_wait_loads() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// two pixels (the two rows of a compute lane) per instruction
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f lerp2(v2f a, v2f b, v2f f) { return fma2(f, b - a, a); }                     // lerp_ (tbrm_device_math.h)
__device__ __forceinline__ v2f decode2(v2f c) { return fma2(c, (v2f) 0x1.010102p-8f, c * (v2f) -0x1.fdfdfep-33f); } // decode_u8f
__device__ __forceinline__ v2f quantize2(v2f x)                                                          // quantize_u8
{
    v2f c;
    c.x = __builtin_amdgcn_fmed3f(x.x, 0.0f, 1.0f);
    c.y = __builtin_amdgcn_fmed3f(x.y, 0.0f, 1.0f);
    c = c * (v2f) 255.0f + (v2f) 0.5f;
    c.x = __builtin_floorf(c.x);
    c.y = __builtin_floorf(c.y);
    return c;
}
__device__ __forceinline__ v2f quantize2_unfloored(v2f x) // (>= 0.5: the conversion to an integer code IS the floor)
{
    v2f c;
    c.x = __builtin_amdgcn_fmed3f(x.x, 0.0f, 1.0f);
    c.y = __builtin_amdgcn_fmed3f(x.y, 0.0f, 1.0f);
    return c * (v2f) 255.0f + (v2f) 0.5f;
}

// MODE: PASS_ADD (stream a), PASS_CHANGE (a added, r removed) or PASS_ADD2 (two lights that leave the same cube face added in one
// sweep: per voxel light a's read-modify-write, then light r's on its result — exactly pass a followed by pass r). PF: slices ahead of their use that the neighbours'
// hand-off words are requested (a tile settles PF + 1 slices and one memory round trip behind its upstream neighbours).
// HC: 64-word chunks of hand-off words per slice. RREC (PASS_CHANGE): stream r's halo comes from the records of an earlier
// PASS_PLANES launch (SweepParams::r_from_records). MODE PASS_PLANES: one stream, the light volume untouched.
// LFMT: the light volume's (and the read / write buffers') format. FMT_U8: planes are UNORM8 codes re-quantised every slice, one
// hand-off word carries a launch tag and both streams' codes, the tile's light-volume bricks are staged in LDS. FMT_F32
// (bLightVolume32Bit, RaymarchVolume.cpp:857-866): planes are floats as they are, a hand-off word is an 8-byte granule {float, launch
// tag} (one per stream, written by one store and read by one load), and the light volume is updated
// in place — fire-and-forget fp32 atomic adds, the removed light's as a second add of -L: (LV + La) - Lr rounds twice, like the
// reference's expression (ChangeDirLightShader.usf:152-154).
// CHAIN: the tile belongs to one of several passes of ONE launch (k_light_sweep_chain; SweepLink says how it waits for the pass
// before and what it publishes for the pass behind); `ticket` is the tile's ticket within its pass.
template <int MODE, int AXIS, int PF, int HC, bool RREC, int LFMT, int TH, bool CHAIN>
__device__ __forceinline__ void sweep_tile(const ChunkParams& p, const SweepParams& q, const SweepLink& link, const int ticket)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(TH == 32 || TH == 16, "a tile is 32 x 32 or 32 x 16 pixels");
    static_assert(!CHAIN || (LFMT == FMT_U8 && !RREC && (MODE == PASS_ADD || MODE == PASS_CHANGE)), "chained passes: Add / fused Change over a UNORM8 light volume");
    constexpr int T = kSweepTile, CS = sweep_col_stride(TH), PLANE = sweep_plane(TH), LVB = kSweepLvBrick;
    constexpr int R = 2, NWC = sweep_compute_waves(TH), NTC = NWC * 64, NT = sweep_threads(MODE, TH);
    constexpr int NB = sweep_blocks(TH), NBR = sweep_bricks(TH);
    constexpr int NS = sweep_two_streams(MODE) ? 2 : 1;
    constexpr bool LV = MODE != PASS_PLANES; // the light volume is updated
    constexpr bool F32 = LFMT == FMT_F32;
    constexpr bool LVS = LV && !F32;         // ... through brick layers staged in LDS
    constexpr int NSH = RREC ? 1 : NS;       // streams handed over from tile to tile in this launch (RREC: stream r comes from records)
    constexpr int NSW = F32 ? NSH : 1;       // hand-off words per cell
    static_assert(!RREC || MODE == PASS_CHANGE, "only a fused Change takes a stream from records");
    static_assert(!F32 || MODE == PASS_ADD || MODE == PASS_CHANGE || MODE == PASS_PLANES, "float light volumes: Add, fused Change, planes");
    constexpr int RING = kSweepRing;
    static_assert(PF >= 1 && PF < RING, "the request ring holds 8 slices");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);

    // ---- which tile: tickets in upstream-first order (taken by the kernel) -------------------------------------------
    const int n_tiles = p.tiles_x * p.tiles_y;
    const int ui = ticket % p.tiles_x, uj = ticket / p.tiles_x;
    const int tile_x = q.sx > 0 ? p.tiles_x - 1 - ui : ui, tile_y = q.sy > 0 ? p.tiles_y - 1 - uj : uj;
    const int tile_lin = tile_y * p.tiles_x + tile_x;
    const int base_x = tile_x * T, base_y = tile_y * TH;
    const int n = p.n_steps, G = n >> 3; // whole brick layers (the launcher's check)
    const int hx = q.hx, hy = q.hy;
    if constexpr (CHAIN) {
        if ((q.debug & 2) != 0 && q.stamps != nullptr && threadIdx.x == 0) q.stamps[4 * tile_lin + 0] = wall_clock64();
    }
    // LDS plane coordinates of tile pixel (0, 0): behind the guard ring and whatever halo lies on the low side
    const int ox = 1 + max(q.sx < 0 ? hx : 0, (RREC && q.r_sx < 0) ? q.r_hx : 0), oy = 1 + max(q.sy < 0 ? hy : 0, (RREC && q.r_sy < 0) ? q.r_hy : 0);
    const bool down = p.dir < 0;
    const int layer0 = p.j0 >> 3;

    // LDS map: plane(buf, si) = the propagated light of stream si before (buf = parity of the slice) / after a slice; three
    // light-volume brick layers (layer g of the span in buffer g % 3); the tile's empty-block flags
    float* const lds = (float*) smem;
    auto plane = [&](int buf, int si) -> float* { return lds + (buf * NS + si) * PLANE; };
    uint8_t* const lvt = (uint8_t*) (lds + 2 * NS * PLANE);
    int32_t* const sslot = (int32_t*) (lvt + (F32 ? 0 : 3 * NBR * LVB)); // [si][2 x TH / 16 blocks][slice group]: rank of the block, < 0: flagged empty
    // the ring of factor slices, [slot][si][2 x 2 blocks][kSweepFBlock]: filled FS - 1 slices ahead by the loader wave
    float* const fring = (float*) (sslot + NS * NB * G);
    constexpr int FS = sweep_factor_slots(MODE); // (divides the loop's eight slices: a slice's slot is a constant)
    // EARLY: what a compute wave reads of the next slice that no other compute wave writes — its factors, the bytes of the voxels it
    // updates — is read at the END of a slice, in front of the barrier, where the wave would wait anyway: the burst of LDS reads
    // behind the barrier, which all eight waves start at once and the arithmetic waits for, is a third shorter
    constexpr bool EARLY_LV = TBRM_SWEEP_EARLY_READS != 0;   // the voxels' bytes
    constexpr bool EARLY = EARLY_LV && FS >= 8;              // ... and the factors (the loader lands a slice one barrier earlier: a ring of eight)
    constexpr int kFSlot = NS * NB * kSweepFBlock; // floats per slot
    auto stream = [&](int si) -> const ChunkStream& { return si == 0 ? p.a : p.r; };

    // ---- the planes before the span's first slice ------------------------------------------------------------------------
    for (int i = threadIdx.x; i < PLANE; i += NT) {
        const int cx = i / CS, cy = i - cx * CS;
        const int gx = base_x + cx - ox, gy = base_y + cy - oy;
        const bool in = (unsigned) gx < (unsigned) p.W && (unsigned) gy < (unsigned) p.H;
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const ChunkStream& s = stream(si);
            // outside the buffer: the read sampler's border colour, for the whole pass (AddDirLightShader.usf:22-25)
            float v = s.border_light;
            if (in) v = p.first_chunk ? s.init_value : s.plane_in[gy * p.W + gx];
            plane(0, si)[i] = v;
            plane(1, si)[i] = s.border_light;
        }
    }
    // the tile's 2 x 2 occlusion blocks in every slice group of the span: their ranks among the pass's live blocks
    for (int i = threadIdx.x; i < NS * NB * G; i += NT) {
        const int si = i / (NB * G), blk = (i / G) % NB, zg = i % G;
        const int bx = (base_x >> 4) + (blk & 1), by = (base_y >> 4) + (blk >> 1);
        int32_t slot = -1;
        if (zg < G && bx < p.occ_blocks_x && by < p.occ_blocks_y) slot = stream(si).fs_slot[((size_t) zg * p.occ_blocks_y + by) * p.occ_blocks_x + bx];
        sslot[i] = slot;
    }

    // ---- light-volume bricks: a layer = the 4 x TH / 8 bricks under the tile, in pieces of 16 bytes, one per compute thread ----
    constexpr int dim_u = AXIS == 0 ? 1 : 0, dim_v = AXIS == 2 ? 1 : 2, dim_s = AXIS; // plane axes -> volume axes
    const int lbn[3] = {p.lv_bnx, p.lv_bnxy / p.lv_bnx, (p.lv_dims[2] + 7) >> 3};
    const int piece = (int) threadIdx.x & (NTC - 1);
    bool piece_exists; // (the tile may hang over the volume's last bricks)
    uint32_t piece_off; // of its brick column's piece in layer 0 of the volume
    uint32_t layer_stride;
    {
        const int lb = piece >> 5;
        const int bu = (base_x >> 3) + (lb & 3), bv = (base_y >> 3) + (lb >> 2);
        piece_exists = bu < lbn[dim_u] && bv < lbn[dim_v];
        int b3[3], l3[3] = {0, 0, 0};
        b3[dim_u] = bu; b3[dim_v] = bv; b3[dim_s] = 0;
        l3[dim_s] = 1;
        piece_off = piece_exists ? (uint32_t) ((b3[2] * lbn[1] + b3[1]) * lbn[0] + b3[0]) * 512u + (uint32_t) (piece & 31) * 16u : 0u;
        layer_stride = (uint32_t) ((l3[2] * lbn[1] + l3[1]) * lbn[0] + l3[0]) * 512u;
    }
    uint4* const piece_lds = (uint4*) (lvt + (piece >> 5) * LVB + (piece & 31) * 16); // in buffer 0
    constexpr int kLvBuf = NBR * LVB;
    auto layer_of = [&](int g) -> int { return down ? layer0 - g : layer0 + g; };
    // (a layer past the span's end is loaded from the span's last layer instead and never used: no branch around the load)
    auto load_layer = [&](int g) -> uint4 {
        const uint8_t* const src = (const uint8_t*) p.light + (piece_off + (uint32_t) layer_of(min(g, G - 1)) * layer_stride);
        if constexpr (CHAIN) {
            // sc1 for every pass but the launch's first: this CU's L1 may still hold a brick as the tile it ran BEFORE read it (no kernel
            // boundary has invalidated it since, and the write-through store that replaced it in memory went past the L1), and from
            // the launch's third pass on this XCD's L2 may hold what a tile of the pass before the last read here
            return link.coherent_loads ? sweep_load16_two_ops<true>(src) : sweep_load16_two_ops<false>(src);
        } else return *(const uint4*) src;
    };
    auto write_back_layer = [&](int g) {
        if (piece_exists) {
            uint8_t* const dst = (uint8_t*) p.light + (piece_off + (uint32_t) layer_of(g) * layer_stride);
            const uint4 v = *(const uint4*) ((const uint8_t*) piece_lds + (g % 3) * kLvBuf);
            if constexpr (CHAIN) sweep_store16_sc1(dst, v);
            else *(uint4*) dst = v;
        }
    };
    // CHAIN: which tile of the pass before owns the bricks of this tile's layer g, and how many of ITS layers it must have written
    // back (SweepLink). The tile's bricks along the pass before's axis a: one layer when a is this pass's axis, else the four (TH / 8)
    // bricks of the tile's extent along a; along the pass before's plane axes they lie inside one of its tiles (a range of bricks
    // under a tile is aligned to it, a single layer lies inside one). Both are affine in the volume layer L = layer_of(g):
    //   progress word  dep_base[dep_c1 * (L >> 2)]      (L >> 2: the tile of the pass before that holds layer L, when this pass's
    //   layers needed  dep_n0 + dep_n1 * L               axis is one of its plane axes — 4 brick layers per 32-pixel tile)
    // worked out once per tile and kept in vector registers (the slice loop has no scalar registers to spare: what it would
    // otherwise fetch from the argument block again costs a scalar-memory round trip inside a slice).
    const uint32_t* dep_base = link.prog_in;
    int dep_c1 = 0, dep_n0 = 0, dep_n1 = 0;
    int dep_G = link.in_G;
    uint32_t dep_epoch = link.in_epoch & 0xffffu;
    uint32_t* prog_word = link.prog_out + tile_lin;
    if constexpr (CHAIN) {
        const int a = link.in_axis;
        const int ua = a == 0 ? 1 : 0, va = a == 2 ? 1 : 2;
        int lo[3]; // first brick of the tile per volume axis (along this pass's axis: filled in per layer)
        lo[dim_u] = base_x >> 3;
        lo[dim_v] = base_y >> 3;
        lo[dim_s] = 0;
        const int lo_a = a == 0 ? lo[0] : (a == 1 ? lo[1] : lo[2]), lo_u = ua == 0 ? lo[0] : lo[1], lo_v = va == 1 ? lo[1] : lo[2];
        static_assert(TH == 32 || !CHAIN, "chained passes: four brick layers per tile along both plane axes");
        int c0;
        if (dim_s == a) { // same axis: the same tile of the pass before, layer by layer
            c0 = (lo_v >> 2) * link.in_tiles_x + (lo_u >> 2);
            dep_n0 = link.in_down ? link.in_layer0 + 1 : 1 - link.in_layer0;
            dep_n1 = link.in_down ? -1 : 1;
        } else {
            // the tile's four bricks along a, [lo_a, lo_a + 4): the last of them in the pass before's order
            dep_n0 = (link.in_down ? link.in_layer0 - lo_a : lo_a + 3 - link.in_layer0) + 1;
            if (dim_s == ua) { c0 = (lo_v >> 2) * link.in_tiles_x; dep_c1 = 1; }
            else { c0 = lo_u >> 2; dep_c1 = link.in_tiles_x; }
        }
        dep_base = link.prog_in + c0; // (the launch's first pass: in_G = 0 — it needs nothing of the words it is pointed at, its own)
        asm volatile("" : "+v"(dep_c1), "+v"(dep_n0), "+v"(dep_n1), "+v"(dep_G), "+v"(dep_epoch));
        asm volatile("" : "+v"(dep_base), "+v"(prog_word));
    }
    auto depends_on = [&](int g, uint32_t& needed) -> const uint32_t* {
        const int L = layer_of(min(g, G - 1));
        needed = (uint32_t) min(max(dep_n0 + dep_n1 * L, 0), dep_G); // (bricks past the volume's end belong to no layer of the pass before)
        return dep_base + dep_c1 * (L >> 2);
    };
    auto wait_for_layer = [&](int g) { // blocking form: the tile's first two layers
        uint32_t needed = 0;
        const uint32_t* const src = depends_on(g, needed);
        if (needed > 0) (void) sweep_poll_progress(src, dep_epoch, needed, q.error, q.give_up_ticks);
    };
    uint4 lv_next = make_uint4(0, 0, 0, 0);
    if constexpr (CHAIN) { // the tile's first two layers: ONE wave waits for the pass before (a poll costs the CU's other workgroups issue slots)
        if (wave == 0) { wait_for_layer(0); wait_for_layer(1); }
        __syncthreads();
    }
    if (LVS && wave < NWC) {
        *piece_lds = load_layer(0);
        lv_next = load_layer(1);
    }

    const uint32_t epoch = q.epoch & 0xffffu, tag = epoch << 16;
    const int RW = TH * hx + T * hy, RWS = RW * NSW;
    const uint32_t rec_slice = (uint32_t) (n_tiles * RWS); // words per slice
    __syncthreads(); // planes, flags and the first brick layer are in LDS
    // Staggered start (SweepParams::stagger_ns): the tile's lag behind its upstream neighbours, taken up front
    if (q.stagger_ns > 0 && !(q.debug & 1)) {
        const int hops = (q.sx != 0 ? ui : 0) + (q.sy != 0 ? uj : 0);
        const unsigned long long until = wall_clock64() + (unsigned long long) hops * (unsigned long long) q.stagger_ns / 10ull; // (100 MHz)
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    if (q.give_up_ticks == 0 && ticket == 0 && threadIdx.x == 0) atomicOr(q.error, 1); // (tunable sweep_timeout_ms < 0, a test hook: "the
                                                                                       // first tile gave up" — what a starved device reports after its timeout)
    const bool stamping = (q.debug & 2) != 0 && q.stamps != nullptr && threadIdx.x == 0;
    if (stamping) q.stamps[4 * tile_lin + (CHAIN ? 1 : 0)] = wall_clock64(); // (CHAIN: [0] is the workgroup's arrival, in front of its wait for the pass before)

    constexpr int HW = kSweepHandoffWaves;
    if (wave >= NWC && wave < NWC + HW) {
        // =================================================== the hand-off wave(s) ================================================
        // A tile publishes E_x = its hx columns and E_y = its hy rows on the side AWAY from the light, words [row][column of
        // E_x] then [row of E_y][column]; its halo is the same cells of the three upstream neighbours.
        const int e0x = q.sx > 0 ? 0 : T - hx, e0y = q.sy > 0 ? 0 : TH - hy;
        int pub_cell[HC];      // LDS cell of the word this lane publishes (chunk h: word 64 h + lane), < 0: none
        bool hal_on[HC];       // this lane fetches halo word 64 h + lane
        int hal_dst[HC];
        uint32_t hal_src[HC];  // neighbour tile * RW + word
#pragma unroll
        for (int h = 0; h < HC; ++h) {
            const int w = h * 64 + lane;
            {
                int cx = 0, cy = 0;
                const bool on = w < RW;
                if (w < TH * hx) { cy = w / max(hx, 1); cx = e0x + (w - cy * hx); }
                else { const int m = w - TH * hx; cy = e0y + m / T; cx = m % T; }
                pub_cell[h] = on ? (ox + cx) * CS + oy + cy : -1;
            }
            const int nx = TH * hx, ny = T * hy, nc = hx * hy;
            int ntx = tile_x, nty = tile_y, word = 0, cxh = 0, cyh = 0; // neighbour tile, its word, the halo cell in tile coordinates
            bool on = false;
            if (w < nx) { const int row = w / max(hx, 1), kx = w - row * hx; ntx += q.sx; word = row * hx + kx; cxh = (q.sx > 0 ? T : -hx) + kx; cyh = row; on = true; }
            else if (w < nx + ny) { const int m = w - nx, ky = m / T, col = m - ky * T; nty += q.sy; word = nx + ky * T + col; cxh = col; cyh = (q.sy > 0 ? TH : -hy) + ky; on = true; }
            else if (w < nx + ny + nc) {
                const int m = w - nx - ny, ky = m / max(hx, 1), kx = m - ky * hx;
                ntx += q.sx; nty += q.sy;
                word = (e0y + ky) * hx + kx;
                cxh = (q.sx > 0 ? T : -hx) + kx; cyh = (q.sy > 0 ? TH : -hy) + ky;
                on = true;
            }
            // (a neighbour's pixels beyond the buffer are not handed over: their cells hold the border colour from the start)
            on = on && (unsigned) ntx < (unsigned) p.tiles_x && (unsigned) nty < (unsigned) p.tiles_y &&
                 (unsigned) (base_x + cxh) < (unsigned) p.W && (unsigned) (base_y + cyh) < (unsigned) p.H;
            if (q.debug & 1) on = false;
            hal_on[h] = on;
            hal_dst[h] = (ox + cxh) * CS + oy + cyh;
            hal_src[h] = on ? (uint32_t) ((nty * p.tiles_x + ntx) * RWS + word) : 0u;
        }
        // RREC: stream r's halo cells, the same construction with stream r's geometry; their words were written by the launch
        // before this one (tag r_epoch), tile by tile in the same [slice][tile][word] layout
        bool rh_on[HC];
        int rh_dst[HC];
        uint32_t rh_src[HC];
        const int r_RW = TH * q.r_hx + T * q.r_hy;
        const uint32_t r_rec_slice = (uint32_t) (n_tiles * r_RW);
        if constexpr (RREC) {
            const int rhx = q.r_hx, rhy = q.r_hy;
            const int re0y = q.r_sy > 0 ? 0 : TH - rhy;
#pragma unroll
            for (int h = 0; h < HC; ++h) {
                const int w = h * 64 + lane;
                const int nx = TH * rhx, ny = T * rhy, nc = rhx * rhy;
                int ntx = tile_x, nty = tile_y, word = 0, cxh = 0, cyh = 0;
                bool on = false;
                if (w < nx) { const int row = w / max(rhx, 1), kx = w - row * rhx; ntx += q.r_sx; word = row * rhx + kx; cxh = (q.r_sx > 0 ? T : -rhx) + kx; cyh = row; on = true; }
                else if (w < nx + ny) { const int m = w - nx, ky = m / T, col = m - ky * T; nty += q.r_sy; word = nx + ky * T + col; cxh = col; cyh = (q.r_sy > 0 ? TH : -rhy) + ky; on = true; }
                else if (w < nx + ny + nc) {
                    const int m = w - nx - ny, ky = m / max(rhx, 1), kx = m - ky * rhx;
                    ntx += q.r_sx; nty += q.r_sy;
                    word = (re0y + ky) * rhx + kx;
                    cxh = (q.r_sx > 0 ? T : -rhx) + kx; cyh = (q.r_sy > 0 ? TH : -rhy) + ky;
                    on = true;
                }
                on = on && (unsigned) ntx < (unsigned) p.tiles_x && (unsigned) nty < (unsigned) p.tiles_y &&
                     (unsigned) (base_x + cxh) < (unsigned) p.W && (unsigned) (base_y + cyh) < (unsigned) p.H;
                rh_on[h] = on;
                rh_dst[h] = (ox + cxh) * CS + oy + cyh;
                rh_src[h] = on ? (uint32_t) ((nty * p.tiles_x + ntx) * r_RW + word) : 0u;
            }
        }
        auto handoff = [&](auto pub_c, auto con_c) {
        constexpr bool PUB = decltype(pub_c)::value, CON = decltype(con_c)::value;
        using RecW = std::conditional_t<F32, uint64_t, uint32_t>;
        RecW hreg[RING][HC][NSW], rreg[RING][RREC ? HC : 1];
        // (every lane loads: the ones without a halo word read word 0 of the slice — a branch around a load whose result is
        // consumed slices later would make the compiler drain every request in flight at the join)
        auto request_halo = [&](int s, auto slot_c) { // the neighbours' slice s
            constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
            for (int h = 0; h < HC; ++h)
#pragma unroll
                for (int sw = 0; sw < NSW; ++sw) hreg[SLOT][h][sw] = sweep_load_word((const RecW*) q.rec[0] + ((uint32_t) s * rec_slice + hal_src[h] + (uint32_t) (sw * RW)));
            if constexpr (RREC) {
#pragma unroll
                for (int h = 0; h < HC; ++h) rreg[SLOT][h] = sweep_load_word((const RecW*) q.rec[1] + ((uint32_t) s * r_rec_slice + rh_src[h]));
            }
        };
        sweep_each_const([&](auto sl) {
            constexpr int SL = decltype(sl)::value;
            if constexpr (SL < PF && CON) request_halo(SL, sl); // (n >= 8 > PF)
        }, std::make_integer_sequence<int, RING>{});

        lds_barrier(); // (the loader's first factor slice is in LDS)
        __builtin_amdgcn_s_setprio(3); // (its few instructions go first: what it publishes is what the neighbours wait for)
        auto group = [&](int g, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
            sweep_each_const([&](auto kc) {
                constexpr int K8 = decltype(kc)::value, CUR = K8 & 1;
                const int s = g * 8 + K8;
                // (a pass that starts inside a brick layer: the slice before its first real one leaves the initial plane behind)
                bool reinit = false;
                if constexpr (FIRST) reinit = K8 + 1 == q.reinit_slice;
                // the boundary cells of the slice before this one, which the compute waves finished at the last barrier
                if (PUB && s > 0) {
                    RecW* const rec = (RecW*) q.rec[0] + ((uint32_t) (s - 1) * rec_slice + (uint32_t) (tile_lin * RWS));
#pragma unroll
                    for (int h = 0; h < HC; ++h)
                        if (pub_cell[h] >= 0) {
                            if constexpr (F32) {
#pragma unroll
                                for (int si = 0; si < NSH; ++si) {
                                    sweep_store_word(rec + (si * RW + h * 64 + lane), ((uint64_t) epoch << 32) | __float_as_uint(plane(CUR, si)[pub_cell[h]]));
                                }
                            } else {
                                uint32_t w = tag;
#pragma unroll
                                for (int si = 0; si < NS; ++si) w |= ((uint32_t) (plane(CUR, si)[pub_cell[h]] * 255.0f + 0.5f) & 255u) << (8 * si); // (v = code / 255: back to the code)
                                sweep_store_word(rec + (h * 64 + lane), (RecW) w);
                            }
                        }
                }
                // the upstream neighbours' cells of THIS slice into the halo of the plane the compute waves are building
                if constexpr (CON && !(LAST && K8 == 7)) {
#pragma unroll
                    for (int h = 0; h < HC; ++h) {
                        if constexpr (F32) {
#pragma unroll
                            for (int si = 0; si < NSH; ++si) {
                                uint64_t w = hreg[K8][h][si];
                                if (hal_on[h] && (uint32_t) (w >> 32) != epoch)
                                    w = sweep_poll((const uint64_t*) q.rec[0] + ((uint32_t) s * rec_slice + hal_src[h] + (uint32_t) (si * RW)), epoch, q.error, q.give_up_ticks);
                                if (hal_on[h]) plane(CUR ^ 1, si)[hal_dst[h]] = reinit ? stream(si).init_value : __uint_as_float((uint32_t) w);
                            }
                        } else {
                            uint32_t w = (uint32_t) hreg[K8][h][0];
                            if (hal_on[h] && (w >> 16) != epoch) w = sweep_poll((const uint32_t*) q.rec[0] + ((uint32_t) s * rec_slice + hal_src[h]), epoch, q.error, q.give_up_ticks);
                            if (hal_on[h]) {
#pragma unroll
                                for (int si = 0; si < (RREC ? 1 : NS); ++si) plane(CUR ^ 1, si)[hal_dst[h]] = reinit ? stream(si).init_value : decode_u8((w >> (8 * si)) & 255u);
                            }
                        }
                    }
                    if constexpr (RREC) { // the removed light's cells: published long ago (a word that is not there is an error)
#pragma unroll
                        for (int h = 0; h < HC; ++h) {
                            const RecW w = rreg[K8][h];
                            if (rh_on[h]) {
                                if ((uint32_t) (w >> (F32 ? 32 : 16)) != (q.r_epoch & 0xffffu)) atomicOr(q.error, 4);
                                plane(CUR ^ 1, 1)[rh_dst[h]] = reinit ? stream(1).init_value : (F32 ? __uint_as_float((uint32_t) w) : decode_u8((uint32_t) w & 255u));
                            }
                        }
                    }
                }
                if constexpr (CON && (!LAST || K8 + PF <= 6)) request_halo(s + PF, std::integral_constant<int, (K8 + PF) & 7>{});
                lds_barrier();
            }, std::make_integer_sequence<int, 8>{});
        };
        if (q.reinit_slice > 0) group(0, std::true_type{}, std::false_type{}); // (G >= 2: the launcher's check)
        for (int g = q.reinit_slice > 0 ? 1 : 0; g < G - 1; ++g) group(g, std::false_type{}, std::false_type{});
        group(G - 1, std::false_type{}, std::true_type{});
        };
        if constexpr (HW == 1) handoff(std::true_type{}, std::true_type{});
        else if (wave == NWC) handoff(std::true_type{}, std::false_type{});
        else handoff(std::false_type{}, std::true_type{});
    } else if (wave >= NWC + HW) {
        // ================================================= the factor loader =====================================================
        // Block-compact hand-over (ChunkStream::fs_*): per slice group each of the tile's 2 x 2 occlusion blocks of a stream
        // is 8 slices x 1 KiB, found through the block's rank among the pass's live blocks — in the cache entry being filled
        // or read, or in the scratch store beyond the entry's capacity; a flagged-empty block (k_occ_flags) was never
        // computed: its factor 1 - 0 comes from a page of ones. One wave per stream, one LDS-DMA per block slice, FS - 1
        // slices ahead of the compute waves; per slice the wave waits until the NEXT slice's 4 loads have landed, then joins
        // the barrier behind which that slice is read. (Past the last slice the last one is requested again, into slots already
        // consumed: the count of loads in flight stays what the waits assume.)
        constexpr int L = NB; // loads per slice
        constexpr int A = FS - 1;
        const int lsi = NS > 1 ? wave - (NWC + HW) : 0;
        const ChunkStream& st = stream(lsi);
        const int32_t* const ranks = sslot + lsi * NB * G;
        const uint8_t* src[NB]; // of the slice requested next
        uint32_t step[NB];
        auto rebase = [&](int zg) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int32_t slot = ranks[b * G + zg];
                const bool one = slot < 0;
                const float* const base = one ? p.ones : ((uint32_t) slot < st.fs_cap ? st.fs_keep + (size_t) (uint32_t) slot * 2048 : st.fs_spill + (size_t) ((uint32_t) slot - st.fs_cap) * 2048);
                src[b] = (const uint8_t*) base + lane * 16;
                step[b] = one ? 0u : 1024u;
            }
        };
        const uint32_t ring_lds = (uint32_t) (uintptr_t) fring + (uint32_t) (lsi * NB * kSweepFBlock * 4); // (a flat LDS address: its low half is the LDS byte address)
        int req = 0;       // slice requested next
        int req_slot = 0;  // its ring slot
        auto request = [&]() {
            {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t) req_slot * (uint32_t) (kFSlot * 4));
#pragma unroll
                for (int b = 0; b < NB; ++b) sweep_dma_block(src[b], dst + (uint32_t) (b * kSweepFBlock * 4));
            }
            req_slot = req_slot + 1 == FS ? 0 : req_slot + 1;
            if (req + 1 < n) {
                ++req;
                if ((req & 7) == 0) rebase(req >> 3);
                else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) src[b] += step[b];
                }
            }
        };
        // EARLY: the compute waves read slice s + 1's factors at the end of slice s, in front of the barrier: a slice has to have
        // landed one barrier earlier
        constexpr int AW = EARLY ? A - 2 : A - 1;
        static_assert(AW >= 1, "at least one slice of factor loads stays in flight");
        auto landed = [&]() { // everything but the last AW slices' loads
            sweep_wait_loads<AW * L>();
        };
        __builtin_amdgcn_s_setprio(3); // (like the hand-off wave: few instructions, and every other wave waits for them at the barrier)
        // CHAIN: this wave — a prefetcher seven slices ahead of everybody, with time to spare — also keeps the barrier behind slice 2
        // of a group until the pass before has written back the brick layer the compute waves load in slice 3. The progress word is
        // asked for a group ahead, like the wave's other loads in inline assembly (none of its waits is the compiler's): eight
        // slices' requests are issued and waited for behind it, and vmcnt retires in order — it has landed when it is looked at.
        // Measured placements (profiles/r06_sweep_chain.txt): in the compute waves a compiler-visible load made the compiler drain
        // the write-back stores two slices after their issue (+ 6 % per slice); in the hand-off consumer its ring of halo requests
        // retires behind the extra load (+ 1.2 us per hop); in the publisher a warm reset was 1.88 ms against 1.80 - 1.83 here.
        uint32_t dep_needed = 0, dep_word = 0;
        const uint32_t* dep_src = depends_on(2, dep_needed);
        constexpr bool DEP = CHAIN;
        const bool dep_mine = DEP && lsi == 0;
        if (dep_mine) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dep_word) : "v"(dep_src) : "memory");
        rebase(0);
        for (int t = 0; t < A; ++t) request();
        landed(); // slice 0
        lds_barrier();
        for (int s = 0; s < n; ++s) {
            request(); // slice s + A, into the slot slice s - 1 was read from
            landed();  // slice s + 1
            if constexpr (DEP) {
                if (dep_mine && (s & 7) == 2 && (s >> 3) < G - 1) {
                    asm volatile("" : "+v"(dep_word)); // (written behind the compiler's back, eight slices ago)
                    if (dep_needed > 0 && ((dep_word >> 16) != dep_epoch || (dep_word & 0xffffu) < dep_needed))
                        (void) sweep_poll_progress(dep_src, dep_epoch, dep_needed, q.error, q.give_up_ticks);
                    dep_src = depends_on((s >> 3) + 3, dep_needed); // the next group's
                    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dep_word) : "v"(dep_src) : "memory");
                }
            }
            lds_barrier();
        }
        sweep_wait_loads<0>();
    } else {
        // =================================================== the compute waves ===================================================
        const int c = lane & 31, r0 = (wave * 2 + (lane >> 5)) * R;
        const int px = base_x + c;
        const bool in_x = px < p.W;
        bool in_pl[R];            // inside the buffer (D3D drops the overhanging threads' writes)
        int own[R];               // LDS index of the pixel inside a plane
        int tap[NS][R];           // LDS index of its first previous-slice tap
        float wfx[NS];
        v2f wfy[NS];
        uint32_t lv_at[R];        // byte of its voxel in a staged layer, slice row 0
        uint32_t own_idx[R];      // pixel inside a buffer plane
        bool bad = false;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int r = r0 + k, py = base_y + r;
            in_pl[k] = in_x && py < p.H;
            own[k] = (ox + c) * CS + oy + r;
            own_idx[k] = in_pl[k] ? (uint32_t) (py * p.W + px) : 0u;
            const uint32_t lb = (uint32_t) ((r >> 3) * 4 + (c >> 3)) * (uint32_t) LVB;
            if (AXIS == 0) lv_at[k] = lb + (uint32_t) (r & 7) * 64u + (uint32_t) (c & 7) * 8u;
            else if (AXIS == 1) lv_at[k] = lb + (uint32_t) (r & 7) * 64u + (uint32_t) (c & 7);
            else lv_at[k] = lb + (uint32_t) (r & 7) * 8u + (uint32_t) (c & 7);
        }
        static_assert(R == 2, "a lane's two rows share a brick (r0 is even)");
        lv_at[1] = lv_at[0] + (AXIS == 2 ? 8u : 64u); // (said so that the second voxel's address is the first one's plus an immediate)
        constexpr uint32_t kLvStep = AXIS == 0 ? 1u : (AXIS == 1 ? 8u : 64u);
        uint32_t lv_voxel[R] = {0, 0}; // F32: the pixel's voxel in the bricked light volume, slice 0 of the volume (elements)
        if constexpr (F32) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int py = base_y + r0 + k;
                if (AXIS == 0) lv_voxel[k] = brick_off_y(px, p.lv_bnx) + brick_off_z(py, p.lv_bnxy);      // (u, v) = (y, z)
                else if (AXIS == 1) lv_voxel[k] = brick_off_x(px) + brick_off_z(py, p.lv_bnxy);           // (x, z)
                else lv_voxel[k] = brick_off_x(px) + brick_off_y(py, p.lv_bnx);                           // (x, y)
            }
        }
        // F32: L of this slice straight into the light volume (:123-126 / ChangeDirLightShader.usf:152-154): returnless fp32 atomic
        // adds — nothing waits for them, and a lane is its voxel's only writer; (LV + La) - Lr as two adds rounds like the expression
        auto lv_update_f32 = [&](int s_idx, const v2f (&l)[NS]) {
            const int j = p.j0 + s_idx * p.dir;
            const uint32_t so = AXIS == 0 ? brick_off_x(j) : (AXIS == 1 ? brick_off_y(j, p.lv_bnx) : brick_off_z(j, p.lv_bnxy));
            float* const lv = (float*) p.light;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float la = k == 0 ? l[0].x : l[0].y;
                if constexpr (MODE == PASS_CHANGE) {
                    const float lr = k == 0 ? l[NS - 1].x : l[NS - 1].y;
                    if (in_pl[k] && fabsf(la - lr) > 1e-3f) {
                        unsafeAtomicAdd(lv + (lv_voxel[k] + so), la);
                        unsafeAtomicAdd(lv + (lv_voxel[k] + so), -lr);
                    }
                } else {
                    if (in_pl[k] && fabsf(la) > 1e-3f) unsafeAtomicAdd(lv + (lv_voxel[k] + so), la * p.b_added);
                }
            }
        };
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
                const int gsx = (RREC && si == 1) ? q.r_sx : q.sx, ghx = (RREC && si == 1) ? q.r_hx : hx;
                bad = bad || (gsx >= 0 ? (lo < 0 || hi > ghx) : (lo < -ghx || hi > 0));
            }
            wfx[si] = fx;
            float fy2[R];
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
                    const int gsy = (RREC && si == 1) ? q.r_sy : q.sy, ghy = (RREC && si == 1) ? q.r_hy : hy;
                    bad = bad || (gsy >= 0 ? (lo < 0 || hi > ghy) : (lo < -ghy || hi > 0));
                }
                fy2[k] = fy;
                tap[si][k] = own[k] + (in_pl[k] ? ix * CS + iy : 0);
            }
            wfy[si].x = fy2[0]; wfy[si].y = fy2[1];
        }
        if (bad) { // (a lane whose taps are not where the host said reads its own cell instead: wrong values, flagged, in bounds)
            atomicOr(q.error, 2);
#pragma unroll
            for (int si = 0; si < NS; ++si)
#pragma unroll
                for (int k = 0; k < R; ++k) tap[si][k] = own[k];
        }

        // ---- occlusion factors: out of the LDS ring the loader wave fills (one block slice = 16 x 16 floats, blocks 272 apart) ----
        const float* const f_lane = fring + (((r0 >> 4) << 1) | (c >> 4)) * kSweepFBlock + (r0 & 15) * 16 + (c & 15);
        lds_barrier();  // (the loader's first factor slice is in LDS)

        const float thresh = 1e-3f;
        // The light-volume update of a slice (:123-126 / ChangeDirLightShader.usf:152-154) runs one slice late, between the
        // next slice's LDS reads and their use: nothing else depends on it, so it fills the time the taps are in flight.
        v2f lv_l[NS];         // L of the slice whose voxels are still to be updated
        uint8_t* lv_prev = lvt; // where its voxels are: layer buffer + slice row
#pragma unroll
        for (int si = 0; si < NS; ++si) lv_l[si] = (v2f) 0.0f;
        auto light_volume_update = [&](const uint32_t (&code_old)[R]) {
            v2f lv_old;
            lv_old.x = (float) code_old[0]; lv_old.y = (float) code_old[1];
            lv_old = decode2(lv_old);
            v2f nv, d;
            if constexpr (MODE == PASS_ADD2) {
                // light a's update (AddDirLightShader.usf:123-126), the result through the volume's format, then light r's on it
                const v2f qa = quantize2(fma2(lv_l[0], (v2f) p.b_added, lv_old));
                v2f mid;
                mid.x = fabsf(lv_l[0].x) > thresh ? qa.x : (float) code_old[0];
                mid.y = fabsf(lv_l[0].y) > thresh ? qa.y : (float) code_old[1];
                const v2f qb = quantize2(fma2(lv_l[1], (v2f) p.b_added2, decode2(mid)));
                const bool w0 = fabsf(lv_l[1].x) > thresh, w1 = fabsf(lv_l[1].y) > thresh;
                if (in_pl[0]) lv_prev[lv_at[0]] = (uint8_t) (uint32_t) (w0 ? qb.x : mid.x);
                if (in_pl[1]) lv_prev[lv_at[1]] = (uint8_t) (uint32_t) (w1 ? qb.y : mid.y);
                return;
            }
            if constexpr (MODE != PASS_CHANGE) { nv = fma2(lv_l[0], (v2f) p.b_added, lv_old); d = lv_l[0]; } // (l * +-1 is exact: the fused form rounds once, like lv + l * b)
            else { d = lv_l[0] - lv_l[NS - 1]; nv = (lv_old + lv_l[0]) - lv_l[NS - 1]; }
            const v2f qn = quantize2_unfloored(nv);
            const bool w0 = fabsf(d.x) > thresh && in_pl[0], w1 = fabsf(d.y) > thresh && in_pl[1];
            lv_prev[lv_at[0]] = (uint8_t) (w0 ? (uint32_t) qn.x : code_old[0]);
            lv_prev[lv_at[1]] = (uint8_t) (w1 ? (uint32_t) qn.y : code_old[1]);
        };
        // EARLY: read at the end of the slice before
        v2f fac_n[NS];
        uint32_t code_n[R] = {0, 0};
#pragma unroll
        for (int si = 0; si < NS; ++si) fac_n[si] = (v2f) 1.0f;
        if constexpr (EARLY) {
#pragma unroll
            for (int si = 0; si < NS; ++si) { fac_n[si].x = f_lane[si * NB * kSweepFBlock]; fac_n[si].y = f_lane[si * NB * kSweepFBlock + 16]; }
        }
        auto group = [&](int g, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
            uint8_t* const lv_layer = lvt + (g % 3) * kLvBuf;
            sweep_each_const([&](auto kc) {
                constexpr int K8 = decltype(kc)::value, CUR = K8 & 1;
                if constexpr (CHAIN && K8 == 0) {
                    // The wave's vector-memory operations of the group before, in order: [slice 1: the write-back of layer g - 2] [slice 3:
                    // the two loads of layer g + 1]. At most the two loads stay in flight: the write-back (seven slices old) has left the
                    // wave (gfx9 counts loads and stores in issue order); the barrier behind this slice collects the compute waves and
                    // slice 1 publishes it. (Whether the pass before has written back the layer loaded in slice 3 is the factor
                    // loader's business: it holds the barrier behind slice 2 until it has.)
                    if (g > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                if constexpr (CHAIN && K8 == 1) { // "g - 1 layers written back" (in front of this slice's own write-back)
                    if (g > 1 && wave == 0 && lane == 0) sweep_store_word(prog_word, tag | (uint32_t) (g - 1));
                }
                if constexpr (LVS && K8 == 1) { // (the last voxels of the layer before were updated in slice 0 of this group)
                    if (g > 0) write_back_layer(g - 1);
                }
                if constexpr (LVS && K8 == 3 && !LAST) { // the next layer: loaded eight slices ago, first used four barriers from now
                    *(uint4*) ((uint8_t*) piece_lds + ((g + 1) % 3) * kLvBuf) = lv_next;
                    lv_next = load_layer(g + 2);
                }
                // LDS reads: the voxels of the slice before, this slice's taps
                uint32_t code_old[R] = {0, 0};
                if constexpr (LVS) {
#pragma unroll
                    for (int k = 0; k < R; ++k) code_old[k] = EARLY_LV ? code_n[k] : (uint32_t) lv_prev[lv_at[k]];
                }
                // per stream and row: the taps' two columns, each (row iy, row iy + 1)
                v2f ca[NS][R], cb[NS][R], fac[NS];
                const float* const f_at = f_lane + (K8 % FS) * kFSlot;
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    if constexpr (EARLY) fac[si] = fac_n[si];
                    else { fac[si].x = f_at[si * NB * kSweepFBlock]; fac[si].y = f_at[si * NB * kSweepFBlock + 16]; }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const float* const pt = plane(CUR, si) + tap[si][k];
                        ca[si][k].x = pt[0]; ca[si][k].y = pt[1];
                        cb[si][k].x = pt[CS]; cb[si][k].y = pt[CS + 1];
                    }
                }
                if constexpr (LVS)
                    if (K8 > 0 || g > 0) light_volume_update(code_old);
                // this slice, operation by operation over the streams: a slice is one dependent chain per stream, and the two
                // chains of a Change are independent — side by side they fill each other's issue gaps
                v2f pval[NS], xa[NS], qc[NS];
                // previous slice, bilinear with border colour (AddDirLightShader.usf:81-82): along x for (top, bottom) at once,
                // then along y; times 1 - CurrentSample (:117)
#pragma unroll
                for (int si = 0; si < NS; ++si)
#pragma unroll
                    for (int k = 0; k < R; ++k) cb[si][k] = cb[si][k] - ca[si][k];
#pragma unroll
                for (int si = 0; si < NS; ++si)
#pragma unroll
                    for (int k = 0; k < R; ++k) ca[si][k] = fma2((v2f) wfx[si], cb[si][k], ca[si][k]);
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    const float d0 = ca[si][0].y - ca[si][0].x, d1 = ca[si][1].y - ca[si][1].x;
                    xa[si].x = __builtin_fmaf(wfy[si].x, d0, ca[si][0].x);
                    xa[si].y = __builtin_fmaf(wfy[si].y, d1, ca[si][1].x);
                }
#pragma unroll
                for (int si = 0; si < NS; ++si) lv_l[si] = xa[si] * fac[si];
                if constexpr (F32) { // a float buffer returns what was written (:120); the light volume takes L at once
                    if constexpr (LV) {
                        bool real = true; // (not the slices a ragged pass is padded with: in front of a downward pass,
                                          // SweepParams::reinit_slice, behind an upward one, SweepParams::n_real)
                        if constexpr (FIRST) real = K8 >= q.reinit_slice;
                        if constexpr (LAST) real = g * 8 + K8 < q.n_real;
                        if (real) lv_update_f32(g * 8 + K8, lv_l);
                    }
#pragma unroll
                    for (int si = 0; si < NS; ++si) pval[si] = lv_l[si];
                } else {
                // WriteBuffer[PixelLoc] = L (:120), as a read of it returns it: quantize_u8, decode_u8f
#pragma unroll
                for (int si = 0; si < NS; ++si) { qc[si].x = __builtin_amdgcn_fmed3f(lv_l[si].x, 0.0f, 1.0f); qc[si].y = __builtin_amdgcn_fmed3f(lv_l[si].y, 0.0f, 1.0f); }
#pragma unroll
                for (int si = 0; si < NS; ++si) qc[si] = qc[si] * (v2f) 255.0f;
#pragma unroll
                for (int si = 0; si < NS; ++si) qc[si] = qc[si] + (v2f) 0.5f;
#pragma unroll
                for (int si = 0; si < NS; ++si) { qc[si].x = __builtin_floorf(qc[si].x); qc[si].y = __builtin_floorf(qc[si].y); }
#pragma unroll
                for (int si = 0; si < NS; ++si) pval[si] = decode2(qc[si]);
                }
                if constexpr (FIRST) { // the slice before the pass's first real one (a pass that starts inside a brick layer) hands on the
                                       // initial plane, whatever the slices in front of the volume made of it (SweepParams::reinit_slice)
                    if (K8 + 1 == q.reinit_slice) {
#pragma unroll
                        for (int si = 0; si < NS; ++si) pval[si] = (v2f) stream(si).init_value;
                    }
                }
#pragma unroll
                for (int si = 0; si < NS; ++si) {
                    const ChunkStream& st = stream(si);
                    plane(CUR ^ 1, si)[own[0]] = in_pl[0] ? pval[si].x : st.border_light;
                    plane(CUR ^ 1, si)[own[1]] = in_pl[1] ? pval[si].y : st.border_light;
                }
                lv_prev = lv_layer + (uint32_t) (down ? 7 - K8 : K8) * kLvStep;
                if constexpr (LVS && (FIRST || LAST)) { // the slices a ragged pass is padded with leave the padding voxels as they are
                    bool pad = false;               // (their factors were never computed: k_light_occlusion walks the volume's own slices)
                    if constexpr (FIRST) pad = K8 < q.reinit_slice;
                    if constexpr (LAST) pad = g * 8 + K8 >= q.n_real;
                    if (pad) {
#pragma unroll
                        for (int si = 0; si < NS; ++si) lv_l[si] = (v2f) 0.0f;
                    }
                }
                if constexpr (EARLY) { // the next slice's factors (landed a barrier ago)
                    const float* const f_nx = f_lane + ((K8 + 1) % FS) * kFSlot;
#pragma unroll
                    for (int si = 0; si < NS; ++si) { fac_n[si].x = f_nx[si * NB * kSweepFBlock]; fac_n[si].y = f_nx[si * NB * kSweepFBlock + 16]; }
                }
                if constexpr (EARLY_LV && LVS) { // the voxels this slice's L goes to
#pragma unroll
                    for (int k = 0; k < R; ++k) code_n[k] = lv_prev[lv_at[k]];
                }
                if constexpr (LV && LAST && K8 == 7 && !CHAIN) { // the state the next span starts from (a chained pass is a whole pass: nothing continues it)
#pragma unroll
                    for (int si = 0; si < NS; ++si) {
                        if (in_pl[0]) stream(si).plane_out[own_idx[0]] = pval[si].x;
                        if (in_pl[1]) stream(si).plane_out[own_idx[1]] = pval[si].y;
                    }
                }
                lds_barrier();
            }, std::make_integer_sequence<int, 8>{});
        };
        __builtin_amdgcn_s_setprio(2); // (ahead of any occlusion workgroup that shares the CU: this loop is one dependent chain)
        if (q.reinit_slice > 0) group(0, std::true_type{}, std::false_type{});
        for (int g = q.reinit_slice > 0 ? 1 : 0; g < G - 1; ++g) {
            group(g, std::false_type{}, std::false_type{});
            if (g == 7 && stamping && !CHAIN) q.stamps[4 * tile_lin + 1] = wall_clock64();
        }
        group(G - 1, std::false_type{}, std::true_type{});
        __builtin_amdgcn_s_setprio(0);
        if (stamping) q.stamps[4 * tile_lin + 2] = wall_clock64();
        if constexpr (LVS) { // the last slice's voxels, then the last layer
            uint32_t code_old[R];
#pragma unroll
            for (int k = 0; k < R; ++k) code_old[k] = EARLY_LV ? code_n[k] : (uint32_t) lv_prev[lv_at[k]];
            light_volume_update(code_old);
        }
    }

    __syncthreads(); // the last slice's voxels are in the layer buffers
    if (LVS && wave < NWC) write_back_layer(G - 1);
    if constexpr (CHAIN) { // every layer of this tile is in memory: say so
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) sweep_store_word(prog_word, tag | (uint32_t) G);
    }
    if (stamping) q.stamps[4 * tile_lin + 3] = wall_clock64();

    // ---- the last tile to finish re-arms the tickets for the next launch -----------------------------------------------------
    if (threadIdx.x == 0) {
        const int done = atomicAdd(q.ticket + 1, 1);
        if (done == (CHAIN ? link.total_tiles : n_tiles) - 1) {
            atomicExch(q.ticket + 1, 0);
            atomicExch(q.ticket, 0);
        }
    }
}

template <int MODE, int AXIS, int PF, int HC, bool RREC, int LFMT, int TH>
__global__ __launch_bounds__(sweep_threads(MODE, TH), (TH == 16 && HC <= 3) ? 4 : 1) void k_light_sweep(const ChunkParams p, const SweepParams q)
{
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(q.ticket, 1);
    __syncthreads();
    sweep_tile<MODE, AXIS, PF, HC, RREC, LFMT, TH, false>(p, q, SweepLink{}, __builtin_amdgcn_readfirstlane(s_ticket));
}

// Up to kSweepChainMax consecutive passes of an operator in ONE launch (SweepChainArgs): tickets run through the passes in order —
// every tile of pass i in front of every tile of pass i + 1 — and a workgroup becomes whatever tile its ticket says, of whatever
// axis that pass runs along (the three bodies side by side: the launch's registers and LDS are the largest body's).
template <int MODE, int PF, int HC, int TH>
__global__ __launch_bounds__(sweep_threads(MODE, TH), 1) void k_light_sweep_chain(const SweepChainArgs)
{
    // The argument block is read where it lies, in the kernel-argument segment (constant address space: scalar loads), through a
    // pointer — a by-value struct indexed with the pass number would be copied to scratch first (3 KB per lane, every field a
    // vector register). It is the kernel's only explicit argument: offset 0 of the segment.
    typedef const __attribute__((address_space(4))) SweepChainArgs* KernArgs;
    const KernArgs c = (KernArgs) __builtin_amdgcn_kernarg_segment_ptr();
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(c->pass[0].q.ticket, 1);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(s_ticket);
    int pi = 0;
#pragma unroll
    for (int k = 1; k < kSweepChainMax; ++k)
        if (k < c->n && t >= c->pass[k].link.ticket0) pi = k;
    // The pass's parameter blocks as VALUES (loaded once, here): read through the pointer inside the slice loops they would be
    // fetched again behind every barrier — a scalar-memory round trip per slice in the hand-off waves, measured as +15 % per slice
    const ChunkParams p = *(const ChunkParams*) &c->pass[pi].p;
    const SweepParams q = *(const SweepParams*) &c->pass[pi].q;
    const SweepLink link = *(const SweepLink*) &c->pass[pi].link;
    const int local = t - link.ticket0;
    if (p.axis == 0) sweep_tile<MODE, 0, PF, HC, false, FMT_U8, TH, true>(p, q, link, local);
    else if (p.axis == 1) sweep_tile<MODE, 1, PF, HC, false, FMT_U8, TH, true>(p, q, link, local);
    else sweep_tile<MODE, 2, PF, HC, false, FMT_U8, TH, true>(p, q, link, local);
}

template <int MODE, int AXIS, int PF, int HC, int TH, bool RREC = false, int LFMT = FMT_U8>
static hipError_t launch_sweep5(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    static std::atomic<uint64_t> attr_done{0};
    if (const hipError_t e = allow_big_lds(k_light_sweep<MODE, AXIS, PF, HC, RREC, LFMT, TH>, attr_done, 159 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_light_sweep<MODE, AXIS, PF, HC, RREC, LFMT, TH>), dim3(p.tiles_x * p.tiles_y), dim3(sweep_threads(MODE, TH)), sweep_lds_bytes(MODE, p.n_steps, LFMT, TH), s, p, q);
    return hipGetLastError();
}
// Requests run two slices ahead of their use (three for the six-chunk records, whose ring of three slices is what fits the
// registers): distances 3, 4 and 6 lost to 2 in every measurement of rounds 3 and 4 and are not instantiated.
template <int MODE, int AXIS, int TH>
static hipError_t launch_sweep4(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    const int hc = sweep_halo_chunks(q.hx, q.hy, TH);
    if (q.lv_f32) { // float light volumes: Add, fused Change, planes; up to three words per lane and stream (sweep_fit)
        if constexpr (MODE == PASS_ADD || MODE == PASS_CHANGE || MODE == PASS_PLANES) {
            if (q.r_from_records) {
                if constexpr (MODE == PASS_CHANGE) {
                    if (std::max(hc, sweep_halo_chunks(q.r_hx, q.r_hy, TH)) <= 3) return launch_sweep5<MODE, AXIS, 2, 3, TH, true, FMT_F32>(p, q, s);
                }
                return hipErrorInvalidConfiguration;
            }
            if (hc <= 2) return launch_sweep5<MODE, AXIS, 2, 2, TH, false, FMT_F32>(p, q, s);
            if (hc <= 3) return launch_sweep5<MODE, AXIS, 2, 3, TH, false, FMT_F32>(p, q, s);
        }
        return hipErrorInvalidConfiguration;
    }
    if constexpr (MODE == PASS_CHANGE) {
        if (q.r_from_records) { // (sweep_fit: both streams' words fit three per lane)
            const int hc2 = std::max(hc, sweep_halo_chunks(q.r_hx, q.r_hy, TH));
            if (hc2 <= 3) return launch_sweep5<MODE, AXIS, 2, 3, TH, true>(p, q, s);
            if (hc2 <= 6) return launch_sweep5<MODE, AXIS, 3, 6, TH, true>(p, q, s);
            return hipErrorInvalidConfiguration;
        }
    }
    if constexpr (TH == 16) { // (a 32 x 16 tile with a reach of one texel hands 49 words on: one per lane)
        if (hc <= 1) return launch_sweep5<MODE, AXIS, 2, 1, TH>(p, q, s);
    }
    if (hc <= 2) return launch_sweep5<MODE, AXIS, 2, 2, TH>(p, q, s);
    if (hc <= 3) return launch_sweep5<MODE, AXIS, 2, 3, TH>(p, q, s);
    if (hc <= 6) return launch_sweep5<MODE, AXIS, 3, 6, TH>(p, q, s); // (six words per lane and stream: a ring of three slices is what fits the registers)
    return hipErrorInvalidConfiguration; // (sweep_fit rules these out)
}
template <int MODE, int TH>
hipError_t launch_sweep_unit(const ChunkParams& p, const SweepParams& q, hipStream_t s)
{
    return p.axis == 0 ? launch_sweep4<MODE, 0, TH>(p, q, s) : (p.axis == 1 ? launch_sweep4<MODE, 1, TH>(p, q, s) : launch_sweep4<MODE, 2, TH>(p, q, s));
}

// the chained form: every pass of the launch with the hand-off geometry of the widest (HC chunks of 64 words per slice)
template <int MODE, int TH>
hipError_t launch_sweep_chain_unit(const SweepChainArgs& c, hipStream_t s)
{
    if constexpr (MODE != PASS_ADD && MODE != PASS_CHANGE) return hipErrorInvalidConfiguration;
    else {
        int hc = 0, grid = 0;
        size_t lds = 0;
        for (int k = 0; k < c.n; ++k) {
            hc = std::max(hc, sweep_halo_chunks(c.pass[k].q.hx, c.pass[k].q.hy, TH));
            grid += c.pass[k].p.tiles_x * c.pass[k].p.tiles_y;
            lds = std::max(lds, sweep_lds_bytes(MODE, c.pass[k].p.n_steps, FMT_U8, TH));
        }
        // ONE workgroup per CU (a one-stream tile takes 80 KiB: two would fit): a tile of the next pass that is resident beside a
        // tile of this one spends the whole pass polling for it (measured: a warm reset 2.94 ms against 1.92) — it is to arrive
        // when this pass's tile retires
        lds = std::max<size_t>(lds, 84 * 1024);
        auto go = [&](auto kernel) -> hipError_t {
            static std::atomic<uint64_t> attr_done{0};
            if (const hipError_t e = allow_big_lds(kernel, attr_done, 159 * 1024); e != hipSuccess) return e;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(sweep_threads(MODE, TH)), lds, s, c);
            return hipGetLastError();
        };
        if (hc <= 2) return go(k_light_sweep_chain<MODE, 2, 2, TH>);
        if (hc <= 3) return go(k_light_sweep_chain<MODE, 2, 3, TH>);
        return hipErrorInvalidConfiguration; // (the host chains passes of up to three chunks)
    }
}

#ifdef TBRM_SWEEP_UNIT_MODE
template hipError_t launch_sweep_unit<TBRM_SWEEP_UNIT_MODE, TBRM_SWEEP_UNIT_TH>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_chain_unit<TBRM_SWEEP_UNIT_MODE, TBRM_SWEEP_UNIT_TH>(const SweepChainArgs&, hipStream_t);
#else
// (no unit named: every mode and tile height in this one translation unit — what a plain `hipcc -c` of this file builds)
template hipError_t launch_sweep_unit<PASS_ADD, 32>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_CHANGE, 32>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_ADD2, 32>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_PLANES, 32>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_ADD, 16>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_CHANGE, 16>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_ADD2, 16>(const ChunkParams&, const SweepParams&, hipStream_t);
template hipError_t launch_sweep_unit<PASS_PLANES, 16>(const ChunkParams&, const SweepParams&, hipStream_t);
#endif

} // namespace tbrm
