// tbrm_internal.h — parameter blocks passed by value from the host C-ABI layer (tbrm_api.cpp) to the gfx950
// kernels (tbrm_kernels.hip). Plain PODs; field meanings follow the reference shader uniforms they replace
// (AddDirLightShader.usf:15-66, ChangeDirLightShader.usf:15-72, M_Raymarch material parameters).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tbrm {

enum : int { FMT_U8 = 0, FMT_U16 = 1, FMT_F32 = 2 };
enum : int { ADDR_WRAP = 0, ADDR_CLAMP = 1, ADDR_BORDER = 2 };

// what a propagation pass does with its light stream(s): Add (one light, stream a), Change (a added, r removed,
// ChangeDirLightShader.usf), or two lights added in one pass (a then r: AddDirLightShader.usf twice, sharing the slice loop).
// PASS_CHANGE_ONE is an occlusion mode only: ONE stream with the Change shader's rules (no uvw == saturate(uvw) guard) — the
// added light of a Change whose removed light's factors come from the factor cache (tbrm_resources.h).
// PASS_PLANES is a sweep mode only: one stream propagated, the light volume untouched — what the launch leaves behind is its
// hand-off records (SweepParams::r_from_records of the launch after it).
enum : int { PASS_ADD = 0, PASS_CHANGE = 1, PASS_ADD2 = 2, PASS_CHANGE_ONE = 3, PASS_PLANES = 5 };

constexpr int kBrick = 8;      // empty-space-skipping brick edge in voxels
constexpr int kBrickShift = 3;

struct VolumeDev {
    const void* data; // bricked layout (tbrm_device_sampling.h)
    int nx, ny, nz;
    int fmt;
    int bnx;          // bricks along x
    int bnxy;         // bricks per z-layer of bricks (bnx * bny)
    int wrap_layer;   // slab-resident volumes: the z brick layer that is not at its own place but at `wrap_shift` voxels
    int wrap_shift;   // further along z (the copy that wrap addressing reaches from the other end); wrap_layer < 0: none
};

struct WindowDev { // WindowingParameters float4 (VolumeInfo.h:49-52)
    float center, width, low_cutoff, high_cutoff;
    // the host's licence to divide by `width` without the IEEE division sequence (tbrm_device_math.h tf_position_fast): RN(1 / width),
    // and whether the window qualifies (else 0: the division). Not part of any cache key: the results are the same bits either way.
    float inv_width;
    int fast_div;
};

// One light stream of one axis pass (what LightingShaders.cpp:100-124 computes per axis).
struct PropStream {
    float border_light;  // ReadBufferSampler border colour
    float off_u, off_v;  // PrevPixelOffset
    float uvw_off[3];    // UVWOffset
    float step100;       // StepSize * VOLUME_DENSITY
    const void* read;    // ReadBuffer
    void* write;         // WriteBuffer
};

struct PropParams {
    VolumeDev data;
    float data_border; // VolumeSampler border colour
    const float4* tf;  // 256 texels, fp16-rounded values held as fp32
    WindowDev win;
    void* light;       // ALightVolume (bricked)
    int lv_dims[3];
    int lv_bnx, lv_bnxy;
    int lv_fmt;        // FMT_U8 or FMT_F32 (buffers share it)
    float cc[3], cd[3];// LocalClippingCenter / LocalClippingDirection
    int clip_mode;     // 0: plane provably leaves every sample weight at exactly 1; 1: general
    int axis;          // 0,1,2: permutation (LightingShaderUtils.cpp:227-249)
    int td[3];         // TransposedDimensions
    int loop;          // Loop (slice index along the axis)
    int row_block0, row_blocks; // 16-row blocks of the slice plane this launch covers (slab-partitioned passes; else all)
    float b_added;     // +1 / -1 (Add only)
    PropStream a;      // the added light (Add: the only stream)
    PropStream r;      // the removed light (Change only)
};

// One light stream of a chunked propagation pass (Add: the added light; Change: a = added, r = removed).
struct ChunkStream {
    float border_light;
    float off_u, off_v;
    float uvw_off[3];
    float step100;
    float init_value;       // what the cleared read/write buffers decode to (first chunk)
    const float* plane_in;  // propagated light after the previous chunk (W x H floats); unused in the first chunk
    float* plane_out;       // propagated light after this chunk
    const float* occ_base;  // chain: the allocation holding this stream's occlusion planes. Its first 4 KiB are a page of ones: the copy
                            // source for flagged-empty blocks, which keeps the copies per wave uniform
    uint32_t occ_off;       // chain: float index, relative to occ_base, of the occlusion plane of the chunk's first slice
    const uint8_t* occ_flags; // chain: this stream's empty-block flags from the slice group holding the chunk's first slice on:
                            // [slice group][block y][block x]; null: none (the two streams of a jointly computed pass share one array)
    float* occ_next;        // occlusion launch: where the span's factors 1 - CurrentSample go, [span slices][H][W]
    // Block-compact hand-over of the occlusion factors (k_light_occlusion writes, k_light_sweep reads; ChunkParams::compact):
    // the live 16 x 16 x 8 block of rank k in the pass's ascending work list holds its factors as [slice 8][row 16][column 16]
    // floats at fs_keep + 2048 k while k < fs_cap (a cache entry: tbrm_resources.h FactorEntry) and at
    // fs_spill + 2048 (k - fs_cap) beyond (the handle's scratch store; fs_cap = 0: nothing is kept).
    float* fs_keep;
    float* fs_spill;
    uint32_t fs_cap;
    const int32_t* fs_slot; // sweep: rank of every block of the pass, [slice group][block y][block x]; -1: flagged empty (factor 1)
};

// Parameters of the chunked propagation kernels (tbrm_light_kernels.hip). One struct serves the per-pass
// launches (k_occ_flags, k_occ_compact), the occlusion launch of a span (j0 / n_steps = the span) and the chain launch
// of a chunk (j0 / n_steps = the chunk).
struct ChunkParams {
    VolumeDev data;
    float data_border;
    const float4* tf;
    WindowDev win;
    void* light;            // bricked
    int lv_dims[3];
    int lv_bnx, lv_bnxy;
    float cc[3], cd[3];
    int clip_mode;
    int axis;
    int W, H;               // TD.X, TD.Y
    int dir;                // +-1
    int j0, n_steps;        // first slice and number of slices of this launch (a span for the occlusion, a chunk for the chain)
    int first_chunk;        // chain: the pass starts here (windows start from the cleared buffers' value)
    int tiles_x, tiles_y;   // chain: 32x32 tiles of the slice plane this launch advances (tiles_y rows from tile_row0 on)
    int tile_row0;          // chain: first tile row (slab-partitioned passes run only the rows of their slab; else 0)
    int roi_by0, roi_by1;   // occlusion: block rows [roi_by0, roi_by1) can be read by those tiles; the rest is never computed
    int dx_lo, dx_hi, dy_lo, dy_hi; // range of (tap index - pixel index) of the previous-slice bilinear fetch, widened to contain 0
    int rect_planes;        // chain: hulls wider than 56 and at most 48 rows high get 72 x 48 LDS planes (chunk_geometry)
    float b_added;          // Add / two adds: +1 / -1 for stream a
    float b_added2;         // two adds: the same for stream r
    // empty-block hand-off: k_occ_flags marks, once per pass, every occlusion workgroup (16x16 pixels x 8 slices) whose
    // samples can only touch data bricks that map every value to opacity 0. Such a workgroup exits at once and the chain
    // stages the factor 1 - 0 for its pixels from a page of ones instead of the plane stack.
    const uint32_t* empty_bits;   // per data brick (k_brick_empty); used by k_occ_flags only
    const uint8_t* occ_flags;     // occlusion launch without a work list: the span's flags (null: every block is computed)
    int occ_phase;                // chain: index of the chunk's first slice within its slice group (ChunkStream::occ_flags)
    uint8_t* occ_flags_out;       // k_occ_flags: the whole pass, [chunk][slice group][block y][block x]
    int occ_blocks_x, occ_blocks_y, occ_groups; // blocks per plane row / column, slice groups per chunk
    int pass_start, pass_slices, chunk_slices;  // k_occ_flags: first slice, slices in the pass, slices per chunk
    // work list: the non-empty workgroups of each chunk in ascending order (k_occ_compact). The occlusion launch keeps its
    // full grid; workgroup i takes entry i of the list or exits, so the live ones are dealt evenly over the CUs instead
    // of landing wherever the dense part of the volume happens to map.
    const uint32_t* occ_list;     // this chunk's list; null: every workgroup takes its own grid position
    const int* occ_count;         // this chunk's number of list entries
    int occ_grid_cap;             // occlusion launch: at most this many workgroups, each walking several blocks (0: one per block)
    uint32_t* occ_list_out;       // k_occ_compact: the whole pass, [chunk][per-chunk capacity]
    int* occ_count_out;           // k_occ_compact: [chunk]
    int* occ_count_host;          // (one-chunk launches) the same count into pinned host memory as well: no copy behind the kernel
    int32_t* occ_slot_out;        // k_occ_compact: rank of every block in its chunk's list, -1 for the flagged ones (null: not wanted)
    int compact;                  // occlusion launch / sweep: the factors are handed over block-compact (ChunkStream::fs_*)
    const float* ones;            // sweep: 1024 floats of 1.0 (the factor of flagged-empty blocks and of pixels beyond the buffer)
    ChunkStream a, r;
};

// The two axis passes of ONE light sample the data volume at the same positions: UVWOffset = normalize(lightPos) / min(TD)
// whichever axis the pass runs along (LightingShaders.cpp:114-124) — only StepSize differs. A "dual" occlusion launch
// (k_light_occlusion<..., DUAL>) therefore filters, windows and looks the transfer function up ONCE per voxel and emits the
// factor 1 - CurrentSample of BOTH passes (AddDirLightShader.usf:85-117 with the two StepSize * 100 exponents, the logarithm
// of the opacity correction shared). Its work unit is 16 x 16 voxels in x and y times 8 in z, a thread walking z — whatever
// the two pass axes are (ChunkParams describes that virtual pass along z): a block of any pass holds its factors as
// [slice][row][column] with x the column axis unless the pass runs along x (then y), so lanes that span x and y write every
// pass's block-compact store in the same 32-byte segments as the pass's own launch would (measured: looping along x or y
// instead scatters one or both stores over 64 lines per instruction and costs 1.3 - 1.9x the launch).
struct DualPass {
    int axis, start, dir;       // the pass: axis, first slice, direction (tbrm_light_pass)
    int blocks_x, blocks_y;     // its 16 x 16 blocks per plane row / column
    float step100[2];           // StepSize * VOLUME_DENSITY per stream (0: a, 1: r)
    float* fs_keep[2];          // per stream: ChunkStream::fs_keep / fs_spill / fs_cap of THIS pass
    float* fs_spill[2];
    uint32_t fs_cap[2];
    const int32_t* fs_slot;     // rank of every block of this pass (k_occ_compact), shared by the streams; -1: flagged empty
    const uint8_t* flags;       // k_unit_flags: the pass's empty-block flags, [slice group][block y][block x]
};
struct DualOcc {
    int on;
    DualPass pass[2];
};
// k_light_sweep (tbrm_light_sweep.hip): one launch advances every 32x32 tile of the slice plane through a whole span of
// slices. The previous-slice taps of a pass lie on ONE side of the pixel per plane axis (the constant PrevPixelOffset,
// AddDirLightShader.usf:81-82), so a tile depends on at most three neighbours — the ones towards the light — and the tiles
// form a pipeline: tile t runs a few slices behind its upstream neighbours and reads the hx columns / hy rows it needs of
// their slice from a hand-off record in global memory (dwords {16-bit launch tag, UNORM8 code of stream r, of stream a},
// written and polled with relaxed agent-scope atomics: no fences, no kernel boundary, no recomputed halo).
struct SweepParams {
    int sx, sy;             // side of the previous-slice taps along the plane's x / y: +1, -1, 0 (none: the tap is the pixel itself)
    int hx, hy;             // how many columns / rows beyond the tile they reach (both streams)
    // A fused Change whose lights pull opposite ways along a plane axis has no tile order that serves both streams. Its
    // removed light is then propagated FIRST, alone, in its own order (a PASS_PLANES launch into rec[1]); the fused launch
    // runs in the added light's order and takes the removed light's halo from those finished records.
    int r_from_records;     // PASS_CHANGE: stream r's halo comes from rec[1] (tag r_epoch), geometry r_sx .. r_hy
    int r_sx, r_sy, r_hx, r_hy;
    uint32_t r_epoch;
    uint32_t* rec[2];       // [0]: the hand-off records, [slice of the launch][tile][32*hx + 32*hy words]; [1]: r_from_records
    uint32_t epoch;         // this launch's tag, 1 .. 2^16 - 1 (records are not cleared between launches)
    int prefetch;           // slices ahead of their use that the neighbours' records are requested
    int stagger_ns;         // a tile d tiles away from the upstream corner starts d * stagger_ns late: the distance it would
                            // otherwise fall behind by polling (a tile can never make up lag, and every poll that finds nothing
                            // costs a memory round trip: arriving late on purpose is cheaper than arriving early)
    int* ticket;            // [0]: next tile to start (tiles are dealt in upstream-first order: a tile only ever waits for tiles
                            // that started before it), [1]: tiles finished (the last one re-arms both)
    int debug;              // diagnostics (sweep_debug tunable): bit 0 = tiles do not wait for each other (WRONG results: slice time alone);
                            // bit 1 = every tile leaves four time stamps (10 ns units) in `stamps`
    unsigned long long* stamps; // [tile][4]: start of slice 0, end of slice 63, end of the last slice, after the write-back
    int tile_rows;          // height of a tile: 32, or 16 (two workgroups per CU; tbrm_light_sweep.h sweep_tile_rows); a tile is 32 wide
    int lv_f32;             // the light volume (and the planes) are floats: k_light_sweep<..., FMT_F32>, record words are 8-byte {float, launch tag} granules
    int reinit_slice;       // > 0: the launch's first reinit_slice slices lie in front of the volume (a pass that runs downwards from a
                            // depth that is no multiple of 8, padded to whole brick layers): slice reinit_slice - 1 hands on the pass's
                            // initial plane instead of what it computed
    int n_real;             // slices of the launch that lie inside the volume when the padding comes last (an upward pass whose depth is
                            // no multiple of 8): the slices from n_real on leave the light volume alone. n_steps: none are padding
    int* error;             // set when a tile gave up waiting (bit 0) or found its taps outside the halo (bit 1)
    unsigned long long give_up_ticks; // how long a poll waits for a neighbour's word, in 10 ns ticks of wall_clock64 (tunable sweep_timeout_ms)
};

// Several axis passes in ONE sweep launch (k_light_sweep_chain): the tiles of pass i + 1 take their tickets behind every tile of
// pass i, so they start as pass i's tiles retire — the fill of a pass runs under the drain of the one before instead of behind a
// kernel boundary (30 hops x ~2.9 us of every pass are fill / drain at 512^3). What orders the two passes' read-modify-writes of
// the light volume is data, not a boundary: a tile writes its brick layers back with write-through (sc1) stores and then
// publishes "k layers written back" in its progress word {16-bit launch tag, count} (relaxed agent-scope store behind a drained
// vmcnt); a tile of the next pass reads a brick layer (sc1 loads) only after the progress word of the ONE tile of the pass before
// that owns those bricks has reached the layer it needs. Forward progress: a tile only ever waits for tiles with smaller tickets.
struct SweepLink {
    uint32_t* prog_out;        // this pass's progress words, [tile]
    const uint32_t* prog_in;   // the pass before's (the launch's first pass: any valid words — in_G = 0, it needs nothing of them)
    uint32_t in_epoch;         // ... and its launch tag
    int in_axis, in_tiles_x;   // ... its axis and tiles per row
    int in_down, in_layer0, in_G; // ... its direction, first brick layer and number of layers
    int coherent_loads;        // this pass reads its brick layers with sc1 loads: every pass but the launch's first
    int ticket0;               // first ticket of this pass
    int total_tiles;           // tiles of the whole launch (the last one to finish re-arms the tickets)
};
constexpr int kSweepChainMax = 4; // passes per launch (the kernel's arguments hold their parameter blocks: 4 x ~0.6 KB)
struct SweepChainPass {
    ChunkParams p;
    SweepParams q;
    SweepLink link;
};
struct SweepChainArgs {
    int n;
    SweepChainPass pass[kSweepChainMax];
};

struct RayParams {
    VolumeDev data;
    int data_addr_mode; // ADDR_WRAP / ADDR_CLAMP
    const float4* tf;
    WindowDev win;
    const void* light;  // bricked
    int lv_dims[3];
    int lv_bnx, lv_bnxy;
    int lv_fmt;
    float cam_pos[3], fwd[3], right[3], up[3];
    float thx, thy;
    int width, height;  // full framebuffer
    float m[12];        // WorldToLocal rows (3x3 + translation)
    float cc[3], cd[3];
    int clip_mode;      // 0: clip plane provably never clips a sample position; 1: general
    int share_grid;     // light volume has the data volume's size and wrap addressing: taps share their offsets
    int wave_skip;      // k_raymarch_lit takes the empty trips a whole wave shares in one go (pays in large volumes)
    int tile_x0, tile_y0, tile_w, tile_h, row_group_step;
    float steps;
    int jitter_frame;
    const float* depth; // may be null
    float* out;         // tile_w*tile_h*4
    const uint32_t* empty_bits; // one bit per brick; null when skipping is off
    int xcd_rows;               // k_raymarch_lit: rows of pixel blocks per band dealt to one XCD (tunable ray_xcd_rows; 0: launch order)
    const uint2* tab;           // k_raymarch_lit TAB: per axis and texel index -2 .. n + 1 the {voxel offset, brick-index part} of the
                                // addressed texel (x, then y, then z); null: none (slab-resident handles)
    const uint8_t* skip_dist;   // per brick: Chebyshev distance (bricks, capped) to the nearest non-empty brick; null when skipping is off
    int bnx, bny, bnz;  // brick grid
    unsigned long long* sample_counter; // count kernel only
    int lv_wrap_layer, lv_wrap_shift; // VolumeDev::wrap_layer / wrap_shift of the light volume
    int slab_on, slab_z0, slab_z1, slab_dir; // slab stage of the lit march: owned light-volume slices, sweep direction (+1 / -1 / 0: all rays)
    const uint16_t* octree;     // octree march only: the level marched (dense, x fastest)
    int oct_dims[3];            // its dimensions
    float oct_depth0;           // depth of octree level 0 (the z coordinate is rescaled by data depth / this)
};

struct BrickParams {
    VolumeDev data;
    int addr_mode;
    int bnx, bny, bnz;
    float2* minmax;
    int bz0, bz1;     // brick layers whose voxels (and +1 apron) may be read; the others get the range [-inf, +inf] (never empty)
};

struct EmptyParams {
    const float2* minmax;
    int n_bricks;
    WindowDev win;
    const int* alpha_prefix; // 257 entries: # of TF texels j < i with alpha > 0 (or NaN)
    uint32_t* bits;
};

hipError_t launch_shell_transparent(const EmptyParams& p, int bnx, int bny, int bnz, float border, int* flag, hipStream_t s);

constexpr int kSkipDistCap = 32;
struct DistParams {
    const uint32_t* bits; // pass 0: the k_brick_empty bits
    const uint8_t* in;    // later passes: the previous pass (null in pass 0)
    uint8_t* out;
    int bn[3];
    int axis;
};

struct OctreeParams {
    VolumeDev data;        // level 0: the bricked data volume
    const uint16_t* lower; // levels 1..3: the level below
    int lower_dims[3];
    uint16_t* out;
    int dims[3];
};

struct RelayoutParams {
    const void* src;
    void* dst;
    int nx, ny, nz;
    int bnx, bnxy, bnz;
    int elem_bytes;
    int to_bricks; // 1: linear -> bricked (padding voxels are zeroed); 0: bricked -> linear
};

constexpr int kOccSlices = 8;      // slices per occlusion workgroup (kOccDepth in tbrm_light_kernels.hip)
constexpr int kChunkTile = 32;      // core tile edge of the chunked propagation kernel (pixels)
constexpr int kChunkThreads = 1024;
constexpr int kChunkMaxHull = 64;   // T + steps * growth must stay within this

// Process-wide tunables (A/B switches of the measurements in DESIGN.md and of the parity tests). Each starts from the
// environment variable TBRM_<NAME> read ONCE when the library is loaded; afterwards tbrm_set_tunable changes it. No launch
// path reads the environment.
enum Tunable : int {
    TUNE_FORCE_SLICE_KERNEL, // 1: every axis pass takes the reference's one-slice-per-launch structure (k_propagate_slice)
    TUNE_CHUNK_STEPS,        // > 0: only this chunk length is tried (16 / 8 / 4 / 2)
    TUNE_OCC_SLICES,         // slices per occlusion span (0: default)
    TUNE_SPARSE_OCC,         // 0: occlusion blocks that can only see empty bricks are computed like the others
    TUNE_OCC_LIST,           // 0: live occlusion blocks keep their grid position instead of being dealt from a work list
    TUNE_LIGHT_CACHE_MB,     // HBM budget of the factor cache in MiB (0: off, < 0: an eighth of the device's memory): a light's occlusion factors, kept per axis pass
    TUNE_LIGHT_BATCHING,     // tbrm_add_dir_lights: 0 never pair passes, 1 pair when it pays, 2 pair whatever fits
    TUNE_SHARE_GRID,         // 0: the raymarch computes the light volume's tap offsets separately even on a shared grid
    TUNE_RAY_WAVE_SKIP,      // RayParams::wave_skip: 1 on, 0 off, -1 = where the data volume is at least 384 voxels a side
    TUNE_RAY_LANES,          // lanes per ray of k_raymarch_lit: 4, 8, or 0 = by load
    TUNE_CHAIN_FAST_LOOP,    // 0: full, aligned chunks run the generic slice loop too (A/B of the unrolled, branch-free loop)
    TUNE_CHAIN_RECT_PLANES,  // 0: no 72 x 48 LDS planes (a pass with taps two texels wide along x runs 8-slice chunks in square planes)
    TUNE_OCC_OVERLAP,        // workgroups per CU of an occlusion launch that runs beside the previous span's chain (0: never beside it)
    TUNE_LIGHT_SWEEP,        // 0: axis passes never take the pipelined sweep kernel (k_light_sweep); 1: where it applies
    TUNE_SWEEP_PREFETCH,     // slices ahead that a sweep tile requests its neighbours' hand-off records (0: default)
    TUNE_STREAM_PRIORITY,    // priority of a handle's own stream, read when the handle is created: 0 = default, 1 = the highest the
                             // device offers, -1 = the lowest
    TUNE_SWEEP_DEBUG,        // diagnostics. 1: sweep tiles do not wait for each other (WRONG results), 2: per-tile time stamps, 4: host
                             // time per operator phase on stderr, 8 / 16: the occlusion stream skips its waits for the scratch
                             // store / the cache entry to be idle (WRONG results: what gates its start?), 32: no timing events
    TUNE_SWEEP_TIMEOUT_MS,   // how long a sweep tile waits for a neighbour's hand-off word before it gives up and raises the handle's error
                             // word (0: 2 s; < 0: not at all — a test hook: every word that is not there yet fails the launch)
    TUNE_FAST_WINDOW_DIV,    // 0: the kernels always divide by the window's width (IEEE sequence); 1: three fmas where the host vouches
                             // for the window (WindowDev::fast_div) — the same bits
    TUNE_SLAB_SWEEP,         // 1: a slab's share of a pass along z runs as one pipelined sweep instead of the chunked chain (measured at
                             // 1024^3: a tie at 2 and 4 slabs, 6 % slower at 8 — a 128-slice sweep over 1024 tiles is mostly pipeline fill; off)
    TUNE_GPU_TIMING,         // 0: operators do not record the HIP events behind tbrm_last_gpu_time_ms (four markers per step between
                             // dependent kernels: a host that never asks for GPU times need not pay them)
    TUNE_RAY_TABLES,         // 0: k_raymarch_lit computes the data taps' offsets per sample even where its LDS offset tables apply
    TUNE_SWEEP_EPOCH_PRESET, // > 0: a handle's first sweep launch continues from this launch tag (a test hook: the 16-bit tags of the
                             // hand-off records start over after 65535 launches)
    TUNE_OCC_DUAL,           // 1: the two axis passes of a light share ONE occlusion launch where their sampling positions are bit-equal
                             // (DualOcc); 0: one launch per pass
    TUNE_SWEEP_CHAIN,        // consecutive sweep passes of an operator per launch (k_light_sweep_chain): 1 = one launch per pass, up to 4
    TUNE_RAY_XCD_ROWS,       // k_raymarch_lit: rows of pixel blocks per band dealt to one XCD (0: blocks in launch order, i.e. round-robin)
    TUNE_COUNT
};
int tune(Tunable t);

// launchers (tbrm_kernels.hip, tbrm_light_kernels.hip)
hipError_t launch_selftest_decode(float* d_u8, float* d_u16, hipStream_t s);
hipError_t launch_selftest_roundtrip(const float* d_in, float* d_out, size_t n, hipStream_t s);
hipError_t launch_selftest_window_division(const WindowDev& w, unsigned long long* d_mismatches, hipStream_t s);
hipError_t launch_selftest_opacity_correction(float step0, float step1, unsigned long long* d_mismatches, hipStream_t s);
hipError_t launch_relayout(const RelayoutParams& p, hipStream_t s);
size_t chunk_lds_bytes(const ChunkParams& p, int mode, int lv_fmt);
size_t occlusion_lds_bytes(const ChunkParams& p); // dynamic LDS of an occlusion workgroup (the bricks it stages)
constexpr int kPlaneGuard = 4096; // floats of slack on both sides of every plane/occlusion buffer (16-byte row copies overrun rows)
hipError_t launch_occ_flags(const ChunkParams& p, int mode, int n_chunks, hipStream_t s); // + the work lists
hipError_t launch_light_occlusion(const ChunkParams& p, int mode, hipStream_t s, const DualOcc* dual = nullptr);
hipError_t launch_unit_flags(const ChunkParams& pc, const DualOcc& d, hipStream_t s); // + the units' work list (pc: the virtual pass along the third axis)
hipError_t launch_light_chain(const ChunkParams& p, int mode, int lv_fmt, hipStream_t s);
hipError_t launch_light_sweep(const ChunkParams& p, const SweepParams& q, int mode, hipStream_t s);
hipError_t launch_light_sweep_chain(const SweepChainArgs& c, int mode, hipStream_t s); // PASS_ADD / PASS_CHANGE, UNORM8 light volumes
hipError_t launch_fill(void* dst, int fmt, size_t n, float value, hipStream_t s);
hipError_t launch_propagate_slice(const PropParams& p, bool change, hipStream_t s);
hipError_t launch_raymarch(const RayParams& p, hipStream_t s);
hipError_t launch_raymarch_intensity(const RayParams& p, hipStream_t s);
hipError_t launch_raymarch_octree(const RayParams& p, hipStream_t s);
hipError_t launch_octree_level(const OctreeParams& p, bool base, hipStream_t s);
hipError_t launch_count_samples(const RayParams& p, hipStream_t s);
hipError_t launch_brick_minmax(const BrickParams& p, hipStream_t s);
hipError_t launch_brick_empty(const EmptyParams& p, hipStream_t s);
hipError_t launch_brick_dist(const DistParams& p, int addr_mode, hipStream_t s);

} // namespace tbrm
