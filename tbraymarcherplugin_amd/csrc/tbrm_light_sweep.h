// tbrm_light_sweep.h — what k_light_sweep (tbrm_light_sweep.hip, compiled as one translation unit per mode and tile height)
// and the host-side planner (tbrm_light_plan.cpp) have to agree on: tile shape, LDS budget, hand-off record layout. Internal.
#pragma once
#include "tbrm_internal.h"

namespace tbrm {

constexpr int kSweepTile = 32;                         // a tile's width; its height TH is 32 or 16 (sweep_tile_rows)
// LDS plane: 48 columns x (TH + 16) cells (tile + halo <= 14 + guard ring), COLUMN-major — a pixel's two taps of one column
// (rows iy, iy + 1) are neighbours in memory and arrive as one register pair, ready for a packed lerp along x over (top, bottom)
// — with an odd column stride of TH + 17 floats: the 32 columns of a wave's lanes fall into 32 different banks
constexpr int kSweepCols = 48;
constexpr int sweep_col_stride(int th) { return th + 17; }
constexpr int sweep_plane(int th) { return kSweepCols * sweep_col_stride(th); }
constexpr int kSweepLvBrick = 528;                     // bytes per staged light-volume brick: 512 + 16, so that the four bricks
                                                       // under a tile row start 4 banks apart
constexpr int kSweepRing = 8;                          // register ring of requested hand-off words (slices)
constexpr int sweep_compute_waves(int th) { return th / 4; } // a wave: 32 columns x 4 rows (two rows per lane)
// + the hand-off wave + one factor loader per stream (an LDS-DMA costs its wave 60 - 180 cycles of issue: the eight of a
// two-stream slice in one wave took longer than the slice)
constexpr bool sweep_two_streams(int mode) { return mode == PASS_CHANGE || mode == PASS_ADD2; }
#ifndef TBRM_SWEEP_HANDOFF_WAVES
#define TBRM_SWEEP_HANDOFF_WAVES 2
#endif
// Hand-off waves per tile: 1 = one wave publishes, consumes and requests; 2 = a publisher and a consumer (each slice's
// chain of LDS read -> convert -> store and of load -> decode -> LDS write then run side by side instead of one after the other)
constexpr int kSweepHandoffWaves = TBRM_SWEEP_HANDOFF_WAVES;
constexpr int sweep_threads(int mode, int th) { return (sweep_compute_waves(th) + kSweepHandoffWaves + (sweep_two_streams(mode) ? 2 : 1)) * 64; }
constexpr int kSweepFlagGroups = 128;                  // slice groups of a span (1024 slices)
constexpr int kSweepFBlock = 256 + 16;                 // floats per staged 16 x 16 block slice of occlusion factors: the four
                                                       // blocks under a tile start 16 banks apart (a lane pair's columns c, c + 16)

// slots of the LDS ring of factor slices; the loader runs one less ahead. Eight for one and for two streams (round 5: with a
// ring of eight a slice's factors land a barrier earlier and the compute waves read them in front of the barrier instead of
// behind it — k_light_sweep EARLY —: a fused Change's sweeps 0.63 -> 0.61 ms; rounds 3 and 4 ran two-stream passes with four
// slots to leave LDS to occlusion workgroups beside them, which never paid). Build-time switches for A/B builds (tools/build_variant.py):
#ifndef TBRM_SWEEP_TWO_STREAM_SLOTS
#define TBRM_SWEEP_TWO_STREAM_SLOTS 8
#endif
#ifndef TBRM_SWEEP_EARLY_READS
#define TBRM_SWEEP_EARLY_READS 1
#endif
constexpr int sweep_factor_slots(int mode) { return sweep_two_streams(mode) ? TBRM_SWEEP_TWO_STREAM_SLOTS : 8; }

constexpr int sweep_blocks(int th) { return 2 * (th / 16); } // 16 x 16 occlusion blocks under a tile
constexpr int sweep_bricks(int th) { return 4 * (th / 8); }  // light-volume bricks under a tile

inline size_t sweep_lds_bytes(int mode, int slices, int lv_fmt, int th)
{
    const int ns = sweep_two_streams(mode) ? 2 : 1;
    const int groups = (slices + 7) / 8; // (the rank table is as long as the pass: every KiB not taken is the occlusion workgroups')
    // planes, three brick layers (UNORM8 light volumes: a float light volume is updated in place), block ranks, the ring of factor slices
    return (size_t) 2 * ns * sweep_plane(th) * 4 + (lv_fmt == FMT_U8 ? 3 * sweep_bricks(th) * kSweepLvBrick : 0) + (size_t) ns * sweep_blocks(th) * groups * 4 +
           (size_t) sweep_factor_slots(mode) * ns * sweep_blocks(th) * kSweepFBlock * 4;
}

inline int sweep_max_slices() { return 8 * kSweepFlagGroups; }

// hand-off words a tile reads per slice and stream, in chunks of 64 (one per lane of the hand-off wave)
inline int sweep_halo_chunks(int hx, int hy, int th) { return (th * hx + kSweepTile * hy + hx * hy + 63) / 64; }
// words of a tile's hand-off record per slice (and per stream of a float light volume): its hx columns and hy rows away from the light
inline int sweep_record_words(int hx, int hy, int th) { return th * hx + kSweepTile * hy; }


// The tile height the library's sweeps are built with: 32, or — a build variant, measured slower — 16: two workgroups on a CU,
// each a lockstep chain of LDS reads -> arithmetic -> LDS writes -> barrier, for 16 more hops of pipeline fill.
#ifndef TBRM_SWEEP_TILE_ROWS
#define TBRM_SWEEP_TILE_ROWS 32
#endif
static_assert(TBRM_SWEEP_TILE_ROWS == 32 || TBRM_SWEEP_TILE_ROWS == 16, "sweep tiles are 32 x 32 or 32 x 16");
int sweep_tile_rows();

// one translation unit per (mode, tile height): tbrm_light_sweep.hip compiled with -DTBRM_SWEEP_UNIT_MODE / _TH
template <int MODE, int TH>
hipError_t launch_sweep_unit(const ChunkParams& p, const SweepParams& q, hipStream_t s);
template <int MODE, int TH>
hipError_t launch_sweep_chain_unit(const SweepChainArgs& c, hipStream_t s);

} // namespace tbrm
