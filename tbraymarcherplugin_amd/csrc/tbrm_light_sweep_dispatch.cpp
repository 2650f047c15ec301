// tbrm_light_sweep_dispatch.cpp — the host-side entry of k_light_sweep: checks a launch's shape and hands it to the translation
// unit that holds its mode and tile height (tbrm_light_sweep.hip is compiled once per pair: tbraymarcherplugin_amd/build.py).
#include "tbrm_light_sweep.h"

namespace tbrm {

// 32 x 32 tiles. The 32 x 16 form (two workgroups per CU) is a build variant (TBRM_BUILD_VARIANTS=1 builds the library with
// -DTBRM_SWEEP_TILE_ROWS=16): measured in round 5, the free tile is 17 % faster per slice, but 16 more hops at 3.5 - 4.3 instead
// of 2.7 - 3.1 us each make every pass slower (profiles/r05_sweep_variants_tile_rows_loader_depth.txt).
int sweep_tile_rows() { return TBRM_SWEEP_TILE_ROWS; }

// advances every tile through the span (j0, n_steps) in one launch; mode PASS_ADD, PASS_CHANGE, PASS_ADD2 or PASS_PLANES, the
// span whole brick layers of the light volume, the occlusion factors handed over block-compact; q.tile_rows: the tiles' height
// (p.tiles_y counts tiles of that height)
hipError_t launch_light_sweep(const ChunkParams& p, const SweepParams& q, int mode, hipStream_t s)
{
    if (p.n_steps <= 0 || p.tiles_x <= 0 || p.tiles_y <= 0) return hipSuccess;
    const bool aligned = (p.n_steps & 7) == 0 && (p.j0 & 7) == (p.dir > 0 ? 0 : 7) && p.occ_phase == 0 && p.n_steps <= sweep_max_slices();
    if (!aligned || !p.compact || !p.ones || !p.a.fs_slot || (sweep_two_streams(mode) && !p.r.fs_slot)) return hipErrorInvalidConfiguration;
    if (q.r_from_records && (mode != PASS_CHANGE || !q.rec[1])) return hipErrorInvalidConfiguration;
    if (q.reinit_slice < 0 || q.reinit_slice > 7 || (q.reinit_slice > 0 && p.n_steps < 16)) return hipErrorInvalidConfiguration;
    if (q.n_real > p.n_steps || q.n_real <= p.n_steps - 8) return hipErrorInvalidConfiguration; // (padding: less than one brick layer)
    if (q.tile_rows != TBRM_SWEEP_TILE_ROWS) return hipErrorInvalidConfiguration;
    if (p.tiles_x != (p.W + kSweepTile - 1) / kSweepTile || p.tiles_y != (p.H + q.tile_rows - 1) / q.tile_rows) return hipErrorInvalidConfiguration;
    constexpr int TH = TBRM_SWEEP_TILE_ROWS;
    if (mode == PASS_ADD) return launch_sweep_unit<PASS_ADD, TH>(p, q, s);
    if (mode == PASS_CHANGE) return launch_sweep_unit<PASS_CHANGE, TH>(p, q, s);
    if (mode == PASS_ADD2) return launch_sweep_unit<PASS_ADD2, TH>(p, q, s);
    if (mode == PASS_PLANES) return launch_sweep_unit<PASS_PLANES, TH>(p, q, s);
    return hipErrorInvalidConfiguration;
}

// Several passes in one launch (tbrm_internal.h SweepChainArgs): every pass checked like a launch of its own
hipError_t launch_light_sweep_chain(const SweepChainArgs& c, int mode, hipStream_t s)
{
    if (c.n < 1 || c.n > kSweepChainMax || (mode != PASS_ADD && mode != PASS_CHANGE)) return hipErrorInvalidConfiguration;
    int ticket0 = 0;
    for (int k = 0; k < c.n; ++k) {
        const ChunkParams& p = c.pass[k].p;
        const SweepParams& q = c.pass[k].q;
        if (p.n_steps <= 0 || p.tiles_x <= 0 || p.tiles_y <= 0) return hipErrorInvalidConfiguration;
        const bool aligned = (p.n_steps & 7) == 0 && (p.j0 & 7) == (p.dir > 0 ? 0 : 7) && p.occ_phase == 0 && p.n_steps <= sweep_max_slices();
        if (!aligned || !p.compact || !p.ones || !p.a.fs_slot || (mode == PASS_CHANGE && !p.r.fs_slot)) return hipErrorInvalidConfiguration;
        if (q.r_from_records || q.lv_f32 || (q.debug & 1)) return hipErrorInvalidConfiguration;
        if (q.reinit_slice < 0 || q.reinit_slice > 7 || (q.reinit_slice > 0 && p.n_steps < 16)) return hipErrorInvalidConfiguration;
        if (q.n_real > p.n_steps || q.n_real <= p.n_steps - 8) return hipErrorInvalidConfiguration;
        if (q.tile_rows != TBRM_SWEEP_TILE_ROWS) return hipErrorInvalidConfiguration;
        if (p.tiles_x != (p.W + kSweepTile - 1) / kSweepTile || p.tiles_y != (p.H + q.tile_rows - 1) / q.tile_rows) return hipErrorInvalidConfiguration;
        const SweepLink& l = c.pass[k].link;
        if (!l.prog_out || !l.prog_in || l.ticket0 != ticket0 || (k == 0) != (l.in_G == 0) || q.ticket != c.pass[0].q.ticket) return hipErrorInvalidConfiguration;
        ticket0 += p.tiles_x * p.tiles_y;
    }
    for (int k = 0; k < c.n; ++k)
        if (c.pass[k].link.total_tiles != ticket0) return hipErrorInvalidConfiguration;
    constexpr int TH = TBRM_SWEEP_TILE_ROWS;
    return mode == PASS_ADD ? launch_sweep_chain_unit<PASS_ADD, TH>(c, s) : launch_sweep_chain_unit<PASS_CHANGE, TH>(c, s);
}

} // namespace tbrm
