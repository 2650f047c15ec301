// tbrm_light_sweep_dispatch.cpp — the host-side entry of k_light_sweep: checks a launch's shape and hands it to the translation
// unit that holds its mode and tile height (tbrm_light_sweep.hip is compiled once per pair: tbraymarcherplugin_amd/build.py).
#include "tbrm_light_sweep.h"

namespace tbrm {

int sweep_tile_rows()
{
    const int t = tune(TUNE_SWEEP_ROWS);
    return t == 16 ? 16 : 32; // (32 x 16 measured: the free tile 17 % faster per slice, 16 more hops at 3.5 - 4.3 instead of 2.7 - 3.1 us each: slower)
}

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
    if (q.tile_rows != 16 && q.tile_rows != 32) return hipErrorInvalidConfiguration;
    if (p.tiles_x != (p.W + kSweepTile - 1) / kSweepTile || p.tiles_y != (p.H + q.tile_rows - 1) / q.tile_rows) return hipErrorInvalidConfiguration;
    const bool half = q.tile_rows == 16;
    if (mode == PASS_ADD) return half ? launch_sweep_unit<PASS_ADD, 16>(p, q, s) : launch_sweep_unit<PASS_ADD, 32>(p, q, s);
    if (mode == PASS_CHANGE) return half ? launch_sweep_unit<PASS_CHANGE, 16>(p, q, s) : launch_sweep_unit<PASS_CHANGE, 32>(p, q, s);
    if (mode == PASS_ADD2) return half ? launch_sweep_unit<PASS_ADD2, 16>(p, q, s) : launch_sweep_unit<PASS_ADD2, 32>(p, q, s);
    if (mode == PASS_PLANES) return half ? launch_sweep_unit<PASS_PLANES, 16>(p, q, s) : launch_sweep_unit<PASS_PLANES, 32>(p, q, s);
    return hipErrorInvalidConfiguration;
}

} // namespace tbrm
