// tbrm_light_plan.cpp — host side of the illumination operators, planning: what LightingShaders.cpp:35-326 does on the render
// thread (per light the axis passes, per pass the slice loop) becomes a PassPlan — the pipelined sweep where it applies
// (sweep_fit), else the chunked chain (chunk_fit), else one slice per launch — with its geometry and the factor cache's part in it.
#include "tbrm_light_passes.h"

namespace tbrm_host {

// flags of the events that order the two streams (experiment: TBRM_EVENT_FLAGS=0 creates them with timing, i.e. with a marker of
// their own in the queue at record time)
unsigned event_flags()
{
    static const unsigned f = [] { const char* e = getenv("TBRM_EVENT_FLAGS"); return e && *e ? (unsigned) atoi(e) : (unsigned) hipEventDisableTiming; }();
    return f;
}

// ---- chunked propagation (tbrm_light_kernels.hip) --------------------------------------------------------------

bool force_slice_kernel() { return tune(TUNE_FORCE_SLICE_KERNEL) == 1; }
int chunk_steps_override() { return tune(TUNE_CHUNK_STEPS); }

TapRange prev_tap_range(int size, float off)
{
    TapRange t;
    if (!std::isfinite(off) || size <= 0) return t;
    t.lo = INT32_MAX; t.hi = INT32_MIN;
    for (int c = 0; c < size; ++c) {
        const float u = (((float) (uint32_t) c + 0.5f) / (float) size) + off;
        float x = u * (float) size - 0.5f;
        x = std::fmin(std::fmax(x, -0x1p30f), 0x1p30f);
        const int d = (int) std::floor(x) - c;
        t.lo = std::min(t.lo, d);
        t.hi = std::max(t.hi, d + 1);
    }
    t.ok = std::abs(t.lo) <= 64 && std::abs(t.hi) <= 64;
    return t;
}


// ---- the pipelined sweep (tbrm_light_sweep.hip) ---------------------------------------------------------------------------
// Which side of a pixel the previous-slice taps of NON-ZERO weight lie on along one buffer axis, and how far, over every
// pixel of the axis, with the kernel's own fp32 sequence. side 0: every pixel reads itself alone. ok false: taps on both
// sides (an offset so small that rounding decides the side pixel by pixel) or out of range — the sweep declines.
struct TapSide { int side = 0, reach = 0; bool ok = false; };
static TapSide prev_tap_side(int size, float off)
{
    TapSide t;
    if (!std::isfinite(off) || size <= 0) return t;
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int c = 0; c < size; ++c) {
        const float u = (((float) (uint32_t) c + 0.5f) / (float) size) + off;
        float x = u * (float) size - 0.5f;
        x = std::fmin(std::fmax(x, -0x1p30f), 0x1p30f);
        const float fl = std::floor(x);
        const float f = x - fl;
        const int d = (int) fl - c;
        lo = std::min(lo, d);
        hi = std::max(hi, f != 0.0f ? d + 1 : d);
    }
    if (lo >= 0) { t.side = hi > 0 ? 1 : 0; t.reach = hi; t.ok = true; }
    else if (hi <= 0) { t.side = -1; t.reach = -lo; t.ok = true; }
    return t;
}

// Can the axis pass (one stream: pr == null) run as pipelined sweeps? The tiles' dependency has to point one way per buffer
// axis, the reach has to fit the kernel's LDS planes and the hand-off wave's six words per lane, and the pass has to consist
// of whole brick layers of the light volume. The two lights of a fused Change whose minor components have opposite signs
// pull opposite ways: no tile order serves both, and the pass runs as TWO sweeps (SweepFit::two_way) if each light's reach
// fits the hand-off wave and both fit the planes side by side.
bool sweep_fit(const tbrm_resources* r, const tbrm_light_pass& pa, const tbrm_light_pass* pr, int mode, SweepFit& fit)
{
    if (tune(TUNE_LIGHT_SWEEP) == 0 || force_slice_kernel() || r->resident || r->sweep_failed_bits) return false;
    const bool f32 = r->lv_fmt != FMT_U8; // (float light volumes: k_light_sweep<..., FMT_F32> — one-way passes of up to three words per lane)
    if (mode != PASS_ADD && mode != PASS_CHANGE) return false;
    // (a depth that is no multiple of 8 is padded to whole brick layers: plan_pass_sweep; a downward pass then needs a second
    // layer behind the ragged one)
    if (pa.td[2] % 8 != 0 && pa.dir < 0 && pa.td[2] < 9) return false;
    fit = SweepFit{};
    TapSide side[2][2];
    bool opposite = false;
    int n = 0;
    for (const tbrm_light_pass* q : {&pa, pr}) {
        if (!q) continue;
        const TapSide tx = prev_tap_side(q->td[0], q->prev_pixel_offset[0]), ty = prev_tap_side(q->td[1], q->prev_pixel_offset[1]);
        if (!tx.ok || !ty.ok) return false;
        side[n][0] = tx; side[n][1] = ty;
        ++n;
        opposite = opposite || tx.side * fit.sx < 0 || ty.side * fit.sy < 0;
        if (tx.side) fit.sx = tx.side;
        if (ty.side) fit.sy = ty.side;
        fit.hx = std::max(fit.hx, tx.reach);
        fit.hy = std::max(fit.hy, ty.reach);
    }
    const int th = sweep_tile_rows();
    if (!opposite) return fit.hx <= 14 && fit.hy <= 14 && sweep_halo_chunks(fit.hx, fit.hy, th) <= (f32 ? 3 : 6);
    if (tune(TUNE_LIGHT_SWEEP) == 2) return false; // (diagnostics: such passes take the chain, as before round 3's last week)
    fit.two_way = true;
    fit.sx = side[0][0].side; fit.hx = side[0][0].reach; fit.sy = side[0][1].side; fit.hy = side[0][1].reach;
    fit.r_sx = side[1][0].side; fit.r_hx = side[1][0].reach; fit.r_sy = side[1][1].side; fit.r_hy = side[1][1].reach;
    // the planes hold the tile, a guard ring and both lights' halos: the low sides' larger reach plus the high sides'
    int room[2];
    for (int ax = 0; ax < 2; ++ax) {
        int lo = 0, hi = 0;
        for (int si = 0; si < 2; ++si) (side[si][ax].side < 0 ? lo : hi) = std::max(side[si][ax].side < 0 ? lo : hi, side[si][ax].reach);
        room[ax] = lo + hi;
    }
    return room[0] <= 14 && room[1] <= 14 && sweep_halo_chunks(fit.hx, fit.hy, th) <= (f32 ? 3 : 6) && sweep_halo_chunks(fit.r_hx, fit.r_hy, th) <= (f32 ? 3 : 6);
}

// Why sweep_fit declines a one-stream pass (tbrm_host_plan_light; diagnostics): 0 it does not, 1 previous-slice taps on both sides of
// the pixel along a plane axis or an offset out of range, 2 taps more than 14 texels from the pixel, 3 more hand-off words per slice
// than a lane of the hand-off wave carries, 4 a downward pass over a ragged depth of fewer than nine slices, 5 sweeps are off for
// the handle (tunable, slab-resident, a failed sweep)
int sweep_decline_reason(const tbrm_resources* r, const tbrm_light_pass& pa)
{
    if (tune(TUNE_LIGHT_SWEEP) == 0 || force_slice_kernel() || r->resident || r->sweep_failed_bits) return 5;
    if (pa.td[2] % 8 != 0 && pa.dir < 0 && pa.td[2] < 9) return 4;
    const TapSide tx = prev_tap_side(pa.td[0], pa.prev_pixel_offset[0]), ty = prev_tap_side(pa.td[1], pa.prev_pixel_offset[1]);
    if (!tx.ok || !ty.ok) return 1;
    if (tx.reach > 14 || ty.reach > 14) return 2;
    if (sweep_halo_chunks(tx.reach, ty.reach, sweep_tile_rows()) > (r->lv_fmt != FMT_U8 ? 3 : 6)) return 3;
    return 0;
}

void release_sweep(tbrm_resources* r)
{
    for (auto& rec : r->sweep_rec) { (void) hipFree(rec); rec = nullptr; }
    r->sweep_rec_words = r->sweep_rec1_words = 0;
    (void) hipFree(r->sweep_ticket);
    r->sweep_ticket = nullptr;
    (void) hipFree(r->sweep_prog);
    r->sweep_prog = nullptr;
    r->sweep_prog_stride = 0;
    if (r->sweep_error) (void) hipHostFree(r->sweep_error);
    r->sweep_error = nullptr;
    (void) hipFree(r->sweep_stamps);
    r->sweep_stamps = nullptr;
}

int sweep_check(tbrm_resources* r)
{
    if ((tune(TUNE_SWEEP_DEBUG) & 2) && r->sweep_stamps && r->sweep_stamp_tiles > 0) { // diagnostics: the last launch's timeline
        std::vector<unsigned long long> t((size_t) r->sweep_stamp_tiles * 4);
        if (r->sweep_stamp_chain[0] > 0) { // a chained launch: per pass, when its tiles started and ended (us from the launch's first stamp)
            if (hipMemcpy(t.data(), r->sweep_stamps, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                unsigned long long t0 = ~0ull;
                for (int i = 0; i < r->sweep_stamp_tiles; ++i) t0 = std::min(t0, t[4 * i]);
                int first = 0;
                for (int k = 0; k < 4 && r->sweep_stamp_chain[k] > 0; ++k) {
                    const int nt = r->sweep_stamp_chain[k];
                    double lo[4] = {1e30, 1e30, 1e30, 1e30}, hi[4] = {0, 0, 0, 0}, sum[4] = {0, 0, 0, 0};
                    for (int i = first; i < first + nt; ++i)
                        for (int s = 0; s < 4; ++s) {
                            const double us = (double) (t[4 * i + s] - t0) * 0.01;
                            lo[s] = std::min(lo[s], us); hi[s] = std::max(hi[s], us); sum[s] += us;
                        }
                    fprintf(stderr, "[tbrm chain stamps] pass %d: %d tiles; arrived %.1f / %.1f / %.1f us (first / mean / last), first slice begun %.1f / %.1f / %.1f, last slice done %.1f / %.1f / %.1f, written back %.1f / %.1f / %.1f\n",
                            k, nt, lo[0], sum[0] / nt, hi[0], lo[1], sum[1] / nt, hi[1], lo[2], sum[2] / nt, hi[2], lo[3], sum[3] / nt, hi[3]);
                    first += nt;
                }
            }
        } else if (hipMemcpy(t.data(), r->sweep_stamps, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            const int tx = r->sweep_stamp_tx, ty = r->sweep_stamp_tiles / tx;
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < r->sweep_stamp_tiles; ++i) t0 = std::min(t0, t[4 * i]);
            fprintf(stderr, "[tbrm sweep stamps] %d x %d tiles, upstream side (%d, %d); per hop distance: tiles, mean us of start / slice 63 / last slice / end\n", tx, ty,
                    r->sweep_stamp_sx, r->sweep_stamp_sy);
            std::vector<double> acc((size_t) (tx + ty) * 5, 0.0);
            for (int j = 0; j < ty; ++j)
                for (int i = 0; i < tx; ++i) {
                    const int d = (r->sweep_stamp_sx > 0 ? tx - 1 - i : (r->sweep_stamp_sx < 0 ? i : 0)) + (r->sweep_stamp_sy > 0 ? ty - 1 - j : (r->sweep_stamp_sy < 0 ? j : 0));
                    acc[5 * d] += 1.0;
                    for (int k = 0; k < 4; ++k) acc[5 * d + 1 + k] += (double) (t[4 * (j * tx + i) + k] - t0) * 0.01;
                }
            for (int d = 0; d < tx + ty; ++d)
                if (acc[5 * d] > 0) fprintf(stderr, "  hop %2d: %3.0f tiles  %7.2f %7.2f %7.2f %7.2f\n", d, acc[5 * d], acc[5 * d + 1] / acc[5 * d], acc[5 * d + 2] / acc[5 * d], acc[5 * d + 3] / acc[5 * d], acc[5 * d + 4] / acc[5 * d]);
        }
        r->sweep_stamp_tiles = 0;
    }
    return sweep_failed(r);
}

// The sweep kernels' error word (pinned host memory: visible as soon as the kernel that raised it has completed, without a
// copy). Latched into the handle: from then on the light volume is undefined, and every entry point that waits for the
// handle's stream, and every light operator, says so — until the light volume is defined again (ClearResourceLightVolumes, an
// upload). Sweeps are not used while it stands (sweep_fit).
int sweep_failed(tbrm_resources* r)
{
    if (r->sweep_error && *r->sweep_error != 0) {
        r->sweep_failed_bits |= *r->sweep_error;
        *r->sweep_error = 0;
    }
    const int e = r->sweep_failed_bits;
    if (e == 0) return TBRM_OK;
    return fail(TBRM_ERR_NO_DEVICE, "a light-propagation sweep failed on the device (%s%s%s): the light volume is undefined until it is cleared",
                (e & 1) ? "a tile gave up waiting for its neighbours" : "", (e & 2) ? " previous-slice taps outside the planned halo" : "",
                (e & 4) ? " a removed light's plane records were not there" : "");
}

void sweep_failure_cleared(tbrm_resources* r)
{
    if (r->sweep_failed_bits == 0 && !(r->sweep_error && *r->sweep_error != 0)) return;
    drain_streams_public(r); // (whatever was in flight when it failed may still raise the word)
    if (r->sweep_error) *r->sweep_error = 0;
    r->sweep_failed_bits = 0;
}

// room for the hand-off records of a pass (words1: of the removed light's own sweep, two-way Changes), the tickets and the
// error word
int ensure_sweep(tbrm_resources* r, size_t words, size_t words1, size_t chain_tiles)
{
    if (chain_tiles + 256 > r->sweep_prog_stride && chain_tiles > 0) { // the progress words of chained passes (tagged with their launch: never cleared)
        ++r->sync_calls;
        count_alloc(r, 2, "progress words of chained sweeps");
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->sweep_prog);
        r->sweep_prog = nullptr;
        r->sweep_prog_stride = 0;
        const size_t stride = ((chain_tiles + 63) & ~(size_t) 63) + 256; // (+ slack: a launch's first pass asks for words near its own, needing nothing of them)
        HIP_TRY(hipMalloc((void**) &r->sweep_prog, stride * (kSweepChainMax + 1) * sizeof(uint32_t))); // (+ the table nobody writes)
        HIP_TRY(hipMemsetAsync(r->sweep_prog, 0, stride * (kSweepChainMax + 1) * sizeof(uint32_t), r->stream));
        r->sweep_prog_stride = stride;
    }
    if (!r->sweep_ticket) {
        count_alloc(r, 2, "sweep tickets and error word");
        HIP_TRY(hipMalloc((void**) &r->sweep_ticket, 2 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(r->sweep_ticket, 0, 2 * sizeof(int), r->stream));
        HIP_TRY(hipHostMalloc((void**) &r->sweep_error, sizeof(int), hipHostMallocMapped));
        *r->sweep_error = 0;
    }
    if (words > r->sweep_rec_words) {
        ++r->sync_calls;
        count_alloc(r, 2, "hand-off records (a pass beyond the reserved reach)");
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->sweep_rec[0]);
        r->sweep_rec[0] = nullptr;
        r->sweep_rec_words = 0;
        HIP_TRY(hipMalloc((void**) &r->sweep_rec[0], words * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(r->sweep_rec[0], 0, words * sizeof(uint32_t), r->stream)); // tag 0: no launch
        r->sweep_rec_words = words;
    }
    if (words1 > r->sweep_rec1_words) {
        ++r->sync_calls;
        count_alloc(r, 2, "hand-off records of a two-way Change (beyond the reserved reach)");
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->sweep_rec[1]);
        r->sweep_rec[1] = nullptr;
        r->sweep_rec1_words = 0;
        HIP_TRY(hipMalloc((void**) &r->sweep_rec[1], words1 * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(r->sweep_rec[1], 0, words1 * sizeof(uint32_t), r->stream));
        r->sweep_rec1_words = words1;
    }
    return TBRM_OK;
}

// (launches: how many consecutive launches must not have the tags start over between them)
int next_sweep_epoch(tbrm_resources* r, uint32_t& epoch, uint32_t launches)
{
    if (!r->sweep_epoch_preset_done && tune(TUNE_SWEEP_EPOCH_PRESET) > 0) { // (a test hook: the 16-bit tags start over after 65535 launches)
        r->sweep_epoch = (uint32_t) tune(TUNE_SWEEP_EPOCH_PRESET) & 0xffffu;
        r->sweep_epoch_preset_done = true;
    }
    if (++r->sweep_epoch + (launches - 1) >= (1u << 16)) { // 2^16 launches later: tags start over
        HIP_TRY(hipMemsetAsync(r->sweep_rec[0], 0, r->sweep_rec_words * sizeof(uint32_t), r->stream));
        if (r->sweep_rec[1]) HIP_TRY(hipMemsetAsync(r->sweep_rec[1], 0, r->sweep_rec1_words * sizeof(uint32_t), r->stream));
        if (r->sweep_prog) HIP_TRY(hipMemsetAsync(r->sweep_prog, 0, r->sweep_prog_stride * kSweepChainMax * sizeof(uint32_t), r->stream));
        r->sweep_epoch = 1;
    }
    epoch = r->sweep_epoch;
    return TBRM_OK;
}

float through_light_format(int lv_fmt, float v)
{
    if (lv_fmt != FMT_U8) return v;
    float x = v;
    if (x != x) return 0.0f;
    x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
    const uint32_t c = (uint32_t) (x * 255.0f + 0.5f);
    return (float) c / 255.0f;
}

void fill_chunk_stream(ChunkStream& s, const tbrm_light_pass& p, int lv_fmt)
{
    s.border_light = p.border_light;
    s.off_u = p.prev_pixel_offset[0];
    s.off_v = p.prev_pixel_offset[1];
    for (int c = 0; c < 3; ++c) s.uvw_off[c] = p.uvw_offset[c];
    s.step100 = p.step_size * 100.0f;
    s.init_value = through_light_format(lv_fmt, p.light_alpha); // Clear2DTexture of the read/write buffers
}


// why plan_pass last declined a pass (diagnostics of the slab entry points, which have no fallback)
thread_local const char* g_plan_note = "";
int declined(const char* why) { g_plan_note = why; return TBRM_ERR_UNSUPPORTED; }

// 72 x 48 LDS planes (tbrm_light_chain.h): the kernels that have them
static int rect_planes_for(const tbrm_resources* r, int mode)
{
    return mode == PASS_ADD && r->lv_fmt == FMT_U8 && tune(TUNE_CHAIN_RECT_PLANES) != 0 ? 1 : 0;
}

// Chunk length of a pass (one stream: pr == null, else two) and the tap ranges its windows have to cover: the longest of
// 16/8/4/2 slices whose window (tile + steps * growth) and staged occlusion fit in LDS. false: the chunk kernels decline.
bool chunk_fit(const tbrm_resources* r, const tbrm_light_pass& pa, const tbrm_light_pass* pr, ChunkFit& fit, int mode)
{
    if (mode < 0) mode = pr ? PASS_CHANGE : PASS_ADD; // (PASS_ADD2 has PASS_CHANGE's shapes)
    g_plan_note = "";
    if (force_slice_kernel()) return declined("the force_slice_kernel tunable is set"), false;
    const int W = pa.td[0], H = pa.td[1], D_pass = pa.td[2];
    TapRange tx = prev_tap_range(W, pa.prev_pixel_offset[0]), ty = prev_tap_range(H, pa.prev_pixel_offset[1]);
    if (!tx.ok || !ty.ok) return declined("previous-slice offset out of range"), false;
    if (pr) {
        const TapRange rx = prev_tap_range(W, pr->prev_pixel_offset[0]), ry = prev_tap_range(H, pr->prev_pixel_offset[1]);
        if (!rx.ok || !ry.ok) return declined("previous-slice offset out of range"), false;
        tx.lo = std::min(tx.lo, rx.lo); tx.hi = std::max(tx.hi, rx.hi);
        ty.lo = std::min(ty.lo, ry.lo); ty.hi = std::max(ty.hi, ry.hi);
    }
    // Unsheared windows: a tile keeps its 32x32 pixels for the whole chunk and its window grows towards the light by
    // the tap range per remaining slice (the range is widened to contain 0 so the window always covers the tile).
    tx.lo = std::min(tx.lo, 0); tx.hi = std::max(tx.hi, 0);
    ty.lo = std::min(ty.lo, 0); ty.hi = std::max(ty.hi, 0);
    ChunkParams p{};
    p.dx_lo = tx.lo; p.dx_hi = tx.hi; p.dy_lo = ty.lo; p.dy_hi = ty.hi;
    p.rect_planes = rect_planes_for(r, mode);
    p.dir = pa.dir;
    p.j0 = pa.start;
    const int g = std::max(tx.hi - tx.lo, ty.hi - ty.lo);
    fit = ChunkFit{};
    // With more tiles than CUs every CU works through several tiles per launch: the per-chunk overhead is paid once per
    // round of tiles while the halo work of a long chunk (windows 1.56x the tile on average at 16 slices, 1.27x at 8) is
    // paid by every tile, and 8-slice chunks win — measured for a fused Change: 640^3 5.2 -> 4.9 ms, 1024^3 16.7 -> 15.6,
    // 1536^3 55.3 -> 49.7; at 512^3 (one tile per CU) 16 and 8 tie and 16 halves the launches.
    const bool many_tiles = ceil_div(W, kChunkTile) * ceil_div(H, kChunkTile) > r->n_cus;
    for (int cand : {16, 8, 4, 2}) { // 2: steep secondary passes (taps up to 16 texels from the pixel), still 5x the slice kernel
        if (chunk_steps_override() > 0 && cand != chunk_steps_override()) continue;
        if (cand == 16 && many_tiles && chunk_steps_override() == 0) continue;
        p.n_steps = std::min(cand, D_pass);
        if (kChunkTile + cand * g <= kChunkMaxHull && chunk_lds_bytes(p, mode, r->lv_fmt) <= 156 * 1024) { fit.M = cand; break; }
    }
    if (fit.M <= 0) return declined("the previous-slice taps reach too far for a 2-slice chunk"), false;
    fit.tx = tx;
    fit.ty = ty;
    return true;
}

// Returns TBRM_ERR_UNSUPPORTED (nothing enqueued) when the pass has to take the slice-per-launch path.
// slab: the light-volume z range this handle owns (null: everything).
// Rows of the slice plane (z, when the pass runs along x or y) a slice's previous-slice taps can lie from the pixel:
// what a slab has to fetch from its neighbours after every slice of a slice-per-launch pass. < 0: offsets out of range.
int slice_tap_reach(const tbrm_light_pass& pa, const tbrm_light_pass* pr)
{
    int reach = 0;
    for (const tbrm_light_pass* q : {&pa, pr}) {
        if (!q) continue;
        const TapRange t = prev_tap_range(q->td[1], q->prev_pixel_offset[1]);
        if (!t.ok || !prev_tap_range(q->td[0], q->prev_pixel_offset[0]).ok) return -1;
        reach = std::max({reach, -t.lo, t.hi});
    }
    return reach;
}

// A slab-partitioned pass with the reference's one-slice-per-launch structure (the chunk kernels declined it): "chunk" c is
// slice c of what this handle runs, the planes are the pass's read / write buffers in the light volume's format.
int plan_pass_sliced(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
                     const tbrm_slab& slab, PassPlan& plan)
{
    const int nz = r->lv_dims[2], D_pass = pa.td[2];
    if (slab.z_begin < 0 || slab.z_end > nz || slab.z_begin >= slab.z_end || slab.z_begin % kChunkTile || slab.z_end % kChunkTile || nz % kChunkTile)
        return fail(TBRM_ERR_INVALID_ARG, "slab [%d, %d) of a light volume %d deep: bounds and depth must be multiples of %d", slab.z_begin,
                    slab.z_end, nz, kChunkTile);
    const int reach = slice_tap_reach(pa, pr);
    if (reach < 0) return declined("previous-slice offset out of range");
    plan = PassPlan{};
    plan.sliced = true;
    plan.mode = pr ? PASS_CHANGE : PASS_ADD;
    plan.M = 1;
    plan.p.W = pa.td[0];
    plan.p.H = pa.td[1];
    plan.p.axis = pa.axis;
    plan.dir = pa.dir;
    plan.D = D_pass;
    plan.start = pa.start;
    plan.chunks_of_pass = D_pass;
    PropParams& p = plan.slice_params;
    p = base;
    p.b_added = b_added;
    p.axis = pa.axis;
    for (int c = 0; c < 3; ++c) p.td[c] = pa.td[c];
    fill_stream(p.a, pa);
    if (pr) fill_stream(p.r, *pr);
    p.row_block0 = 0;
    p.row_blocks = 0;
    if (pa.axis == 2) {
        plan.D = slab.z_end - slab.z_begin;
        plan.start = pa.dir > 0 ? slab.z_begin : slab.z_end - 1;
        plan.first_chunk_of_pass = pa.dir > 0 ? slab.z_begin : nz - slab.z_end;
        plan.pass_begins_here = plan.first_chunk_of_pass == 0;
    } else {
        if (reach > slab.z_end - slab.z_begin) return declined("a slice's taps reach beyond the neighbouring slab");
        plan.lateral = true;
        plan.halo_rows = reach;
        p.row_block0 = slab.z_begin / 16;
        p.row_blocks = (slab.z_end - slab.z_begin) / 16;
    }
    plan.n_chunks = plan.D;
    if (r->resident) {
        if (slab.z_begin != r->owned.z_begin || slab.z_end != r->owned.z_end)
            return fail(TBRM_ERR_INVALID_ARG, "a slab-resident handle runs its own slab [%d, %d) only", r->owned.z_begin, r->owned.z_end);
    }
    if (plan.pass_begins_here) { // the buffers start from the light's initial value (LightingShaders.cpp:74-79)
        const size_t npx = (size_t) pa.td[0] * pa.td[1];
        const int ax = pa.axis;
        if (!pr) {
            HIP_TRY(launch_fill(r->d_buf[ax][0], r->lv_fmt, npx, pa.light_alpha, r->stream));
            HIP_TRY(launch_fill(r->d_buf[ax][1], r->lv_fmt, npx, pa.light_alpha, r->stream));
        } else {
            HIP_TRY(launch_fill(r->d_buf[ax][0], r->lv_fmt, npx, pr->light_alpha, r->stream));
            HIP_TRY(launch_fill(r->d_buf[ax][1], r->lv_fmt, npx, pr->light_alpha, r->stream));
            HIP_TRY(launch_fill(r->d_buf[ax][2], r->lv_fmt, npx, pa.light_alpha, r->stream));
            HIP_TRY(launch_fill(r->d_buf[ax][3], r->lv_fmt, npx, pa.light_alpha, r->stream));
        }
    }
    return TBRM_OK;
}

// the read buffer of stream si (0: a, 1: r) before this handle's slice number `boundary` (== n_chunks: what its last slice wrote)
void* sliced_plane(const tbrm_resources* r, const PassPlan& plan, int boundary, int si)
{
    const int j = plan.start + boundary * plan.dir;
    const int e = (j % 2 == 0) ? 0 : 1; // LightingShaders.cpp:149-156
    const int ax = plan.p.axis;
    if (plan.mode == PASS_ADD) return r->d_buf[ax][e];
    return si == 0 ? r->d_buf[ax][2 + e] : r->d_buf[ax][e];
}

SpanRange span_range(const PassPlan& plan, int sp)
{
    const ChunkParams& p = plan.p;
    const int M = plan.M, S = plan.S, D = plan.D;
    SpanRange q;
    q.s0 = sp * S;
    q.sn = std::min(S, D - q.s0);
    q.c0 = q.s0 / M;
    q.c1 = ceil_div(q.s0 + q.sn, M);
    // the chain stages its window in groups of 4 pixels starting at tile_x - n*|dx_lo|: only when that is a multiple of 4
    // does a group never straddle two 16-pixel occlusion blocks (always true for full chunks of 16/8/4 slices)
    q.sparse = plan.sparse;
    for (int cc = q.c0; cc < q.c1; ++cc) q.sparse = q.sparse && (std::min(M, D - cc * M) * -p.dx_lo) % 4 == 0; // else the whole span runs dense
    return q;
}

// The common part of a chunked / swept pass's parameters
static void fill_pass_params(const tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
                             float b_added2, PassPlan& plan)
{
    ChunkParams& p = plan.p;
    const int W = pa.td[0], H = pa.td[1];
    p.data = base.data;
    p.data_border = base.data_border;
    p.tf = base.tf;
    p.win = base.win;
    p.light = base.light;
    for (int c = 0; c < 3; ++c) { p.lv_dims[c] = base.lv_dims[c]; p.cc[c] = base.cc[c]; p.cd[c] = base.cd[c]; }
    p.lv_bnx = base.lv_bnx; p.lv_bnxy = base.lv_bnxy;
    p.clip_mode = base.clip_mode;
    p.axis = pa.axis;
    p.W = W; p.H = H;
    p.dir = pa.dir;
    p.b_added = b_added;
    p.b_added2 = b_added2;
    fill_chunk_stream(p.a, pa, r->lv_fmt);
    if (pr) fill_chunk_stream(p.r, *pr, r->lv_fmt);
    plan.D = pa.td[2];
    plan.start = pa.start;
    plan.dir = pa.dir;
    p.tiles_x = ceil_div(W, kChunkTile);
    p.tiles_y = ceil_div(H, kChunkTile);
    p.tile_row0 = 0;
    p.occ_blocks_x = ceil_div(W, 16);
    p.occ_blocks_y = ceil_div(H, 16);
    p.roi_by0 = 0;
    p.roi_by1 = p.occ_blocks_y;
}

// A whole, unpartitioned pass over a UNORM8 light volume as ONE pipelined sweep (tbrm_light_sweep.hip): the occlusion of the
// whole pass is computed block-compact on the occlusion stream — or comes from the factor cache — and one launch propagates.
// TBRM_ERR_UNSUPPORTED (nothing changed): the pass takes the chunked chain.
// slab (a pass along z of a slab-partitioned operator, round 4): the handle's own slices [z_begin, z_end) as one sweep that starts
// from the planes the slab before handed on (or the pass's initial value) and leaves its last planes for the slab behind.
static int plan_pass_sweep(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
                           PassPlan& plan, int mode, const tbrm_slab* slab = nullptr)
{
    SweepFit sfit;
    if (!sweep_fit(r, pa, pr, mode, sfit)) return TBRM_ERR_UNSUPPORTED;
    if (slab) {
        const int nz = r->lv_dims[2];
        if (pa.axis != 2 || sfit.two_way) return TBRM_ERR_UNSUPPORTED; // (lateral slab passes need a hand-off per slice across handles: the chain)
        if (slab->z_begin < 0 || slab->z_end > nz || slab->z_begin >= slab->z_end || slab->z_begin % kChunkTile || slab->z_end % kChunkTile || nz % kChunkTile)
            return TBRM_ERR_UNSUPPORTED; // (the chain's planner words the error)
    }
    // The pass over the light volume padded to whole brick layers along its axis (the bricked layout has the padding voxels):
    // D slices from `start`, of which the `pad` slices beyond the volume come last when the pass runs upwards — garbage in,
    // garbage out, into voxels nothing reads — and first when it runs downwards, where the last of them hands on the initial
    // plane (SweepParams::reinit_slice).
    const int D = slab ? slab->z_end - slab->z_begin : ceil_div(pa.td[2], 8) * 8, pad = slab ? 0 : D - pa.td[2];
    const int start = slab ? (pa.dir > 0 ? slab->z_begin : slab->z_end - 1) : (pa.dir > 0 ? 0 : D - 1);
    if (D > sweep_max_slices() || tune(TUNE_SPARSE_OCC) == 0 || tune(TUNE_OCC_LIST) == 0) return TBRM_ERR_UNSUPPORTED;
    if (int e = ensure_skipping(r)) return e; // (the work list needs the per-brick emptiness bits)
    if (int e = ensure_occ_stream(r)) return e;
    const bool change = pr != nullptr;
    plan = PassPlan{};
    plan.mode = mode;
    plan.sweep = true;
    fill_pass_params(r, base, pa, pr, b_added, 0.0f, plan);
    ChunkParams& p = plan.p;
    plan.D = D;
    plan.start = start;
    plan.M = plan.S = D;
    plan.n_chunks = plan.n_spans = plan.chunks_of_pass = 1;
    if (slab) { // (the drivers order the slabs by first_chunk_of_pass; chunk = this slab's depth)
        const int nz = r->lv_dims[2], before = pa.dir > 0 ? slab->z_begin : nz - slab->z_end;
        plan.chunks_of_pass = ceil_div(nz, D);
        plan.first_chunk_of_pass = before / D;
        plan.pass_begins_here = before == 0;
    }
    plan.sparse = plan.work_list = true;
    p.occ_groups = D / 8;
    plan.flags_per_group = (size_t) p.occ_blocks_y * p.occ_blocks_x;
    plan.flags_per_span = (size_t) p.occ_groups * plan.flags_per_group;
    p.empty_bits = r->d_empty;
    p.pass_start = plan.start;
    p.pass_slices = D;
    p.chunk_slices = D;
    p.compact = 1;
    const size_t blocks = plan.flags_per_span;

    // the factor cache: which streams' occlusion is at hand, which is computed (and kept)
    estimate_scope(r, base);
    for (FactorEntry* e : r->kept) resolve_entry(r, e, false); // (counts that have arrived sharpen the estimate)
    const bool cache_on = cache_usable(r);
    FactorEntry *have_a = nullptr, *have_r = nullptr, *refill = nullptr;
    FactorKey key_a{};
    if (cache_on) {
        key_a = factor_key(r, base, pa, mode == PASS_ADD, start, D);
        have_a = kept_find(r, key_a);
        if (change) have_r = kept_find(r, factor_key(r, base, *pr, false, start, D));
        // (the removed light alone is not computed: both are — and the added light's factors go into the entry that already
        // holds them, not into a second one with the same key)
        if (change && have_a && !have_r) { refill = have_a; have_a = nullptr; }
    }
    // whatever is kept of a light that this pass takes out of the scene (the removed side of a Change, a removal) is of no
    // further use unless the light comes back: first in line when a buffer is needed (kept_new)
    auto retire = [&](const tbrm_light_pass& q) {
        for (int guard = 0; guard < 2; ++guard) {
            const FactorKey k = factor_key(r, base, q, guard != 0, start, D);
            for (FactorEntry* e : r->kept)
                if (!memcmp(&e->key, &k, sizeof(k))) e->spent = true;
        }
    };
    plan.f_buf = (r->f_buf + 1) % tbrm_resources::kFScratch;
    // (a pass that computes BOTH streams' occlusion — an uncached fused Change — keeps to the first four buffers: the ones a reserved
    // handle holds a second stream's store for; consecutive passes still get different buffers)
    if (change && !have_a && !have_r) plan.f_buf %= 4;
    plan.occ_mode = -1;
    if (have_a) { plan.f_entry[0] = have_a; plan.f_hit[0] = true; use_kept(r, have_a, !change && b_added < 0.0f); ++r->kept_hits; }
    if (have_r) { plan.f_entry[1] = have_r; plan.f_hit[1] = true; use_kept(r, have_r, true); ++r->kept_hits; }
    if (!have_a) {
        // the added light's occlusion: alone (Add rules, or the Change shader's when the removed light's is at hand) or both
        plan.occ_mode = !change ? PASS_ADD : (have_r ? PASS_CHANGE_ONE : PASS_CHANGE);
        r->kept_computed += plan.occ_mode == PASS_CHANGE ? 2 : 1;
        if (int e = ensure_factor_scratch(r, plan.f_buf, blocks, plan.occ_mode == PASS_CHANGE ? 2 : 1)) return e;
        // the pass's block lists: the handle's, if a pass with the same signature has been here under this volume / transfer
        // function / window (tbrm_block_lists.cpp) — then nothing is launched for them, and the live-block count may be known
        plan.lists = block_lists_for_pass(r, p, plan.occ_mode);
        if (!plan.lists) return TBRM_ERR_OUT_OF_MEMORY;
        size_t live = 0;
        const bool live_known = block_lists_count(plan.lists, false, &live);
        if (live_known) r->f_est_blocks = std::max(r->f_est_blocks, live); // (estimate_scope: the estimate is this state's)
        // what is added stays in the scene: its factors are worth keeping (null: no room). What is removed does not.
        if (cache_on && !(mode == PASS_ADD && b_added < 0.0f)) {
            // sized for what passes under this volume / transfer function / window have needed so far; before the first count
            // has arrived: every block of a small pass (an entry that overflows is dropped and its pass sampled again), half
            // the blocks of a large one (CT-like volumes are mostly air: 512^3 of the benchmark keeps 40 %)
            const size_t unknown = blocks * 2048 * sizeof(float) <= ((size_t) 256 << 20) ? blocks : blocks / 2;
            const size_t want = live_known ? std::max<size_t>(live, 1)
                                           : (r->f_est_blocks ? std::min(blocks, r->f_est_blocks + r->f_est_blocks / 32 + 64) : std::max<size_t>(unknown, 1));
            if (refill && refill->cap_blocks >= want && !refill->pinned) {
                refill->resolved = refill->valid = refill->enqueued = refill->spent = false;
                refill->pinned = true;
                refill->last_use = ++r->kept_clock;
                if (refill->lists) --refill->lists->users;
                refill->lists = plan.lists;
                ++plan.lists->users;
                plan.f_entry[0] = refill;
            } else plan.f_entry[0] = kept_new(r, key_a, want, plan.lists);
        }
    } else if (int e = ensure_factor_scratch(r, plan.f_buf, blocks, 0)) return e; // (its events order the buffers' reuse)
    if (tune(TUNE_SWEEP_DEBUG) & 4)
        fprintf(stderr, "[tbrm plan] op %llu axis %d mode %d occ_mode %d buf %d (used %d) entry a %p (hit %d, read_yet %d, last read op %llu) entry r %p; pool %zu\n",
                (unsigned long long) r->op_serial, pa.axis, mode, plan.occ_mode, plan.f_buf, (int) r->f_scratch[plan.f_buf].used, (void*) plan.f_entry[0], (int) plan.f_hit[0],
                plan.f_entry[0] ? (int) plan.f_entry[0]->read_yet : -1, plan.f_entry[0] ? (unsigned long long) plan.f_entry[0]->last_read_op : 0ull, (void*) plan.f_entry[1],
                r->kept.size());
    if (cache_on && change) retire(*pr);
    if (cache_on && !change && b_added < 0.0f) retire(pa);
    r->f_buf = plan.f_buf;

    SweepParams& q = plan.sq;
    q.sx = sfit.sx; q.sy = sfit.sy; q.hx = sfit.hx; q.hy = sfit.hy;
    q.r_from_records = sfit.two_way ? 1 : 0;
    q.r_sx = sfit.r_sx; q.r_sy = sfit.r_sy; q.r_hx = sfit.r_hx; q.r_hy = sfit.r_hy;
    q.tile_rows = sweep_tile_rows(); // (the sweep's tiles: 32 wide, 16 or 32 high)
    p.tiles_y = ceil_div(p.H, q.tile_rows);
    const size_t words = (size_t) D * p.tiles_x * p.tiles_y * (size_t) sweep_record_words(sfit.hx, sfit.hy, q.tile_rows) * (size_t) (r->lv_fmt != FMT_U8 && change && !sfit.two_way ? 2 : 1); // (float records: a granule per stream handed over)
    const size_t words1 = sfit.two_way ? (size_t) D * p.tiles_x * p.tiles_y * (size_t) sweep_record_words(sfit.r_hx, sfit.r_hy, q.tile_rows) : 0;
    if (words >= ((size_t) 1 << 32) || words1 >= ((size_t) 1 << 32)) return declined("hand-off records too large");
    const size_t gw = r->lv_fmt != FMT_U8 ? 2 : 1; // 32-bit words per record word (float light volumes: {float, launch tag})
    if (words * gw >= ((size_t) 1 << 32) || words1 * gw >= ((size_t) 1 << 32)) return declined("hand-off records too large");
    if (int e = ensure_sweep(r, std::max<size_t>(words * gw, 1), words1 * gw)) return e;
    // (the record buffers may still grow while the operator's other passes are planned: taken at enqueue time)
    // (measured at 512^3, profiles/r03_sweep_ablation.txt: requests two slices ahead beat three by 3 - 4 %, start delays of 1 - 2 us
    // per hop tie and beat 3 - 4 us)
    q.prefetch = tune(TUNE_SWEEP_PREFETCH) > 0 ? std::min(tune(TUNE_SWEEP_PREFETCH), 6) : 2;
    q.stagger_ns = 1500;
    q.debug = tune(TUNE_SWEEP_DEBUG);
    q.reinit_slice = pa.dir < 0 ? pad : 0;
    q.n_real = pa.dir > 0 ? D - pad : D;
    q.lv_f32 = r->lv_fmt != FMT_U8 ? 1 : 0;
    {
        const int ms = tune(TUNE_SWEEP_TIMEOUT_MS);
        q.give_up_ticks = ms < 0 ? 0ull : (unsigned long long) (ms == 0 ? 2000 : ms) * 100000ull;
    }
    plan.serial = ++r->plan_serial;
    return TBRM_OK;
}

// pr == null: Add of pa (b_added = +-1). Else two streams: mode PASS_CHANGE (pa added, pr removed) or PASS_ADD2 (pa, then
// pr, both added with b_added / b_added2).
int plan_pass(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
              const tbrm_slab* slab, PassPlan& plan, int two_stream_mode, float b_added2)
{
    g_plan_note = "";
    const bool change = pr != nullptr;
    const int mode = change ? two_stream_mode : PASS_ADD;
    if (!slab || (pa.axis == 2 && !r->resident && mode != PASS_ADD2 && tune(TUNE_SLAB_SWEEP) != 0)) {
        const int e = plan_pass_sweep(r, base, pa, pr, b_added, plan, mode, slab);
        if (e != TBRM_ERR_UNSUPPORTED) return e;
    }
    // the chunked chain: whatever the sweep declines (float light volumes, passes that are not whole brick layers, taps on
    // both sides of the pixel, slab-partitioned passes); its occlusion is not cached
    ChunkFit fit;
    if (!chunk_fit(r, pa, pr, fit, mode)) {
        if (!slab || two_stream_mode == PASS_ADD2) return TBRM_ERR_UNSUPPORTED;
        return plan_pass_sliced(r, base, pa, pr, b_added, *slab, plan);
    }
    r->kept_computed += mode == PASS_ADD ? 1 : 2;
    const int W = pa.td[0], H = pa.td[1], D_pass = pa.td[2];
    plan = PassPlan{};
    plan.mode = mode;
    fill_pass_params(r, base, pa, pr, b_added, b_added2, plan);
    ChunkParams& p = plan.p;
    p.dx_lo = fit.tx.lo; p.dx_hi = fit.tx.hi; p.dy_lo = fit.ty.lo; p.dy_hi = fit.ty.hi;
    p.rect_planes = rect_planes_for(r, mode);
    const int M = fit.M;
    plan.M = M;
    plan.chunks_of_pass = ceil_div(D_pass, M);
    if (slab) {
        const int nz = r->lv_dims[2];
        if (slab->z_begin < 0 || slab->z_end > nz || slab->z_begin >= slab->z_end || slab->z_begin % kChunkTile || slab->z_end % kChunkTile ||
            nz % kChunkTile)
            return fail(TBRM_ERR_INVALID_ARG, "slab [%d, %d) of a light volume %d deep: bounds and depth must be multiples of %d",
                        slab->z_begin, slab->z_end, nz, kChunkTile);
        if (pa.axis == 2) { // the pass runs along the slab axis: this handle advances its own slices, planes are handed on
            plan.D = slab->z_end - slab->z_begin;
            plan.start = pa.dir > 0 ? slab->z_begin : slab->z_end - 1;
            plan.first_chunk_of_pass = (pa.dir > 0 ? slab->z_begin : nz - slab->z_end) / M;
            plan.pass_begins_here = plan.first_chunk_of_pass == 0;
        } else { // z is the plane's row axis: the slab's tile rows, and the occlusion of every row their windows can reach
            plan.lateral = true;
            p.tile_row0 = slab->z_begin / kChunkTile;
            p.tiles_y = (slab->z_end - slab->z_begin) / kChunkTile;
            p.roi_by0 = std::max(slab->z_begin - kChunkTile, 0) / 16;
            p.roi_by1 = std::min(ceil_div(slab->z_end + kChunkTile, 16), p.occ_blocks_y);
        }
    }
    if (r->resident) {
        if (!slab || slab->z_begin != r->owned.z_begin || slab->z_end != r->owned.z_end)
            return fail(TBRM_ERR_INVALID_ARG, "a slab-resident handle runs its own slab [%d, %d) only", r->owned.z_begin, r->owned.z_end);
        // every data texel the occlusion of this handle's rows / slices can sample has to be resident: z range of the taps of
        // light-volume slices [za, zb), with the kernel's own arithmetic (GetUVW + UVWOffset, texel split)
        const int za = plan.lateral ? std::max(slab->z_begin - kChunkTile, 0) : slab->z_begin;
        const int zb = plan.lateral ? std::min(slab->z_end + kChunkTile, r->lv_dims[2]) : slab->z_end;
        int lo = INT32_MAX, hi = INT32_MIN;
        for (const tbrm_light_pass* q : {&pa, pr}) {
            if (!q) continue;
            for (int z : {za, zb - 1}) {
                const float w = (((float) (uint32_t) z + 0.5f) / (float) (uint32_t) r->lv_dims[2]) + q->uvw_offset[2];
                float x = w * (float) r->desc.dim_z - 0.5f;
                x = std::fmin(std::fmax(x, -0x1p30f), 0x1p30f);
                const int i0 = (int) std::floor(x);
                lo = std::min(lo, i0);
                hi = std::max(hi, i0 + 1);
            }
        }
        lo = clamp_int(lo, 0, r->desc.dim_z - 1);
        hi = clamp_int(hi, 0, r->desc.dim_z - 1);
        if ((lo >> 3) < r->res_data.lo || (hi >> 3) >= r->res_data.hi)
            return fail(TBRM_ERR_UNSUPPORTED, "this pass samples data slices %d..%d, the handle holds %d..%d", lo, hi, r->res_data.lo * 8,
                        r->res_data.hi * 8 - 1);
    }
    const int D = plan.D;
    plan.n_chunks = ceil_div(D, M);

    // The occlusion launches are decoupled from the chain's chunk length: one launch covers a "span" of S slices (several
    // chunks), so that it has enough workgroups to fill 256 CUs even when the chain has to run short chunks (a strongly
    // slanted pass runs M = 8) and the live-workgroup list of a span deals an even share to every CU.
    int S = 128; // measured on MI355X, fused Change at 512^3: S = 32 2.80 ms, 64 2.61, 128 2.53, 256 2.52
    if (tune(TUNE_OCC_SLICES) > 0) S = tune(TUNE_OCC_SLICES);
    S = std::max(M, (S / M) * M);

    const size_t slice_elems = (size_t) W * H;
    while (S > M && ((size_t) S * slice_elems + 2 * kPlaneGuard) * sizeof(float) >= ((size_t) 1 << 32)) S -= M;
    if (((size_t) S * slice_elems + 2 * kPlaneGuard) * sizeof(float) >= ((size_t) 1 << 32)) return declined("slice plane too large for the occlusion scratch");
    plan.S = S;
    plan.n_spans = ceil_div(D, S);

    // empty-block hand-off (needs the per-brick emptiness bits of the current TF/window): one flag per occlusion
    // workgroup of the whole pass and per span the ascending list of the workgroups with work, computed in front of the
    // pass's first occlusion launch (enqueue_plan_chunk)
    plan.sparse = tune(TUNE_SPARSE_OCC) != 0;
    plan.work_list = plan.sparse && tune(TUNE_OCC_LIST) != 0;
    p.occ_groups = ceil_div(S, kOccSlices);
    plan.flags_per_group = (size_t) p.occ_blocks_y * p.occ_blocks_x;
    plan.flags_per_span = (size_t) p.occ_groups * plan.flags_per_group;
    if (plan.sparse) {
        if (plan.n_spans > 4096) return declined("too many occlusion spans");
        if (int e = ensure_skipping(r)) return e;
        p.empty_bits = r->d_empty;
        p.pass_start = plan.start;
        p.pass_slices = D;
        p.chunk_slices = S;
    }
    // the occlusion stores of the stream(s) this pass propagates
    const size_t flag_bytes = plan.sparse ? plan.flags_per_span * plan.n_spans : 0;
    for (int b = 0; b < 2; ++b)
        for (int si = 0; si < (plan.two_streams() ? 2 : 1); ++si)
            if (int e = ensure_store(r, &r->occ_tmp[b][si], S, slice_elems, si == 0 ? flag_bytes : 0)) return e;
    plan.serial = ++r->plan_serial;
    return TBRM_OK;
}

// The plane holding the propagated light of stream `si` (0: a, 1: r) BEFORE chunk `boundary` (boundary = n_chunks: after
// the last one): chunk c reads the planes of parity c & 1 and writes the others.
float* plan_plane(const tbrm_resources* r, int boundary, int si) { return r->d_plane[2 * si + (boundary & 1)] + kPlaneGuard; }

} // namespace tbrm_host
