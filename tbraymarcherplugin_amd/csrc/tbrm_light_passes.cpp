// tbrm_light_passes.cpp — host side of the illumination operators: what LightingShaders.cpp:35-326 does on the render
// thread (per light the axis passes, per pass the slice loop), planned here as chunks for the gfx950 kernels of
// tbrm_light_kernels.hip and enqueued on the handle's stream. tbrm_api.cpp's entry points call enqueue_add /
// enqueue_add_batch / enqueue_change for the whole-volume operators and plan_pass / enqueue_plan_chunk step by step for the
// slab-partitioned ones.
#include "tbrm_resources.h"
#include "tbrm_light_chain.h"
#include "tbrm_light_sweep.h"

#include <algorithm>
#include <climits>
#include <chrono>
#include <cmath>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace tbrm_host {

// flags of the events that order the two streams (experiment: TBRM_EVENT_FLAGS=0 creates them with timing, i.e. with a marker of
// their own in the queue at record time)
static unsigned event_flags()
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
    if (tune(TUNE_LIGHT_SWEEP) == 2 || f32) return false; // (diagnostics: such passes take the chain, as before round 3's last week)
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
    return room[0] <= 14 && room[1] <= 14 && sweep_halo_chunks(fit.hx, fit.hy, th) <= 6 && sweep_halo_chunks(fit.r_hx, fit.r_hy, th) <= 6;
}

void release_sweep(tbrm_resources* r)
{
    for (auto& rec : r->sweep_rec) { (void) hipFree(rec); rec = nullptr; }
    r->sweep_rec_words = r->sweep_rec1_words = 0;
    (void) hipFree(r->sweep_ticket);
    r->sweep_ticket = nullptr;
    if (r->sweep_error) (void) hipHostFree(r->sweep_error);
    r->sweep_error = nullptr;
    (void) hipFree(r->sweep_stamps);
    r->sweep_stamps = nullptr;
}

int sweep_check(tbrm_resources* r)
{
    if ((tune(TUNE_SWEEP_DEBUG) & 2) && r->sweep_stamps && r->sweep_stamp_tiles > 0) { // diagnostics: the last launch's timeline
        std::vector<unsigned long long> t((size_t) r->sweep_stamp_tiles * 4);
        if (hipMemcpy(t.data(), r->sweep_stamps, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
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
static int ensure_sweep(tbrm_resources* r, size_t words, size_t words1 = 0)
{
    if (!r->sweep_ticket) {
        HIP_TRY(hipMalloc((void**) &r->sweep_ticket, 2 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(r->sweep_ticket, 0, 2 * sizeof(int), r->stream));
        HIP_TRY(hipHostMalloc((void**) &r->sweep_error, sizeof(int), hipHostMallocMapped));
        *r->sweep_error = 0;
    }
    if (words > r->sweep_rec_words) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->sweep_rec[0]);
        r->sweep_rec[0] = nullptr;
        r->sweep_rec_words = 0;
        HIP_TRY(hipMalloc((void**) &r->sweep_rec[0], words * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(r->sweep_rec[0], 0, words * sizeof(uint32_t), r->stream)); // tag 0: no launch
        r->sweep_rec_words = words;
    }
    if (words1 > r->sweep_rec1_words) {
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
static int next_sweep_epoch(tbrm_resources* r, uint32_t& epoch, uint32_t launches = 1)
{
    if (!r->sweep_epoch_preset_done && tune(TUNE_SWEEP_EPOCH_PRESET) > 0) { // (a test hook: the 16-bit tags start over after 65535 launches)
        r->sweep_epoch = (uint32_t) tune(TUNE_SWEEP_EPOCH_PRESET) & 0xffffu;
        r->sweep_epoch_preset_done = true;
    }
    if (++r->sweep_epoch + (launches - 1) >= (1u << 16)) { // 2^16 launches later: tags start over
        HIP_TRY(hipMemsetAsync(r->sweep_rec[0], 0, r->sweep_rec_words * sizeof(uint32_t), r->stream));
        if (r->sweep_rec[1]) HIP_TRY(hipMemsetAsync(r->sweep_rec[1], 0, r->sweep_rec1_words * sizeof(uint32_t), r->stream));
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

struct SpanRange { int s0, sn, c0, c1; bool sparse; };
static SpanRange span_range(const PassPlan& plan, int sp)
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

// ---- occlusion stores and the factor cache (tbrm_resources.h) ------------------------------------------------------------

static void drain_streams(tbrm_resources* r)
{
    (void) hipStreamSynchronize(r->stream);
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream);
}
void drain_streams_public(tbrm_resources* r) { drain_streams(r); }

static void free_entry(FactorEntry* e)
{
    (void) hipFree(e->base);
    if (e->lists) --e->lists->users;
    for (hipEvent_t ev : {e->ev_filled, e->ev_idle})
        if (ev) (void) hipEventDestroy(ev);
    delete e;
}

void release_kept(tbrm_resources* r)
{
    if (!r->kept.empty()) drain_streams(r);
    for (FactorEntry* e : r->kept) free_entry(e);
    r->kept.clear();
}

size_t kept_bytes(const tbrm_resources* r)
{
    size_t n = 0;
    for (const FactorEntry* e : r->kept) n += e->bytes();
    return n;
}

void release_occ_stores(tbrm_resources* r)
{
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream);
    for (auto& buf : r->occ_tmp)
        for (OccStore& st : buf) {
            (void) hipFree(st.base);
            (void) hipFree(st.flags);
            (void) hipFree(st.list);
            st = OccStore{};
        }
    for (auto& slot : r->occ_slot) slot = tbrm_resources::OccSlot{};
    for (FactorScratch& f : r->f_scratch) {
        for (float*& st : f.store) { (void) hipFree(st); st = nullptr; }
        for (hipEvent_t ev : {f.ev_ready, f.ev_idle})
            if (ev) (void) hipEventDestroy(ev);
        f = FactorScratch{};
    }
    (void) hipFree(r->d_ones);
    r->d_ones = nullptr;
    release_kept(r);
    release_block_lists(r);
}

// room for `slices` planes of slice_elems floats behind the page of ones, and for the flags / lists of a pass
static int ensure_store(tbrm_resources* r, OccStore* st, int slices, size_t slice_elems, size_t flag_bytes)
{
    const size_t elems = (size_t) slices * slice_elems; // (the passes of a non-cubic volume have planes of different sizes)
    if (elems > st->capacity || !st->base) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        if (r->occ_stream) HIP_TRY(hipStreamSynchronize(r->occ_stream));
        (void) hipFree(st->base);
        st->base = nullptr;
        st->capacity = 0;
        HIP_TRY(hipMalloc((void**) &st->base, (elems + 2 * kPlaneGuard) * sizeof(float)));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) st->base, 0x3f800000, 1024, r->stream)); // the page of ones
        st->capacity = elems;
    }
    if (flag_bytes > st->flag_bytes) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        if (r->occ_stream) HIP_TRY(hipStreamSynchronize(r->occ_stream));
        (void) hipFree(st->flags);
        (void) hipFree(st->list);
        st->flags = nullptr;
        st->list = nullptr;
        st->flag_bytes = 0;
        HIP_TRY(hipMalloc((void**) &st->flags, flag_bytes));
        HIP_TRY(hipMalloc((void**) &st->list, flag_bytes * sizeof(uint32_t) + 4096 * sizeof(int))); // lists + per-span counts
        st->flag_bytes = flag_bytes;
    }
    return TBRM_OK;
}

static int ensure_occ_stream(tbrm_resources* r)
{
    if (r->occ_stream) return TBRM_OK;
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&r->occ_stream, hipStreamNonBlocking, tune(TUNE_OCC_PRIORITY) == 1 ? 0 : least));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&r->occ_ev_fork[k], event_flags()));
        HIP_TRY(hipEventCreateWithFlags(&r->occ_ev_ready[k], event_flags()));
    }
    for (hipEvent_t& ev : r->op_done) HIP_TRY(hipEventCreateWithFlags(&ev, event_flags()));
    return TBRM_OK;
}

// the scratch of a sweep pass with `blocks` occlusion blocks: stores of `streams` streams (every block could be live), the
// page of ones
static int ensure_factor_scratch(tbrm_resources* r, int b, size_t blocks, int streams)
{
    FactorScratch& f = r->f_scratch[b];
    if (!r->d_ones) {
        HIP_TRY(hipMalloc((void**) &r->d_ones, 1024 * sizeof(float)));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) r->d_ones, 0x3f800000, 1024, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream)); // (read from the occlusion stream's sweeps' predecessors: simplest to have it done)
    }
    if (!f.ev_ready) {
        HIP_TRY(hipEventCreateWithFlags(&f.ev_ready, event_flags()));
        HIP_TRY(hipEventCreateWithFlags(&f.ev_idle, event_flags()));
    }
    const bool grow_store = blocks > f.store_blocks;
    bool need = false;
    for (int si = 0; si < streams; ++si) need = need || grow_store || !f.store[si];
    if (!need) return TBRM_OK;
    drain_streams(r);
    if (grow_store) {
        for (float*& st : f.store) { (void) hipFree(st); st = nullptr; }
        f.store_blocks = 0;
    }
    const size_t cap = std::max(blocks, f.store_blocks);
    for (int si = 0; si < streams; ++si)
        if (!f.store[si]) HIP_TRY(hipMalloc((void**) &f.store[si], cap * 2048 * sizeof(float)));
    f.store_blocks = cap;
    return TBRM_OK;
}

// start / D: the slices the planned sweep covers (a slab's share of a pass along z has the whole pass's geometry but its own
// slices: its blocks and their ranks are not the whole pass's)
static FactorKey factor_key(const tbrm_resources* r, const PropParams& base, const tbrm_light_pass& q, bool guard, int start, int D)
{
    FactorKey k;
    memset(&k, 0, sizeof(k)); // compared bytewise
    k.data_gen = r->data_gen;
    k.tf_gen = r->tf_gen;
    k.win[0] = base.win.center; k.win[1] = base.win.width; k.win[2] = base.win.low_cutoff; k.win[3] = base.win.high_cutoff;
    for (int c = 0; c < 3; ++c) { k.cc[c] = base.cc[c]; k.cd[c] = base.cd[c]; k.uvw_off[c] = q.uvw_offset[c]; }
    k.data_border = base.data_border;
    k.clip_mode = base.clip_mode;
    k.axis = q.axis; k.dir = q.dir; k.start = start; k.D = D; k.W = q.td[0]; k.H = q.td[1];
    // the Add shader's uvw == saturate(uvw) guard only matters where a sample outside the cube could be opaque
    // (k_shell_transparent): else both shaders compute the same factors and one entry serves both
    k.guard = (guard && !r->shell_transparent) ? 1 : 0;
    k.step100 = q.step_size * 100.0f;
    return k;
}

// what is known about the live blocks of a pass under the current volume / transfer function / window
static void estimate_scope(tbrm_resources* r, const PropParams& base)
{
    const float win[4] = {base.win.center, base.win.width, base.win.low_cutoff, base.win.high_cutoff};
    if (r->f_est_key[0] != r->data_gen || r->f_est_key[1] != r->tf_gen || memcmp(r->f_est_win, win, sizeof(win))) {
        r->f_est_key[0] = r->data_gen;
        r->f_est_key[1] = r->tf_gen;
        memcpy(r->f_est_win, win, sizeof(win));
        r->f_est_blocks = 0;
        // what was kept under another volume / transfer function / window is out of reach unless the host comes back to exactly
        // that state: first in line when a buffer is needed
        for (FactorEntry* e : r->kept) e->spent = true;
    }
}

// reads an entry's live-block count once it has arrived (wait: block until it has); an entry that overflowed is dropped
static void resolve_entry(tbrm_resources* r, FactorEntry* e, bool wait)
{
    if (e->resolved || !e->enqueued) return;
    size_t count = 0;
    if (!block_lists_count(e->lists, wait, &count)) return;
    e->resolved = true;
    e->valid = count <= e->cap_blocks;
    if (e->key.data_gen == r->f_est_key[0] && e->key.tf_gen == r->f_est_key[1] && !memcmp(e->key.win, r->f_est_win, sizeof(e->key.win)))
        r->f_est_blocks = std::max(r->f_est_blocks, count);
}

static FactorEntry* kept_find(tbrm_resources* r, const FactorKey& key)
{
    for (FactorEntry* e : r->kept) {
        if (!e->enqueued || memcmp(&e->key, &key, sizeof(key)) || (e->resolved && !e->valid)) continue;
        resolve_entry(r, e, true);
        if (e->valid) return e;
    }
    return nullptr;
}

// the budget of the factor cache in bytes: the tunable, or (auto) an eighth of the device's memory
static size_t kept_budget(const tbrm_resources* r)
{
    const int mb = tune(TUNE_LIGHT_CACHE_MB);
    if (mb >= 0) return (size_t) mb << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return total_b / 8;
}

// An entry for a pass about to be computed, sized for `want` blocks: the buffer of an entry whose light has left the scene
// if one is large enough (no allocation while lights merely move), else a fresh allocation while the budget lasts and the
// device has room to spare, else the least recently used entry's. null: the cache is off, or nothing can be had.
static FactorEntry* kept_new(tbrm_resources* r, const FactorKey& key, size_t want, BlockLists* lists)
{
    const size_t bytes = want * 2048 * sizeof(float);
    FactorEntry* e = nullptr;
    auto fits = [&](const FactorEntry* c) { return !c->pinned && c->cap_blocks >= want && c->cap_blocks <= want + want / 2 + 64; };
    auto reusable = [&](const FactorEntry* c) { return fits(c) && ((c->resolved && !c->valid) || c->spent || !c->enqueued); };
    // An entry that the operator just before this one read (the removed side of its Change) is still being read by that
    // operator's sweeps when this operator's occlusion could start — beside those very sweeps, which leave two thirds of a
    // CU's issue slots idle. Reusing it would make the occlusion wait for them (measured: the whole 0.38 ms of it exposed
    // in front of every operator of the benchmark's loop); an entry retired an operator earlier is free by then.
    auto settled = [&](const FactorEntry* c) { return !(c->read_yet && c->last_read_op + 1 >= r->op_serial); };
    for (FactorEntry* c : r->kept) // dropped and spent entries first, oldest first
        if (reusable(c) && settled(c) && (!e || c->last_use < e->last_use)) e = c;
    size_t budget = 0; // (asked for only when an allocation is on the cards: hipMemGetInfo takes milliseconds)
    auto room_for_a_new_one = [&]() {
        budget = kept_budget(r);
        if (kept_bytes(r) + bytes > budget) return false;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return false; }
        return free_b >= 2 * bytes + ((size_t) 1 << 30);
    };
    if (!e && !room_for_a_new_one()) // (no room to grow: the entry that is still being read, and the wait)
        for (FactorEntry* c : r->kept)
            if (reusable(c) && (!e || c->last_use < e->last_use)) e = c;
    if (!e) {
        if (budget == 0) budget = kept_budget(r);
        // make room: entries that are of no use go first, then the least recently used
        auto victim = [&]() -> FactorEntry* {
            FactorEntry* v = nullptr;
            for (FactorEntry* c : r->kept)
                if (!c->pinned && ((c->resolved && !c->valid) || c->spent || !c->enqueued) && (!v || c->last_use < v->last_use)) v = c;
            if (v) return v;
            for (FactorEntry* c : r->kept)
                if (!c->pinned && (!v || c->last_use < v->last_use)) v = c;
            return v;
        };
        if (bytes > budget) return nullptr;
        while (kept_bytes(r) + bytes > budget) {
            FactorEntry* v = victim();
            if (!v) return nullptr;
            drain_streams(r);
            r->kept.erase(std::find(r->kept.begin(), r->kept.end(), v));
            free_entry(v);
        }
        // never the last of the device's memory: whoever else allocates on this device (a renderer, torch, another handle)
        // must not find it gone (asked at every allocation: a few per scene, hipMemGetInfo takes milliseconds)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bytes + ((size_t) 1 << 30)) { (void) hipGetLastError(); return nullptr; }
        e = new FactorEntry{};
        bool ok = hipMalloc((void**) &e->base, bytes) == hipSuccess;
        for (hipEvent_t* ev : {&e->ev_filled, &e->ev_idle}) ok = ok && hipEventCreateWithFlags(ev, event_flags()) == hipSuccess;
        if (!ok) { // out of memory: do without
            (void) hipGetLastError();
            free_entry(e);
            return nullptr;
        }
        e->cap_blocks = want;
        r->kept.push_back(e);
    }
    e->key = key;
    e->resolved = false;
    e->valid = false;
    e->enqueued = false;
    e->spent = false;
    e->pinned = true;
    e->last_use = ++r->kept_clock;
    if (e->lists) --e->lists->users;
    e->lists = lists; // (its blocks will be stored under these ranks)
    ++lists->users;
    return e;
}

static void use_kept(tbrm_resources* r, FactorEntry* e, bool leaves_the_scene)
{
    e->spent = leaves_the_scene;
    e->pinned = true;
    e->last_use = ++r->kept_clock;
}

static void unpin_kept(tbrm_resources* r)
{
    for (FactorEntry* e : r->kept) e->pinned = false;
}

static bool cache_usable(const tbrm_resources* r) { return !force_slice_kernel() && tune(TUNE_LIGHT_CACHE_MB) != 0; }

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
    const size_t words = (size_t) D * p.tiles_x * p.tiles_y * (size_t) sweep_record_words(sfit.hx, sfit.hy, q.tile_rows) * (size_t) (r->lv_fmt != FMT_U8 && change ? 2 : 1);
    const size_t words1 = sfit.two_way ? (size_t) D * p.tiles_x * p.tiles_y * (size_t) sweep_record_words(sfit.r_hx, sfit.r_hy, q.tile_rows) : 0;
    if (words >= ((size_t) 1 << 32) || words1 >= ((size_t) 1 << 32)) return declined("hand-off records too large");
    if (int e = ensure_sweep(r, std::max<size_t>(words, 1), words1)) return e;
    // (the record buffers may still grow while the operator's other passes are planned: taken at enqueue time)
    // (measured at 512^3, profiles/r03_sweep_ablation.txt: requests two slices ahead beat three by 3 - 4 %, start delays of 1 - 2 us
    // per hop tie and beat 3 - 4 us)
    q.prefetch = tune(TUNE_SWEEP_PREFETCH) > 0 ? std::min(tune(TUNE_SWEEP_PREFETCH), 6) : 2;
    q.stagger_ns = tune(TUNE_SWEEP_STAGGER_NS) != 0 ? std::max(tune(TUNE_SWEEP_STAGGER_NS), 0) : 1500;
    q.debug = tune(TUNE_SWEEP_DEBUG);
    q.reinit_slice = pa.dir < 0 ? pad : 0;
    q.lv_f32 = r->lv_fmt != FMT_U8 ? 1 : 0;
    plan.rec_words = words;
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

// The occlusion of span sp of the plan into buffer b ({a,r}.occ_next = that buffer's stores), with — in front of the pass's
// first span — the empty-block flags and work lists of the whole pass. beside: on occ_stream, ordered behind everything
// enqueued on the handle's stream so far (the data volume, the transfer function's tables, the chains that read the
// buffer's previous contents), with a grid small enough to be resident beside the chain's workgroups; the chain waits for
// occ_ev_ready[b] (enqueue_plan_chunk).
static int enqueue_occlusion(tbrm_resources* r, const PassPlan& plan, int sp, int b, bool beside, int beside_wgs_per_cu = 0)
{
    hipStream_t s = r->stream;
    if (beside) {
        if (int e = ensure_occ_stream(r)) return e;
        s = r->occ_stream;
        HIP_TRY(hipEventRecord(r->occ_ev_fork[b], r->stream));
        HIP_TRY(hipStreamWaitEvent(s, r->occ_ev_fork[b], 0));
    }
    ChunkParams p = plan.p;
    const SpanRange q = span_range(plan, sp);
    // the propagated streams' occlusion: one launch per span computes both (a block is flagged empty when it is empty for both)
    const int occ_mode = plan.mode;
    OccStore* const fs = &r->occ_tmp[plan.serial & 1][0];
    int* const counts = (int*) (fs->list + fs->flag_bytes);
    p.a.occ_next = r->occ_tmp[b][0].base + kPlaneGuard;
    p.r.occ_next = plan.n_streams() == 2 ? r->occ_tmp[b][1].base + kPlaneGuard : nullptr;
    if (sp == 0 && plan.sparse) {
        p.occ_flags_out = fs->flags;
        p.occ_list_out = plan.work_list ? fs->list : nullptr;
        p.occ_count_out = counts;
        HIP_TRY(launch_occ_flags(p, occ_mode, plan.n_spans, s));
    }
    p.j0 = plan.start + q.s0 * plan.dir;
    p.n_steps = q.sn;
    p.occ_flags = nullptr;
    p.occ_list = q.sparse && plan.work_list ? fs->list + (size_t) sp * plan.flags_per_span : nullptr;
    p.occ_count = q.sparse && plan.work_list ? counts + sp : nullptr;
    if (q.sparse && !plan.work_list) p.occ_flags = fs->flags + (size_t) sp * plan.flags_per_span;
    p.occ_grid_cap = beside ? beside_wgs_per_cu * r->n_cus : 0;
    HIP_TRY(launch_light_occlusion(p, occ_mode, s));
    ++r->occ_launches;
    if (beside) HIP_TRY(hipEventRecord(r->occ_ev_ready[b], s));
    r->occ_slot[b].plan_serial = plan.serial;
    r->occ_slot[b].span = sp;
    r->occ_slot_async[b] = beside;
    return TBRM_OK;
}

// What the sweep passes' occlusion reads — the data volume, the transfer function's tables, the per-brick emptiness bits — is
// written on the handle's stream; the occlusion stream waits for it ONCE after every change (not per operator: an event
// recorded behind the previous operator's sweeps would take the occlusion out from beside them).
static int order_behind_inputs(tbrm_resources* r)
{
    if (!r->occ_inputs_changed) return TBRM_OK;
    HIP_TRY(hipEventRecord(r->occ_ev_fork[0], r->stream));
    HIP_TRY(hipStreamWaitEvent(r->occ_stream, r->occ_ev_fork[0], 0));
    r->occ_inputs_changed = false;
    return TBRM_OK;
}

// The occlusion stream is about to overwrite the scratch buffer / cache entry of `plan`, which earlier sweeps may still be
// reading. Readers of an EARLIER operator are waited for through that operator's "sweeps done" event — one wait per occlusion
// launch whatever the number of buffers (wait_for_readers; *op collects the latest such operator) —, readers of THIS operator
// (a reset of many lights runs through all four buffers within one run_passes) through the buffer's own event.
static int note_readers(tbrm_resources* r, const PassPlan& plan, uint64_t* op)
{
    FactorScratch& f = r->f_scratch[plan.f_buf];
    FactorEntry* const e = plan.f_hit[0] ? nullptr : plan.f_entry[0];
    if (f.used && !(tune(TUNE_SWEEP_DEBUG) & 8)) {
        if (f.last_read_op >= r->op_serial && f.idle_recorded) HIP_TRY(hipStreamWaitEvent(r->occ_stream, f.ev_idle, 0));
        else if (f.last_read_op >= r->op_serial) *op = UINT64_MAX; // (no event of its own: everything enqueued so far)
        else *op = std::max(*op, f.last_read_op);
    }
    if (e && e->read_yet && !(tune(TUNE_SWEEP_DEBUG) & 16)) {
        if (e->last_read_op >= r->op_serial && e->idle_recorded) HIP_TRY(hipStreamWaitEvent(r->occ_stream, e->ev_idle, 0));
        else if (e->last_read_op >= r->op_serial) *op = UINT64_MAX;
        else *op = std::max(*op, e->last_read_op);
    }
    return TBRM_OK;
}
static int wait_for_readers(tbrm_resources* r, uint64_t op)
{
    if (r->frame_pending && tune(TUNE_OCC_AFTER_FRAME)) { // (tunable occ_after_frame: not beside the frame that is on its way)
        HIP_TRY(hipStreamWaitEvent(r->occ_stream, r->frame_done, 0));
        r->frame_pending = false;
    }
    if (op == 0) return TBRM_OK;
    const int k = (int) (op % tbrm_resources::kOpEvents);
    if (op != UINT64_MAX && r->op_done_serial[k] >= op) { // (what a later operator recorded in the same slot is later still)
        HIP_TRY(hipStreamWaitEvent(r->occ_stream, r->op_done[k], 0));
        return TBRM_OK;
    }
    // that operator never recorded its event (it failed half way): everything enqueued on the handle's stream so far
    HIP_TRY(hipEventRecord(r->occ_ev_fork[1], r->stream));
    HIP_TRY(hipStreamWaitEvent(r->occ_stream, r->occ_ev_fork[1], 0));
    return TBRM_OK;
}

// The block lists of a sweep pass that computes occlusion, unless the handle has them (BlockLists::enqueued): empty-block flags,
// work list, block ranks, count — on the occlusion stream, in front of the occlusion launch that walks the list. How many
// blocks a cache entry has to hold is known on the device only: the compaction leaves the count in pinned host memory too
// (block_lists_count, resolve_entry).
static int enqueue_block_lists(tbrm_resources* r, const PassPlan& plan)
{
    BlockLists* const l = plan.lists;
    if (!l) return fail(TBRM_ERR_INVALID_ARG, "a pass that computes occlusion has no block lists");
    if (l->enqueued) return TBRM_OK;
    ChunkParams p = plan.p;
    p.occ_flags_out = l->flags;
    p.occ_list_out = l->list;
    p.occ_count_out = l->count;
    p.occ_slot_out = l->slot;
    p.occ_count_host = l->count_host;
    HIP_TRY(launch_occ_flags(p, plan.occ_mode, 1, r->occ_stream));
    HIP_TRY(hipEventRecord(l->ev_done, r->occ_stream));
    l->enqueued = true;
    ++r->lists_launches;
    return TBRM_OK;
}

// The occlusion of a sweep pass (plan_pass_sweep): the whole pass's empty-block flags, work list and block ranks, then one
// launch that leaves the factors of the live blocks block-compact in the cache entry being filled and / or the scratch
// buffer — all on the occlusion stream, beside whatever the handle's stream is running (the sweep of the pass before, a
// frame); the sweep waits for FactorScratch::ev_ready. Nothing to do when both streams' factors come from the cache.
int enqueue_sweep_occlusion(tbrm_resources* r, const PassPlan& plan)
{
    if (!plan.sweep || plan.occ_mode < 0 || plan.occ_enqueued) return TBRM_OK;
    FactorScratch& f = r->f_scratch[plan.f_buf];
    FactorEntry* const e = plan.f_hit[0] ? nullptr : plan.f_entry[0];
    hipStream_t s = r->occ_stream;
    if (int e2 = order_behind_inputs(r)) return e2;
    // the buffers about to be overwritten may still be read by earlier sweeps
    {
        uint64_t op = 0;
        if (int e2 = note_readers(r, plan, &op)) return e2;
        if (int e2 = wait_for_readers(r, op)) return e2;
    }
    ChunkParams p = plan.p;
    if (int e2 = enqueue_block_lists(r, plan)) return e2;
    p.j0 = plan.start;
    p.n_steps = plan.D;
    p.occ_flags = nullptr;
    p.occ_list = plan.lists->list;
    p.occ_count = plan.lists->count;
    // an ordinary grid, one workgroup per live block: beside a sweep (nine waves and a third of the LDS per CU) the dispatcher
    // fills what is free; the resident grids that pay beside the chunked chain (occ_overlap) only slow this pair down
    // (measured: cached Change 1.48 ms, 1.58 - 1.68 with 4 - 8 resident workgroups per CU)
    p.occ_grid_cap = 0;
    p.a.fs_keep = e ? e->base : nullptr;
    p.a.fs_cap = e ? (uint32_t) e->cap_blocks : 0u;
    p.a.fs_spill = f.store[0];
    p.r.fs_keep = nullptr;
    p.r.fs_cap = 0;
    p.r.fs_spill = f.store[1];
    HIP_TRY(launch_light_occlusion(p, plan.occ_mode, s));
    ++r->occ_launches;
    HIP_TRY(hipEventRecord(f.ev_ready, s));
    if (e) {
        HIP_TRY(hipEventRecord(e->ev_filled, s));
        e->enqueued = true;
    }
    plan.occ_enqueued = true;
    return TBRM_OK;
}

// May ONE occlusion launch serve both passes (tbrm_internal.h DualOcc)? They are passes of the same operator (same volume,
// window, transfer function, clip plane); what has to hold is that they sample at the same positions under the same rules —
// UVWOffset bit-equal per stream, which the reference's host math gives the two passes of a light (LightingShaders.cpp:114-124:
// normalize(lightPos) / min(TD) whatever the axis) —, that both still have their occlusion to compute, and that they run along
// different axes.
bool dual_fit(const PassPlan& a, const PassPlan& b)
{
    if (tune(TUNE_OCC_DUAL) == 0 || !a.sweep || !b.sweep || a.occ_mode < 0 || a.occ_mode != b.occ_mode || a.occ_enqueued || b.occ_enqueued) return false;
    if (a.p.axis == b.p.axis || a.f_buf == b.f_buf) return false;
    if (memcmp(a.p.a.uvw_off, b.p.a.uvw_off, sizeof(a.p.a.uvw_off))) return false;
    if (a.occ_mode == PASS_CHANGE && memcmp(a.p.r.uvw_off, b.p.r.uvw_off, sizeof(a.p.r.uvw_off))) return false;
    return true;
}

// Both passes' occlusion in one launch: each pass's empty-block flags, work list and block ranks as for its own launch
// (the sweeps and the cache entries need them), the work units' flags and list, then k_light_occlusion<..., DUAL> over the
// virtual pass along z.
int enqueue_dual_occlusion(tbrm_resources* r, const PassPlan& pa, const PassPlan& pb)
{
    hipStream_t s = r->occ_stream;
    if (int e2 = order_behind_inputs(r)) return e2;
    const PassPlan* const plans[2] = {&pa, &pb};
    const int axc = 2;
    // the virtual pass: along z, lanes over x and y, whatever the two pass axes are (tbrm_internal.h DualOcc)
    ChunkParams pc = pa.p;
    pc.axis = axc;
    const int dim_u = axc == 0 ? 1 : 0, dim_v = axc == 2 ? 1 : 2;
    pc.W = pc.lv_dims[dim_u];
    pc.H = pc.lv_dims[dim_v];
    pc.dir = 1;
    pc.j0 = 0;
    pc.n_steps = pc.lv_dims[axc];
    pc.pass_start = 0;
    pc.pass_slices = pc.chunk_slices = pc.n_steps;
    pc.occ_blocks_x = ceil_div(pc.W, 16);
    pc.occ_blocks_y = ceil_div(pc.H, 16);
    pc.occ_groups = ceil_div(pc.n_steps, kOccSlices);
    pc.roi_by0 = 0;
    pc.roi_by1 = pc.occ_blocks_y;
    pc.compact = 1;
    const size_t units = (size_t) pc.occ_groups * pc.occ_blocks_y * pc.occ_blocks_x;
    if (!pa.lists || !pb.lists) return fail(TBRM_ERR_INVALID_ARG, "a pass that computes occlusion has no block lists");
    // the work units' flags and list: a function of the two passes' lists — the handle's, if this pair has been here
    BlockLists* const ul = block_lists_for_dual(r, pa.lists, pb.lists, units);
    if (!ul) return TBRM_ERR_OUT_OF_MEMORY;
    DualOcc d{};
    d.on = 1;
    // the buffers about to be overwritten may still be read by earlier sweeps
    {
        uint64_t op = 0;
        if (int e2 = note_readers(r, pa, &op)) return e2;
        if (int e2 = note_readers(r, pb, &op)) return e2;
        if (int e2 = wait_for_readers(r, op)) return e2;
    }
    for (int k = 0; k < 2; ++k) {
        const PassPlan& plan = *plans[k];
        FactorScratch& f = r->f_scratch[plan.f_buf];
        FactorEntry* const e = plan.f_hit[0] ? nullptr : plan.f_entry[0];
        if (int e2 = enqueue_block_lists(r, plan)) return e2;
        DualPass& P = d.pass[k];
        P.axis = plan.p.axis; P.start = plan.start; P.dir = plan.dir;
        P.blocks_x = plan.p.occ_blocks_x; P.blocks_y = plan.p.occ_blocks_y;
        P.step100[0] = plan.p.a.step100; P.step100[1] = plan.p.r.step100;
        P.fs_keep[0] = e ? e->base : nullptr;
        P.fs_cap[0] = e ? (uint32_t) e->cap_blocks : 0u;
        P.fs_spill[0] = f.store[0];
        P.fs_keep[1] = nullptr;
        P.fs_cap[1] = 0;
        P.fs_spill[1] = f.store[1];
        P.fs_slot = plan.lists->slot;
        P.flags = plan.lists->flags;
    }
    if (!ul->enqueued) {
        pc.occ_flags_out = ul->flags;
        pc.occ_list_out = ul->list;
        pc.occ_count_out = ul->count;
        pc.occ_slot_out = nullptr;
        pc.occ_count_host = nullptr;
        HIP_TRY(launch_unit_flags(pc, d, s));
        ul->enqueued = true;
        ++r->lists_launches;
    }
    pc.occ_flags = nullptr;
    pc.occ_list = ul->list;
    pc.occ_count = ul->count;
    pc.occ_grid_cap = 0;
    HIP_TRY(launch_light_occlusion(pc, pa.occ_mode, s, &d));
    ++r->dual_launches;
    for (int k = 0; k < 2; ++k) {
        const PassPlan& plan = *plans[k];
        FactorEntry* const e = plan.f_hit[0] ? nullptr : plan.f_entry[0];
        HIP_TRY(hipEventRecord(r->f_scratch[plan.f_buf].ev_ready, s));
        if (e) {
            HIP_TRY(hipEventRecord(e->ev_filled, s));
            e->enqueued = true;
        }
        plan.occ_enqueued = true;
    }
    return TBRM_OK;
}

// The sweep of a sweep pass on the handle's stream, behind the occlusion it consumes
static int enqueue_sweep(tbrm_resources* r, const PassPlan& plan)
{
    if (int e = enqueue_sweep_occlusion(r, plan)) return e;
    FactorScratch& f = r->f_scratch[plan.f_buf];
    const int ns = plan.n_streams();
    if (plan.occ_mode >= 0) HIP_TRY(hipStreamWaitEvent(r->stream, f.ev_ready, 0));
    for (int si = 0; si < ns; ++si)
        if (plan.f_hit[si]) HIP_TRY(hipStreamWaitEvent(r->stream, plan.f_entry[si]->ev_filled, 0)); // (it may still be being filled)
    ChunkParams p = plan.p;
    p.j0 = plan.start;
    p.n_steps = plan.D;
    p.first_chunk = plan.pass_begins_here ? 1 : 0; // (a slab behind the first continues from the planes it was handed)
    p.occ_phase = 0;
    p.a.plane_in = plan_plane(r, 0, 0); p.a.plane_out = plan_plane(r, 1, 0);
    p.r.plane_in = plan_plane(r, 0, 1); p.r.plane_out = plan_plane(r, 1, 1);
    p.ones = r->d_ones;
    ChunkStream* const streams[2] = {&p.a, &p.r};
    for (int si = 0; si < ns; ++si) {
        FactorEntry* const e = plan.f_entry[si];
        ChunkStream& st = *streams[si];
        if (plan.f_hit[si]) { // every live block is in the entry
            st.fs_keep = e->base; st.fs_cap = (uint32_t) e->cap_blocks; st.fs_spill = nullptr; st.fs_slot = e->lists->slot;
        } else { // computed by this pass: stream a into its entry (if it has one) and the scratch, stream r into the scratch,
                 // both under the ranks of the jointly computed work list
            FactorEntry* const filled = plan.f_entry[0];
            st.fs_keep = (si == 0 && filled) ? filled->base : nullptr;
            st.fs_cap = (si == 0 && filled) ? (uint32_t) filled->cap_blocks : 0u;
            st.fs_spill = f.store[si];
            st.fs_slot = plan.lists->slot;
        }
    }
    SweepParams q = plan.sq;
    q.rec[0] = r->sweep_rec[0];
    q.rec[1] = r->sweep_rec[1];
    q.ticket = r->sweep_ticket;
    q.error = r->sweep_error;
    if (q.r_from_records) {
        // the removed light's planes first: one stream in its own tile order, the light volume untouched, its hand-off
        // records (which the fused launch reads instead of waiting for them) in the second buffer
        ChunkParams pr1 = p;
        pr1.a = p.r;
        SweepParams q1 = q;
        q1.sx = q.r_sx; q1.sy = q.r_sy; q1.hx = q.r_hx; q1.hy = q.r_hy;
        q1.r_from_records = 0;
        q1.rec[0] = r->sweep_rec[1];
        q1.rec[1] = nullptr;
        q1.stamps = nullptr;
        q1.debug &= ~2;
        if (int e = next_sweep_epoch(r, q1.epoch, 2)) return e;
        HIP_TRY(launch_light_sweep(pr1, q1, PASS_PLANES, r->stream));
        ++r->launches[0];
        ++r->sweep_launches;
        q.r_epoch = q1.epoch;
    }
    if (int e = next_sweep_epoch(r, q.epoch)) return e;
    if (q.lv_f32) // float hand-off words carry no tag: "not published yet" is written over the launch's records first
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) r->sweep_rec[0], 0xffffffffu, plan.rec_words, r->stream));
    q.stamps = nullptr;
    if (q.debug & 2) { // diagnostics: per-tile time stamps of this launch (printed by tbrm_flush)
        const int tiles = p.tiles_x * p.tiles_y;
        if (tiles > r->sweep_stamp_tiles || !r->sweep_stamps) {
            drain_streams(r);
            (void) hipFree(r->sweep_stamps);
            r->sweep_stamps = nullptr;
            HIP_TRY(hipMalloc((void**) &r->sweep_stamps, (size_t) tiles * 4 * sizeof(unsigned long long)));
        }
        r->sweep_stamp_tiles = tiles;
        r->sweep_stamp_tx = p.tiles_x;
        r->sweep_stamp_sx = q.sx;
        r->sweep_stamp_sy = q.sy;
        q.stamps = r->sweep_stamps;
    }
    HIP_TRY(launch_light_sweep(p, q, plan.mode, r->stream));
    ++r->launches[0];
    ++r->sweep_launches;
    // who read what: later operators wait for this operator's "sweeps done" event (wait_for_readers); the buffers' own events
    // are recorded only where a later pass of THIS operator could take the buffer again (an operator of more than two sweep
    // passes: every marker between two dependent kernels costs the stream a few microseconds)
    if (r->op_many_passes) HIP_TRY(hipEventRecord(f.ev_idle, r->stream));
    f.used = true;
    f.last_read_op = r->op_serial;
    f.idle_recorded = r->op_many_passes;
    for (int si = 0; si < ns; ++si)
        if (FactorEntry* const e = plan.f_entry[si]) {
            if (r->op_many_passes) HIP_TRY(hipEventRecord(e->ev_idle, r->stream));
            e->read_yet = true;
            e->last_read_op = r->op_serial;
            e->idle_recorded = r->op_many_passes;
        }
    return TBRM_OK;
}

// Two Add passes of DIFFERENT lights that leave the same cube face as ONE sweep (PASS_ADD2; SURVEY.md 8f N4, the multi-light
// optimisation the reference lists as not done, Readme.md:186-187): both streams share the slice loop, its latency chain, the
// hand-off words and the light volume's read-modify-write (light a's, then light b's on its result: exactly pass a followed by
// pass b). Each stream keeps its own factors (the lights' own occlusion launches, the cache entries); `fit` is the two
// streams' common tile order and reach (sweep_fit, not two_way).
static int enqueue_sweep_pair(tbrm_resources* r, const PassPlan& pa, const PassPlan& pb, const SweepFit& fit)
{
    const PassPlan* const plans[2] = {&pa, &pb};
    for (const PassPlan* plan : plans) {
        if (int e = enqueue_sweep_occlusion(r, *plan)) return e;
        if (plan->occ_mode >= 0) HIP_TRY(hipStreamWaitEvent(r->stream, r->f_scratch[plan->f_buf].ev_ready, 0));
        if (plan->f_hit[0]) HIP_TRY(hipStreamWaitEvent(r->stream, plan->f_entry[0]->ev_filled, 0));
    }
    ChunkParams p = pa.p;
    p.r = pb.p.a;
    p.b_added2 = pb.p.b_added;
    p.j0 = pa.start;
    p.n_steps = pa.D;
    p.first_chunk = 1;
    p.occ_phase = 0;
    p.a.plane_in = plan_plane(r, 0, 0); p.a.plane_out = plan_plane(r, 1, 0);
    p.r.plane_in = plan_plane(r, 0, 1); p.r.plane_out = plan_plane(r, 1, 1);
    p.ones = r->d_ones;
    ChunkStream* const streams[2] = {&p.a, &p.r};
    for (int si = 0; si < 2; ++si) { // each stream from its own pass's factors (enqueue_sweep, stream a)
        const PassPlan& plan = *plans[si];
        FactorScratch& f = r->f_scratch[plan.f_buf];
        FactorEntry* const e = plan.f_entry[0];
        ChunkStream& st = *streams[si];
        if (plan.f_hit[0]) { st.fs_keep = e->base; st.fs_cap = (uint32_t) e->cap_blocks; st.fs_spill = nullptr; st.fs_slot = e->lists->slot; }
        else {
            st.fs_keep = e ? e->base : nullptr;
            st.fs_cap = e ? (uint32_t) e->cap_blocks : 0u;
            st.fs_spill = f.store[0];
            st.fs_slot = plan.lists->slot;
        }
    }
    SweepParams q = pa.sq;
    q.sx = fit.sx; q.sy = fit.sy; q.hx = fit.hx; q.hy = fit.hy;
    q.r_from_records = 0;
    const size_t words = (size_t) pa.D * p.tiles_x * p.tiles_y * (size_t) sweep_record_words(fit.hx, fit.hy, q.tile_rows);
    if (words >= ((size_t) 1 << 32)) return fail(TBRM_ERR_UNSUPPORTED, "hand-off records too large");
    if (int e = ensure_sweep(r, std::max<size_t>(words, 1), 0)) return e;
    q.rec[0] = r->sweep_rec[0];
    q.rec[1] = r->sweep_rec[1];
    q.ticket = r->sweep_ticket;
    q.error = r->sweep_error;
    q.stamps = nullptr;
    q.debug &= ~2;
    if (int e = next_sweep_epoch(r, q.epoch)) return e;
    HIP_TRY(launch_light_sweep(p, q, PASS_ADD2, r->stream));
    ++r->launches[0];
    ++r->sweep_launches;
    ++r->pair_sweeps;
    for (const PassPlan* plan : plans) {
        FactorScratch& f = r->f_scratch[plan->f_buf];
        if (r->op_many_passes) HIP_TRY(hipEventRecord(f.ev_idle, r->stream));
        f.used = true;
        f.last_read_op = r->op_serial;
        f.idle_recorded = r->op_many_passes;
        if (FactorEntry* const e = plan->f_entry[0]) {
            if (r->op_many_passes) HIP_TRY(hipEventRecord(e->ev_idle, r->stream));
            e->read_yet = true;
            e->last_read_op = r->op_serial;
            e->idle_recorded = r->op_many_passes;
        }
    }
    return TBRM_OK;
}

static bool plan_has_occlusion(const PassPlan& plan) { return !plan.sliced && !plan.sweep && plan.n_chunks > 0; }

static int enqueue_plan_chunk_impl(tbrm_resources* r, const PassPlan& plan, int c, const PassPlan* next);

// Nothing may outlive a failed operator on the second stream: an occlusion launched beside the chain would still be reading
// the data volume and the skipping metadata when the caller uploads or frees them, and its buffer would keep a label that
// no chain will ever wait for.
void quiesce_occ_stream(tbrm_resources* r)
{
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream);
    for (int b = 0; b < 2; ++b) {
        r->occ_slot_async[b] = false;
        r->occ_slot[b] = tbrm_resources::OccSlot{};
    }
}

// Enqueues chunk c of the plan: in front of a span's first chunk the occlusion of the span unless it is already under way,
// and the occlusion of the span after it (of this plan, or the first of `next`) beside this span's chain; then the chain.
int enqueue_plan_chunk(tbrm_resources* r, const PassPlan& plan, int c, const PassPlan* next)
{
    const int e = enqueue_plan_chunk_impl(r, plan, c, next);
    if (e != TBRM_OK) quiesce_occ_stream(r);
    return e;
}

static int enqueue_plan_chunk_impl(tbrm_resources* r, const PassPlan& plan, int c, const PassPlan* next)
{
    if (plan.sliced) {
        PropParams sp = plan.slice_params;
        const int j = plan.start + c * plan.dir;
        const int e = (j % 2 == 0) ? 0 : 1, ax = plan.p.axis;
        sp.loop = j;
        if (plan.mode == PASS_ADD) {
            sp.a.read = r->d_buf[ax][e];
            sp.a.write = r->d_buf[ax][1 - e];
        } else {
            sp.r.read = r->d_buf[ax][e];
            sp.r.write = r->d_buf[ax][1 - e];
            sp.a.read = r->d_buf[ax][2 + e];
            sp.a.write = r->d_buf[ax][3 - e];
        }
        HIP_TRY(launch_propagate_slice(sp, plan.mode != PASS_ADD, r->stream));
        ++r->launches[1];
        return TBRM_OK;
    }
    if (plan.sweep) return enqueue_sweep(r, plan);
    ChunkParams p = plan.p;
    const int M = plan.M, D = plan.D, W = p.W, H = p.H;
    const int sp = (c * M) / plan.S;
    const SpanRange q = span_range(plan, sp);
    const size_t slice_elems = (size_t) W * H;
    const int ns = plan.n_streams();
    OccStore* const fs = &r->occ_tmp[plan.serial & 1][0];
    auto holds = [&](int b) { return r->occ_slot[b].plan_serial == plan.serial && r->occ_slot[b].span == sp; };
    int ob = holds(0) ? 0 : (holds(1) ? 1 : -1); // the buffer with this span's occlusion
    if (c == q.c0) {
        if (ob < 0) {
            ob = r->occ_last ^ 1;
            if (int e = enqueue_occlusion(r, plan, sp, ob, false)) return e;
        }
        if (r->occ_slot_async[ob]) {
            HIP_TRY(hipStreamWaitEvent(r->stream, r->occ_ev_ready[ob], 0));
            r->occ_slot_async[ob] = false;
        }
        r->occ_last = ob;
        // the span after this one — beside a chain that propagates one stream: the LDS of a two-stream chain leaves an
        // occlusion workgroup no room on its CU — and one tile per CU at most: with several rounds of tiles the chain's own
        // workgroups are what fills a CU's spare slots (1024^3: 16.2 ms per Change one after the other, 17.1 beside)
        if (tune(TUNE_OCC_OVERLAP) > 0 && ns == 1 && p.tiles_x * p.tiles_y <= r->n_cus) {
            const PassPlan* np = sp + 1 < plan.n_spans ? &plan : (next && plan_has_occlusion(*next) ? next : nullptr);
            if (np) {
                // only when the requested occlusion workgroups per CU all fit beside this plan's chain workgroup: every one of
                // them has to be resident from the start (one that waits takes the slot the next chain launch needs), and
                // fewer than two per CU do not finish a span's occlusion in the time of its chain (the 72 x 48 planes of a
                // cached Change leave room for one: 2.07 ms beside, 1.90 one after the other)
                ChunkParams full = plan.p;
                full.n_steps = M;
                full.j0 = plan.start;
                const size_t chain_lds = chunk_lds_bytes(full, plan.mode, r->lv_fmt), occ_lds = occlusion_lds_bytes(np->p) + 2560;
                const int room = chain_lds < 160 * 1024 ? (int) ((160 * 1024 - chain_lds) / occ_lds) : 0;
                const int want = std::min(tune(TUNE_OCC_OVERLAP), 2); // (two per CU beside a chain: measured)
                const int wgs = room >= want ? want : 0;
                if (wgs > 0)
                    if (int e = enqueue_occlusion(r, *np, np == &plan ? sp + 1 : 0, ob ^ 1, true, wgs)) return e;
            }
        }
    }
    if (ob < 0) return fail(TBRM_ERR_INVALID_ARG, "chunk %d enqueued before the first chunk of its span", c);
    const int k0 = c * M - q.s0; // first slice of the chunk within the span
    p.n_steps = std::min(M, D - c * M);
    p.j0 = plan.start + c * M * plan.dir;
    p.first_chunk = c == 0 && plan.pass_begins_here;
    p.a.plane_in = plan_plane(r, c, 0); p.a.plane_out = plan_plane(r, c + 1, 0);
    p.r.plane_in = plan_plane(r, c, 1); p.r.plane_out = plan_plane(r, c + 1, 1);
    ChunkStream* const streams[2] = {&p.a, &p.r};
    const uint8_t* const chunk_flags = q.sparse ? fs->flags + (size_t) sp * plan.flags_per_span + (size_t) (k0 / kOccSlices) * plan.flags_per_group : nullptr;
    for (int si = 0; si < ns; ++si) {
        streams[si]->occ_base = r->occ_tmp[ob][si].base;
        streams[si]->occ_off = (uint32_t) (kPlaneGuard + (size_t) k0 * slice_elems);
        streams[si]->occ_flags = chunk_flags;
    }
    p.occ_phase = k0 % kOccSlices;
    p.occ_list = nullptr;
    p.occ_count = nullptr;
    p.occ_flags = nullptr;
    HIP_TRY(launch_light_chain(p, plan.mode, r->lv_fmt, r->stream));
    ++r->launches[0];
    return TBRM_OK;
}

// the reference's structure: one launch per slice (LightingShaders.cpp:132-158 / :289-318)
int enqueue_pass_sliced(tbrm_resources* r, PropParams p, const tbrm_light_pass& pa, const tbrm_light_pass* pr)
{
    const bool change = pr != nullptr;
    const size_t npx = (size_t) pa.td[0] * pa.td[1];
    const int ax = pa.axis;
    if (!change) {
        HIP_TRY(launch_fill(r->d_buf[ax][0], r->lv_fmt, npx, pa.light_alpha, r->stream));
        HIP_TRY(launch_fill(r->d_buf[ax][1], r->lv_fmt, npx, pa.light_alpha, r->stream));
    } else {
        HIP_TRY(launch_fill(r->d_buf[ax][0], r->lv_fmt, npx, pr->light_alpha, r->stream));
        HIP_TRY(launch_fill(r->d_buf[ax][1], r->lv_fmt, npx, pr->light_alpha, r->stream));
        HIP_TRY(launch_fill(r->d_buf[ax][2], r->lv_fmt, npx, pa.light_alpha, r->stream));
        HIP_TRY(launch_fill(r->d_buf[ax][3], r->lv_fmt, npx, pa.light_alpha, r->stream));
    }
    p.axis = ax;
    for (int c = 0; c < 3; ++c) p.td[c] = pa.td[c];
    fill_stream(p.a, pa);
    if (change) fill_stream(p.r, *pr);
    for (int j = pa.start; j != pa.stop; j += pa.dir) {
        p.loop = j;
        const int e = (j % 2 == 0) ? 0 : 1; // switch read and write buffers each slice
        if (!change) {
            p.a.read = r->d_buf[ax][e];
            p.a.write = r->d_buf[ax][1 - e];
        } else {
            p.r.read = r->d_buf[ax][e];
            p.r.write = r->d_buf[ax][1 - e];
            p.a.read = r->d_buf[ax][2 + e];
            p.a.write = r->d_buf[ax][3 - e];
        }
        HIP_TRY(launch_propagate_slice(p, change, r->stream));
        ++r->launches[1];
    }
    return TBRM_OK;
}

// One axis pass of an operator as the entry points hand it over: the added stream, optionally a second stream (Change: the
// removed light; PASS_ADD2: a second added light).
struct PassSpec {
    tbrm_light_pass a{}, r{};
    bool two = false;
    int mode = PASS_ADD;
    float b_added = 0.0f, b_added2 = 0.0f;
};

// Runs the axis passes of one operator in order. Every pass is planned before anything is enqueued: a pass the chunk
// kernels decline takes the one-slice-per-launch path, any other planning failure leaves the light volume untouched.
// diagnostics (sweep_debug bit 2): where the host's time goes while an operator is enqueued
struct HostProbe {
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t;
    std::string line;
    explicit HostProbe(const char* w) : what(w), on((tune(TUNE_SWEEP_DEBUG) & 4) != 0), t(std::chrono::steady_clock::now()) {}
    void lap(const char* name)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof(buf), " %s %.0f us", name, std::chrono::duration<double, std::micro>(now - t).count());
        line += buf;
        t = now;
    }
    ~HostProbe() { if (on) fprintf(stderr, "[tbrm host] %s:%s\n", what, line.c_str()); }
};

// partner (optional, one entry per spec): the spec that is swept TOGETHER with this one (enqueue_sweep_pair), -1: none. The
// partner's own turn is skipped.
int run_passes(tbrm_resources* r, const PropParams& base, std::vector<PassSpec> specs, std::vector<int> partner = {})
{
    if (int e = sweep_failed(r)) return e; // (an earlier sweep left the light volume undefined: nothing to build on)
    ++r->op_serial;
    if (cache_usable(r)) // (the cache's keys depend on tbrm_resources::shell_transparent)
        if (int e = ensure_skipping(r)) return e;
    struct Unpin { // planning pins cache entries (use_kept / kept_new): released on every way out
        tbrm_resources* r;
        ~Unpin() { unpin_kept(r); }
    } unpin{r};
    std::vector<PassPlan> plans;
    std::vector<char> chunked;
    HostProbe probe("run_passes");
    for (size_t i = 0; i < specs.size(); ++i) {
        const PassSpec q = specs[i];
        PassPlan plan;
        const int e = plan_pass(r, base, q.a, q.two ? &q.r : nullptr, q.b_added, nullptr, plan, q.mode, q.b_added2);
        if (e == TBRM_ERR_UNSUPPORTED && q.mode == PASS_ADD2) { // the pair has no chunked form: light a's pass, then light b's
            PassSpec first = q, second = q;
            first.two = second.two = false;
            first.mode = second.mode = PASS_ADD;
            second.a = q.r;
            second.b_added = q.b_added2;
            specs[i] = first;
            specs.insert(specs.begin() + (long) i + 1, second);
            --i;
            continue;
        }
        if (e != TBRM_OK && e != TBRM_ERR_UNSUPPORTED) return e;
        plans.push_back(plan);
        chunked.push_back(e == TBRM_OK ? 1 : 0);
        probe.lap("plan");
    }
    r->op_many_passes = specs.size() > 2;
    std::vector<char> is_second(specs.size(), 0);
    if (partner.size() != specs.size()) partner.assign(specs.size(), -1); // (also after an ADD2 spec was split: chain batches carry none)
    for (size_t i = 0; i < specs.size(); ++i) {
        const int j = partner[i];
        // a pair needs both passes on the sweep; anything else: each on its own
        if (j < 0 || (size_t) j >= specs.size() || !chunked[i] || !chunked[(size_t) j] || !plans[i].sweep || !plans[(size_t) j].sweep) partner[i] = -1;
        else is_second[(size_t) j] = 1;
    }
    bool any_pair = false;
    for (int j : partner) any_pair = any_pair || j >= 0;
    if (any_pair) // (a group of two lights: four passes, four scratch buffers — every occlusion can go first)
        for (size_t k = 0; k < specs.size(); ++k) {
            if (!chunked[k]) continue;
            const int e = (k + 1 < specs.size() && chunked[k + 1] && dual_fit(plans[k], plans[k + 1])) ? enqueue_dual_occlusion(r, plans[k], plans[k + 1])
                                                                                                       : enqueue_sweep_occlusion(r, plans[k]);
            if (e) { quiesce_occ_stream(r); return e; }
        }
    for (size_t i = 0; i < specs.size(); ++i) {
        const PassSpec& q = specs[i];
        if (is_second[i]) continue; // (swept together with its partner)
        if (partner[i] >= 0) {
            const size_t j = (size_t) partner[i];
            SweepFit fit;
            if (sweep_fit(r, specs[i].a, &specs[j].a, PASS_CHANGE, fit) && !fit.two_way) {
                if (int e = enqueue_sweep_pair(r, plans[i], plans[j], fit)) { quiesce_occ_stream(r); return e; }
                r->passes[0] += 2;
                probe.lap("pair");
                continue;
            }
            is_second[j] = 0; // (does not fit after all: both on their own, in the order given)
        }
        if (!chunked[i]) {
            PropParams p = base;
            p.b_added = q.b_added;
            if (int e = enqueue_pass_sliced(r, p, q.a, q.two ? &q.r : nullptr)) return e;
            ++r->passes[2];
            continue;
        }
        ++r->passes[plans[i].sweep ? 0 : 1];
        const PassPlan* next = i + 1 < specs.size() && chunked[i + 1] ? &plans[i + 1] : nullptr;
        // sweep passes: this pass's occlusion, and the next pass's behind it on the occlusion stream, so that it runs beside
        // this pass's sweep — the two passes of a light in ONE launch where they sample the same positions (dual_fit)
        auto occlusion_of = [&](size_t k) -> int {
            if (k >= specs.size() || !chunked[k]) return TBRM_OK;
            if (k + 1 < specs.size() && chunked[k + 1] && dual_fit(plans[k], plans[k + 1])) return enqueue_dual_occlusion(r, plans[k], plans[k + 1]);
            return enqueue_sweep_occlusion(r, plans[k]);
        };
        if (int e = occlusion_of(i)) { quiesce_occ_stream(r); return e; }
        if (int e = occlusion_of(i + 1)) { quiesce_occ_stream(r); return e; }
        probe.lap("occlusion");
        for (int c = 0; c < plans[i].n_chunks; ++c)
            if (int e = enqueue_plan_chunk(r, plans[i], c, next)) return e; // (enqueue_plan_chunk has drained the second stream)
        probe.lap("pass");
    }
    if (r->occ_stream) { // "this operator's sweeps are done" (wait_for_readers)
        const int k = (int) (r->op_serial % tbrm_resources::kOpEvents);
        HIP_TRY(hipEventRecord(r->op_done[k], r->stream));
        r->op_done_serial[k] = r->op_serial;
    }
    return TBRM_OK;
}

// the axis passes of AddDirLightToSingleLightVolume_RenderThread (LightingShaders.cpp:35-166) appended to `specs`
static void add_light_specs(const tbrm_resources* r, const tbrm_dir_light_params& light, bool added, const tbrm_world_params& world,
                            std::vector<PassSpec>& specs)
{
    tbrm_light_pass passes[2];
    int n = 0;
    if (!host_light_passes(light, world, r->lv_dims, r->desc.border_mode, passes, &n)) return; // :41-46
    for (int i = 0; i < n; ++i) { // breaks on weight == 0 (:65,:94)
        PassSpec q;
        q.a = passes[i];
        q.b_added = added ? 1.0f : -1.0f;
        specs.push_back(q);
    }
}

// AddDirLightToSingleLightVolume_RenderThread (LightingShaders.cpp:35-166)
int enqueue_add(tbrm_resources* r, const tbrm_dir_light_params& light, bool added, const tbrm_world_params& world)
{
    std::vector<PassSpec> specs;
    add_light_specs(r, light, added, world, specs);
    return run_passes(r, base_prop_params(r, world), specs);
}

// Several AddDirLightToSingleLightVolume calls as one (the multi-light optimisation of the Sunden/Ropinski
// scheme the reference left out, Readme.md:166,186-187). The axis passes of all lights are collected; two passes of
// different lights that leave the same cube face (same axis, same direction) share one slice loop — the data volume's
// bricks, the plane geometry and the per-chunk overhead are paid once for both (PASS_ADD2: light a's read-modify-write,
// then light b's on its result, exactly as if pass a and then pass b had run over the volume). Passes are taken in
// the lights' order; each pairs with the first later pass of the same face. The order of the per-voxel updates thus
// differs from adding the lights one after the other; `schedule` (4 ints per entry: light and pass of a, light and
// pass of b or -1 -1) reports it so that a checker can replay it.
int enqueue_add_batch(tbrm_resources* r, const tbrm_dir_light_params* lights, int n_lights, bool added, const tbrm_world_params& world,
                      int32_t* schedule, int32_t* n_entries)
{
    struct Entry { int light, pass; tbrm_light_pass p; bool done; };
    std::vector<Entry> all;
    for (int i = 0; i < n_lights; ++i) {
        tbrm_light_pass passes[2];
        int n = 0;
        if (!host_light_passes(lights[i], world, r->lv_dims, r->desc.border_mode, passes, &n)) continue; // zero direction
        for (int k = 0; k < n; ++k) all.push_back(Entry{i, k, passes[k], false});
    }
    const PropParams base = base_prop_params(r, world);
    if (cache_usable(r))
        if (int e = ensure_skipping(r)) return e;
    const float b = added ? 1.0f : -1.0f;
    const bool pairing = tune(TUNE_LIGHT_BATCHING) != 0;
    // ---- every pass on the pipelined sweep (UNORM8 light volumes: the production path) --------------------------------------
    // Lights are taken two at a time: a group's passes (four at most) each get their own occlusion — one launch per light
    // (DualOcc) — into the four scratch buffers / their cache entries, and passes of the two lights that leave the same cube face
    // and pull the same way are swept TOGETHER (enqueue_sweep_pair). A light's partner is the later light it shares most faces
    // with. Groups of two keep every factor store of a pair resident without more scratch than single operators use.
    auto sweepable = [&](const tbrm_light_pass& q) {
        SweepFit sf;
        return sweep_fit(r, q, nullptr, PASS_ADD, sf) && ceil_div(q.td[2], 8) * 8 <= sweep_max_slices() && tune(TUNE_SPARSE_OCC) != 0 && tune(TUNE_OCC_LIST) != 0;
    };
    bool all_sweep = pairing && !all.empty() && r->lv_fmt == FMT_U8; // (the two-light sweep is built for UNORM8 light volumes)
    for (const Entry& e : all) all_sweep = all_sweep && sweepable(e.p);
    if (all_sweep) {
        auto pair_fits = [&](const tbrm_light_pass& x, const tbrm_light_pass& y) {
            SweepFit sf;
            return x.face == y.face && sweep_fit(r, x, &y, PASS_CHANGE, sf) && !sf.two_way;
        };
        std::vector<std::vector<size_t>> of_light((size_t) n_lights);
        for (size_t k = 0; k < all.size(); ++k) of_light[(size_t) all[k].light].push_back(k);
        std::vector<char> light_done((size_t) n_lights, 0);
        int entries = 0;
        for (int la = 0; la < n_lights; ++la) {
            if (light_done[(size_t) la] || of_light[(size_t) la].empty()) continue;
            light_done[(size_t) la] = 1;
            // the partner light: most pairs (each pass in at most one)
            int best = -1, best_n = 0;
            std::vector<std::pair<size_t, size_t>> best_pairs;
            for (int lb = la + 1; lb < n_lights; ++lb) {
                if (light_done[(size_t) lb] || of_light[(size_t) lb].empty()) continue;
                std::vector<std::pair<size_t, size_t>> pairs;
                std::vector<char> used_b(of_light[(size_t) lb].size(), 0);
                for (size_t ka : of_light[(size_t) la])
                    for (size_t ib = 0; ib < of_light[(size_t) lb].size(); ++ib) {
                        const size_t kb = of_light[(size_t) lb][ib];
                        if (!used_b[ib] && pair_fits(all[ka].p, all[kb].p)) { used_b[ib] = 1; pairs.emplace_back(ka, kb); break; }
                    }
                if ((int) pairs.size() > best_n) { best_n = (int) pairs.size(); best = lb; best_pairs = pairs; }
            }
            std::vector<size_t> group = of_light[(size_t) la];
            if (best >= 0) {
                light_done[(size_t) best] = 1;
                group.insert(group.end(), of_light[(size_t) best].begin(), of_light[(size_t) best].end());
            }
            std::vector<PassSpec> specs;
            std::vector<int> partner(group.size(), -1);
            for (size_t g = 0; g < group.size(); ++g) {
                PassSpec q;
                q.a = all[group[g]].p;
                q.b_added = b;
                specs.push_back(q);
                for (const auto& pr : best_pairs)
                    if (pr.first == group[g])
                        for (size_t h = 0; h < group.size(); ++h)
                            if (group[h] == pr.second) partner[g] = (int) h;
            }
            std::vector<char> second(group.size(), 0);
            for (int j : partner)
                if (j >= 0) second[(size_t) j] = 1;
            for (size_t g = 0; g < group.size(); ++g) { // the order the sweeps run in (run_passes)
                if (second[g]) continue;
                if (schedule) {
                    const Entry& ea = all[group[g]];
                    schedule[4 * entries + 0] = ea.light; schedule[4 * entries + 1] = ea.pass;
                    schedule[4 * entries + 2] = partner[g] >= 0 ? all[group[(size_t) partner[g]]].light : -1;
                    schedule[4 * entries + 3] = partner[g] >= 0 ? all[group[(size_t) partner[g]]].pass : -1;
                }
                ++entries;
            }
            if (n_entries) *n_entries = entries;
            if (int e = run_passes(r, base, specs, partner)) return e;
        }
        if (n_entries) *n_entries = entries;
        return TBRM_OK;
    }
    std::vector<PassSpec> specs;
    int entries = 0;
    for (size_t ia = 0; ia < all.size(); ++ia) {
        Entry& a = all[ia];
        if (a.done) continue;
        a.done = true;
        // Partner: a later pass of the same face whose previous-slice taps fall inside this pass's tap range or the other
        // way round. Two lights in one slice loop share the per-chunk overhead and the per-slice latency of the chain, but
        // every window has to cover both lights' taps: measured on MI355X (512^3, all pairs of the 8 config lights), a pair
        // only pays when the union of the two tap ranges is no wider than the wider of the two — then 0.3 to 0.5 ms per
        // paired pass (lights 1 and 7: 3.61 -> 2.53 ms for both passes); with diverging directions the wider windows and
        // shorter chunks cost up to 0.2 ms more than they save.
        // A pass that the pipelined sweep takes is not paired: a sweep of its own costs less than its half of a paired chain.
        auto kept = [&](const tbrm_light_pass& q) {
            SweepFit sf;
            return sweep_fit(r, q, nullptr, PASS_ADD, sf) && ceil_div(q.td[2], 8) * 8 <= sweep_max_slices() && tune(TUNE_SPARSE_OCC) != 0 && tune(TUNE_OCC_LIST) != 0;
        };
        Entry* partner = nullptr;
        ChunkFit fa;
        if (pairing && !kept(a.p) && chunk_fit(r, a.p, nullptr, fa)) {
            int best_area = INT32_MAX;
            for (size_t ib = ia + 1; ib < all.size(); ++ib) {
                Entry& b2 = all[ib];
                ChunkFit fb, fp;
                if (b2.done || b2.light == a.light || b2.p.face != a.p.face || kept(b2.p)) continue;
                if (!chunk_fit(r, b2.p, nullptr, fb) || !chunk_fit(r, a.p, &b2.p, fp)) continue;
                if (tune(TUNE_LIGHT_BATCHING) == 2) { partner = &b2; break; } // diagnostics: pair whatever fits
                const int sx = fp.tx.hi - fp.tx.lo, sy = fp.ty.hi - fp.ty.lo;
                const bool contained = sx <= std::max(fa.tx.hi - fa.tx.lo, fb.tx.hi - fb.tx.lo) && sy <= std::max(fa.ty.hi - fa.ty.lo, fb.ty.hi - fb.ty.lo);
                if (!contained || fp.M < std::min(fa.M, fb.M)) continue;
                if (sx * sy < best_area) { best_area = sx * sy; partner = &b2; }
            }
        }
        if (schedule) {
            schedule[4 * entries + 0] = a.light; schedule[4 * entries + 1] = a.pass;
            schedule[4 * entries + 2] = partner ? partner->light : -1; schedule[4 * entries + 3] = partner ? partner->pass : -1;
        }
        ++entries;
        PassSpec q;
        q.a = a.p;
        q.b_added = b;
        if (partner) {
            partner->done = true;
            q.r = partner->p;
            q.two = true;
            q.mode = PASS_ADD2;
            q.b_added2 = b;
        }
        specs.push_back(q);
    }
    if (n_entries) *n_entries = entries;
    return run_passes(r, base, specs);
}

// ChangeDirLightInSingleLightVolume_RenderThread (LightingShaders.cpp:168-326)
int enqueue_change(tbrm_resources* r, const tbrm_dir_light_params& removed, const tbrm_dir_light_params& added_light,
                   const tbrm_world_params& world)
{
    tbrm_light_pass rp[2], ap[2];
    int rn = 0, an = 0;
    const bool r_ok = host_light_passes(removed, world, r->lv_dims, r->desc.border_mode, rp, &rn);
    const bool a_ok = host_light_passes(added_light, world, r->lv_dims, r->desc.border_mode, ap, &an);
    if (!r_ok || !a_ok) return TBRM_OK; // :173-179
    const PropParams base = base_prop_params(r, world);
    std::vector<PassSpec> specs;
    if (rp[0].face != ap[0].face || rp[1].face != ap[1].face) { // :192-198: remove the old light, add the new one
        add_light_specs(r, removed, false, world, specs);
        add_light_specs(r, added_light, true, world, specs);
        return run_passes(r, base, specs);
    }
    for (int i = 0; i < 2; ++i) { // no break on weight 0 (:238)
        // Both streams dark (weight 0 on this axis for old and new light): buffers and borders are 0, every
        // propagated value is 0*(1-s) = 0 and |0-0| > 1e-3 never holds: the pass cannot touch the light volume.
        if (rp[i].light_alpha == 0.0f && ap[i].light_alpha == 0.0f && rp[i].border_light == 0.0f && ap[i].border_light == 0.0f)
            continue;
        PassSpec q;
        q.a = ap[i];
        q.r = rp[i];
        q.two = true;
        q.mode = PASS_CHANGE;
        specs.push_back(q);
    }
    return run_passes(r, base, specs);
}

} // namespace tbrm_host
