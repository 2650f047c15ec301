// tbrm_api.cpp — the C-ABI of include/tbrm.h over the gfx950 kernels.
//
// Plays the role of the reference's game-thread operator library + render-thread drivers:
//   URaymarchUtils::AddDirLightToSingleVolume / ChangeDirLightInSingleVolume / ClearResourceLightVolumes
//       Source/Raymarcher/Private/Util/RaymarchUtils.cpp:35-111
//   AddDirLightToSingleLightVolume_RenderThread / ChangeDirLightInSingleLightVolume_RenderThread
//       Source/Raymarcher/Private/Rendering/LightingShaders.cpp:35-326
//   ARaymarchVolume::InitializeRaymarchResources / FreeRaymarchResources
//       Source/Raymarcher/Private/Actor/RaymarchVolume.cpp:821-949
// Every call enqueues on the handle's HIP stream (FIFO, like ENQUEUE_RENDER_COMMAND) and returns; parameter
// structs are copied at call time (the reference captures them by value, RaymarchUtils.cpp:63-66).
// There is no CPU path: without a HIP device every data call fails with TBRM_ERR_NO_DEVICE.
#include "../../include/tbrm.h"
#include "tbrm_host_math.h"
#include "tbrm_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace tbrm;

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        const hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? TBRM_ERR_OUT_OF_MEMORY : TBRM_ERR_NO_DEVICE, "%s failed: %s", \
                #expr, hipGetErrorString(e_));                                                                \
    } while (0)

size_t format_bytes(int fmt) { return fmt == TBRM_FMT_G8 ? 1 : (fmt == TBRM_FMT_G16 ? 2 : 4); }

} // namespace

struct tbrm_resources {
    tbrm_resources_desc desc{};
    int32_t lv_dims[3]{};
    int lv_fmt = FMT_U8;
    int n_cus = 256;               // compute units of the device (chunk length heuristics)
    hipStream_t stream = nullptr;

    void* d_data = nullptr;        // bricked data volume; slab-resident handles: rebased so that global brick offsets apply
    size_t data_bytes = 0;
    bool has_volume = false;

    // Slab-resident handles (tbrm_resources_create_slab) hold only some z brick layers of the two volumes: layers
    // [lo, hi) contiguously, then one more layer holding a copy of layer `wrap_src` (what wrap addressing reaches from the
    // first / last slice; -1: none). d_data / d_light point lo * layer_bytes BEFORE the allocation, so a kernel that only
    // touches resident layers addresses them with the global brick offsets, unchanged.
    struct Residency { int lo = 0, hi = 0, wrap_src = -1; size_t layer_bytes = 0; void* alloc = nullptr; };
    bool resident = false;
    tbrm_slab owned{};
    Residency res_data, res_light;

    float4* d_tf = nullptr;
    float tf_host[1024]{};
    bool has_tf = false;

    tbrm_windowing_params win{0.5f, 1.0f, 1, 1};

    void* d_light = nullptr;
    size_t light_bytes = 0;        // linear size (what download/upload exchange)
    size_t light_bricked_bytes = 0;
    size_t data_bricked_bytes = 0;
    int dbn[3]{};                  // data volume bricks per axis
    int lbn[3]{};                  // light volume bricks per axis
    void* d_buf[3][4]{};           // the reference's read/write buffers (slice-per-launch fallback path)
    float* d_plane[4]{};           // chunk kernel: propagated-light planes, 2 per stream
    float* d_occ = nullptr;        // chunk kernel: page of ones + the occlusion plane stacks of a span (allocated on first use)
    size_t occ_elems = 0;
    uint8_t* d_occ_zero[2]{};      // empty-block flags of the two occlusion buffers
    uint32_t* d_occ_list = nullptr; // work lists of the pass (one uint32 per flag) followed by 4096 per-chunk counts
    size_t occ_zero_bytes = 0;

    // empty-space-skipping metadata
    int bn[3]{};
    float2* d_minmax = nullptr;
    uint32_t* d_empty = nullptr;
    uint8_t* d_dist[2]{};          // empty-space leaping: per-brick distance field (ping-pong of the separable passes; [0] is final)
    int* d_alpha_prefix = nullptr;
    bool minmax_valid = false, empty_valid = false;

    // Octree render mode: 4-level UNORM16 max pyramid (allocated by the first tbrm_generate_octree)
    uint16_t* d_octree[4]{};
    int oct_dims[4][3]{};
    bool octree_valid = false;

    unsigned long long* d_counter = nullptr;
    float* d_out = nullptr; // staging for the host-pointer raymarch variant
    size_t out_bytes = 0;

    struct SlabOp* slab_op = nullptr; // the slab-partitioned light operation in flight (tbrm_slab_*)

    hipEvent_t ev[2][2]{};
    bool ev_valid[2]{};
    uint64_t launches[3]{}; // chunk, slice, raymarch
};

namespace {

int bind(const tbrm_resources* r)
{
    HIP_TRY(hipSetDevice(r->desc.device));
    return TBRM_OK;
}

bool initialized(const tbrm_resources* r) { return r && r->has_volume && r->has_tf && r->d_light; }

VolumeDev data_view(const tbrm_resources* r)
{
    const tbrm_resources::Residency& q = r->res_data;
    return VolumeDev{r->d_data, r->desc.dim_x, r->desc.dim_y, r->desc.dim_z, r->desc.data_format, r->dbn[0], r->dbn[0] * r->dbn[1],
                     q.wrap_src, (q.hi - q.wrap_src) * 8};
}
WindowDev window_dev(const tbrm_resources* r)
{
    return WindowDev{r->win.center, r->win.width, r->win.low_cutoff ? 1.0f : 0.0f, r->win.high_cutoff ? 1.0f : 0.0f};
}

// The clip plane is inert for the propagation when every sample position (uvw + UVWOffset, inside
// [-1/min(res), 1+1/min(res)]^3) sits >= 2 light-volume voxels on the kept side: AlphaWeight then clamps to
// exactly 1 (AddDirLightShader.usf:105). Requires a unit direction (a zero direction gives weight 0.5).
int propagation_clip_mode(const float cc[3], const float cd[3], const int32_t lv[3])
{
    const double n2 = (double) cd[0] * cd[0] + (double) cd[1] * cd[1] + (double) cd[2] * cd[2];
    if (!(n2 > 0.98 && n2 < 1.02)) return 1;
    const int rmin = std::min({lv[0], lv[1], lv[2]});
    const double pad = 1.0 / (double) rmin + 0.01;
    const double dmin = host_min_plane_distance(cc, cd, -pad, 1.0 + pad);
    return (dmin * (double) rmin >= 2.0) ? 0 : 1;
}
// Raymarch positions stay within one step of the unit cube; IsCurPosClipped is never true when the whole
// box [-1,2]^3 is strictly on the kept side.
int raymarch_clip_mode(const float cc[3], const float cd[3])
{
    const double dmin = host_min_plane_distance(cc, cd, -1.0, 2.0);
    return (dmin > 1e-3) ? 0 : 1;
}

void fill_stream(PropStream& s, const tbrm_light_pass& p)
{
    s.border_light = p.border_light;
    s.off_u = p.prev_pixel_offset[0];
    s.off_v = p.prev_pixel_offset[1];
    s.uvw_off[0] = p.uvw_offset[0];
    s.uvw_off[1] = p.uvw_offset[1];
    s.uvw_off[2] = p.uvw_offset[2];
    s.step100 = p.step_size * 100.0f; // StepSize * VOLUME_DENSITY (AddDirLightShader.usf:112)
}

PropParams base_prop_params(const tbrm_resources* r, const tbrm_world_params& world)
{
    PropParams p{};
    p.data = data_view(r);
    p.data_border = host_data_border(r->win, r->desc.border_mode);
    p.tf = r->d_tf;
    p.win = window_dev(r);
    p.light = r->d_light;
    for (int c = 0; c < 3; ++c) p.lv_dims[c] = r->lv_dims[c];
    p.lv_bnx = r->lbn[0];
    p.lv_bnxy = r->lbn[0] * r->lbn[1];
    p.lv_fmt = r->lv_fmt;
    host_local_clipping(world, p.cc, p.cd);
    p.clip_mode = propagation_clip_mode(p.cc, p.cd, r->lv_dims);
    return p;
}

int begin_timed(tbrm_resources* r, int kind)
{
    HIP_TRY(hipEventRecord(r->ev[kind][0], r->stream));
    return TBRM_OK;
}
int end_timed(tbrm_resources* r, int kind)
{
    HIP_TRY(hipEventRecord(r->ev[kind][1], r->stream));
    r->ev_valid[kind] = true;
    return TBRM_OK;
}

int ensure_skipping(tbrm_resources* r);

// ---- chunked propagation (tbrm_light_kernels.hip) --------------------------------------------------------------

bool force_slice_kernel()
{
    const char* e = getenv("TBRM_FORCE_SLICE_KERNEL"); // read per call so tests can A/B the two kernels
    return e && e[0] == '1';
}
int chunk_steps_override()
{
    const char* e = getenv("TBRM_CHUNK_STEPS");
    return e ? atoi(e) : 0;
}

// Range of (tap index - pixel index) of the previous-slice bilinear fetch over one buffer axis, evaluated with the
// kernel's own fp32 sequence (texel_split of ((c+0.5)/size + offset)); hi includes the +1 tap.
struct TapRange { int lo = 0, hi = 0; bool ok = false; };
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

inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int ceil_div(int a, int b) { return -floor_div(-a, b); }
inline int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

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

// One axis pass (Add: stream a only; Change: a = added, r = removed) as a plan: everything that is constant over the
// pass, worked out once, and the chunks then enqueued one by one (plan_pass / enqueue_plan_chunk). A single-GPU pass
// enqueues all of them back to back; a slab-partitioned pass (tbrm_slab_*) stops after each chunk so that the host can
// exchange the propagated planes between ranks.
struct PassPlan {
    ChunkParams p{};
    int mode = PASS_ADD;        // PASS_ADD / PASS_CHANGE / PASS_ADD2
    bool two_streams() const { return mode != PASS_ADD; }
    int M = 0, S = 0;           // slices per chain chunk / per occlusion span
    int D = 0;                  // slices this handle runs (the whole pass, or its slab's part of a pass along z)
    int start = 0, dir = 1;     // first of them
    int n_chunks = 0, n_spans = 0;
    bool pass_begins_here = true; // chunk 0 starts from the cleared buffers' value (else from imported planes)
    bool sparse = false, work_list = false;
    size_t flags_per_group = 0, flags_per_span = 0, occ_off_a = 0, occ_off_r = 0;
    // slab-partitioned passes
    bool lateral = false;       // the slices contain the slab axis: every rank runs every chunk on its rows
    int first_chunk_of_pass = 0, chunks_of_pass = 0;
    // a slab-partitioned pass whose taps reach too far for the chunk kernels: one slice per "chunk" with the reference's
    // kernel structure (k_propagate_slice) on the read / write buffers, which then are the planes the host exchanges
    bool sliced = false;
    PropParams slice_params{};
    int halo_rows = 0;          // lateral: rows a slice's taps can reach beyond a slab
};

// why plan_pass last declined a pass (diagnostics of the slab entry points, which have no fallback)
thread_local const char* g_plan_note = "";
int declined(const char* why) { g_plan_note = why; return TBRM_ERR_UNSUPPORTED; }

// Chunk length of a pass (one stream: pr == null, else two) and the tap ranges its windows have to cover: the longest of
// 16/8/4/2 slices whose window (tile + steps * growth) and staged occlusion fit in LDS. false: the chunk kernels decline.
struct ChunkFit { int M = 0; TapRange tx, ty; };
bool chunk_fit(const tbrm_resources* r, const tbrm_light_pass& pa, const tbrm_light_pass* pr, ChunkFit& fit)
{
    g_plan_note = "";
    if (force_slice_kernel()) return declined("TBRM_FORCE_SLICE_KERNEL is set"), false;
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
        if (kChunkTile + cand * g <= kChunkMaxHull && chunk_lds_bytes(p, pr != nullptr, r->lv_fmt) <= 156 * 1024) { fit.M = cand; break; }
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

// pr == null: Add of pa (b_added = +-1). Else two streams: mode PASS_CHANGE (pa added, pr removed) or PASS_ADD2 (pa, then
// pr, both added with b_added / b_added2).
int plan_pass(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
              const tbrm_slab* slab, PassPlan& plan, int two_stream_mode = PASS_CHANGE, float b_added2 = 0.0f)
{
    g_plan_note = "";
    ChunkFit fit;
    if (!chunk_fit(r, pa, pr, fit)) {
        if (!slab || two_stream_mode == PASS_ADD2) return TBRM_ERR_UNSUPPORTED;
        return plan_pass_sliced(r, base, pa, pr, b_added, *slab, plan);
    }
    const bool change = pr != nullptr;
    const int W = pa.td[0], H = pa.td[1], D_pass = pa.td[2];
    plan = PassPlan{};
    plan.mode = change ? two_stream_mode : PASS_ADD;
    ChunkParams& p = plan.p;
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
    p.dx_lo = fit.tx.lo; p.dx_hi = fit.tx.hi; p.dy_lo = fit.ty.lo; p.dy_hi = fit.ty.hi;
    p.b_added = b_added;
    p.b_added2 = b_added2;
    fill_chunk_stream(p.a, pa, r->lv_fmt);
    if (change) fill_chunk_stream(p.r, *pr, r->lv_fmt);
    const int M = fit.M;
    plan.M = M;

    // what this handle runs: the whole pass, or (slab-partitioned) its rows of every slice / its slices of a pass along z
    plan.D = D_pass;
    plan.start = pa.start;
    plan.dir = pa.dir;
    p.tiles_x = ceil_div(W, kChunkTile);
    p.tiles_y = ceil_div(H, kChunkTile);
    p.tile_row0 = 0;
    p.occ_blocks_x = ceil_div(W, 16);
    p.occ_blocks_y = ceil_div(H, 16);
    p.roi_by0 = 0;
    p.roi_by1 = p.occ_blocks_y;
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
    if (const char* e = getenv("TBRM_OCC_SLICES")) S = atoi(e);
    S = std::max(M, (S / M) * M);

    // occlusion scratch, allocated on first use: ONE allocation = [page of ones | guard][stream a: S planes][guard]
    // [stream r: S planes][guard], so that the chain addresses every copy source as base + 32-bit offset
    size_t occ_elems = (size_t) S * W * H;
    while (S > M && (2 * occ_elems + 3 * kPlaneGuard) * sizeof(float) >= ((size_t) 1 << 32)) { S -= M; occ_elems = (size_t) S * W * H; }
    const size_t occ_total = 2 * occ_elems + 3 * kPlaneGuard;
    if (occ_total * sizeof(float) >= ((size_t) 1 << 32)) return declined("slice plane too large for the occlusion scratch");
    if (occ_elems > r->occ_elems) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->d_occ);
        r->d_occ = nullptr;
        r->occ_elems = 0;
        HIP_TRY(hipMalloc((void**) &r->d_occ, occ_total * sizeof(float)));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) r->d_occ, 0x3f800000, 1024, r->stream)); // the page of ones
        r->occ_elems = occ_elems;
    }
    plan.S = S;
    plan.n_spans = ceil_div(D, S);
    plan.occ_off_a = kPlaneGuard;
    plan.occ_off_r = kPlaneGuard + r->occ_elems + kPlaneGuard;

    // empty-block hand-off (needs the per-brick emptiness bits of the current TF/window): one flag per occlusion
    // workgroup of the whole pass, computed up front, and per span the ascending list of the workgroups with work
    plan.sparse = !getenv("TBRM_NO_SPARSE_OCC");
    plan.work_list = plan.sparse && !getenv("TBRM_NO_OCC_LIST");
    p.occ_groups = ceil_div(S, kOccSlices);
    plan.flags_per_group = (size_t) p.occ_blocks_y * p.occ_blocks_x;
    plan.flags_per_span = (size_t) p.occ_groups * plan.flags_per_group;
    if (plan.sparse) {
        if (plan.n_spans > 4096) return declined("too many occlusion spans");
        if (int e = ensure_skipping(r)) return e;
        const size_t zbytes = plan.flags_per_span * plan.n_spans;
        if (zbytes > r->occ_zero_bytes) {
            HIP_TRY(hipStreamSynchronize(r->stream));
            (void) hipFree(r->d_occ_zero[0]);
            (void) hipFree(r->d_occ_list);
            r->d_occ_zero[0] = nullptr;
            r->d_occ_list = nullptr;
            r->occ_zero_bytes = 0;
            HIP_TRY(hipMalloc((void**) &r->d_occ_zero[0], zbytes));
            HIP_TRY(hipMalloc((void**) &r->d_occ_list, zbytes * sizeof(uint32_t) + 4096 * sizeof(int))); // lists + counts
            r->occ_zero_bytes = zbytes;
        }
        p.empty_bits = r->d_empty;
        p.occ_flags_out = r->d_occ_zero[0];
        p.occ_list_out = plan.work_list ? r->d_occ_list : nullptr;
        p.occ_count_out = (int*) (r->d_occ_list + r->occ_zero_bytes);
        p.pass_start = plan.start;
        p.pass_slices = D;
        p.chunk_slices = S;
        HIP_TRY(launch_occ_flags(p, plan.mode, plan.n_spans, r->stream));
    }
    p.occ_base = r->d_occ;
    p.a.occ_next = r->d_occ + plan.occ_off_a;
    p.r.occ_next = r->d_occ + plan.occ_off_r;
    return TBRM_OK;
}

// The plane holding the propagated light of stream `si` (0: a, 1: r) BEFORE chunk `boundary` (boundary = n_chunks: after
// the last one): chunk c reads the planes of parity c & 1 and writes the others.
float* plan_plane(const tbrm_resources* r, int boundary, int si) { return r->d_plane[2 * si + (boundary & 1)] + kPlaneGuard; }

// Enqueues chunk c of the plan: the occlusion of its span first if the span starts here, then the chain.
int enqueue_plan_chunk(tbrm_resources* r, const PassPlan& plan, int c)
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
    ChunkParams p = plan.p;
    const int M = plan.M, S = plan.S, D = plan.D, W = p.W, H = p.H;
    const int sp = (c * M) / S;
    const int s0 = sp * S, sn = std::min(S, D - s0);
    const int c0 = s0 / M, c1 = ceil_div(s0 + sn, M);
    // the chain stages its window in groups of 4 pixels starting at tile_x - n*|dx_lo|: only when that is a multiple of 4
    // does a group never straddle two 16-pixel occlusion blocks (always true for full chunks of 16/8/4 slices)
    auto chunk_sparse_ok = [&](int cc) { return (std::min(M, D - cc * M) * -p.dx_lo) % 4 == 0; };
    bool span_sparse = plan.sparse;
    for (int cc = c0; cc < c1; ++cc) span_sparse = span_sparse && chunk_sparse_ok(cc); // else the whole span runs dense
    const bool work_list = plan.work_list;

    if (c == c0) { // occlusion of the span: fills {a,r}.occ_next with sn planes
        p.j0 = plan.start + s0 * plan.dir;
        p.n_steps = sn;
        p.occ_flags = nullptr;
        p.occ_list = span_sparse && work_list ? r->d_occ_list + (size_t) sp * plan.flags_per_span : nullptr;
        p.occ_count = span_sparse && work_list ? (const int*) (r->d_occ_list + r->occ_zero_bytes) + sp : nullptr;
        if (span_sparse && !work_list) p.occ_flags = r->d_occ_zero[0] + (size_t) sp * plan.flags_per_span;
        HIP_TRY(launch_light_occlusion(p, plan.mode, r->stream));
    }
    const int k0 = c * M - s0; // first slice of the chunk within the span
    p.n_steps = std::min(M, D - c * M);
    p.j0 = plan.start + c * M * plan.dir;
    p.first_chunk = c == 0 && plan.pass_begins_here;
    p.a.plane_in = plan_plane(r, c, 0); p.a.plane_out = plan_plane(r, c + 1, 0);
    p.r.plane_in = plan_plane(r, c, 1); p.r.plane_out = plan_plane(r, c + 1, 1);
    p.a.occ_off = (uint32_t) (plan.occ_off_a + (size_t) k0 * W * H);
    p.r.occ_off = (uint32_t) (plan.occ_off_r + (size_t) k0 * W * H);
    p.occ_phase = k0 % kOccSlices;
    p.occ_list = nullptr;
    p.occ_count = nullptr;
    p.occ_flags = span_sparse ? r->d_occ_zero[0] + (size_t) sp * plan.flags_per_span + (size_t) (k0 / kOccSlices) * plan.flags_per_group : nullptr;
    HIP_TRY(launch_light_chain(p, plan.mode, r->lv_fmt, r->stream));
    ++r->launches[0];
    return TBRM_OK;
}

int enqueue_pass_chunked(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr,
                         float b_added, int two_stream_mode = PASS_CHANGE, float b_added2 = 0.0f)
{
    PassPlan plan;
    if (int e = plan_pass(r, base, pa, pr, b_added, nullptr, plan, two_stream_mode, b_added2)) return e;
    for (int c = 0; c < plan.n_chunks; ++c)
        if (int e = enqueue_plan_chunk(r, plan, c)) return e;
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

int enqueue_pass(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added)
{
    const int e = enqueue_pass_chunked(r, base, pa, pr, b_added);
    if (e != TBRM_ERR_UNSUPPORTED) return e;
    PropParams p = base;
    p.b_added = b_added;
    return enqueue_pass_sliced(r, p, pa, pr);
}

// AddDirLightToSingleLightVolume_RenderThread (LightingShaders.cpp:35-166)
int enqueue_add(tbrm_resources* r, const tbrm_dir_light_params& light, bool added, const tbrm_world_params& world)
{
    tbrm_light_pass passes[2];
    int n = 0;
    if (!host_light_passes(light, world, r->lv_dims, r->desc.border_mode, passes, &n)) return TBRM_OK; // :41-46
    const PropParams base = base_prop_params(r, world);
    for (int i = 0; i < n; ++i) // breaks on weight == 0 (:65,:94)
        if (int e = enqueue_pass(r, base, passes[i], nullptr, added ? 1.0f : -1.0f)) return e;
    return TBRM_OK;
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
    const float b = added ? 1.0f : -1.0f;
    const bool pairing = !getenv("TBRM_NO_LIGHT_BATCHING");
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
        Entry* partner = nullptr;
        ChunkFit fa;
        if (pairing && chunk_fit(r, a.p, nullptr, fa)) {
            int best_area = INT32_MAX;
            for (size_t ib = ia + 1; ib < all.size(); ++ib) {
                Entry& b2 = all[ib];
                ChunkFit fb, fp;
                if (b2.done || b2.light == a.light || b2.p.face != a.p.face) continue;
                if (!chunk_fit(r, b2.p, nullptr, fb) || !chunk_fit(r, a.p, &b2.p, fp)) continue;
                if (getenv("TBRM_LIGHT_BATCHING_FORCE")) { partner = &b2; break; } // diagnostics: pair whatever fits
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
        if (partner) {
            partner->done = true;
            const int e = enqueue_pass_chunked(r, base, a.p, &partner->p, b, PASS_ADD2, b);
            if (e == TBRM_OK) continue;
            if (e != TBRM_ERR_UNSUPPORTED) return e;
            if (int e2 = enqueue_pass(r, base, a.p, nullptr, b)) return e2; // the pair does not fit one launch: a, then b
            if (int e2 = enqueue_pass(r, base, partner->p, nullptr, b)) return e2;
        } else if (int e = enqueue_pass(r, base, a.p, nullptr, b)) return e;
    }
    if (n_entries) *n_entries = entries;
    return TBRM_OK;
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
    if (rp[0].face != ap[0].face || rp[1].face != ap[1].face) { // :192-198
        const int e = enqueue_add(r, removed, false, world);
        if (e != TBRM_OK) return e;
        return enqueue_add(r, added_light, true, world);
    }
    const PropParams base = base_prop_params(r, world);
    for (int i = 0; i < 2; ++i) { // no break on weight 0 (:238)
        // Both streams dark (weight 0 on this axis for old and new light): buffers and borders are 0, every
        // propagated value is 0*(1-s) = 0 and |0-0| > 1e-3 never holds: the pass cannot touch the light volume.
        if (rp[i].light_alpha == 0.0f && ap[i].light_alpha == 0.0f && rp[i].border_light == 0.0f && ap[i].border_light == 0.0f)
            continue;
        if (int e = enqueue_pass(r, base, ap[i], &rp[i], 0.0f)) return e;
    }
    return TBRM_OK;
}

} // namespace

// A light operation taken apart for slab-partitioned execution: its axis passes, and the plan of the one being stepped.
struct SlabOp {
    tbrm_slab slab{};
    bool change = false;
    float b_added = 0.0f;
    int n = 0;
    tbrm_light_pass a[2]{}, r[2]{};
    PropParams base{};
    int current = -1; // pass being stepped
    PassPlan plan;
};

namespace {

RelayoutParams relayout_params(const void* src, void* dst, const int dims[3], const int bn[3], size_t elem, bool to_bricks)
{
    return RelayoutParams{src, dst, dims[0], dims[1], dims[2], bn[0], bn[0] * bn[1], bn[2], (int) elem, to_bricks ? 1 : 0};
}

int ensure_skipping(tbrm_resources* r)
{
    const int nb = r->bn[0] * r->bn[1] * r->bn[2];
    if (!r->minmax_valid) {
        // a brick's range covers its +1 apron: of a slab-resident volume the last resident layer has none (unless it is the
        // volume's last layer and the apron clamps onto it, or wraps onto a resident layer 0)
        const tbrm_resources::Residency& q = r->res_data;
        const bool clamp = r->desc.data_address_mode == TBRM_ADDRESS_CLAMP;
        const int bz1 = (q.hi == r->bn[2] && (clamp || q.lo == 0)) ? q.hi : q.hi - 1;
        BrickParams bp{data_view(r), clamp ? ADDR_CLAMP : ADDR_WRAP, r->bn[0], r->bn[1], r->bn[2], r->d_minmax, q.lo, bz1};
        HIP_TRY(launch_brick_minmax(bp, r->stream));
        r->minmax_valid = true;
        r->empty_valid = false;
    }
    if (!r->empty_valid) {
        int prefix[257];
        prefix[0] = 0;
        for (int i = 0; i < 256; ++i) {
            const float a = r->tf_host[i * 4 + 3];
            prefix[i + 1] = prefix[i] + ((a > 0.0f || a != a) ? 1 : 0);
        }
        // stream-ordered copy from a stack buffer: stage through a pageable memcpy that completes before return
        HIP_TRY(hipMemcpyAsync(r->d_alpha_prefix, prefix, sizeof(prefix), hipMemcpyHostToDevice, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));
        EmptyParams ep{r->d_minmax, nb, window_dev(r), r->d_alpha_prefix, r->d_empty};
        HIP_TRY(launch_brick_empty(ep, r->stream));
        // distance field for empty-space leaping: three separable passes, x then y then z
        const int mode = r->desc.data_address_mode == TBRM_ADDRESS_CLAMP ? ADDR_CLAMP : ADDR_WRAP;
        for (int axis = 0; axis < 3; ++axis) {
            DistParams dp{r->d_empty, axis == 0 ? nullptr : r->d_dist[(axis + 1) & 1], r->d_dist[axis & 1],
                {r->bn[0], r->bn[1], r->bn[2]}, axis};
            HIP_TRY(launch_brick_dist(dp, mode, r->stream));
        }
        r->empty_valid = true;
    }
    return TBRM_OK;
}

int build_ray_params(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                     const tbrm_world_params* world, RayParams& p)
{
    if (!(rp->steps > 0.0f)) return fail(TBRM_ERR_INVALID_ARG, "steps must be > 0");
    if (tile->w < 0 || tile->h < 0 || cam->width <= 0 || cam->height <= 0) return fail(TBRM_ERR_INVALID_ARG, "bad tile/camera size");
    p = RayParams{};
    p.data = data_view(r);
    p.data_addr_mode = r->desc.data_address_mode == TBRM_ADDRESS_CLAMP ? ADDR_CLAMP : ADDR_WRAP;
    p.tf = r->d_tf;
    p.win = window_dev(r);
    p.light = r->d_light;
    for (int c = 0; c < 3; ++c) p.lv_dims[c] = r->lv_dims[c];
    p.lv_bnx = r->lbn[0];
    p.lv_bnxy = r->lbn[0] * r->lbn[1];
    p.lv_fmt = r->lv_fmt;
    p.lv_wrap_layer = r->res_light.wrap_src;
    p.lv_wrap_shift = (r->res_light.hi - r->res_light.wrap_src) * 8;
    const tbrm_vec3d* v[4] = {&cam->position, &cam->forward, &cam->right, &cam->up};
    float* dst[4] = {p.cam_pos, p.fwd, p.right, p.up};
    for (int k = 0; k < 4; ++k) {
        dst[k][0] = (float) v[k]->x; dst[k][1] = (float) v[k]->y; dst[k][2] = (float) v[k]->z;
    }
    p.thx = (float) cam->tan_half_fov_x;
    p.thy = (float) cam->tan_half_fov_y;
    p.width = cam->width;
    p.height = cam->height;
    host_world_to_local(world->volume_transform, p.m);
    host_local_clipping(*world, p.cc, p.cd);
    p.clip_mode = raymarch_clip_mode(p.cc, p.cd);
    p.share_grid = (r->lv_dims[0] == r->desc.dim_x && r->lv_dims[1] == r->desc.dim_y && r->lv_dims[2] == r->desc.dim_z &&
                    !r->resident && !getenv("TBRM_NO_SHARE_GRID")) ? 1 : 0; // (the two volumes of a slab-resident handle relocate different layers)
    p.tile_x0 = tile->x0; p.tile_y0 = tile->y0; p.tile_w = tile->w; p.tile_h = tile->h;
    p.row_group_step = tile->row_group_step > 0 ? tile->row_group_step : 1;
    p.steps = rp->steps;
    p.jitter_frame = rp->jitter_frame;
    p.bnx = r->bn[0]; p.bny = r->bn[1]; p.bnz = r->bn[2];
    return TBRM_OK;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------

extern "C" {

const char* tbrm_version(void) { return "tbrm-mi355x 0.1.0 (gfx950)"; }
const char* tbrm_last_error(void) { return g_error; }

int tbrm_device_count(int* out_count)
{
    if (!out_count) return fail(TBRM_ERR_INVALID_ARG, "out_count is null");
    *out_count = 0;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(TBRM_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    *out_count = n;
    return TBRM_OK;
}

static int create_impl(const tbrm_resources_desc* desc, const tbrm_slab* owned, tbrm_resources** out);
int tbrm_resources_create(const tbrm_resources_desc* desc, tbrm_resources** out) { return create_impl(desc, nullptr, out); }
int tbrm_resources_create_slab(const tbrm_resources_desc* desc, const tbrm_slab* owned, tbrm_resources** out)
{
    if (!owned) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    return create_impl(desc, owned, out);
}

static int create_impl(const tbrm_resources_desc* desc, const tbrm_slab* owned, tbrm_resources** out)
{
    if (!desc || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (desc->dim_x <= 0 || desc->dim_y <= 0 || desc->dim_z <= 0) return fail(TBRM_ERR_INVALID_ARG, "volume size must be > 0");
    if (desc->data_format < TBRM_FMT_G8 || desc->data_format > TBRM_FMT_R32_FLOAT) return fail(TBRM_ERR_INVALID_ARG, "bad data_format");
    int ndev = 0;
    const int dc = tbrm_device_count(&ndev);
    if (dc != TBRM_OK) return dc;
    if (desc->device < 0 || desc->device >= ndev) return fail(TBRM_ERR_INVALID_ARG, "device %d out of range (%d devices)", desc->device, ndev);

    tbrm_resources* r = new (std::nothrow) tbrm_resources();
    if (!r) return fail(TBRM_ERR_OUT_OF_MEMORY, "host allocation failed");
    r->desc = *desc;
    // RaymarchVolume.cpp:850-861
    r->lv_dims[0] = desc->light_volume_half_resolution ? (desc->dim_x + 1) / 2 : desc->dim_x;
    r->lv_dims[1] = desc->light_volume_half_resolution ? (desc->dim_y + 1) / 2 : desc->dim_y;
    r->lv_dims[2] = desc->light_volume_half_resolution ? (desc->dim_z + 1) / 2 : desc->dim_z;
    r->lv_fmt = desc->light_volume_32bit ? FMT_F32 : FMT_U8;
    r->data_bytes = (size_t) desc->dim_x * desc->dim_y * desc->dim_z * format_bytes(desc->data_format);
    const size_t lv_elem = desc->light_volume_32bit ? 4 : 1;
    r->light_bytes = (size_t) r->lv_dims[0] * r->lv_dims[1] * r->lv_dims[2] * lv_elem;
    for (int c = 0; c < 3; ++c) {
        const int d = c == 0 ? desc->dim_x : (c == 1 ? desc->dim_y : desc->dim_z);
        r->bn[c] = r->dbn[c] = (d + kBrick - 1) / kBrick;
        r->lbn[c] = (r->lv_dims[c] + kBrick - 1) / kBrick;
    }
    const size_t data_bricks = (size_t) r->dbn[0] * r->dbn[1] * r->dbn[2], light_bricks = (size_t) r->lbn[0] * r->lbn[1] * r->lbn[2];
    if (data_bricks * 512 >= (1ull << 32) || light_bricks * 512 >= (1ull << 32)) {
        delete r;
        return fail(TBRM_ERR_UNSUPPORTED, "volumes of 2^32 or more (padded) voxels are not supported");
    }
    r->data_bricked_bytes = data_bricks * 512 * format_bytes(desc->data_format);
    r->light_bricked_bytes = light_bricks * 512 * lv_elem;
    r->res_data.layer_bytes = (size_t) r->dbn[0] * r->dbn[1] * 512 * format_bytes(desc->data_format);
    r->res_light.layer_bytes = (size_t) r->lbn[0] * r->lbn[1] * 512 * lv_elem;
    r->res_data.hi = r->dbn[2];
    r->res_light.hi = r->lbn[2];
    if (owned) { // slab-resident: which brick layers this handle keeps
        const int lz = r->lv_dims[2], dz = desc->dim_z;
        if (owned->z_begin < 0 || owned->z_end > lz || owned->z_begin >= owned->z_end || owned->z_begin % kChunkTile || owned->z_end % kChunkTile ||
            lz % kChunkTile) {
            delete r;
            return fail(TBRM_ERR_INVALID_ARG, "slab [%d, %d) of a light volume %d deep: bounds and depth must be multiples of %d",
                        owned->z_begin, owned->z_end, lz, kChunkTile);
        }
        r->resident = true;
        r->owned = *owned;
        // light volume: the owned slices and one brick layer either side (the raymarch's taps); wrap addressing reaches
        // the far end of the volume from its first / last slice
        auto set = [](tbrm_resources::Residency& q, int lo, int hi, int layers) {
            q.lo = std::max(lo, 0);
            q.hi = std::min(hi, layers);
            q.wrap_src = (q.lo == 0 && q.hi < layers) ? layers - 1 : ((q.hi == layers && q.lo > 0) ? 0 : -1);
        };
        set(r->res_light, owned->z_begin / 8 - 1, owned->z_end / 8 + 1, r->lbn[2]);
        // data volume: what the occlusion of the slab's rows and of the 32 rows either side can sample (lateral passes,
        // tbrm_slab_*: taps reach UVWOffset + 1 texel beyond a row's own position), in data texels
        const double ratio = (double) dz / (double) lz;
        const int halo = (int) std::ceil(ratio * (kChunkTile + 8)) + 8;
        const int d0 = (int) std::floor(owned->z_begin * ratio) - halo, d1 = (int) std::ceil(owned->z_end * ratio) + halo;
        set(r->res_data, floor_div(d0, 8), ceil_div(d1, 8), r->dbn[2]);
        if (desc->data_address_mode == TBRM_ADDRESS_CLAMP) r->res_data.wrap_src = -1;
    }
    const size_t nb = (size_t) r->bn[0] * r->bn[1] * r->bn[2];
    const size_t nb_pad = (nb + 255) / 256 * 256;

#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        const hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                                    \
            const int code_ = fail(e_ == hipErrorOutOfMemory ? TBRM_ERR_OUT_OF_MEMORY : TBRM_ERR_NO_DEVICE, \
                "%s failed: %s", #expr, hipGetErrorString(e_));                                    \
            tbrm_resources_destroy(r);                                                             \
            return code_;                                                                          \
        }                                                                                          \
    } while (0)

    CREATE_TRY(hipSetDevice(desc->device));
    CREATE_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    if (hipDeviceGetAttribute(&r->n_cus, hipDeviceAttributeMultiprocessorCount, desc->device) != hipSuccess || r->n_cus <= 0) r->n_cus = 256;
    {
        tbrm_resources::Residency& q = r->res_data;
        CREATE_TRY(hipMalloc(&q.alloc, (size_t) (q.hi - q.lo + (q.wrap_src >= 0 ? 1 : 0)) * q.layer_bytes));
        r->d_data = (char*) q.alloc - (size_t) q.lo * q.layer_bytes;
    }
    CREATE_TRY(hipMalloc((void**) &r->d_tf, 256 * sizeof(float4)));
    {
        tbrm_resources::Residency& q = r->res_light;
        CREATE_TRY(hipMalloc(&q.alloc, (size_t) (q.hi - q.lo + (q.wrap_src >= 0 ? 1 : 0)) * q.layer_bytes));
        r->d_light = (char*) q.alloc - (size_t) q.lo * q.layer_bytes;
    }
    // XYZReadWriteBuffers: 4 buffers per axis in the light volume's format (RaymarchVolume.cpp:864-866,:889-891)
    const size_t buf_px[3] = {(size_t) r->lv_dims[1] * r->lv_dims[2], (size_t) r->lv_dims[0] * r->lv_dims[2],
        (size_t) r->lv_dims[0] * r->lv_dims[1]};
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 4; ++k) CREATE_TRY(hipMalloc(&r->d_buf[a][k], buf_px[a] * lv_elem));
    const size_t plane_px = std::max({buf_px[0], buf_px[1], buf_px[2]});
    for (int k = 0; k < 4; ++k) CREATE_TRY(hipMalloc((void**) &r->d_plane[k], (plane_px + 2 * kPlaneGuard) * sizeof(float)));
    CREATE_TRY(hipMalloc((void**) &r->d_minmax, nb * sizeof(float2)));
    CREATE_TRY(hipMalloc((void**) &r->d_empty, nb_pad / 8));
    for (int k = 0; k < 2; ++k) CREATE_TRY(hipMalloc((void**) &r->d_dist[k], nb_pad));
    CREATE_TRY(hipMalloc((void**) &r->d_alpha_prefix, 257 * sizeof(int)));
    CREATE_TRY(hipMalloc((void**) &r->d_counter, sizeof(unsigned long long)));
    for (int k = 0; k < 2; ++k)
        for (int e = 0; e < 2; ++e) CREATE_TRY(hipEventCreate(&r->ev[k][e]));
    // the light volume render target starts cleared
    CREATE_TRY(hipMemsetAsync(r->res_light.alloc, 0, (size_t) (r->res_light.hi - r->res_light.lo + (r->res_light.wrap_src >= 0 ? 1 : 0)) * r->res_light.layer_bytes, r->stream));
#undef CREATE_TRY
    *out = r;
    return TBRM_OK;
}

int tbrm_resources_destroy(tbrm_resources* r)
{
    if (!r) return TBRM_OK;
    (void) hipSetDevice(r->desc.device);
    if (r->stream) (void) hipStreamSynchronize(r->stream);
    (void) hipFree(r->res_data.alloc);
    (void) hipFree(r->d_tf);
    (void) hipFree(r->res_light.alloc);
    for (auto& axis : r->d_buf)
        for (void* b : axis) (void) hipFree(b);
    for (float* pl : r->d_plane) (void) hipFree(pl);
    delete r->slab_op;
    (void) hipFree(r->d_occ);
    for (uint8_t* z : r->d_occ_zero) (void) hipFree(z);
    (void) hipFree(r->d_occ_list);
    (void) hipFree(r->d_minmax);
    (void) hipFree(r->d_empty);
    for (uint16_t* o : r->d_octree) (void) hipFree(o);
    for (uint8_t* d : r->d_dist) (void) hipFree(d);
    (void) hipFree(r->d_alpha_prefix);
    (void) hipFree(r->d_counter);
    (void) hipFree(r->d_out);
    for (auto& k : r->ev)
        for (hipEvent_t e : k)
            if (e) (void) hipEventDestroy(e);
    if (r->stream) (void) hipStreamDestroy(r->stream);
    delete r;
    return TBRM_OK;
}

int tbrm_resources_light_volume_dims(const tbrm_resources* r, int32_t out_dims[3])
{
    if (!r || !out_dims) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int c = 0; c < 3; ++c) out_dims[c] = r->lv_dims[c];
    return TBRM_OK;
}

int tbrm_resources_is_initialized(const tbrm_resources* r) { return initialized(r) ? 1 : 0; }

int tbrm_upload_volume(tbrm_resources* r, const void* host_voxels, size_t n_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: upload its layers with tbrm_upload_volume_slices");
    if (!r || !host_voxels) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_bytes != r->data_bytes) return fail(TBRM_ERR_INVALID_ARG, "volume is %zu bytes, expected %zu", n_bytes, r->data_bytes);
    if (int e = bind(r)) return e;
    void* staging = nullptr; // linear copy in HBM, re-laid out into bricks by the GPU
    HIP_TRY(hipMalloc(&staging, n_bytes));
    const int dims[3] = {r->desc.dim_x, r->desc.dim_y, r->desc.dim_z};
    hipError_t e1 = hipMemcpyAsync(staging, host_voxels, n_bytes, hipMemcpyHostToDevice, r->stream);
    if (e1 == hipSuccess) e1 = launch_relayout(relayout_params(staging, r->d_data, dims, r->dbn, format_bytes(r->desc.data_format), true), r->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(r->stream); // the caller may free its buffer on return
    (void) hipFree(staging);
    HIP_TRY(e1);
    r->has_volume = true;
    r->octree_valid = false;
    r->minmax_valid = false;
    return TBRM_OK;
}

int tbrm_upload_volume_device(tbrm_resources* r, const void* device_voxels, size_t n_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: upload its layers with tbrm_upload_volume_slices");
    if (!r || !device_voxels) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_bytes != r->data_bytes) return fail(TBRM_ERR_INVALID_ARG, "volume is %zu bytes, expected %zu", n_bytes, r->data_bytes);
    if (int e = bind(r)) return e;
    const int dims[3] = {r->desc.dim_x, r->desc.dim_y, r->desc.dim_z};
    HIP_TRY(launch_relayout(relayout_params(device_voxels, r->d_data, dims, r->dbn, format_bytes(r->desc.data_format), true), r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->has_volume = true;
    r->octree_valid = false;
    r->minmax_valid = false;
    return TBRM_OK;
}

int tbrm_set_tf_lut(tbrm_resources* r, const float* rgba_256x4)
{
    if (!r || !rgba_256x4) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (int e = bind(r)) return e;
    host_bake_tf(rgba_256x4, r->tf_host); // FFloat16 storage (RaymarchUtils.cpp:151-161)
    HIP_TRY(hipMemcpyAsync(r->d_tf, r->tf_host, sizeof(r->tf_host), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->has_tf = true;
    r->empty_valid = false;
    return TBRM_OK;
}

int tbrm_color_curve_to_lut(const float* key_times[4], const float* key_values[4], const int32_t n_keys[4], float* out)
{
    if (!key_times || !key_values || !n_keys || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int c = 0; c < 4; ++c)
        if (n_keys[c] > 0 && (!key_times[c] || !key_values[c])) return fail(TBRM_ERR_INVALID_ARG, "null key array");
    host_color_curve_to_lut(key_times, key_values, n_keys, out);
    return TBRM_OK;
}

int tbrm_make_default_tf_lut(float* out)
{
    if (!out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_default_tf_lut(out);
    return TBRM_OK;
}

int tbrm_host_bake_tf_lut(const float* rgba_256x4, float* out)
{
    if (!rgba_256x4 || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_bake_tf(rgba_256x4, out);
    return TBRM_OK;
}

int tbrm_set_windowing(tbrm_resources* r, const tbrm_windowing_params* w)
{
    if (!r || !w) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    r->win = *w;
    r->empty_valid = false;
    return TBRM_OK;
}

int tbrm_add_dir_light(tbrm_resources* r, const tbrm_dir_light_params* light, int added, const tbrm_world_params* world,
                       int* light_added, int gpu_sync)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: light operators run through tbrm_slab_*");
    (void) gpu_sync; // accepted and ignored (RaymarchUtils.cpp:51-59)
    if (light_added) *light_added = 0;
    if (!r || !light || !world) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function"); // :39-45
    if (light_added) *light_added = 1;
    if (int e = bind(r)) return e;
    if (int e = begin_timed(r, 0)) return e;
    if (int e = enqueue_add(r, *light, added != 0, *world)) return e;
    return end_timed(r, 0);
}

int tbrm_add_dir_lights(tbrm_resources* r, const tbrm_dir_light_params* lights, int32_t n_lights, int added, const tbrm_world_params* world,
                        int32_t* schedule, int32_t* n_entries)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: light operators run through tbrm_slab_*");
    if (n_entries) *n_entries = 0;
    if (!r || !world || (n_lights > 0 && !lights) || n_lights < 0) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    if (int e = bind(r)) return e;
    if (int e = begin_timed(r, 0)) return e;
    if (int e = enqueue_add_batch(r, lights, n_lights, added != 0, *world, schedule, n_entries)) return e;
    return end_timed(r, 0);
}

int tbrm_change_dir_light(tbrm_resources* r, const tbrm_dir_light_params* old_light, const tbrm_dir_light_params* new_light,
                          const tbrm_world_params* world, int* light_added)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: light operators run through tbrm_slab_*");
    if (light_added) *light_added = 0;
    if (!r || !old_light || !new_light || !world) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function"); // :74-80
    if (light_added) *light_added = 1;
    if (int e = bind(r)) return e;
    if (int e = begin_timed(r, 0)) return e;
    if (int e = enqueue_change(r, *old_light, *new_light, *world)) return e;
    return end_timed(r, 0);
}

// ---- slab-partitioned illumination (tbrm.h "slabs") ------------------------------------------------------------

int tbrm_slab_light_begin(tbrm_resources* r, const tbrm_dir_light_params* removed, const tbrm_dir_light_params* light, int added,
                          const tbrm_world_params* world, const tbrm_slab* slab, int32_t* n_passes)
{
    if (!r || !light || !world || !slab || !n_passes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    *n_passes = 0;
    if (!r->slab_op) r->slab_op = new SlabOp;
    SlabOp& op = *r->slab_op;
    op.slab = *slab;
    op.change = removed != nullptr;
    op.n = 0;
    op.current = -1;
    op.base = base_prop_params(r, *world);
    if (!op.change) { // enqueue_add
        int n = 0;
        if (!host_light_passes(*light, *world, r->lv_dims, r->desc.border_mode, op.a, &n)) return TBRM_OK;
        op.n = n;
        op.b_added = added ? 1.0f : -1.0f;
    } else { // enqueue_change
        tbrm_light_pass rp[2], ap[2];
        int rn = 0, an = 0;
        const bool r_ok = host_light_passes(*removed, *world, r->lv_dims, r->desc.border_mode, rp, &rn);
        const bool a_ok = host_light_passes(*light, *world, r->lv_dims, r->desc.border_mode, ap, &an);
        if (!r_ok || !a_ok) return TBRM_OK;
        if (rp[0].face != ap[0].face || rp[1].face != ap[1].face)
            return fail(TBRM_ERR_AXES_DIFFER, "the two lights' major axes differ: remove the old light and add the new one "
                                              "(LightingShaders.cpp:192-198)");
        for (int i = 0; i < 2; ++i) {
            if (rp[i].light_alpha == 0.0f && ap[i].light_alpha == 0.0f && rp[i].border_light == 0.0f && ap[i].border_light == 0.0f)
                continue; // both streams dark: the pass cannot touch the light volume (enqueue_change)
            op.a[op.n] = ap[i];
            op.r[op.n] = rp[i];
            ++op.n;
        }
        op.b_added = 0.0f;
    }
    for (int i = 0; i < op.n; ++i) { // all or nothing: every pass has to have a chunked (slab-capable) form
        ChunkFit fit;
        const tbrm_light_pass* pr_i = op.change ? &op.r[i] : nullptr;
        const int reach = slice_tap_reach(op.a[i], pr_i);
        const bool slice_form = reach >= 0 && (op.a[i].axis == 2 || reach <= slab->z_end - slab->z_begin); // one slice per step
        if (!chunk_fit(r, op.a[i], pr_i, fit) && !slice_form) {
            const int n = op.n;
            op.n = 0;
            return fail(TBRM_ERR_UNSUPPORTED, "pass %d of %d (axis %d) has no slab-partitioned form: its taps reach %d rows from the pixel (%s)",
                        i, n, (int) op.a[i].axis, reach, g_plan_note);
        }
    }
    *n_passes = op.n;
    return TBRM_OK;
}

int tbrm_slab_pass_begin(tbrm_resources* r, int32_t pass, tbrm_slab_pass* out)
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->slab_op || pass < 0 || pass >= r->slab_op->n) return fail(TBRM_ERR_INVALID_ARG, "no such pass (tbrm_slab_light_begin first)");
    if (int e = bind(r)) return e;
    SlabOp& op = *r->slab_op;
    op.current = -1;
    const int e = plan_pass(r, op.base, op.a[pass], op.change ? &op.r[pass] : nullptr, op.b_added, &op.slab, op.plan);
    if (e == TBRM_ERR_UNSUPPORTED)
        return fail(e, "pass %d (axis %d) has no slab-partitioned form: %s", (int) pass, (int) op.a[pass].axis, g_plan_note);
    if (e) return e;
    op.current = pass;
    const PassPlan& pl = op.plan;
    out->axis = pl.p.axis;
    out->dir = pl.dir;
    out->lateral = pl.lateral ? 1 : 0;
    out->streams = pl.two_streams() ? 2 : 1;
    out->plane_w = pl.p.W;
    out->plane_h = pl.p.H;
    out->chunk_slices = pl.M;
    out->chunks_of_pass = pl.chunks_of_pass;
    out->first_chunk = pl.first_chunk_of_pass;
    out->n_chunks = pl.n_chunks;
    out->halo_rows = pl.lateral ? (pl.sliced ? pl.halo_rows : kChunkTile) : 0;
    out->plane_elem_bytes = pl.sliced ? (r->lv_fmt == FMT_U8 ? 1 : 4) : 4;
    return TBRM_OK;
}

int tbrm_slab_pass_chunk(tbrm_resources* r, int32_t chunk)
{
    if (!r || !r->slab_op || r->slab_op->current < 0) return fail(TBRM_ERR_INVALID_ARG, "no pass in flight (tbrm_slab_pass_begin first)");
    const PassPlan& pl = r->slab_op->plan;
    if (chunk < 0 || chunk >= pl.n_chunks) return fail(TBRM_ERR_INVALID_ARG, "chunk %d of %d", chunk, pl.n_chunks);
    if (int e = bind(r)) return e;
    return enqueue_plan_chunk(r, pl, chunk);
}

int tbrm_slab_pass_plane(tbrm_resources* r, int32_t boundary, int32_t stream, void** device_plane)
{
    if (!r || !device_plane || !r->slab_op || r->slab_op->current < 0) return fail(TBRM_ERR_INVALID_ARG, "no pass in flight");
    const PassPlan& pl = r->slab_op->plan;
    if (boundary < 0 || boundary > pl.n_chunks || stream < 0 || stream >= (pl.two_streams() ? 2 : 1))
        return fail(TBRM_ERR_INVALID_ARG, "boundary %d / stream %d out of range", boundary, stream);
    *device_plane = pl.sliced ? sliced_plane(r, pl, boundary, stream) : (void*) plan_plane(r, boundary, stream);
    return TBRM_OK;
}

// ---- slab-resident handles: moving their layers in and out -------------------------------------------------------------

namespace {
// where brick layer `layer` of a volume lives, or null when the handle does not hold it
char* layer_address(const tbrm_resources::Residency& q, int layer)
{
    if (layer >= q.lo && layer < q.hi) return (char*) q.alloc + (size_t) (layer - q.lo) * q.layer_bytes;
    if (layer == q.wrap_src) return (char*) q.alloc + (size_t) (q.hi - q.lo) * q.layer_bytes;
    return nullptr;
}
} // namespace

int tbrm_slab_resident_slices(const tbrm_resources* r, int32_t data[3], int32_t light[3])
{
    if (!r || !data || !light) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    const tbrm_resources::Residency* q[2] = {&r->res_data, &r->res_light};
    const int depth[2] = {r->desc.dim_z, r->lv_dims[2]};
    int32_t* out[2] = {data, light};
    for (int k = 0; k < 2; ++k) {
        out[k][0] = q[k]->lo * 8;
        out[k][1] = std::min(q[k]->hi * 8, depth[k]);
        out[k][2] = q[k]->wrap_src >= 0 ? q[k]->wrap_src * 8 : -1;
    }
    return TBRM_OK;
}

int tbrm_upload_volume_slices(tbrm_resources* r, int32_t z_begin, int32_t z_count, const void* host_voxels, size_t n_bytes)
{
    if (!r || !host_voxels) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    const int nz = r->desc.dim_z;
    const size_t esz = format_bytes(r->desc.data_format), slice = (size_t) r->desc.dim_x * r->desc.dim_y * esz;
    if (z_begin < 0 || z_count <= 0 || z_begin + z_count > nz || z_begin % 8 || ((z_begin + z_count) % 8 && z_begin + z_count != nz))
        return fail(TBRM_ERR_INVALID_ARG, "slices [%d, %d): whole brick layers (multiples of 8) of a volume %d deep", z_begin, z_begin + z_count, nz);
    if (n_bytes != slice * (size_t) z_count) return fail(TBRM_ERR_INVALID_ARG, "%d slices are %zu bytes, got %zu", z_count, slice * (size_t) z_count, n_bytes);
    if (int e = bind(r)) return e;
    void* staging = nullptr;
    HIP_TRY(hipMalloc(&staging, n_bytes));
    hipError_t e1 = hipMemcpyAsync(staging, host_voxels, n_bytes, hipMemcpyHostToDevice, r->stream);
    int code = TBRM_OK;
    for (int layer = z_begin / 8; e1 == hipSuccess && layer < ceil_div(z_begin + z_count, 8); ++layer) { // layer by layer: the wrap copy lives elsewhere
        char* dst = layer_address(r->res_data, layer);
        if (!dst) { code = fail(TBRM_ERR_INVALID_ARG, "data slices %d.. are not resident on this handle", layer * 8); break; }
        const int lz = std::min(8, nz - layer * 8);
        const int dims[3] = {r->desc.dim_x, r->desc.dim_y, lz}, bn[3] = {r->dbn[0], r->dbn[1], 1};
        e1 = launch_relayout(relayout_params((const char*) staging + (size_t) (layer * 8 - z_begin) * slice, dst, dims, bn, esz, true), r->stream);
    }
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(r->stream);
    (void) hipFree(staging);
    if (code != TBRM_OK) return code;
    HIP_TRY(e1);
    r->has_volume = true;
    r->minmax_valid = false;
    return TBRM_OK;
}

int tbrm_download_light_slices(tbrm_resources* r, int32_t z_begin, int32_t z_count, void* host_out, size_t n_bytes)
{
    if (!r || !host_out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    const int nz = r->lv_dims[2];
    const size_t esz = r->lv_fmt == FMT_U8 ? 1 : 4, slice = (size_t) r->lv_dims[0] * r->lv_dims[1] * esz;
    if (z_begin < 0 || z_count <= 0 || z_begin + z_count > nz || z_begin % 8 || ((z_begin + z_count) % 8 && z_begin + z_count != nz))
        return fail(TBRM_ERR_INVALID_ARG, "slices [%d, %d): whole brick layers (multiples of 8) of a light volume %d deep", z_begin, z_begin + z_count, nz);
    if (n_bytes != slice * (size_t) z_count) return fail(TBRM_ERR_INVALID_ARG, "%d slices are %zu bytes, got %zu", z_count, slice * (size_t) z_count, n_bytes);
    if (int e = bind(r)) return e;
    void* staging = nullptr;
    HIP_TRY(hipMalloc(&staging, n_bytes));
    hipError_t e1 = hipSuccess;
    int code = TBRM_OK;
    for (int layer = z_begin / 8; e1 == hipSuccess && layer < ceil_div(z_begin + z_count, 8); ++layer) {
        const tbrm_resources::Residency& q = r->res_light;
        char* src = (layer >= q.lo && layer < q.hi) ? layer_address(q, layer) : nullptr; // the layer itself, not a wrap copy of it
        if (!src) { code = fail(TBRM_ERR_INVALID_ARG, "light-volume slices %d.. are not resident on this handle", layer * 8); break; }
        const int lz = std::min(8, nz - layer * 8);
        const int dims[3] = {r->lv_dims[0], r->lv_dims[1], lz}, bn[3] = {r->lbn[0], r->lbn[1], 1};
        e1 = launch_relayout(relayout_params(src, (char*) staging + (size_t) (layer * 8 - z_begin) * slice, dims, bn, esz, false), r->stream);
    }
    if (e1 == hipSuccess && code == TBRM_OK) e1 = hipMemcpyAsync(host_out, staging, n_bytes, hipMemcpyDeviceToHost, r->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(r->stream);
    (void) hipFree(staging);
    if (code != TBRM_OK) return code;
    HIP_TRY(e1);
    return TBRM_OK;
}

int tbrm_slab_light_halo(tbrm_resources* r, int32_t side, void** send_layer, void** recv_layer, size_t* layer_bytes)
{
    if (!r || !send_layer || !recv_layer || !layer_bytes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->resident) return fail(TBRM_ERR_INVALID_ARG, "not a slab-resident handle");
    if (side != 0 && side != 1) return fail(TBRM_ERR_INVALID_ARG, "side is 0 (towards z = 0) or 1");
    const tbrm_resources::Residency& q = r->res_light;
    const int first = r->owned.z_begin / 8, last = r->owned.z_end / 8 - 1, layers = r->lbn[2];
    const int send = side == 0 ? first : last;
    const int recv = side == 0 ? (first == 0 ? layers - 1 : first - 1) : (last == layers - 1 ? 0 : last + 1); // across the ends: the wrap copy
    *send_layer = layer_address(q, send);
    *recv_layer = (recv >= first && recv <= last) ? nullptr : layer_address(q, recv); // a handle that owns everything has no halo
    *layer_bytes = q.layer_bytes;
    return TBRM_OK;
}

int tbrm_clear_light_volume(tbrm_resources* r, float clear_value)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->d_light) return TBRM_OK; // RaymarchUtils.cpp:106-109
    if (int e = bind(r)) return e;
    if (int e = begin_timed(r, 0)) return e;
    const tbrm_resources::Residency& q = r->res_light; // (all layers of an ordinary handle); padding voxels are never sampled
    const size_t n = (size_t) r->lbn[0] * r->lbn[1] * 512 * (size_t) (q.hi - q.lo + (q.wrap_src >= 0 ? 1 : 0));
    HIP_TRY(launch_fill(q.alloc, r->lv_fmt, n, clear_value, r->stream));
    return end_timed(r, 0);
}

int tbrm_raymarch_lit_device(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                             const tbrm_world_params* world, const float* device_scene_depth, float* device_out_rgba)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: frames are marched with tbrm_raymarch_lit_slab_device");
    if (!r || !cam || !tile || !rp || !world || !device_out_rgba) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    if (int e = bind(r)) return e;
    RayParams p;
    if (int e = build_ray_params(r, cam, tile, rp, world, p)) return e;
    p.depth = device_scene_depth;
    p.out = device_out_rgba;
    if (rp->enable_skipping) {
        if (int e = ensure_skipping(r)) return e;
        p.empty_bits = r->d_empty;
        p.skip_dist = r->d_dist[0];
    }
    if (int e = begin_timed(r, 1)) return e;
    HIP_TRY(launch_raymarch(p, r->stream));
    ++r->launches[2];
    return end_timed(r, 1);
}

int tbrm_raymarch_lit_slab_device(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                                  const tbrm_world_params* world, const float* device_scene_depth, float* device_state_rgba,
                                  const tbrm_slab* slab, int direction)
{
    if (!r || !cam || !tile || !rp || !world || !device_state_rgba || !slab) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    if (slab->z_begin < 0 || slab->z_end > r->lv_dims[2] || slab->z_begin >= slab->z_end)
        return fail(TBRM_ERR_INVALID_ARG, "slab [%d, %d) of a light volume %d deep", slab->z_begin, slab->z_end, r->lv_dims[2]);
    if (int e = bind(r)) return e;
    RayParams p;
    if (int e = build_ray_params(r, cam, tile, rp, world, p)) return e;
    p.depth = device_scene_depth;
    p.out = device_state_rgba;
    p.slab_on = 1;
    p.slab_z0 = slab->z_begin;
    p.slab_z1 = slab->z_end;
    p.slab_dir = direction > 0 ? 1 : (direction < 0 ? -1 : 0);
    if (r->resident && (slab->z_begin != r->owned.z_begin || slab->z_end != r->owned.z_end))
        return fail(TBRM_ERR_INVALID_ARG, "a slab-resident handle marches its own slab [%d, %d) only", r->owned.z_begin, r->owned.z_end);
    if (rp->enable_skipping) {
        if (int e = ensure_skipping(r)) return e;
        p.empty_bits = r->d_empty;
        p.skip_dist = r->d_dist[0];
    }
    if (int e = begin_timed(r, 1)) return e;
    HIP_TRY(launch_raymarch(p, r->stream));
    ++r->launches[2];
    return end_timed(r, 1);
}

int tbrm_raymarch_lit(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                      const tbrm_world_params* world, float* host_out_rgba)
{
    if (!r || !tile || !host_out_rgba) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (tile->w < 0 || tile->h < 0) return fail(TBRM_ERR_INVALID_ARG, "bad tile size");
    const size_t bytes = (size_t) tile->w * tile->h * 4 * sizeof(float);
    if (bytes == 0) return TBRM_OK;
    if (int e = bind(r)) return e;
    if (bytes > r->out_bytes) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->d_out);
        r->d_out = nullptr;
        r->out_bytes = 0;
        HIP_TRY(hipMalloc((void**) &r->d_out, bytes));
        r->out_bytes = bytes;
    }
    if (int e = tbrm_raymarch_lit_device(r, cam, tile, rp, world, nullptr, r->d_out)) return e;
    HIP_TRY(hipMemcpyAsync(host_out_rgba, r->d_out, bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return TBRM_OK;
}

int tbrm_raymarch_intensity_device(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                                   const tbrm_world_params* world, const float* device_scene_depth, float* device_out_rgba)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: only the lit march has a slab form");
    if (!r || !cam || !tile || !rp || !world || !device_out_rgba) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->has_volume) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume");
    if (int e = bind(r)) return e;
    RayParams p;
    if (int e = build_ray_params(r, cam, tile, rp, world, p)) return e;
    p.depth = device_scene_depth;
    p.out = device_out_rgba;
    if (int e = begin_timed(r, 1)) return e;
    HIP_TRY(launch_raymarch_intensity(p, r->stream));
    ++r->launches[2];
    return end_timed(r, 1);
}

int tbrm_raymarch_intensity(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                            const tbrm_world_params* world, float* host_out_rgba)
{
    if (!r || !tile || !host_out_rgba) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (tile->w < 0 || tile->h < 0) return fail(TBRM_ERR_INVALID_ARG, "bad tile size");
    const size_t bytes = (size_t) tile->w * tile->h * 4 * sizeof(float);
    if (bytes == 0) return TBRM_OK;
    if (int e = bind(r)) return e;
    if (bytes > r->out_bytes) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->d_out);
        r->d_out = nullptr;
        r->out_bytes = 0;
        HIP_TRY(hipMalloc((void**) &r->d_out, bytes));
        r->out_bytes = bytes;
    }
    if (int e = tbrm_raymarch_intensity_device(r, cam, tile, rp, world, nullptr, r->d_out)) return e;
    HIP_TRY(hipMemcpyAsync(host_out_rgba, r->d_out, bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return TBRM_OK;
}

int tbrm_octree_mip_dims(const tbrm_resources* r, int mip, int32_t out_dims[3])
{
    if (!r || !out_dims || mip < 0 || mip > 3) return fail(TBRM_ERR_INVALID_ARG, "bad argument");
    const int d[3] = {r->desc.dim_x, r->desc.dim_y, r->desc.dim_z};
    for (int c = 0; c < 3; ++c) {
        int p2 = 1;
        while (p2 < d[c]) p2 <<= 1; // FMath::RoundUpToPowerOfTwo (RaymarchVolume.cpp:876-877)
        out_dims[c] = std::max(p2 >> mip, 1);
    }
    return TBRM_OK;
}

int tbrm_generate_octree(tbrm_resources* r)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: only the lit march has a slab form");
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->has_volume) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume");
    if (int e = bind(r)) return e;
    for (int m = 0; m < 4; ++m) {
        int32_t d[3];
        (void) tbrm_octree_mip_dims(r, m, d);
        for (int c = 0; c < 3; ++c) r->oct_dims[m][c] = d[c];
        if (!r->d_octree[m]) HIP_TRY(hipMalloc((void**) &r->d_octree[m], (size_t) d[0] * d[1] * d[2] * sizeof(uint16_t)));
        OctreeParams op{};
        op.data = data_view(r);
        op.lower = m ? r->d_octree[m - 1] : nullptr;
        for (int c = 0; c < 3; ++c) { op.dims[c] = d[c]; op.lower_dims[c] = m ? r->oct_dims[m - 1][c] : 0; }
        op.out = r->d_octree[m];
        HIP_TRY(launch_octree_level(op, m == 0, r->stream));
    }
    r->octree_valid = true;
    return TBRM_OK;
}

int tbrm_download_octree_mip(tbrm_resources* r, int mip, uint16_t* host_out, size_t bytes)
{
    if (!r || !host_out || mip < 0 || mip > 3) return fail(TBRM_ERR_INVALID_ARG, "bad argument");
    if (!r->octree_valid) return fail(TBRM_ERR_NOT_INITIALIZED, "no octree: call tbrm_generate_octree after uploading the volume");
    if (int e = bind(r)) return e;
    const size_t need = (size_t) r->oct_dims[mip][0] * r->oct_dims[mip][1] * r->oct_dims[mip][2] * sizeof(uint16_t);
    if (bytes != need) return fail(TBRM_ERR_INVALID_ARG, "octree level %d is %zu bytes, got %zu", mip, need, bytes);
    HIP_TRY(hipMemcpyAsync(host_out, r->d_octree[mip], need, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return TBRM_OK;
}

int tbrm_raymarch_octree_device(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                                const tbrm_world_params* world, int octree_mip, const float* device_scene_depth, float* device_out_rgba)
{
    if (!r || !cam || !tile || !rp || !world || !device_out_rgba || octree_mip < 0 || octree_mip > 3) return fail(TBRM_ERR_INVALID_ARG, "bad argument");
    if (!r->has_volume || !r->has_tf) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    if (!r->octree_valid) return fail(TBRM_ERR_NOT_INITIALIZED, "no octree: call tbrm_generate_octree after uploading the volume");
    if (int e = bind(r)) return e;
    RayParams p;
    if (int e = build_ray_params(r, cam, tile, rp, world, p)) return e;
    p.depth = device_scene_depth;
    p.out = device_out_rgba;
    p.octree = r->d_octree[octree_mip];
    for (int c = 0; c < 3; ++c) p.oct_dims[c] = r->oct_dims[octree_mip][c];
    p.oct_depth0 = (float) r->oct_dims[0][2];
    if (int e = begin_timed(r, 1)) return e;
    HIP_TRY(launch_raymarch_octree(p, r->stream));
    ++r->launches[2];
    return end_timed(r, 1);
}

int tbrm_raymarch_octree(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                         const tbrm_world_params* world, int octree_mip, float* host_out_rgba)
{
    if (!r || !tile || !host_out_rgba) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (tile->w < 0 || tile->h < 0) return fail(TBRM_ERR_INVALID_ARG, "bad tile size");
    const size_t bytes = (size_t) tile->w * tile->h * 4 * sizeof(float);
    if (bytes == 0) return TBRM_OK;
    if (int e = bind(r)) return e;
    if (bytes > r->out_bytes) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        (void) hipFree(r->d_out);
        r->d_out = nullptr;
        r->out_bytes = 0;
        HIP_TRY(hipMalloc((void**) &r->d_out, bytes));
        r->out_bytes = bytes;
    }
    if (int e = tbrm_raymarch_octree_device(r, cam, tile, rp, world, octree_mip, nullptr, r->d_out)) return e;
    HIP_TRY(hipMemcpyAsync(host_out_rgba, r->d_out, bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return TBRM_OK;
}

int tbrm_count_nominal_samples(tbrm_resources* r, const tbrm_camera* cam, const tbrm_tile* tile, const tbrm_raymarch_params* rp,
                               const tbrm_world_params* world, uint64_t* out_samples)
{
    if (!r || !cam || !tile || !rp || !world || !out_samples) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (int e = bind(r)) return e;
    RayParams p;
    if (int e = build_ray_params(r, cam, tile, rp, world, p)) return e;
    p.sample_counter = r->d_counter;
    HIP_TRY(hipMemsetAsync(r->d_counter, 0, sizeof(unsigned long long), r->stream));
    HIP_TRY(launch_count_samples(p, r->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, r->d_counter, sizeof(v), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    *out_samples = v;
    return TBRM_OK;
}

int tbrm_download_light_volume(tbrm_resources* r, void* host_out, size_t n_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: read its slices with tbrm_download_light_slices");
    if (!r || !host_out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_bytes != r->light_bytes) return fail(TBRM_ERR_INVALID_ARG, "light volume is %zu bytes, got %zu", r->light_bytes, n_bytes);
    if (int e = bind(r)) return e;
    void* staging = nullptr;
    HIP_TRY(hipMalloc(&staging, n_bytes));
    const int dims[3] = {r->lv_dims[0], r->lv_dims[1], r->lv_dims[2]};
    hipError_t e1 = launch_relayout(relayout_params(r->d_light, staging, dims, r->lbn, r->lv_fmt == FMT_U8 ? 1 : 4, false), r->stream);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(host_out, staging, n_bytes, hipMemcpyDeviceToHost, r->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(r->stream);
    (void) hipFree(staging);
    HIP_TRY(e1);
    return TBRM_OK;
}

int tbrm_upload_light_volume(tbrm_resources* r, const void* host_in, size_t n_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: the light volume is written by tbrm_slab_* only");
    if (!r || !host_in) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_bytes != r->light_bytes) return fail(TBRM_ERR_INVALID_ARG, "light volume is %zu bytes, got %zu", r->light_bytes, n_bytes);
    if (int e = bind(r)) return e;
    void* staging = nullptr;
    HIP_TRY(hipMalloc(&staging, n_bytes));
    const int dims[3] = {r->lv_dims[0], r->lv_dims[1], r->lv_dims[2]};
    hipError_t e1 = hipMemcpyAsync(staging, host_in, n_bytes, hipMemcpyHostToDevice, r->stream);
    if (e1 == hipSuccess) e1 = launch_relayout(relayout_params(staging, r->d_light, dims, r->lbn, r->lv_fmt == FMT_U8 ? 1 : 4, true), r->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(r->stream);
    (void) hipFree(staging);
    HIP_TRY(e1);
    return TBRM_OK;
}

int tbrm_light_volume_device_ptr(tbrm_resources* r, void** out_ptr, size_t* out_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: exchange its boundary layers with tbrm_slab_light_halo");
    if (!r || !out_ptr) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    *out_ptr = r->d_light; // bricked layout (DESIGN.md "Data layout")
    if (out_bytes) *out_bytes = r->light_bricked_bytes;
    return TBRM_OK;
}

int tbrm_selftest_unorm_decode(int device, float* out_u8_256, float* out_u16_65536)
{
    if (!out_u8_256 || !out_u16_65536) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**) &d, (256 + 65536) * sizeof(float)));
    hipError_t e = launch_selftest_decode(d, d + 256, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_u8_256, d, 256 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_u16_65536, d + 256, 65536 * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIP_TRY(e);
    return TBRM_OK;
}

int tbrm_selftest_unorm8_roundtrip(int device, const float* in, size_t n, float* out)
{
    if (!in || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n == 0) return TBRM_OK;
    HIP_TRY(hipSetDevice(device));
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**) &d, 2 * n * sizeof(float)));
    hipError_t e = hipMemcpy(d, in, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest_roundtrip(d, d + n, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d + n, n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIP_TRY(e);
    return TBRM_OK;
}

int tbrm_launch_counters(const tbrm_resources* r, uint64_t out[3])
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 3; ++k) out[k] = r->launches[k];
    return TBRM_OK;
}

int tbrm_flush(tbrm_resources* r)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (int e = bind(r)) return e;
    HIP_TRY(hipStreamSynchronize(r->stream));
    return TBRM_OK;
}

int tbrm_stream(tbrm_resources* r, void** out_hip_stream)
{
    if (!r || !out_hip_stream) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    *out_hip_stream = (void*) r->stream;
    return TBRM_OK;
}

int tbrm_last_gpu_time_ms(tbrm_resources* r, int kind, float* out_ms)
{
    if (!r || !out_ms || kind < 0 || kind > 1) return fail(TBRM_ERR_INVALID_ARG, "bad argument");
    if (!r->ev_valid[kind]) return fail(TBRM_ERR_INVALID_ARG, "no timed call of kind %d yet", kind);
    if (int e = bind(r)) return e;
    HIP_TRY(hipEventSynchronize(r->ev[kind][1]));
    HIP_TRY(hipEventElapsedTime(out_ms, r->ev[kind][0], r->ev[kind][1]));
    return TBRM_OK;
}

int tbrm_host_light_passes(const tbrm_dir_light_params* light, const tbrm_world_params* world, const int32_t lv_dims[3],
                           int border_mode, tbrm_light_pass out[2], int* n_passes)
{
    if (!light || !world || !lv_dims || !out || !n_passes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_light_passes(*light, *world, lv_dims, border_mode, out, n_passes);
    return TBRM_OK;
}

int tbrm_host_local_clipping(const tbrm_world_params* world, float out_center[3], float out_dir[3])
{
    if (!world || !out_center || !out_dir) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_local_clipping(*world, out_center, out_dir);
    return TBRM_OK;
}

float tbrm_host_data_border(const tbrm_windowing_params* w, int border_mode)
{
    if (!w) return 0.0f;
    return host_data_border(*w, border_mode);
}

int tbrm_host_world_to_local(const tbrm_transform* t, float out_m[12])
{
    if (!t || !out_m) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_world_to_local(*t, out_m);
    return TBRM_OK;
}

} // extern "C"
