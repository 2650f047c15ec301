// tbrm_api.cpp — the C-ABI of include/tbrm.h over the gfx950 kernels.
//
// Plays the role of the reference's game-thread operator library:
//   URaymarchUtils::AddDirLightToSingleVolume / ChangeDirLightInSingleVolume / ClearResourceLightVolumes
//       Source/Raymarcher/Private/Util/RaymarchUtils.cpp:35-111
//   (their render-thread drivers, LightingShaders.cpp:35-326, are tbrm_light_operators.cpp)
//   ARaymarchVolume::InitializeRaymarchResources / FreeRaymarchResources
//       Source/Raymarcher/Private/Actor/RaymarchVolume.cpp:821-949
// Every call enqueues on the handle's HIP stream (FIFO, like ENQUEUE_RENDER_COMMAND) and returns; parameter
// structs are copied at call time (the reference captures them by value, RaymarchUtils.cpp:63-66).
// There is no CPU path: without a HIP device every data call fails with TBRM_ERR_NO_DEVICE.
#include "tbrm_resources.h"
#include "tbrm_light_sweep.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace tbrm;
using namespace tbrm_host;

namespace {
thread_local char g_error[512] = "";

// ---- tunables (tbrm_internal.h): name, default; initialised from TBRM_<NAME> when the library is loaded ------------
struct TunableDef { const char* name; int def; };
const TunableDef kTunables[TUNE_COUNT] = {
    {"force_slice_kernel", 0}, {"chunk_steps", 0}, {"occ_slices", 0}, {"sparse_occ", 1}, {"occ_list", 1}, {"light_cache_mb", -1},
    {"light_batching", 1}, {"share_grid", 1}, {"ray_wave_skip", -1}, {"ray_lanes", 0}, {"chain_fast_loop", 1},
    {"chain_rect_planes", 1}, {"occ_overlap", 2}, {"light_sweep", 1}, {"sweep_prefetch", 0}, {"stream_priority", 0}, {"sweep_debug", 0},
    {"sweep_timeout_ms", 0}, {"fast_window_div", 1}, {"slab_sweep", 0}, {"gpu_timing", 1}, {"ray_tables", 1}, {"sweep_epoch_preset", 0}, {"occ_dual", 1}, {"sweep_chain", 4}, {"ray_xcd_rows", 1},
};
struct TunableStore {
    std::atomic<int> v[TUNE_COUNT];
    TunableStore()
    {
        for (int t = 0; t < TUNE_COUNT; ++t) {
            char env[64] = "TBRM_";
            size_t n = strlen(env);
            for (const char* c = kTunables[t].name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char) toupper((unsigned char) *c);
            env[n] = 0;
            const char* e = getenv(env);
            v[t].store(e && *e ? atoi(e) : kTunables[t].def, std::memory_order_relaxed);
        }
    }
};
TunableStore g_tunables;
} // namespace

namespace tbrm {
int tune(Tunable t) { return g_tunables.v[t].load(std::memory_order_relaxed); }
} // namespace tbrm

int tbrm_set_tunable(const char* name, int32_t value)
{
    if (!name) return tbrm_host::fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int t = 0; t < TUNE_COUNT; ++t)
        if (!strcmp(name, kTunables[t].name)) { g_tunables.v[t].store(value, std::memory_order_relaxed); return TBRM_OK; }
    return tbrm_host::fail(TBRM_ERR_INVALID_ARG, "no tunable named '%s'", name);
}

int tbrm_get_tunable(const char* name, int32_t* value)
{
    if (!name || !value) return tbrm_host::fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int t = 0; t < TUNE_COUNT; ++t)
        if (!strcmp(name, kTunables[t].name)) { *value = tune((Tunable) t); return TBRM_OK; }
    return tbrm_host::fail(TBRM_ERR_INVALID_ARG, "no tunable named '%s'", name);
}

namespace tbrm_host {

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

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
// May the kernels divide by this window's width without the division sequence (tbrm_device_math.h tf_position_fast)? The width's
// reciprocal correctly rounded (IEEE float division here), a significand that is not all ones, centre and width of moderate size
// (no over- / underflow between a = value - centre + width / 2 in [-2^41, 2^41] and the quotient's correction term).
static void window_division(float center, float width, float* inv_width, int* fast_div)
{
    *inv_width = 0.0f;
    *fast_div = 0;
    if (tune(TUNE_FAST_WINDOW_DIV) == 0 || !std::isfinite(center) || !std::isfinite(width)) return;
    const float aw = std::fabs(width), ac = std::fabs(center);
    if (!(aw >= 0x1p-40f && aw <= 0x1p40f) || ac > 0x1p40f) return;
    uint32_t bits;
    memcpy(&bits, &width, 4);
    if ((bits & 0x007fffffu) == 0x007fffffu) return;
    *inv_width = 1.0f / width;
    *fast_div = 1;
}

WindowDev window_dev_of(float center, float width, int low_cutoff, int high_cutoff)
{
    WindowDev w{center, width, low_cutoff ? 1.0f : 0.0f, high_cutoff ? 1.0f : 0.0f, 0.0f, 0};
    window_division(center, width, &w.inv_width, &w.fast_div);
    return w;
}

WindowDev window_dev(const tbrm_resources* r) { return window_dev_of(r->win.center, r->win.width, r->win.low_cutoff, r->win.high_cutoff); }

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
    if ((tune(TUNE_SWEEP_DEBUG) & 32) || tune(TUNE_GPU_TIMING) == 0) { r->ev_valid[kind] = false; return TBRM_OK; }
    HIP_TRY(hipEventRecord(r->ev[kind][0], r->stream));
    return TBRM_OK;
}
int end_timed(tbrm_resources* r, int kind)
{
    if ((tune(TUNE_SWEEP_DEBUG) & 32) || tune(TUNE_GPU_TIMING) == 0) return TBRM_OK;
    HIP_TRY(hipEventRecord(r->ev[kind][1], r->stream));
    r->ev_valid[kind] = true;
    return TBRM_OK;
}

} // namespace tbrm_host

namespace tbrm_host {

RelayoutParams relayout_params(const void* src, void* dst, const int dims[3], const int bn[3], size_t elem, bool to_bricks)
{
    return RelayoutParams{src, dst, dims[0], dims[1], dims[2], bn[0], bn[0] * bn[1], bn[2], (int) elem, to_bricks ? 1 : 0};
}

} // namespace tbrm_host


namespace tbrm_host {

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
        int prefix[258]; // [257]: the shell-transparency flag (k_shell_transparent clears it)
        prefix[257] = 1;
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
        // may the contribution cache serve the Add shader from what the Change shader propagated and the other way round?
        // (one more host round trip per new transfer function / window / volume, next to the one above)
        r->shell_transparent = false;
        if (r->res_data.lo == 0 && r->res_data.hi == r->bn[2]) {
            int flag = 0;
            HIP_TRY(launch_shell_transparent(ep, r->bn[0], r->bn[1], r->bn[2], host_data_border(r->win, r->desc.border_mode),
                                             r->d_alpha_prefix + 257, r->stream));
            HIP_TRY(hipMemcpyAsync(&flag, r->d_alpha_prefix + 257, sizeof(int), hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
            r->shell_transparent = flag != 0;
        }
        // distance field for empty-space leaping: three separable passes, x then y then z
        const int mode = r->desc.data_address_mode == TBRM_ADDRESS_CLAMP ? ADDR_CLAMP : ADDR_WRAP;
        for (int axis = 0; axis < 3; ++axis) {
            DistParams dp{r->d_empty, axis == 0 ? nullptr : r->d_dist[(axis + 1) & 1], r->d_dist[axis & 1],
                {r->bn[0], r->bn[1], r->bn[2]}, axis};
            HIP_TRY(launch_brick_dist(dp, mode, r->stream));
        }
        r->empty_valid = true;
        ++r->empty_gen;
        r->occ_inputs_changed = true; // (the occlusion stream orders itself behind this: order_behind_inputs)
    }
    return TBRM_OK;
}

} // namespace tbrm_host

// ---------------------------------------------------------------------------------------------------------------

extern "C" {

const char* tbrm_version(void) { return "tbrm-mi355x 0.4.0 (gfx950)"; }
int tbrm_abi_version(void) { return TBRM_ABI_VERSION; }
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
    {
        int least = 0, greatest = 0;
        CREATE_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int want = tune(TUNE_STREAM_PRIORITY);
        CREATE_TRY(hipStreamCreateWithPriority(&r->stream, hipStreamNonBlocking, want > 0 ? greatest : (want < 0 ? least : 0)));
    }
    if (hipDeviceGetAttribute(&r->n_cus, hipDeviceAttributeMultiprocessorCount, desc->device) != hipSuccess || r->n_cus <= 0) r->n_cus = 256;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) r->device_total_bytes = total_b; // (the factor cache's automatic budget: asked once)
        else (void) hipGetLastError();
    }
    {
        tbrm_resources::Residency& q = r->res_data;
        const size_t bytes = (size_t) (q.hi - q.lo + (q.wrap_src >= 0 ? 1 : 0)) * q.layer_bytes;
        CREATE_TRY(hipMalloc(&q.alloc, bytes));
        r->d_data = (char*) q.alloc - (size_t) q.lo * q.layer_bytes;
        // a slab-resident handle is filled layer by layer (tbrm_upload_volume_slices): what has not arrived yet reads as 0,
        // not as whatever the allocation held
        if (r->resident) CREATE_TRY(hipMemsetAsync(q.alloc, 0, bytes, r->stream));
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
    CREATE_TRY(hipMalloc((void**) &r->d_alpha_prefix, 258 * sizeof(int)));
    CREATE_TRY(hipMalloc((void**) &r->d_counter, sizeof(unsigned long long)));
    if (!r->resident) { // k_raymarch_lit's offset tables (RayParams::tab): per axis, for texel index i = -2 .. n + 1, where the data
                        // sampler's address mode puts it in the bricked layout, and its brick's share of the linear brick index
        const int dn[3] = {desc->dim_x, desc->dim_y, desc->dim_z};
        std::vector<uint2> tab;
        for (int c = 0; c < 3; ++c)
            for (int i = -2; i < dn[c] + 2; ++i) {
                int w = i;
                if (desc->data_address_mode == TBRM_ADDRESS_CLAMP) w = clamp_int(i, 0, dn[c] - 1);
                else { w %= dn[c]; if (w < 0) w += dn[c]; }
                const uint32_t in_brick = (uint32_t) (w & 7) << (3 * c);
                const uint32_t brick = (uint32_t) (w >> 3) * (c == 0 ? 1u : (c == 1 ? (uint32_t) r->dbn[0] : (uint32_t) (r->dbn[0] * r->dbn[1])));
                tab.push_back(make_uint2((brick << 9) | in_brick, brick));
            }
        if (tab.size() & 1) tab.push_back(make_uint2(0, 0));
        CREATE_TRY(hipMalloc((void**) &r->d_ray_tab, tab.size() * sizeof(uint2)));
        CREATE_TRY(hipMemcpy(r->d_ray_tab, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
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
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream); // (an occlusion beside a chain reads the volume and the skipping metadata)
    (void) hipFree(r->res_data.alloc);
    (void) hipFree(r->d_tf);
    (void) hipFree(r->res_light.alloc);
    for (auto& axis : r->d_buf)
        for (void* b : axis) (void) hipFree(b);
    for (float* pl : r->d_plane) (void) hipFree(pl);
    delete r->slab_op;
    release_occ_stores(r);
    release_sweep(r);
    (void) hipFree(r->d_minmax);
    (void) hipFree(r->d_empty);
    if (r->occ_stream) {
        (void) hipStreamSynchronize(r->occ_stream);
        (void) hipStreamDestroy(r->occ_stream);
        for (int b = 0; b < 2; ++b) { (void) hipEventDestroy(r->occ_ev_fork[b]); (void) hipEventDestroy(r->occ_ev_ready[b]); }
        for (hipEvent_t e : r->op_done)
            if (e) (void) hipEventDestroy(e);
    }
    for (uint16_t* o : r->d_octree) (void) hipFree(o);
    for (uint8_t* d : r->d_dist) (void) hipFree(d);
    (void) hipFree(r->d_alpha_prefix);
    (void) hipFree(r->d_counter);
    (void) hipFree(r->d_ray_tab);
    (void) hipFree(r->d_out);
    for (auto& k : r->ev)
        for (hipEvent_t e : k)
            if (e) (void) hipEventDestroy(e);
    if (r->stream) (void) hipStreamDestroy(r->stream);
    delete r;
    return TBRM_OK;
}

int tbrm_resources_reserve(tbrm_resources* r, int32_t n_lights, uint32_t flags)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_lights < 0 || (flags & ~1u)) return fail(TBRM_ERR_INVALID_ARG, "n_lights %d / flags %u", (int) n_lights, (unsigned) flags);
    if (int e = bind(r)) return e;
    return reserve_resources(r, n_lights, flags);
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
    quiesce_occ_stream(r); // (nothing on the second stream may still be reading the volume)
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
    ++r->data_gen;
    return TBRM_OK;
}

int tbrm_upload_volume_device(tbrm_resources* r, const void* device_voxels, size_t n_bytes)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: upload its layers with tbrm_upload_volume_slices");
    if (!r || !device_voxels) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (n_bytes != r->data_bytes) return fail(TBRM_ERR_INVALID_ARG, "volume is %zu bytes, expected %zu", n_bytes, r->data_bytes);
    if (int e = bind(r)) return e;
    quiesce_occ_stream(r);
    const int dims[3] = {r->desc.dim_x, r->desc.dim_y, r->desc.dim_z};
    HIP_TRY(launch_relayout(relayout_params(device_voxels, r->d_data, dims, r->dbn, format_bytes(r->desc.data_format), true), r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->has_volume = true;
    r->octree_valid = false;
    r->minmax_valid = false;
    ++r->data_gen;
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
    ++r->tf_gen;
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
    const int e = enqueue_add(r, *light, added != 0, *world);
    const int e2 = end_timed(r, 0); // also after a failure: the events then bracket whatever was enqueued
    return e ? e : e2;
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
    const int e = enqueue_add_batch(r, lights, n_lights, added != 0, *world, schedule, n_entries);
    const int e2 = end_timed(r, 0);
    return e ? e : e2;
}

int tbrm_change_dir_light(tbrm_resources* r, const tbrm_dir_light_params* old_light, const tbrm_dir_light_params* new_light,
                          const tbrm_world_params* world, int* light_added, int gpu_sync)
{
    if (r && r->resident) return fail(TBRM_ERR_UNSUPPORTED, "slab-resident handle: light operators run through tbrm_slab_*");
    (void) gpu_sync; // accepted and ignored (RaymarchUtils.cpp:70-92 never reads it)
    if (light_added) *light_added = 0;
    if (!r || !old_light || !new_light || !world) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function"); // :74-80
    if (light_added) *light_added = 1;
    if (int e = bind(r)) return e;
    if (int e = begin_timed(r, 0)) return e;
    const int e = enqueue_change(r, *old_light, *new_light, *world);
    const int e2 = end_timed(r, 0);
    return e ? e : e2;
}

int tbrm_clear_light_volume(tbrm_resources* r, float clear_value)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!r->d_light) return TBRM_OK; // RaymarchUtils.cpp:106-109
    if (int e = bind(r)) return e;
    sweep_failure_cleared(r); // (a sweep that failed left the light volume undefined: this call defines it again)
    if (int e = begin_timed(r, 0)) return e;
    const tbrm_resources::Residency& q = r->res_light; // (all layers of an ordinary handle); padding voxels are never sampled
    const size_t n = (size_t) r->lbn[0] * r->lbn[1] * 512 * (size_t) (q.hi - q.lo + (q.wrap_src >= 0 ? 1 : 0));
    HIP_TRY(launch_fill(q.alloc, r->lv_fmt, n, clear_value, r->stream));
    return end_timed(r, 0);
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
    return sweep_failed(r); // (the slices of a light volume a failed sweep left undefined are not handed out as good)
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
    sweep_failure_cleared(r); // (the light volume is defined again)
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

int tbrm_selftest_window_division(int device, float center, float width, uint64_t* out_mismatches, int* out_fast_path)
{
    if (!out_mismatches) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    WindowDev w = window_dev_of(center, width, 1, 1);
    if (out_fast_path) *out_fast_path = w.fast_div;
    *out_mismatches = 0;
    if (!w.fast_div) return TBRM_OK; // (the kernels divide for this window: nothing to compare)
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**) &d, sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = launch_selftest_window_division(w, d, nullptr);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d, sizeof(bad), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIP_TRY(e);
    *out_mismatches = bad;
    return TBRM_OK;
}

int tbrm_selftest_opacity_correction(int device, float step0, float step1, uint64_t* out_mismatches)
{
    if (!out_mismatches) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!(step0 >= 0.0f) || !(step1 >= 0.0f) || std::isinf(step0) || std::isinf(step1)) return fail(TBRM_ERR_INVALID_ARG, "the short form is used for finite step sizes >= 0 only");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**) &d, sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = launch_selftest_opacity_correction(step0, step1, d, nullptr);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d, sizeof(bad), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIP_TRY(e);
    *out_mismatches = bad;
    return TBRM_OK;
}

int tbrm_launch_counters(const tbrm_resources* r, uint64_t out[3])
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 3; ++k) out[k] = r->launches[k];
    return TBRM_OK;
}

int tbrm_sweep_launches(const tbrm_resources* r, uint64_t* out)
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    *out = r->sweep_launches;
    return TBRM_OK;
}

int tbrm_path_counters(const tbrm_resources* r, uint64_t out[TBRM_PATH_COUNTERS])
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < TBRM_PATH_COUNTERS; ++k) out[k] = 0;
    out[0] = r->passes[0]; out[1] = r->passes[1]; out[2] = r->passes[2];
    out[3] = r->sweep_launches; out[4] = r->launches[0] - r->sweep_launches; out[5] = r->launches[1];
    out[6] = r->occ_launches; out[7] = r->dual_launches;
    out[8] = r->kept_hits;
    out[9] = r->launches[2];
    out[10] = r->pair_sweeps;
    out[11] = r->lists_launches;
    out[12] = r->alloc_calls;
    out[13] = r->sync_calls;
    out[14] = r->chain_launches;
    return TBRM_OK;
}

int tbrm_light_cache_stats(const tbrm_resources* r, uint64_t out[4])
{
    if (!r || !out) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    out[0] = r->kept_hits;
    out[1] = r->kept_computed;
    out[2] = r->kept.size();
    // an axis pass covers the light volume once, whichever axis it runs along
    out[3] = (uint64_t) kept_bytes(r);
    return TBRM_OK;
}

int tbrm_light_cache_clear(tbrm_resources* r)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (int e = bind(r)) return e;
    HIP_TRY(hipStreamSynchronize(r->stream)); // (every occlusion launch beside the chain has been waited for by a chain)
    release_kept(r);
    return TBRM_OK;
}

int tbrm_flush(tbrm_resources* r)
{
    if (!r) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (int e = bind(r)) return e;
    HIP_TRY(hipStreamSynchronize(r->stream));
    return sweep_check(r);
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
    return sweep_failed(r); // (the time of an operator whose sweep gave up is not a time worth reporting)
}

int tbrm_host_light_passes(const tbrm_dir_light_params* light, const tbrm_world_params* world, const int32_t lv_dims[3],
                           int border_mode, tbrm_light_pass out[2], int* n_passes)
{
    if (!light || !world || !lv_dims || !out || !n_passes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    host_light_passes(*light, *world, lv_dims, border_mode, out, n_passes);
    return TBRM_OK;
}

// Which kernel each axis pass of AddDirLight(light) would take for a light volume of these dimensions — the planner alone, no device
// (tools/planner_anisotropic.py: how often do real CT shapes fall off the sweep, and why?)
int tbrm_host_plan_light(const tbrm_dir_light_params* light, const tbrm_world_params* world, const int32_t lv_dims[3], int light_volume_32bit,
                         int32_t out[8], int* n_passes)
{
    if (!light || !world || !lv_dims || !out || !n_passes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    tbrm_resources fake{}; // (the planner reads the dimensions, the format and the CU count of a handle: no device behind this one)
    for (int c = 0; c < 3; ++c) fake.lv_dims[c] = lv_dims[c];
    fake.lv_fmt = light_volume_32bit ? FMT_F32 : FMT_U8;
    fake.desc.border_mode = TBRM_BORDER_ENGINE_8BIT;
    tbrm_light_pass passes[2];
    int n = 0;
    host_light_passes(*light, *world, lv_dims, TBRM_BORDER_ENGINE_8BIT, passes, &n);
    *n_passes = n;
    for (int k = 0; k < 2; ++k) {
        int32_t* o = out + 4 * k;
        o[0] = -1; o[1] = o[2] = o[3] = 0;
        if (k >= n) continue;
        SweepFit sf;
        ChunkFit cf;
        if (sweep_fit(&fake, passes[k], nullptr, PASS_ADD, sf) && ceil_div(passes[k].td[2], 8) * 8 <= sweep_max_slices()) { o[0] = 0; o[1] = sf.hx; o[2] = sf.hy; }
        else if (chunk_fit(&fake, passes[k], nullptr, cf)) { o[0] = 1; o[1] = cf.M; o[3] = ceil_div(passes[k].td[2], 8) * 8 > sweep_max_slices() ? 6 : sweep_decline_reason(&fake, passes[k]); }
        else { o[0] = 2; o[3] = ceil_div(passes[k].td[2], 8) * 8 > sweep_max_slices() ? 6 : sweep_decline_reason(&fake, passes[k]); }
    }
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
