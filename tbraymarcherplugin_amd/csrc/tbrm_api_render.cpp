// tbrm_api_render.cpp — the C-ABI's render entry points (include/tbrm.h): the lit raymarch (PerformWindowedLitRaymarch,
// WindowedRaymarchMaterials.usf:36-96, with its cube setup RaymarchMaterialCommon.usf:23-69), its slab stage, the Intensity
// slice view (:187-242), the Octree mode's pyramid and march (GenerateOctreeShader.usf:28-107, :99-183) and the nominal-sample
// count of the benchmark's metric. Handle life cycle, inputs and light operators: tbrm_api.cpp.
#include "tbrm_resources.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace tbrm;
using namespace tbrm_host;

namespace {

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
    p.xcd_rows = tune(TUNE_RAY_XCD_ROWS);
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
    {   // (measured: frames of 128^3 / 256^3 volumes lose 15 % to the bookkeeping, 512^3 is even, 2048^2 rays through 512^3 gain 12 %)
        const int ws = tune(TUNE_RAY_WAVE_SKIP);
        p.wave_skip = ws > 0 || (ws < 0 && std::min(r->desc.dim_x, std::min(r->desc.dim_y, r->desc.dim_z)) >= 384) ? 1 : 0;
    }
    p.share_grid = (r->lv_dims[0] == r->desc.dim_x && r->lv_dims[1] == r->desc.dim_y && r->lv_dims[2] == r->desc.dim_z &&
                    !r->resident && tune(TUNE_SHARE_GRID)) ? 1 : 0; // (the two volumes of a slab-resident handle relocate different layers)
    p.tile_x0 = tile->x0; p.tile_y0 = tile->y0; p.tile_w = tile->w; p.tile_h = tile->h;
    p.row_group_step = tile->row_group_step > 0 ? tile->row_group_step : 1;
    p.steps = rp->steps;
    p.jitter_frame = rp->jitter_frame;
    p.bnx = r->bn[0]; p.bny = r->bn[1]; p.bnz = r->bn[2];
    p.tab = r->d_ray_tab;
    return TBRM_OK;
}

} // namespace

extern "C" {

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
    return sweep_failed(r); // (a frame lit by a light volume that a failed sweep left undefined is not handed out as good)
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

} // extern "C"
