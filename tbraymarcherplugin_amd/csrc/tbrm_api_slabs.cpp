// tbrm_api_slabs.cpp — the C-ABI's slab entry points (include/tbrm.h "slabs"): a light operator partitioned over handles in
// light-volume z slabs, stepped pass by pass and chunk by chunk by the host that exchanges the planes, and the
// slab-resident handles' layer traffic (upload / download / halo layers). The planning underneath: tbrm_light_plan.cpp.
#include "tbrm_resources.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace tbrm;
using namespace tbrm_host;

extern "C" {

int tbrm_slab_light_begin(tbrm_resources* r, const tbrm_dir_light_params* removed, const tbrm_dir_light_params* light, int added,
                          const tbrm_world_params* world, const tbrm_slab* slab, int32_t* n_passes)
{
    if (!r || !light || !world || !slab || !n_passes) return fail(TBRM_ERR_INVALID_ARG, "null argument");
    if (!initialized(r)) return fail(TBRM_ERR_NOT_INITIALIZED, "resources have no volume or transfer function");
    *n_passes = 0;
    if (int e = bind(r)) return e;
    if (int e = ensure_reserved(r)) return e; // (a handle nobody reserved: once, with the defaults — tbrm_resources_reserve)
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
    // (a slab-partitioned operator is a light operator of its own for the sweep's buffer bookkeeping: its sweeps record their
    // buffers' own idle events — no "operator done" event is recorded for it, a later operator's occlusion then waits for
    // everything enqueued so far: wait_for_readers)
    ++r->op_serial;
    r->block_lists_op_floor = r->block_lists_serial; // (lists the stored plan points at are younger: never pruned or recycled under it)
    r->op_many_passes = true;
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
    if (op.held_lists) { --op.held_lists->users; op.held_lists = nullptr; }
    const int e = plan_pass(r, op.base, op.a[pass], op.change ? &op.r[pass] : nullptr, op.b_added, &op.slab, op.plan);
    if (e == TBRM_OK && op.plan.lists) { // the plan is stored across API calls (tbrm_slab_pass_chunk enqueues later): its lists stay put
        op.held_lists = op.plan.lists;
        ++op.held_lists->users;
    }
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
    quiesce_occ_stream(r);
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
    r->octree_valid = false;
    r->minmax_valid = false;
    ++r->data_gen;
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
    return sweep_failed(r); // (the slices of a light volume a failed sweep left undefined are not handed out as good)
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

} // extern "C"
