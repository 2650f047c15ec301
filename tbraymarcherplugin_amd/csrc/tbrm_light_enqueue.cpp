// tbrm_light_enqueue.cpp — a PassPlan's launches on the handle's two streams: block lists and occlusion on the occlusion stream,
// sweeps / chain chunks / slices on the handle's stream, and the events that order them.
#include "tbrm_light_passes.h"

namespace tbrm_host {

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

// What a sweep launch of the plan is handed: the pass's parameters with its factor stores, the sweep's with the handle's
// records, tickets and error word (the launch tag is the caller's: next_sweep_epoch)
static void sweep_launch_params(tbrm_resources* r, const PassPlan& plan, ChunkParams& p, SweepParams& q)
{
    FactorScratch& f = r->f_scratch[plan.f_buf];
    const int ns = plan.n_streams();
    p = plan.p;
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
            e->lists->last_read_op = std::max(e->lists->last_read_op, r->op_serial); // (new_lists: not rewritten under this sweep)
        } else { // computed by this pass: stream a into its entry (if it has one) and the scratch, stream r into the scratch,
                 // both under the ranks of the jointly computed work list
            FactorEntry* const filled = plan.f_entry[0];
            st.fs_keep = (si == 0 && filled) ? filled->base : nullptr;
            st.fs_cap = (si == 0 && filled) ? (uint32_t) filled->cap_blocks : 0u;
            st.fs_spill = f.store[si];
            st.fs_slot = plan.lists->slot;
        }
    }
    q = plan.sq;
    q.rec[0] = r->sweep_rec[0];
    q.rec[1] = r->sweep_rec[1];
    q.ticket = r->sweep_ticket;
    q.error = r->sweep_error;
}

// who read what: later operators wait for this operator's "sweeps done" event (wait_for_readers); the buffers' own events
// are recorded only where a later pass of THIS operator could take the buffer again (an operator of more than two sweep
// passes: every marker between two dependent kernels costs the stream a few microseconds)
static int sweep_read_marks(tbrm_resources* r, const PassPlan& plan)
{
    FactorScratch& f = r->f_scratch[plan.f_buf];
    if (r->op_many_passes) HIP_TRY(hipEventRecord(f.ev_idle, r->stream));
    f.used = true;
    f.last_read_op = r->op_serial;
    f.idle_recorded = r->op_many_passes;
    for (int si = 0; si < plan.n_streams(); ++si)
        if (FactorEntry* const e = plan.f_entry[si]) {
            if (r->op_many_passes) HIP_TRY(hipEventRecord(e->ev_idle, r->stream));
            e->read_yet = true;
            e->last_read_op = r->op_serial;
            e->idle_recorded = r->op_many_passes;
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
    ChunkParams p;
    SweepParams q;
    sweep_launch_params(r, plan, p, q);
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
    q.stamps = nullptr;
    if (q.debug & 2) { // diagnostics: per-tile time stamps of this launch (printed by tbrm_flush)
        const int tiles = p.tiles_x * p.tiles_y;
        if (tiles > r->sweep_stamp_tiles || !r->sweep_stamps) {
            drain_streams(r);
            count_alloc(r, 2, "sweep stamps (diagnostics)");
            (void) hipFree(r->sweep_stamps);
            r->sweep_stamps = nullptr;
            HIP_TRY(hipMalloc((void**) &r->sweep_stamps, (size_t) tiles * 4 * sizeof(unsigned long long)));
        }
        r->sweep_stamp_tiles = tiles;
        r->sweep_stamp_chain[0] = 0;
        r->sweep_stamp_tx = p.tiles_x;
        r->sweep_stamp_sx = q.sx;
        r->sweep_stamp_sy = q.sy;
        q.stamps = r->sweep_stamps;
    }
    HIP_TRY(launch_light_sweep(p, q, plan.mode, r->stream));
    ++r->launches[0];
    ++r->sweep_launches;
    return sweep_read_marks(r, plan);
}

// Two Add passes of DIFFERENT lights that leave the same cube face as ONE sweep (PASS_ADD2; SURVEY.md 8f N4, the multi-light
// optimisation the reference lists as not done, Readme.md:186-187): both streams share the slice loop, its latency chain, the
// hand-off words and the light volume's read-modify-write (light a's, then light b's on its result: exactly pass a followed by
// pass b). Each stream keeps its own factors (the lights' own occlusion launches, the cache entries); `fit` is the two
// streams' common tile order and reach (sweep_fit, not two_way).
int enqueue_sweep_pair(tbrm_resources* r, const PassPlan& pa, const PassPlan& pb, const SweepFit& fit)
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
        if (plan.f_hit[0]) {
            st.fs_keep = e->base; st.fs_cap = (uint32_t) e->cap_blocks; st.fs_spill = nullptr; st.fs_slot = e->lists->slot;
            e->lists->last_read_op = std::max(e->lists->last_read_op, r->op_serial);
        } else {
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

// Several consecutive sweep passes of an operator as ONE launch (k_light_sweep_chain, tbrm_internal.h SweepLink): the fill of
// pass i + 1 runs under the drain of pass i. Every pass's occlusion goes first (the launch waits for all of them); each pass has
// its own region of the hand-off records and its own launch tag, and a table of progress words orders the passes' read-modify-writes
// of the light volume, brick layer by brick layer. plans: n passes (2 .. kSweepChainMax) of one mode (PASS_ADD or PASS_CHANGE),
// one-way, over a UNORM8 light volume, in different scratch buffers.
int enqueue_sweep_chain(tbrm_resources* r, const PassPlan* plans, int n)
{
    if (n < 1 || n > kSweepChainMax) return fail(TBRM_ERR_INVALID_ARG, "a sweep chain of %d passes", n);
    for (int k = 0; k < n; ++k) {
        const int e = (k + 1 < n && dual_fit(plans[k], plans[k + 1])) ? enqueue_dual_occlusion(r, plans[k], plans[k + 1]) : enqueue_sweep_occlusion(r, plans[k]);
        if (e) return e;
    }
    size_t offset[kSweepChainMax + 1] = {0};
    int tiles_max = 0;
    for (int k = 0; k < n; ++k) {
        const PassPlan& plan = plans[k];
        const size_t words = (size_t) plan.D * plan.p.tiles_x * plan.p.tiles_y * (size_t) sweep_record_words(plan.sq.hx, plan.sq.hy, plan.sq.tile_rows);
        offset[k + 1] = offset[k] + ((words + 63) & ~(size_t) 63);
        tiles_max = std::max(tiles_max, plan.p.tiles_x * plan.p.tiles_y);
    }
    if (offset[n] >= ((size_t) 1 << 32)) return fail(TBRM_ERR_UNSUPPORTED, "hand-off records too large");
    if (int e = ensure_sweep(r, std::max<size_t>(offset[n], 1), 0, (size_t) tiles_max)) return e;
    SweepChainArgs c{};
    c.n = n;
    int ticket0 = 0;
    const bool stamps = (tune(TUNE_SWEEP_DEBUG) & 2) != 0; // diagnostics: per-tile time stamps of this launch (printed by tbrm_flush)
    if (stamps) {
        int total = 0;
        for (int k = 0; k < n; ++k) total += plans[k].p.tiles_x * plans[k].p.tiles_y;
        if (total > r->sweep_stamp_tiles || !r->sweep_stamps) {
            drain_streams(r);
            count_alloc(r, 2, "sweep stamps (diagnostics)");
            (void) hipFree(r->sweep_stamps);
            r->sweep_stamps = nullptr;
            HIP_TRY(hipMalloc((void**) &r->sweep_stamps, (size_t) total * 4 * sizeof(unsigned long long)));
        }
        r->sweep_stamp_tiles = total;
        for (int k = 0; k < 4; ++k) r->sweep_stamp_chain[k] = k < n ? plans[k].p.tiles_x * plans[k].p.tiles_y : 0;
    }
    for (int k = 0; k < n; ++k) {
        const PassPlan& plan = plans[k];
        FactorScratch& f = r->f_scratch[plan.f_buf];
        if (plan.occ_mode >= 0) HIP_TRY(hipStreamWaitEvent(r->stream, f.ev_ready, 0));
        for (int si = 0; si < plan.n_streams(); ++si)
            if (plan.f_hit[si]) HIP_TRY(hipStreamWaitEvent(r->stream, plan.f_entry[si]->ev_filled, 0));
        SweepChainPass& P = c.pass[k];
        sweep_launch_params(r, plan, P.p, P.q);
        P.q.rec[0] = r->sweep_rec[0] + offset[k];
        P.q.rec[1] = nullptr;
        P.q.stamps = stamps ? r->sweep_stamps + (size_t) 4 * ticket0 : nullptr;
        P.q.debug &= stamps ? ~1 : ~3;
        if (k > 0) P.q.stagger_ns = 0; // (its tiles start as the pass before's retire: that IS their stagger)
        if (int e = next_sweep_epoch(r, P.q.epoch, k == 0 ? (uint32_t) n : 1u)) return e;
        SweepLink& l = P.link;
        l.prog_out = r->sweep_prog + (size_t) k * r->sweep_prog_stride;
        // (the first pass: in_G = 0 — it needs nothing of the words it asks for; a table nobody writes, so that asking costs nothing)
        l.prog_in = k > 0 ? c.pass[k - 1].link.prog_out : r->sweep_prog + (size_t) kSweepChainMax * r->sweep_prog_stride;
        l.in_epoch = k > 0 ? c.pass[k - 1].q.epoch : 0u;
        l.in_axis = k > 0 ? c.pass[k - 1].p.axis : 0;
        l.in_tiles_x = k > 0 ? c.pass[k - 1].p.tiles_x : 1;
        l.in_down = k > 0 && c.pass[k - 1].p.dir < 0 ? 1 : 0;
        l.in_layer0 = k > 0 ? c.pass[k - 1].p.j0 >> 3 : 0;
        l.in_G = k > 0 ? c.pass[k - 1].p.n_steps >> 3 : 0;
        l.coherent_loads = k >= 1 ? 1 : 0;
        l.ticket0 = ticket0;
        ticket0 += P.p.tiles_x * P.p.tiles_y;
    }
    for (int k = 0; k < n; ++k) c.pass[k].link.total_tiles = ticket0;
    HIP_TRY(launch_light_sweep_chain(c, plans[0].mode, r->stream));
    ++r->launches[0];
    ++r->sweep_launches;
    ++r->chain_launches;
    for (int k = 0; k < n; ++k) sweep_read_marks(r, plans[k]);
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

} // namespace tbrm_host
