// tbrm_factor_cache.cpp — the occlusion stores of the chunked chain, the scratch stores of the sweep and the factor cache
// (tbrm_resources.h FactorEntry): allocation, look-up, reuse and release.
#include "tbrm_light_passes.h"

namespace tbrm_host {

// ---- occlusion stores and the factor cache (tbrm_resources.h) ------------------------------------------------------------

void drain_streams(tbrm_resources* r)
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
int ensure_store(tbrm_resources* r, OccStore* st, int slices, size_t slice_elems, size_t flag_bytes)
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

int ensure_occ_stream(tbrm_resources* r)
{
    if (r->occ_stream) return TBRM_OK;
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&r->occ_stream, hipStreamNonBlocking, least)); // (the handle's stream's priority instead: measured, no gain)
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&r->occ_ev_fork[k], event_flags()));
        HIP_TRY(hipEventCreateWithFlags(&r->occ_ev_ready[k], event_flags()));
    }
    for (hipEvent_t& ev : r->op_done) HIP_TRY(hipEventCreateWithFlags(&ev, event_flags()));
    return TBRM_OK;
}

// the scratch of a sweep pass with `blocks` occlusion blocks: stores of `streams` streams (every block could be live), the
// page of ones
int ensure_factor_scratch(tbrm_resources* r, int b, size_t blocks, int streams)
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
FactorKey factor_key(const tbrm_resources* r, const PropParams& base, const tbrm_light_pass& q, bool guard, int start, int D)
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
void estimate_scope(tbrm_resources* r, const PropParams& base)
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
void resolve_entry(tbrm_resources* r, FactorEntry* e, bool wait)
{
    if (e->resolved || !e->enqueued) return;
    size_t count = 0;
    if (!block_lists_count(e->lists, wait, &count)) return;
    e->resolved = true;
    e->valid = count <= e->cap_blocks;
    if (e->key.data_gen == r->f_est_key[0] && e->key.tf_gen == r->f_est_key[1] && !memcmp(e->key.win, r->f_est_win, sizeof(e->key.win)))
        r->f_est_blocks = std::max(r->f_est_blocks, count);
}

FactorEntry* kept_find(tbrm_resources* r, const FactorKey& key)
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
FactorEntry* kept_new(tbrm_resources* r, const FactorKey& key, size_t want, BlockLists* lists)
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

void use_kept(tbrm_resources* r, FactorEntry* e, bool leaves_the_scene)
{
    e->spent = leaves_the_scene;
    e->pinned = true;
    e->last_use = ++r->kept_clock;
}

void unpin_kept(tbrm_resources* r)
{
    for (FactorEntry* e : r->kept) e->pinned = false;
}

bool cache_usable(const tbrm_resources* r) { return !force_slice_kernel() && tune(TUNE_LIGHT_CACHE_MB) != 0; }

} // namespace tbrm_host
