// tbrm_factor_cache.cpp — the occlusion stores of the chunked chain, the scratch stores of the sweep and the factor cache
// (tbrm_resources.h FactorEntry): allocation, look-up, reuse and release.
#include "tbrm_light_passes.h"

namespace tbrm_host {

// ---- occlusion stores and the factor cache (tbrm_resources.h) ------------------------------------------------------------

void drain_streams(tbrm_resources* r)
{
    ++r->sync_calls;
    (void) hipStreamSynchronize(r->stream);
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream);
}
void drain_streams_public(tbrm_resources* r) { drain_streams(r); }

void count_alloc(tbrm_resources* r, int calls, const char* what)
{
    r->alloc_calls += (uint64_t) calls;
    if (tune(TUNE_SWEEP_DEBUG) & 64) fprintf(stderr, "[tbrm alloc] op %llu: %d calls for %s\n", (unsigned long long) r->op_serial, calls, what);
}

// ordering events come from a pool the handle keeps (made by tbrm_resources_reserve, refilled when an entry goes)
hipEvent_t take_event(tbrm_resources* r)
{
    if (!r->event_pool.empty()) {
        const hipEvent_t ev = r->event_pool.back();
        r->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    count_alloc(r, 1, "an ordering event (pool empty)");
    if (hipEventCreateWithFlags(&ev, event_flags()) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return ev;
}
void give_event(tbrm_resources* r, hipEvent_t ev)
{
    if (ev) r->event_pool.push_back(ev);
}

// Has every sweep of operator `op` completed? Operators record "my sweeps are done" in one of kOpEvents slots (run_passes); a slot
// that a LATER operator has taken answers for the earlier one too (one stream, in order). Never blocks; false when in doubt.
bool op_finished(tbrm_resources* r, uint64_t op)
{
    if (op == 0) return true;
    for (int k = 0; k < tbrm_resources::kOpEvents; ++k) { // (any operator from `op` on that has recorded its event and seen it fire)
        if (!r->op_done[k] || r->op_done_serial[k] < op) continue;
        if (hipEventQuery(r->op_done[k]) == hipSuccess) return true;
        (void) hipGetLastError();
    }
    return false; // (not recorded: the operator is being enqueued, was a slab operation, or failed half way)
}

static void free_entry(tbrm_resources* r, FactorEntry* e)
{
    if (r->cache_arena.owns(e->base)) r->cache_arena.give(e->base, e->bytes());
    else if (e->base) { count_alloc(r, 1, "freeing a cache entry allocated outside the arena"); (void) hipFree(e->base); }
    if (e->lists) --e->lists->users;
    give_event(r, e->ev_filled);
    give_event(r, e->ev_idle);
    delete e;
}

void release_kept(tbrm_resources* r)
{
    if (!r->kept.empty()) drain_streams(r);
    for (FactorEntry* e : r->kept) free_entry(r, e);
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
    (void) hipFree(r->cache_arena_alloc);
    r->cache_arena_alloc = nullptr;
    r->cache_arena.reset(nullptr, 0);
    for (hipEvent_t ev : r->event_pool) (void) hipEventDestroy(ev);
    r->event_pool.clear();
    r->reserved = false;
}

// room for `slices` planes of slice_elems floats behind the page of ones, and for the flags / lists of a pass
int ensure_store(tbrm_resources* r, OccStore* st, int slices, size_t slice_elems, size_t flag_bytes)
{
    const size_t elems = (size_t) slices * slice_elems; // (the passes of a non-cubic volume have planes of different sizes)
    if (elems > st->capacity || !st->base) {
        ++r->sync_calls;
        count_alloc(r, 2, "a chunked-chain occlusion store");
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
        ++r->sync_calls;
        count_alloc(r, 4, "a chunked-chain flag store");
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
    count_alloc(r, 1 + 4 + tbrm_resources::kOpEvents, "the occlusion stream and its events");
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
        count_alloc(r, 1, "the page of ones");
        ++r->sync_calls;
        HIP_TRY(hipMalloc((void**) &r->d_ones, 1024 * sizeof(float)));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) r->d_ones, 0x3f800000, 1024, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream)); // (read from the occlusion stream's sweeps' predecessors: simplest to have it done)
    }
    if (!f.ev_ready) {
        count_alloc(r, 2, "a scratch buffer's events");
        HIP_TRY(hipEventCreateWithFlags(&f.ev_ready, event_flags()));
        HIP_TRY(hipEventCreateWithFlags(&f.ev_idle, event_flags()));
    }
    const bool grow_store = blocks > f.store_blocks;
    bool need = false;
    for (int si = 0; si < streams; ++si) need = need || grow_store || !f.store[si];
    if (!need) return TBRM_OK;
    drain_streams(r);
    count_alloc(r, std::max(streams, 1), "a factor scratch store");
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
    return r->device_total_bytes / 8; // (asked of the device once, when the handle was created)
}

// An entry for a pass about to be computed, sized for `want` blocks: the buffer of an entry whose light has left the scene if one
// is large enough (lights that merely move recycle each other's buffers), else a piece of the handle's arena
// (tbrm_resources_reserve) — after giving back the pieces of entries that are of no further use and that nothing in flight
// reads (op_finished: asked, never waited for), least recently used first. null: the cache is off, or nothing can be had
// without waiting — the pass then computes into the scratch store alone and is sampled again next time. Nothing here
// allocates device memory, asks the device how much it has, or waits for a stream.
FactorEntry* kept_new(tbrm_resources* r, const FactorKey& key, size_t want, BlockLists* lists)
{
    const size_t bytes = want * 2048 * sizeof(float);
    FactorEntry* e = nullptr;
    auto fits = [&](const FactorEntry* c) { return !c->pinned && c->cap_blocks >= want && c->cap_blocks <= want + want / 2 + 64; };
    auto useless = [&](const FactorEntry* c) { return (c->resolved && !c->valid) || c->spent || !c->enqueued; };
    auto reusable = [&](const FactorEntry* c) { return fits(c) && useless(c); };
    // An entry that the operator just before this one read (the removed side of its Change) is still being read by that
    // operator's sweeps when this operator's occlusion could start — beside those very sweeps, which leave two thirds of a
    // CU's issue slots idle. Reusing it would make the occlusion wait for them (measured: the whole 0.38 ms of it exposed
    // in front of every operator of the benchmark's loop); an entry retired an operator earlier is free by then.
    auto settled = [&](const FactorEntry* c) { return !(c->read_yet && c->last_read_op + 1 >= r->op_serial); };
    for (FactorEntry* c : r->kept) // dropped and spent entries first, oldest first
        if (reusable(c) && settled(c) && (!e || c->last_use < e->last_use)) e = c;
    const size_t budget = kept_budget(r);
    if (bytes > budget) return nullptr;
    // nothing in flight writes or reads it: its fill has completed (or never started), its last reader's operator is done
    auto quiet = [&](const FactorEntry* c) {
        if (c->pinned) return false;
        if (c->enqueued && c->ev_filled && hipEventQuery(c->ev_filled) != hipSuccess) { (void) hipGetLastError(); return false; }
        return !c->read_yet || op_finished(r, c->last_read_op);
    };
    void* piece = nullptr;
    if (!e) {
        auto drop = [&](FactorEntry* v) {
            r->kept.erase(std::find(r->kept.begin(), r->kept.end(), v));
            free_entry(r, v);
        };
        for (int pass = 0; pass < 2 && !piece; ++pass) { // 0: what is of no use anyway; 1: the least recently used
            for (;;) {
                if (kept_bytes(r) + bytes <= budget && (piece = r->cache_arena.take(bytes))) break;
                FactorEntry* v = nullptr;
                for (FactorEntry* c : r->kept)
                    if ((pass == 1 || useless(c)) && quiet(c) && (!v || c->last_use < v->last_use)) v = c;
                if (!v) break;
                drop(v);
            }
        }
        if (!piece) // no room without waiting: the spent entry that is still being read after all (its readers are waited for on the
                    // occlusion stream, not here: note_readers), else do without
            for (FactorEntry* c : r->kept)
                if (reusable(c) && (!e || c->last_use < e->last_use)) e = c;
        if (!piece && !e) return nullptr;
    }
    if (!e) {
        e = new FactorEntry{};
        e->base = (float*) piece;
        e->ev_filled = take_event(r);
        e->ev_idle = take_event(r);
        if (!e->ev_filled || !e->ev_idle) {
            free_entry(r, e);
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

// ---- tbrm_resources_reserve ------------------------------------------------------------------------------------------------
// Everything the whole-volume light operators of this handle need, sized from the volume's dimensions, the number of lights the
// scene will hold and the factor cache's budget — what the reference does in InitializeRaymarchResources (RaymarchVolume.cpp:821-920):
// the occlusion stream and its events, the page of ones, the four factor scratch buffers (both streams: every block could be live),
// hand-off records for previous-slice taps up to two texels from the pixel, tickets, block lists, ordering events, and ONE arena
// for the factor cache's entries. Afterwards the operators allocate nothing (tbrm_path_counters [12] stands still); only a pass
// outside that envelope (taps further away, the chunked-chain fallback without flag 1) still allocates what it needs, once.
static size_t pass_blocks_max(const tbrm_resources* r, size_t* slice_elems_max, int* tiles_max, int* depth_max)
{
    size_t best = 0, elems = 0;
    int tiles = 0, depth = 0;
    for (int a = 0; a < 3; ++a) {
        const int u = a == 0 ? 1 : 0, v = a == 2 ? 1 : 2;
        const int W = r->lv_dims[u], H = r->lv_dims[v], D = ceil_div(r->lv_dims[a], 8) * 8;
        best = std::max(best, (size_t) ceil_div(W, 16) * ceil_div(H, 16) * (size_t) (D / 8));
        elems = std::max(elems, (size_t) W * H);
        tiles = std::max(tiles, ceil_div(W, kSweepTile) * ceil_div(H, sweep_tile_rows()));
        depth = std::max(depth, D);
    }
    if (slice_elems_max) *slice_elems_max = elems;
    if (tiles_max) *tiles_max = tiles;
    if (depth_max) *depth_max = depth;
    return best;
}

int reserve_resources(tbrm_resources* r, int n_lights, unsigned flags)
{
    if (r->resident) { r->reserved = true; return TBRM_OK; } // (slab-resident handles run the chunked chain on their own stores)
    n_lights = std::max(n_lights, 1);
    // flags bit 31 (internal: ensure_reserved): what EVERY handle needs to run its operators without allocating in the steady
    // state — the arena, the event pool, the spare block lists — but not the scratch stores and hand-off records of the largest
    // pass the volume could ever see (32 GiB at 1024^3: a host with several handles, or one that only ever runs slab passes, says
    // so itself by calling tbrm_resources_reserve); those are then allocated by the first pass that needs them, as before round 6
    const bool eager = !(flags & 0x80000000u);
    flags &= 0x7fffffffu;
    if (r->reserved && n_lights <= r->reserved_lights && !(flags & 1u) && (!eager || r->reserved_eagerly)) return TBRM_OK;
    if (int e = ensure_occ_stream(r)) return e;
    size_t slice_elems = 0;
    int tiles = 0, depth = 0;
    const size_t blocks = pass_blocks_max(r, &slice_elems, &tiles, &depth);
    // the page of ones, the scratch stores (ensure_factor_scratch drains and allocates only what is missing)
    if (eager)
        for (int b = 0; b < tbrm_resources::kFScratch; ++b) // (both streams for four of them: an uncached Change computes two streams per pass)
            if (int e = ensure_factor_scratch(r, b, blocks, b < 4 ? 2 : 1)) return e;
    // hand-off records: reach 2 x 2 for both record buffers (two-way Changes), float light volumes: 8-byte granules per stream
    if (eager) {
        const size_t gw = r->lv_fmt != FMT_U8 ? 4 : 1;
        // (reach 4 where that is small — a volume of up to 256^3 —, else 2)
        const int reach = (size_t) depth * (size_t) tiles * (size_t) sweep_record_words(4, 4, sweep_tile_rows()) * gw * 4 <= ((size_t) 64 << 20) ? 4 : 2;
        const size_t words = (size_t) depth * (size_t) tiles * (size_t) sweep_record_words(reach, reach, sweep_tile_rows()) * gw;
        // (chained passes have a region of the first buffer each: kSweepChainMax passes of reach 1, or two of reach 2)
        const size_t first = std::max(words, (size_t) kSweepChainMax * (((size_t) depth * (size_t) tiles * (size_t) sweep_record_words(reach, reach, sweep_tile_rows()) + 63) & ~(size_t) 63));
        if (first < ((size_t) 1 << 32))
            if (int e = ensure_sweep(r, std::max<size_t>(first, 1), words, (size_t) tiles)) return e;
    }
    // ordering events and block lists for n_lights lights' passes
    const size_t want_events = (size_t) 16 * n_lights + 32;
    while (r->event_pool.size() < want_events) {
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, event_flags()));
        r->event_pool.push_back(ev);
    }
    {
        size_t have_ranked = 0, have_plain = 0;
        for (const BlockLists* l : r->spare_lists) (l->slot ? have_ranked : have_plain) += 1;
        const size_t units = (size_t) ceil_div(r->lv_dims[0], 16) * ceil_div(r->lv_dims[1], 16) * (size_t) ceil_div(r->lv_dims[2], 8);
        for (size_t k = have_ranked; k < (size_t) 16 * n_lights + 32; ++k)
            if (!make_spare_lists(r, blocks, true)) return TBRM_ERR_OUT_OF_MEMORY;
        for (size_t k = have_plain; k < (size_t) 8 * n_lights + 16; ++k)
            if (!make_spare_lists(r, std::max(units, blocks), false)) return TBRM_ERR_OUT_OF_MEMORY;
    }
    // the factor cache's arena: per light two passes, twice (a light that moves fills new entries while the old ones are still read),
    // sized like an entry whose live-block count is not known yet (kept_new's caller: every block of a small pass, half of a large one)
    if (tune(TUNE_LIGHT_CACHE_MB) != 0) {
        const size_t entry = (blocks * 2048 * sizeof(float) <= ((size_t) 256 << 20) ? blocks : blocks / 2) * 2048 * sizeof(float);
        size_t want = std::min(kept_budget(r), DeviceArena::rounded(entry) * (size_t) (4 * n_lights));
        size_t free_b = 0, total_b = 0;
        // never the last of the device's memory — whoever else allocates on it (a renderer, torch, other handles, the runtime's own
        // scratch for kernels in flight: a queue that cannot get it aborts the process) must not find it gone: half of what is free
        // beyond 8 GiB for a host that reserves, an eighth for a handle that was never reserved (there may be many of them)
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t spare = free_b > ((size_t) 8 << 30) ? free_b - ((size_t) 8 << 30) : 0;
            want = std::min(want, eager ? spare / 2 : spare / 8);
        } else (void) hipGetLastError();
        if (want > r->cache_arena.bytes && want >= DeviceArena::rounded(entry)) {
            drain_streams(r);
            release_kept(r); // (entries of a smaller arena, or allocated one by one before the handle was reserved)
            (void) hipFree(r->cache_arena_alloc);
            r->cache_arena_alloc = nullptr;
            r->cache_arena.reset(nullptr, 0);
            if (hipMalloc(&r->cache_arena_alloc, want) == hipSuccess) r->cache_arena.reset((char*) r->cache_arena_alloc, want);
            else (void) hipGetLastError(); // (no arena: no cache — the operators still run)
        }
    }
    if (flags & 1u) { // the chunked chain's occlusion stores too (passes the sweep declines: reaches beyond 14 texels, ...)
        const size_t flag_bytes = blocks + 16 * (blocks / std::max<size_t>((size_t) depth / 8, 1)); // (whole spans of 128 slices: up to 16 slice groups more than the pass has)
        for (int b = 0; b < 2; ++b)
            for (int si = 0; si < 2; ++si)
                if (int e = ensure_store(r, &r->occ_tmp[b][si], 128, slice_elems, si == 0 ? flag_bytes : 0)) return e;
    }
    r->reserved = true;
    r->reserved_eagerly = r->reserved_eagerly || eager;
    r->reserved_lights = std::max(r->reserved_lights, n_lights);
    return TBRM_OK;
}

int ensure_reserved(tbrm_resources* r)
{
    if (r->reserved) return TBRM_OK;
    return reserve_resources(r, 4, 0x80000000u); // (a host that never said how many lights it has: the reference scenes' four)
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
