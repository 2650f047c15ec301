// tbrm_light_operators.cpp — the whole-volume light operators (AddDirLightToSingleVolume / ChangeDirLightInSingleVolume,
// LightingShaders.cpp:35-326) as sequences of planned passes; tbrm_api.cpp's entry points call enqueue_add / enqueue_add_batch /
// enqueue_change for these and plan_pass / enqueue_plan_chunk step by step for the slab-partitioned ones.
#include "tbrm_light_passes.h"

namespace tbrm_host {

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
static int run_passes(tbrm_resources* r, const PropParams& base, std::vector<PassSpec> specs, std::vector<int> partner = {})
{
    if (int e = sweep_failed(r)) return e; // (an earlier sweep left the light volume undefined: nothing to build on)
    if (int e = ensure_reserved(r)) return e; // (a handle nobody reserved: once, with the defaults — tbrm_resources_reserve)
    ++r->op_serial;
    r->block_lists_op_floor = r->block_lists_serial;
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
    // A pair is ONE entry of the schedule the caller was given (enqueue_add_batch: light a's pass, then light b's, voxel by voxel):
    // whether the two passes share a sweep or — should one of them not be on the sweep after all, or the pair not fit one tile
    // order — run one after the other, the partner takes its turn HERE, right behind the first, never at its own later index.
    std::vector<char> is_second(specs.size(), 0), pair_sweeps(specs.size(), 0);
    if (partner.size() != specs.size()) partner.assign(specs.size(), -1); // (also after an ADD2 spec was split: chain batches carry none)
    for (size_t i = 0; i < specs.size(); ++i) {
        const int j = partner[i];
        if (j < 0 || (size_t) j >= specs.size() || (size_t) j == i || is_second[i]) { partner[i] = -1; continue; }
        is_second[(size_t) j] = 1;
        // sharing a sweep needs both passes on the sweep
        pair_sweeps[i] = chunked[i] && chunked[(size_t) j] && plans[i].sweep && plans[(size_t) j].sweep;
    }
    bool any_pair = false;
    for (size_t i = 0; i < specs.size(); ++i) any_pair = any_pair || (partner[i] >= 0 && pair_sweeps[i]);
    if (any_pair) // (a group of two lights: four passes, four scratch buffers — every occlusion can go first)
        for (size_t k = 0; k < specs.size(); ++k) {
            if (!chunked[k]) continue;
            const int e = (k + 1 < specs.size() && chunked[k + 1] && dual_fit(plans[k], plans[k + 1])) ? enqueue_dual_occlusion(r, plans[k], plans[k + 1])
                                                                                                       : enqueue_sweep_occlusion(r, plans[k]);
            if (e) { quiesce_occ_stream(r); return e; }
        }
    auto run_single = [&](size_t i) -> int {
        const PassSpec& q = specs[i];
        if (!chunked[i]) {
            PropParams p = base;
            p.b_added = q.b_added;
            if (int e = enqueue_pass_sliced(r, p, q.a, q.two ? &q.r : nullptr)) return e;
            ++r->passes[2];
            return TBRM_OK;
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
        return TBRM_OK;
    };
    // consecutive sweep passes in ONE launch (k_light_sweep_chain): the next pass's tiles fill while this pass's drain
    const int chain_max = std::min(tune(TUNE_SWEEP_CHAIN), kSweepChainMax);
    auto chainable = [&](size_t k) {
        if (k >= specs.size() || !chunked[k] || is_second[k] || partner[k] >= 0) return false;
        const PassPlan& pl = plans[k];
        return pl.sweep && (pl.mode == PASS_ADD || pl.mode == PASS_CHANGE) && !pl.sq.r_from_records && !pl.sq.lv_f32 && !(pl.sq.debug & 1) &&
               sweep_halo_chunks(pl.sq.hx, pl.sq.hy, pl.sq.tile_rows) <= 3;
    };
    for (size_t i = 0; i < specs.size(); ++i) {
        if (is_second[i]) continue; // (took its turn with its partner)
        if (chain_max >= 2 && chainable(i)) {
            size_t len = 1;
            while ((int) len < chain_max && chainable(i + len) && plans[i + len].mode == plans[i].mode) {
                // A launch waits for the occlusion of ALL its passes: a pass whose factors are still to be computed joins only the
                // pass it shares its occlusion launch with (the two passes of a light, dual_fit) — the occlusion of the light after
                // runs beside this launch instead of in front of it (measured, cold reset of four lights: 3.18 ms light by light,
                // 3.25 with two lights' occlusion in front of one launch of four passes)
                const PassPlan& nx = plans[i + len];
                if (nx.occ_mode >= 0 && !nx.occ_enqueued && !dual_fit(plans[i + len - 1], nx)) break;
                bool fresh_buffer = true; // (every pass of a launch reads its own scratch buffer)
                for (size_t k = 0; k < len; ++k) fresh_buffer = fresh_buffer && plans[i + k].f_buf != plans[i + len].f_buf;
                if (!fresh_buffer) break;
                ++len;
            }
            if (len >= 2) {
                if (int e = enqueue_sweep_chain(r, &plans[i], (int) len)) { quiesce_occ_stream(r); return e; }
                r->passes[0] += len;
                probe.lap("chain");
                i += len - 1;
                continue;
            }
        }
        if (partner[i] >= 0) {
            const size_t j = (size_t) partner[i];
            SweepFit fit;
            if (pair_sweeps[i] && sweep_fit(r, specs[i].a, &specs[j].a, PASS_CHANGE, fit) && !fit.two_way) {
                if (int e = enqueue_sweep_pair(r, plans[i], plans[j], fit)) { quiesce_occ_stream(r); return e; }
                r->passes[0] += 2;
                probe.lap("pair");
                continue;
            }
            if (int e = run_single(i)) return e; // (no shared sweep after all: the same two passes, in the same order)
            if (int e = run_single(j)) return e;
            continue;
        }
        if (int e = run_single(i)) return e;
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
        // Nothing pairs (every same-face couple pulls opposite ways along a plane axis — config 3's four lights —, or none shares a
        // face): the lights four at a time, in the caller's order — exactly light after light, voxel by voxel —, their sweeps chained
        // (k_light_sweep_chain: up to four passes per launch where the factors are at hand). Eight scratch buffers hold a group's
        // passes. (One occlusion launch for all four lights was built and measured: 0.46 ms per light against 0.28 for the dual
        // launch — 4 x 4 x 3 staged bricks per unit instead of 3 x 3 x 2 cost two of five workgroups per CU;
        // tools/diagnostics/multi_light_occlusion.patch, profiles/EXPERIMENTS.md.)
        bool any_pair_fits = false;
        for (size_t x = 0; x < all.size() && !any_pair_fits; ++x)
            for (size_t y = x + 1; y < all.size() && !any_pair_fits; ++y) any_pair_fits = all[x].light != all[y].light && pair_fits(all[x].p, all[y].p);
        if (!any_pair_fits) {
            int entries = 0;
            constexpr int kGroup = tbrm_resources::kFScratch / 2; // lights per run_passes: two passes each, a scratch buffer per pass
            for (int l0 = 0; l0 < n_lights; l0 += kGroup) {
                std::vector<PassSpec> specs;
                for (int li = l0; li < std::min(l0 + kGroup, n_lights); ++li)
                    for (size_t k : of_light[(size_t) li]) {
                        PassSpec q;
                        q.a = all[k].p;
                        q.b_added = b;
                        specs.push_back(q);
                        if (schedule) {
                            schedule[4 * entries + 0] = all[k].light; schedule[4 * entries + 1] = all[k].pass;
                            schedule[4 * entries + 2] = -1; schedule[4 * entries + 3] = -1;
                        }
                        ++entries;
                    }
                if (n_entries) *n_entries = entries;
                if (specs.empty()) continue;
                if (int e = run_passes(r, base, specs)) return e;
            }
            if (n_entries) *n_entries = entries;
            return TBRM_OK;
        }
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
