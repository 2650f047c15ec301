// tbrm_block_lists.cpp — the block lists of the sparse occlusion (tbrm_resources.h BlockLists): which 16 x 16 x 8 blocks of an
// axis pass can see anything but empty bricks, the ascending list of those that can, their ranks, their count.
//
// Until round 5 every light operator rebuilt them (k_occ_flags + k_occ_compact per pass, k_unit_flags + k_occ_compact per dual
// launch: six dependent launches in front of every occlusion kernel, ~0.1 ms of every benchmark step on the occlusion stream).
// What k_occ_flags computes depends on the emptiness bits of the data bricks (volume, transfer function, window) and on the
// pass only through INTEGERS: per block column / row / slice group the range of data texels its samples' base taps fall into
// (texel_split of the block's first and last position + UVWOffset, tbrm_light_kernels.hip k_occ_flags). A light that turns by a
// few degrees keeps those integers — UVWOffset stays within the same texel — so the lists are kept under that signature and a
// pass that finds its signature launches nothing. The signature is evaluated here with the kernel's own fp32 sequence
// (IEEE single operations, -ffp-contract=off on both sides).
#include "tbrm_resources.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tbrm_host {

using namespace tbrm;

static constexpr int kBlock = 16, kDepth = 8; // kOccTile, kOccDepth (tbrm_light_kernels.hip)
static constexpr size_t kMaxLists = 384;      // per handle (a reserved handle of eight lights starts with 240 spare ones); beyond: the least recently used ones nobody refers to go

// texel_split (tbrm_device_math.h): the index of the lower tap
static int base_tap(float u, float n)
{
    float x = u * n - 0.5f;
    x = std::fmin(std::fmax(x, -0x1p30f), 0x1p30f);
    return (int) std::floor(x);
}

// Everything k_occ_flags<MODE, AXIS> reads of a one-chunk pass besides ChunkParams::empty_bits, as integers: the geometry and,
// per block column, block row and slice group, the lowest base tap and the highest upper tap over the streams the mode computes.
static void block_lists_signature(const ChunkParams& p, int occ_mode, std::vector<int32_t>& sig)
{
    const int ns = (occ_mode == PASS_ADD || occ_mode == PASS_CHANGE_ONE) ? 1 : 2;
    const int dim_u = p.axis == 0 ? 1 : 0, dim_v = p.axis == 2 ? 1 : 2, dim_s = p.axis;
    const int dn[3] = {p.data.nx, p.data.ny, p.data.nz};
    sig.clear();
    for (int v : {ns, p.axis, p.W, p.H, p.dir, p.pass_start, p.pass_slices, p.chunk_slices, p.occ_groups, p.occ_blocks_x, p.occ_blocks_y, p.roi_by0,
                  p.roi_by1, dn[0], dn[1], dn[2], p.data.bnx, p.data.bnxy, p.lv_dims[0], p.lv_dims[1], p.lv_dims[2]})
        sig.push_back(v);
    auto range = [&](int dim, int first, int last) {
        int lo = INT32_MAX, hi = INT32_MIN;
        for (int si = 0; si < ns; ++si) {
            const ChunkStream& s = si == 0 ? p.a : p.r;
            for (int pos : {first, last}) {
                const float cc = (((float) (uint32_t) pos + 0.5f) / (float) (uint32_t) p.lv_dims[dim]) + s.uvw_off[dim];
                const int i0 = base_tap(cc, (float) dn[dim]);
                lo = std::min(lo, i0);
                hi = std::max(hi, i0 + 1);
            }
        }
        sig.push_back(lo);
        sig.push_back(hi);
    };
    for (int bx = 0; bx < p.occ_blocks_x; ++bx) range(dim_u, bx * kBlock, std::min(bx * kBlock + kBlock, p.W) - 1);
    for (int by = 0; by < p.occ_blocks_y; ++by) range(dim_v, by * kBlock, std::min(by * kBlock + kBlock, p.H) - 1);
    const int n = std::min(p.chunk_slices, p.pass_slices);
    for (int zg = 0; zg < p.occ_groups; ++zg) {
        const int k0 = zg * kDepth;
        if (k0 >= n) { sig.push_back(0); sig.push_back(-1); continue; } // (a group past the pass's last slice: never computed)
        const int nk = std::min(kDepth, n - k0);
        const int j0 = p.pass_start + k0 * p.dir, j1 = j0 + (nk - 1) * p.dir;
        range(dim_s, j0, j1);
    }
}

static void free_lists(BlockLists* l)
{
    if (!l) return;
    (void) hipFree(l->flags);
    (void) hipFree(l->list);
    (void) hipFree(l->slot);
    (void) hipFree(l->count);
    if (l->count_host) (void) hipHostFree(l->count_host);
    if (l->ev_done) (void) hipEventDestroy(l->ev_done);
    delete l;
}

void release_block_lists(tbrm_resources* r)
{
    for (BlockLists* l : r->block_lists) free_lists(l);
    r->block_lists.clear();
    for (BlockLists* l : r->spare_lists) free_lists(l);
    r->spare_lists.clear();
}

static void drain(tbrm_resources* r)
{
    ++r->sync_calls;
    (void) hipStreamSynchronize(r->stream);
    if (r->occ_stream) (void) hipStreamSynchronize(r->occ_stream);
}

static BlockLists* allocate_lists(size_t blocks, bool with_ranks)
{
    BlockLists* l = new BlockLists{};
    bool ok = hipMalloc((void**) &l->flags, blocks) == hipSuccess && hipMalloc((void**) &l->list, blocks * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void**) &l->count, 16 * sizeof(int)) == hipSuccess;
    if (with_ranks)
        ok = ok && hipMalloc((void**) &l->slot, blocks * sizeof(int32_t)) == hipSuccess &&
             hipHostMalloc((void**) &l->count_host, sizeof(int), hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&l->ev_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void) hipGetLastError();
        free_lists(l);
        fail(TBRM_ERR_OUT_OF_MEMORY, "no memory for the block lists of a pass (%zu blocks)", blocks);
        return nullptr;
    }
    l->cap = blocks;
    return l;
}

// tbrm_resources_reserve: lists made ahead of their first use
BlockLists* make_spare_lists(tbrm_resources* r, size_t blocks, bool with_ranks)
{
    BlockLists* l = allocate_lists(blocks, with_ranks);
    if (l) r->spare_lists.push_back(l);
    return l;
}

// Beyond kMaxLists (many distinct light directions under one volume / transfer function / window): the least recently used
// lists go, as far as no factor cache entry stores its blocks under their ranks. Buffers that launches in flight may still
// read are never freed under them: the streams are drained first.
static void prune(tbrm_resources* r)
{
    if (r->block_lists.size() < kMaxLists) return;
    drain(r);
    count_alloc(r, 4, "pruned block lists");
    std::vector<BlockLists*> keep;
    std::sort(r->block_lists.begin(), r->block_lists.end(), [](const BlockLists* a, const BlockLists* b) { return a->last_use > b->last_use; });
    for (BlockLists* l : r->block_lists) {
        if (l->users > 0 || l->last_use > r->block_lists_op_floor || keep.size() < kMaxLists / 2) keep.push_back(l); // (never the ones the operator being planned holds)
        else free_lists(l);
    }
    // (units lists whose passes' lists went: ids are never reused, so they are merely never found again)
    r->block_lists.swap(keep);
}

// New lists for `blocks` blocks: the buffers of lists that were computed from skipping metadata which is gone (a new volume,
// transfer function or window: nothing will ask for them again) if some are large enough — no allocation while only the
// window moves (APerformanceTest1's sweep) —, else fresh ones.
// A list's buffers are rewritten only when nothing in flight can still read them: every launch that is handed a list's flags,
// work list or ranks notes its operator in BlockLists::last_read_op (the planner, for every list a plan points at), and a stale
// list — nobody's ranks (users == 0), computed from metadata that is gone — is taken again once that operator's "sweeps done"
// event has FIRED (op_finished: asked, not waited for). Round 5 drained the streams once per metadata change instead, which left a
// window: a list pinned by a cache entry across the change could be read by a sweep enqueued AFTER that drain and recycled when the
// entry let go of it.
static BlockLists* new_lists(tbrm_resources* r, size_t blocks, bool with_ranks)
{
    BlockLists* l = nullptr;
    // a spare one first (tbrm_resources_reserve): never used, nothing reads it
    for (size_t i = 0; i < r->spare_lists.size(); ++i) {
        BlockLists* c = r->spare_lists[i];
        if (c->cap >= blocks && (c->slot != nullptr) == with_ranks) {
            l = c;
            r->spare_lists.erase(r->spare_lists.begin() + (long) i);
            r->block_lists.push_back(l);
            break;
        }
    }
    if (!l) {
        // then one computed from metadata that is gone (nothing will ask for it again); then — a reserved handle keeps to the lists it
        // was given — the least recently used one of the current metadata (a light that turns leaves a trail of signatures behind)
        auto free_to_take = [&](const BlockLists* c) {
            return c->users == 0 && c->cap >= blocks && (c->slot != nullptr) == with_ranks && c->last_use <= r->block_lists_op_floor;
        };
        for (BlockLists* c : r->block_lists)
            if (c->empty_gen != r->empty_gen && free_to_take(c) && op_finished(r, c->last_read_op) && (!l || c->cap < l->cap)) l = c;
        if (!l && r->reserved)
            for (BlockLists* c : r->block_lists)
                if (free_to_take(c) && (!l || c->last_use < l->last_use) && op_finished(r, c->last_read_op)) l = c;
        if (l) {
            l->sig.clear();
            l->a_id = l->b_id = 0;
            l->enqueued = false;
        }
    }
    if (!l) {
        prune(r);
        count_alloc(r, with_ranks ? 5 : 3, "new block lists (no spare or quiet list to take)");
        l = allocate_lists(blocks, with_ranks);
        if (!l) return nullptr;
        r->block_lists.push_back(l);
    }
    l->blocks = blocks;
    if (l->count_host) *l->count_host = 0;
    l->empty_gen = r->empty_gen;
    l->id = ++r->block_lists_serial;
    l->last_use = l->id;
    l->last_read_op = r->op_serial; // (the operator being planned will launch over them)
    return l;
}

BlockLists* block_lists_for_pass(tbrm_resources* r, const ChunkParams& p, int occ_mode)
{
    std::vector<int32_t> sig;
    block_lists_signature(p, occ_mode, sig);
    for (BlockLists* l : r->block_lists)
        if (l->a_id == 0 && l->empty_gen == r->empty_gen && l->sig == sig) {
            l->last_use = ++r->block_lists_serial;
            l->last_read_op = std::max(l->last_read_op, r->op_serial);
            return l;
        }
    BlockLists* l = new_lists(r, (size_t) p.occ_groups * p.occ_blocks_y * p.occ_blocks_x, true);
    if (l) l->sig.swap(sig);
    return l;
}

BlockLists* block_lists_for_dual(tbrm_resources* r, const BlockLists* a, const BlockLists* b, size_t units)
{
    for (BlockLists* l : r->block_lists)
        if (l->a_id == a->id && l->b_id == b->id && l->blocks == units) {
            l->last_use = ++r->block_lists_serial;
            l->last_read_op = std::max(l->last_read_op, r->op_serial);
            return l;
        }
    BlockLists* l = new_lists(r, units, false);
    if (l) { l->a_id = a->id; l->b_id = b->id; }
    return l;
}

bool block_lists_count(BlockLists* l, bool wait, size_t* count)
{
    if (!l || !l->enqueued || !l->ev_done) return false;
    if (wait) (void) hipEventSynchronize(l->ev_done);
    else if (hipEventQuery(l->ev_done) != hipSuccess) { (void) hipGetLastError(); return false; }
    *count = (size_t) std::max(*l->count_host, 0);
    return true;
}

} // namespace tbrm_host
