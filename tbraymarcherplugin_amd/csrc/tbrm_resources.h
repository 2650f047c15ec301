// tbrm_resources.h — what the host side of the library shares between its translation units: the handle behind the
// C-ABI's opaque tbrm_resources, error reporting, and the light-pass layer (planning and enqueueing the axis passes of the
// Add / Change operators, tbrm_light_plan.cpp / tbrm_light_enqueue.cpp / tbrm_light_operators.cpp) that tbrm_api.cpp's entry points drive. Internal; not installed.
#pragma once

#include "../../include/tbrm.h"
#include "tbrm_host_math.h"
#include "tbrm_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace tbrm_host {

int fail(int code, const char* fmt, ...); // sets tbrm_last_error() of the calling thread, returns code
inline size_t format_bytes(int fmt) { return fmt == TBRM_FMT_G8 ? 1 : (fmt == TBRM_FMT_G16 ? 2 : 4); }

} // namespace tbrm_host

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        const hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                                 \
            return tbrm_host::fail(e_ == hipErrorOutOfMemory ? TBRM_ERR_OUT_OF_MEMORY : TBRM_ERR_NO_DEVICE, "%s failed: %s", \
                #expr, hipGetErrorString(e_));                                                                \
    } while (0)

// The occlusion factors 1 - CurrentSample (AddDirLightShader.usf:85-117) of ONE light stream for a span of S slices of an
// axis pass, as the chain kernel consumes them: [page of ones | guard][S x H x W floats][guard], plus the pass's
// empty-block flags and work lists. Overwritten span by span; a handle has two per stream, so that the next span's can be
// written while the chain reads the current one's (tbrm_resources::occ_tmp).
struct OccStore {
    float* base = nullptr;          // the allocation
    size_t capacity = 0;            // floats of planes it holds (slices x H x W of the pass it was sized for)
    uint8_t* flags = nullptr;       // empty-block flags of the pass: [span][slice group][block y][block x]
    uint32_t* list = nullptr;       // work lists of the pass (one uint32 per flag) followed by 4096 per-span counts
    size_t flag_bytes = 0;
};

// The factor cache. The expensive half of an axis pass is its occlusion: the factors 1 - CurrentSample
// (AddDirLightShader.usf:85-117) of every voxel, which depend on the data volume, the transfer function, the window, the clip
// plane and the light's sampling offsets — not on the light volume and not on the light's intensity. A pass that propagates a
// light which stays in the scene keeps its factors, in the block-compact form the sweep kernel consumes
// (tbrm_internal.h ChunkStream::fs_*): only the 16 x 16 x 8 blocks that can be opaque at all are stored, 8 KiB each, behind
// the table of their ranks. Later operators on that light skip its occlusion: the removed side of a ChangeDirLight, a removal,
// a re-add after ClearResourceLightVolumes propagate from the kept factors (the propagation itself is cheap: one sweep).
// This is the Sunden / Ropinski selective update taken one step further than the reference takes it — the reference samples
// the volume again for the removed light; here a light's samples are taken once.
//
// An entry is allocated before its pass runs, for an estimated number of live blocks (the count is computed on the device);
// blocks beyond the capacity go to the handle's scratch store, and an entry that overflowed is dropped when the host learns
// the count (a tiny read-back, waited for only when the entry is looked up).
struct FactorKey {
    uint64_t data_gen, tf_gen;
    float win[4];
    float cc[3], cd[3], data_border;
    int32_t clip_mode, axis, dir, start, D, W, H, guard;
    float uvw_off[3], step100;
};
// One device allocation handed out in pieces (host-side bookkeeping only: first fit over an offset-sorted free list, neighbours
// merged on release). What the factor cache's entries live in once a handle is reserved (tbrm_resources_reserve): an operator that
// needs a new entry takes a piece and gives back the pieces of entries nothing in flight reads — no hipMalloc, no hipFree, no
// hipMemGetInfo on the operator path (the reference creates its buffers once, in InitializeRaymarchResources,
// RaymarchVolume.cpp:821-920, never inside AddDirLightToSingleVolume).
struct DeviceArena {
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<std::pair<size_t, size_t>> free_; // (offset, size), ascending, never adjacent
    static constexpr size_t kAlign = 4096;
    void reset(char* b, size_t n)
    {
        base = b;
        bytes = n;
        free_.clear();
        if (b && n) free_.emplace_back(0, n);
    }
    bool owns(const void* p) const { return base && (const char*) p >= base && (const char*) p < base + bytes; }
    static size_t rounded(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }
    void* take(size_t n)
    {
        n = rounded(std::max<size_t>(n, 1));
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= n) {
                char* const p = base + free_[i].first;
                free_[i].first += n;
                free_[i].second -= n;
                if (free_[i].second == 0) free_.erase(free_.begin() + (long) i);
                return p;
            }
        return nullptr;
    }
    void give(void* p, size_t n)
    {
        if (!owns(p)) return;
        n = rounded(std::max<size_t>(n, 1));
        const size_t off = (size_t) ((char*) p - base);
        size_t i = 0;
        while (i < free_.size() && free_[i].first < off) ++i;
        free_.insert(free_.begin() + (long) i, std::make_pair(off, n));
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) { free_[i].second += free_[i + 1].second; free_.erase(free_.begin() + (long) i + 1); }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) { free_[i - 1].second += free_[i].second; free_.erase(free_.begin() + (long) i); }
    }
    size_t free_bytes() const
    {
        size_t n = 0;
        for (const auto& f : free_) n += f.second;
        return n;
    }
};

// The block lists of a pass (tbrm_block_lists.cpp): which of its 16 x 16 x 8 occlusion blocks can see anything but empty bricks
// (k_occ_flags), the ascending list of those that can and every block's rank in it (k_occ_compact), the count — or the same for
// the work units of a dual occlusion launch (k_unit_flags). They depend on the skipping metadata (volume, transfer function,
// window: tbrm_resources::empty_gen) and on the light only through the integer range of data texels a block's samples touch,
// which a light that turns by a few degrees does not change: computed once per `sig`, kept with the handle, never rewritten
// (a later pass with the same signature launches nothing), freed when the metadata they were computed from is gone.
struct BlockLists {
    std::vector<int32_t> sig;       // block_lists_signature: everything k_occ_flags reads but the emptiness bits
    uint64_t empty_gen = 0;         // tbrm_resources::empty_gen they were computed under
    uint8_t* flags = nullptr;       // [slice group][block y][block x], 1 = nothing to compute
    uint32_t* list = nullptr;       // the live blocks, ascending
    int32_t* slot = nullptr;        // rank of every block in `list`, -1 for the flagged ones (null: units of a dual launch)
    int* count = nullptr;           // device: live blocks
    int* count_host = nullptr;      // pinned: the same, written by k_occ_compact
    hipEvent_t ev_done = nullptr;   // ... which this event follows (occlusion stream)
    size_t blocks = 0, cap = 0;     // blocks of the pass; what the buffers hold
    bool enqueued = false;          // its kernels are on the occlusion stream
    uint64_t id = 0;                // never reused within a handle
    uint64_t a_id = 0, b_id = 0;    // units of a dual launch: the two passes' lists (0: a pass's own lists)
    int users = 0;                  // factor cache entries whose ranks these are (+ the slab operation that holds a plan over them)
    uint64_t last_use = 0;
    uint64_t last_read_op = 0;      // tbrm_resources::op_serial of the last operator whose launches (occlusion, sweeps) were handed these
                                    // buffers: they are rewritten only once that operator's "done" event has fired (new_lists)
    size_t bytes() const { return blocks * (sizeof(uint8_t) + sizeof(uint32_t) + (slot ? sizeof(int32_t) : 0)); }
};

struct FactorEntry {
    float* base = nullptr;          // cap_blocks x 2048 floats
    size_t cap_blocks = 0;
    BlockLists* lists = nullptr;    // the ranks its blocks are stored under (borrowed: BlockLists::users)
    hipEvent_t ev_filled = nullptr; // the occlusion that fills the entry is done (occlusion stream)
    hipEvent_t ev_idle = nullptr;   // the last sweep that reads the entry is done (the handle's stream)
    bool read_yet = false;          // (ev_idle has been recorded)
    uint64_t last_read_op = 0;      // tbrm_resources::op_serial of the operator whose sweep read it last
    bool idle_recorded = false;     // ... and whether that sweep recorded ev_idle (else: the operator's op_done event)
    FactorKey key{};
    bool enqueued = false;          // the occlusion that fills it is on the occlusion stream
    bool resolved = false;          // the count has been read: valid / dropped
    bool valid = false;
    bool pinned = false;            // in use by the operator being planned
    bool spent = false;             // its light has left the scene: first in line for reuse
    uint64_t last_use = 0;
    size_t bytes() const { return cap_blocks * 2048 * sizeof(float); }
};

// Scratch of the block-compact hand-over: per buffer (axis passes take four in rotation) the factor stores of the two streams
// (the pass's block lists — flags, work list, ranks, count — are BlockLists)
struct FactorScratch {
    float* store[2] = {nullptr, nullptr};
    size_t store_blocks = 0;
    hipEvent_t ev_ready = nullptr;  // the occlusion into this buffer is done (occlusion stream)
    hipEvent_t ev_idle = nullptr;   // the sweep that read this buffer is done (the handle's stream)
    bool used = false;              // (ev_idle has been recorded)
    uint64_t last_read_op = 0;      // tbrm_resources::op_serial of the operator whose sweep read it
    bool idle_recorded = false;     // ... and whether that sweep recorded ev_idle (else: the operator's op_done event)
};

struct tbrm_resources {
    tbrm_resources_desc desc{};
    int32_t lv_dims[3]{};
    int lv_fmt = tbrm::FMT_U8;
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
    // chunk kernels: the occlusion stores, [buffer][stream] (allocated on first use). Spans alternate between the two
    // buffers so that the occlusion of the next span — which depends on the data volume alone — can run on occ_stream beside
    // the chain of the current one (enqueue_plan_chunk); the flags / work lists of a pass live in occ_tmp[pass serial & 1][0].
    OccStore occ_tmp[2][2];
    struct OccSlot { uint64_t plan_serial = 0; int span = -1; } occ_slot[2]; // what each buffer holds (or will, once occ_ev_ready fires)
    bool occ_slot_async[2] = {false, false};                                 // ... computed on occ_stream
    int occ_last = 1;              // buffer of the most recent span
    uint64_t plan_serial = 0;      // PassPlan::serial of the last plan made
    hipStream_t occ_stream = nullptr; // low priority; created with the first overlapped launch
    hipEvent_t occ_ev_fork[2]{}, occ_ev_ready[2]{};
    bool occ_inputs_changed = true; // the volume / transfer function / skipping metadata were (re)written on `stream` since the
                                    // occlusion stream last ordered itself behind it (ensure_skipping sets, order_behind_inputs clears)
    // the pipelined sweep (k_light_sweep, tbrm_light_sweep.hip): hand-off records of the two streams ([slice][tile][word],
    // never cleared: words carry the tag of the launch that wrote them), the tile tickets and the error word
    uint32_t* sweep_rec[2] = {nullptr, nullptr};
    size_t sweep_rec_words = 0;    // capacity of [0]
    size_t sweep_rec1_words = 0;   // capacity of [1] (two-way Changes only)
    uint32_t* sweep_prog = nullptr; // chained sweeps (SweepLink): kSweepChainMax tables of per-tile progress words, sweep_prog_stride apart
    size_t sweep_prog_stride = 0;
    uint64_t chain_launches = 0;   // sweep launches that ran several passes (k_light_sweep_chain)
    int* sweep_ticket = nullptr;   // device: [0] next tile, [1] tiles finished (re-armed by the last tile of every launch)
    int* sweep_error = nullptr;    // pinned host memory the kernels write to: a tile gave up waiting (1) / taps outside its halo (2)
    uint32_t sweep_epoch = 0;      // tag of the last sweep launch
    bool sweep_epoch_preset_done = false; // (tunable sweep_epoch_preset has been applied to this handle)
    int sweep_failed_bits = 0;     // latched error word (sweep_failed): the light volume is undefined until it is cleared
    unsigned long long* sweep_stamps = nullptr; // diagnostics (sweep_debug & 2): the last launch's per-tile time stamps
    int sweep_stamp_tiles = 0, sweep_stamp_tx = 0, sweep_stamp_sx = 0, sweep_stamp_sy = 0;
    int sweep_stamp_chain[4] = {0, 0, 0, 0}; // a chained launch's stamps: tiles per pass (0: not a chain)
    std::vector<FactorEntry*> kept; // the factor cache
    uint64_t kept_clock = 0;       // its LRU clock
    uint64_t op_serial = 0;        // whole-volume light operators run so far (run_passes)
    // "every sweep of operator k is done" (the handle's stream), k mod 8: what the occlusion stream waits for before it
    // overwrites a scratch buffer or a cache entry that operator k's sweeps read — ONE wait per occlusion launch whatever the
    // number of buffers involved
    static constexpr int kOpEvents = 8;
    hipEvent_t op_done[kOpEvents]{};
    uint64_t op_done_serial[kOpEvents]{}; // which operator each was last recorded for (0: never)
    bool op_many_passes = false;   // the operator being enqueued has more than two sweep passes: its buffers may come round again
    uint64_t kept_hits = 0, kept_computed = 0; // stream-passes whose occlusion came from the cache / was computed (tbrm_light_cache_stats)
    size_t f_est_blocks = 0;       // live blocks per pass seen under f_est_key (what a new entry is sized for)
    uint64_t f_est_key[2] = {0, 0};
    float f_est_win[4] = {0, 0, 0, 0};
    // Taken in rotation: a light's two passes are filled by ONE occlusion launch (DualOcc) while the sweeps of the operator
    // before may still be reading the two before them
    static constexpr int kFScratch = 8; // (round 6: eight — tbrm_add_dir_lights plans the passes of four lights together, each pass its own buffer)
    FactorScratch f_scratch[kFScratch];
    int f_buf = 0;                 // buffer of the most recent sweep pass
    std::vector<BlockLists*> block_lists; // passes' and dual launches' block lists computed so far (tbrm_block_lists.cpp)
    uint64_t block_lists_serial = 0;
    uint64_t block_lists_op_floor = 0;  // block_lists_serial when the operator being planned began: its plans point at younger lists
    std::vector<BlockLists*> spare_lists;  // made by tbrm_resources_reserve, never used yet: [with ranks], then [without]
    uint64_t lists_launches = 0;   // passes / dual launches whose lists had to be computed (tbrm_path_counters)
    // tbrm_resources_reserve: everything the light operators would otherwise allocate as they go
    bool reserved = false;
    bool reserved_eagerly = false; // (by tbrm_resources_reserve: scratch stores and hand-off records too)
    int reserved_lights = 0;
    DeviceArena cache_arena;       // the factor cache's entries
    void* cache_arena_alloc = nullptr;
    std::vector<hipEvent_t> event_pool; // ordering events (event_flags()) not in use
    size_t device_total_bytes = 0; // hipMemGetInfo at creation (the factor cache's automatic budget: an eighth of it)
    uint64_t alloc_calls = 0;      // hipMalloc / hipHostMalloc / hipFree / hipHostFree / hipMemGetInfo made inside operators (tbrm_path_counters [12])
    uint64_t sync_calls = 0;       // host-side waits for a stream made inside operators (tbrm_path_counters [13])
    uint64_t dual_launches = 0;    // occlusion launches that served two passes (tbrm_launch_counters)
    float* d_ones = nullptr;       // 1024 floats of 1.0
    uint64_t data_gen = 1, tf_gen = 1; // bumped by volume uploads / tbrm_set_tf_lut: what cached occlusion was computed from

    // empty-space-skipping metadata
    int bn[3]{};
    float2* d_minmax = nullptr;
    uint32_t* d_empty = nullptr;
    uint8_t* d_dist[2]{};          // empty-space leaping: per-brick distance field (ping-pong of the separable passes; [0] is final)
    int* d_alpha_prefix = nullptr;
    bool shell_transparent = false; // valid with empty_valid (ensure_skipping): the Add and the Change shader propagate the same L
    bool minmax_valid = false, empty_valid = false;
    uint64_t empty_gen = 0;        // bumped whenever d_empty is rewritten (ensure_skipping): what BlockLists were computed from

    // Octree render mode: 4-level UNORM16 max pyramid (allocated by the first tbrm_generate_octree)
    uint16_t* d_octree[4]{};
    int oct_dims[4][3]{};
    bool octree_valid = false;

    uint2* d_ray_tab = nullptr;    // k_raymarch_lit's texel -> offset tables (RayParams::tab), built once per handle
    unsigned long long* d_counter = nullptr;
    float* d_out = nullptr; // staging for the host-pointer raymarch variant
    size_t out_bytes = 0;

    struct SlabOp* slab_op = nullptr; // the slab-partitioned light operation in flight (tbrm_slab_*)

    hipEvent_t ev[2][2]{};
    bool ev_valid[2]{};
    uint64_t launches[3]{}; // chunk, slice, raymarch
    uint64_t sweep_launches = 0; // (of the chunk launches: the pipelined sweep kernel's)
    uint64_t passes[3]{};        // axis passes run as a sweep / as the chunked chain / one slice per launch (tbrm_path_counters)
    uint64_t occ_launches = 0;   // occlusion launches that served one pass (dual_launches: both passes of a light)
    uint64_t pair_sweeps = 0;    // sweep launches that propagated two lights' passes at once (PASS_ADD2)
};


namespace tbrm_host {

using namespace tbrm;

inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int ceil_div(int a, int b) { return -floor_div(-a, b); }
inline int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- handle helpers (tbrm_api.cpp) ------------------------------------------------------------------------------------
int bind(const tbrm_resources* r);
bool initialized(const tbrm_resources* r);
VolumeDev data_view(const tbrm_resources* r);
WindowDev window_dev(const tbrm_resources* r);
PropParams base_prop_params(const tbrm_resources* r, const tbrm_world_params& world);
void fill_stream(PropStream& s, const tbrm_light_pass& p);
int begin_timed(tbrm_resources* r, int kind);
int end_timed(tbrm_resources* r, int kind);
int ensure_skipping(tbrm_resources* r);
int raymarch_clip_mode(const float cc[3], const float cd[3]);
RelayoutParams relayout_params(const void* src, void* dst, const int dims[3], const int bn[3], size_t elem, bool to_bricks);

// ---- the light-pass layer (tbrm_light_plan.cpp, tbrm_factor_cache.cpp, tbrm_light_enqueue.cpp, tbrm_light_operators.cpp) ------------------------------------------------------------------------
// Range of (tap index - pixel index) of the previous-slice bilinear fetch over one buffer axis, evaluated with the
// kernel's own fp32 sequence (texel_split of ((c+0.5)/size + offset)); hi includes the +1 tap.
struct TapRange { int lo = 0, hi = 0; bool ok = false; };

struct ChunkFit { int M = 0; TapRange tx, ty; };

// One axis pass (Add: stream a only; Change: a = added, r = removed) as a plan: everything that is constant over the
// pass, worked out once, and the chunks then enqueued one by one (plan_pass / enqueue_plan_chunk). A single-GPU pass
// enqueues all of them back to back; a slab-partitioned pass (tbrm_slab_*) stops after each chunk so that the host can
// exchange the propagated planes between ranks.
struct PassPlan {
    ChunkParams p{};
    int mode = PASS_ADD;        // PASS_ADD / PASS_CHANGE / PASS_ADD2
    uint64_t serial = 0;        // identifies the plan (occlusion buffers are labelled with it)
    int n_streams() const { return mode == PASS_ADD ? 1 : 2; } // streams propagated
    bool two_streams() const { return n_streams() == 2; }
    int M = 0, S = 0;           // slices per chain chunk / per occlusion span
    int D = 0;                  // slices this handle runs (the whole pass, or its slab's part of a pass along z)
    int start = 0, dir = 1;     // first of them
    int n_chunks = 0, n_spans = 0;
    bool pass_begins_here = true; // chunk 0 starts from the cleared buffers' value (else from imported planes)
    bool sparse = false, work_list = false;
    size_t flags_per_group = 0, flags_per_span = 0;
    // sweep passes and the factor cache: per stream the entry its factors come from (a hit) or go to (being filled; null:
    // scratch only), and which streams' occlusion this pass computes (occ_mode: -1 none, else the occlusion kernel's mode)
    FactorEntry* f_entry[2] = {nullptr, nullptr};
    bool f_hit[2] = {false, false};
    int occ_mode = -1;
    BlockLists* lists = nullptr; // occ_mode >= 0: the pass's block lists (found among the handle's, or new: BlockLists::enqueued)
    int f_buf = 0;              // scratch buffer of this pass
    mutable bool occ_enqueued = false;
    // slab-partitioned passes
    bool lateral = false;       // the slices contain the slab axis: every rank runs every chunk on its rows
    int first_chunk_of_pass = 0, chunks_of_pass = 0;
    // a slab-partitioned pass whose taps reach too far for the chunk kernels: one slice per "chunk" with the reference's
    // kernel structure (k_propagate_slice) on the read / write buffers, which then are the planes the host exchanges
    bool sliced = false;
    PropParams slice_params{};
    // the pass's chunks are spans advanced by the pipelined sweep kernel (one launch per span; sweep_fit)
    bool sweep = false;
    SweepParams sq{};
    int halo_rows = 0;          // lateral: rows a slice's taps can reach beyond a slab
};

extern thread_local const char* g_plan_note; // why chunk_fit / plan_pass last declined a pass
bool chunk_fit(const tbrm_resources* r, const tbrm_light_pass& pa, const tbrm_light_pass* pr, ChunkFit& fit, int mode = -1);
int slice_tap_reach(const tbrm_light_pass& pa, const tbrm_light_pass* pr);
int plan_pass(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
              const tbrm_slab* slab, PassPlan& plan, int two_stream_mode = PASS_CHANGE, float b_added2 = 0.0f);
// two_way (fused Change only): the removed light's taps lie on the other side along some axis — its planes are swept first,
// on their own (r_*), and the fused sweep runs in the added light's order (SweepParams::r_from_records)
struct SweepFit { int sx = 0, sy = 0, hx = 0, hy = 0; bool two_way = false; int r_sx = 0, r_sy = 0, r_hx = 0, r_hy = 0; };
bool sweep_fit(const tbrm_resources* r, const tbrm_light_pass& pa, const tbrm_light_pass* pr, int mode, SweepFit& fit);
int sweep_decline_reason(const tbrm_resources* r, const tbrm_light_pass& pa);
void release_sweep(tbrm_resources* r);
int sweep_check(tbrm_resources* r);  // after the stream has drained: did a sweep kernel raise its error word? (+ the stamps' print-out)
int sweep_failed(tbrm_resources* r); // latches the error word; TBRM_OK or the (sticky) error
void sweep_failure_cleared(tbrm_resources* r); // the light volume has been defined anew
void drain_streams_public(tbrm_resources* r);
float* plan_plane(const tbrm_resources* r, int boundary, int si);
void* sliced_plane(const tbrm_resources* r, const PassPlan& plan, int boundary, int si);
int enqueue_plan_chunk(tbrm_resources* r, const PassPlan& plan, int c, const PassPlan* next = nullptr); // next: the plan enqueued after this one
int enqueue_sweep_occlusion(tbrm_resources* r, const PassPlan& plan); // a sweep pass's occlusion, ahead of its sweep (else: nothing)
bool dual_fit(const PassPlan& a, const PassPlan& b);                    // may ONE occlusion launch serve both passes?
int enqueue_dual_occlusion(tbrm_resources* r, const PassPlan& a, const PassPlan& b);
void quiesce_occ_stream(tbrm_resources* r);  // waits for the occlusion stream and forgets what its buffers hold
void release_kept(tbrm_resources* r);       // frees the factor cache (the streams must be idle)
void count_alloc(tbrm_resources* r, int calls, const char* what); // tbrm_path_counters [12] (sweep_debug bit 6: says what, on stderr)
int reserve_resources(tbrm_resources* r, int n_lights, unsigned flags); // tbrm_resources_reserve
int ensure_reserved(tbrm_resources* r);     // the first light operator of a handle nobody reserved: reserve_resources with the defaults
bool op_finished(tbrm_resources* r, uint64_t op); // has every sweep of operator `op` completed? (never blocks)
hipEvent_t take_event(tbrm_resources* r);   // from the handle's pool (null: creation failed)
void give_event(tbrm_resources* r, hipEvent_t ev);
BlockLists* make_spare_lists(tbrm_resources* r, size_t blocks, bool with_ranks); // tbrm_block_lists.cpp
// tbrm_block_lists.cpp
BlockLists* block_lists_for_pass(tbrm_resources* r, const ChunkParams& p, int occ_mode);  // null: allocation failed (tbrm_last_error)
BlockLists* block_lists_for_dual(tbrm_resources* r, const BlockLists* a, const BlockLists* b, size_t units);
bool block_lists_count(BlockLists* l, bool wait, size_t* count); // the live-block count, once it has arrived
void release_block_lists(tbrm_resources* r); // (the streams must be idle)
void release_occ_stores(tbrm_resources* r); // frees the occlusion stores and the factor cache (the streams must be idle)
size_t kept_bytes(const tbrm_resources* r);
int enqueue_add(tbrm_resources* r, const tbrm_dir_light_params& light, bool added, const tbrm_world_params& world);
int enqueue_add_batch(tbrm_resources* r, const tbrm_dir_light_params* lights, int n_lights, bool added, const tbrm_world_params& world,
                      int32_t* schedule, int32_t* n_entries);
int enqueue_change(tbrm_resources* r, const tbrm_dir_light_params& removed, const tbrm_dir_light_params& added_light,
                   const tbrm_world_params& world);

} // namespace tbrm_host

// A light operation taken apart for slab-partitioned execution: its axis passes, and the plan of the one being stepped.
struct SlabOp {
    tbrm_slab slab{};
    bool change = false;
    float b_added = 0.0f;
    int n = 0;
    tbrm_light_pass a[2]{}, r[2]{};
    tbrm::PropParams base{};
    int current = -1; // pass being stepped
    tbrm_host::PassPlan plan;
    BlockLists* held_lists = nullptr; // plan.lists, kept from being recycled while the plan is stored (BlockLists::users)
};

