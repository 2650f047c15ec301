// tbrm_light_passes.h — what the four translation units of the host side of the illumination operators share beyond
// tbrm_resources.h: tbrm_light_plan.cpp (which kernel a pass takes, its geometry), tbrm_factor_cache.cpp (occlusion stores, the
// factor cache), tbrm_light_enqueue.cpp (launches and stream ordering), tbrm_light_operators.cpp (Add / Change / batches). Internal.
#pragma once
#include "tbrm_resources.h"
#include "tbrm_light_chain.h"
#include "tbrm_light_sweep.h"

#include <algorithm>
#include <climits>
#include <chrono>
#include <cmath>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace tbrm_host {

struct SpanRange { int s0, sn, c0, c1; bool sparse; };

// tbrm_light_plan.cpp
unsigned event_flags();
bool force_slice_kernel();
int chunk_steps_override();
int ensure_sweep(tbrm_resources* r, size_t words, size_t words1 = 0, size_t chain_tiles = 0);
int next_sweep_epoch(tbrm_resources* r, uint32_t& epoch, uint32_t launches = 1);
float through_light_format(int lv_fmt, float v);
void fill_chunk_stream(ChunkStream& s, const tbrm_light_pass& p, int lv_fmt);
int plan_pass_sliced(tbrm_resources* r, const PropParams& base, const tbrm_light_pass& pa, const tbrm_light_pass* pr, float b_added,
                     const tbrm_slab& slab, PassPlan& plan);
SpanRange span_range(const PassPlan& plan, int sp);
// tbrm_factor_cache.cpp
void drain_streams(tbrm_resources* r);
int ensure_store(tbrm_resources* r, OccStore* st, int slices, size_t slice_elems, size_t flag_bytes);
int ensure_factor_scratch(tbrm_resources* r, int b, size_t blocks, int streams);
FactorKey factor_key(const tbrm_resources* r, const PropParams& base, const tbrm_light_pass& q, bool guard, int start, int D);
void estimate_scope(tbrm_resources* r, const PropParams& base);
void resolve_entry(tbrm_resources* r, FactorEntry* e, bool wait);
FactorEntry* kept_find(tbrm_resources* r, const FactorKey& key);
FactorEntry* kept_new(tbrm_resources* r, const FactorKey& key, size_t want, BlockLists* lists);
void use_kept(tbrm_resources* r, FactorEntry* e, bool leaves_the_scene);
void unpin_kept(tbrm_resources* r);
bool cache_usable(const tbrm_resources* r);
int ensure_occ_stream(tbrm_resources* r);
// tbrm_light_enqueue.cpp
int enqueue_sweep_pair(tbrm_resources* r, const PassPlan& pa, const PassPlan& pb, const SweepFit& fit);
int enqueue_sweep_chain(tbrm_resources* r, const PassPlan* plans, int n);
int enqueue_pass_sliced(tbrm_resources* r, PropParams p, const tbrm_light_pass& pa, const tbrm_light_pass* pr);
} // namespace tbrm_host
