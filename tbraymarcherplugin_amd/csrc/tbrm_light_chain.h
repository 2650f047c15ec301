// tbrm_light_chain.h — what the chain kernel (tbrm_light_chain.hip), the occlusion kernel (tbrm_light_kernels.hip) and
// the host-side planner (tbrm_light_plan.cpp) have to agree on: the window geometry of a chain workgroup, the shape of
// its LDS planes and the global->LDS copy helpers. Internal.
#pragma once
#include "tbrm_internal.h"

#include <atomic>

namespace tbrm {

// Window geometry of one chain workgroup. A tile keeps its 32x32 pixels for the whole chunk; with r slices still to
// go its window is [r*lox, T + r*hix) x [r*loy, T + r*hiy) in tile coordinates (lox <= 0 <= hix: the range of the
// previous-slice taps, widened to contain 0), i.e. it grows towards the light by the tap range per remaining slice.
struct ChunkGeom {
    int n;                  // steps in this chunk
    int lox, hix, loy, hiy;
    int HX, HY;             // hull = window at r = n (the input state)
    int RS;                 // LDS plane row stride in floats (0: the hull does not fit any instantiation)
    int RR;                 // LDS plane rows: RS, or (ChunkParams::rect_planes) 48 under a 72-float stride / 64 under a 56-float one
    int padx, pady;         // plane coordinates of tile pixel (0,0)
    int lv_layers;          // 8-slice brick layers of the light volume the chunk touches
    int lv_layer0;          // first of them
};

// LDS planes are RS x RR floats, RS an odd multiple of 8 (bank-conflict-free 8x8 patches, see k_light_chain); RR = RS except
// for the two rectangular shapes: 72 x 48 and 56 x 64, the hulls of a 16-slice chunk whose taps are two texels wide along x
// (y) and one along the other axis — the usual second pass of a slanted light — and of an 8-slice chunk with taps four
// texels wide. Eight square 72 x 72 planes (two windows, three ring slots of occlusion factors and kept L) exceed the
// LDS, eight of 72 x 48 take 111 KB, so such a pass runs chunks twice as long.
__host__ __device__ constexpr int chain_plane_elems(int RS, int RR) { return RS * RR + 8; } // + slack for inactive slots' reads
__host__ __device__ constexpr int chain_row_stride(int hull) { return hull <= 40 ? 40 : (hull <= 56 ? 56 : (hull <= 72 ? 72 : 0)); }

__host__ __device__ inline ChunkGeom chunk_geometry(const ChunkParams& p)
{
    ChunkGeom g;
    g.n = p.n_steps;
    g.lox = p.dx_lo; g.hix = p.dx_hi; g.loy = p.dy_lo; g.hiy = p.dy_hi;
    g.HX = kChunkTile + g.n * (g.hix - g.lox);
    g.HY = kChunkTile + g.n * (g.hiy - g.loy);
    g.RS = chain_row_stride(g.HX > g.HY ? g.HX : g.HY);
    g.RR = g.RS;
    if (p.rect_planes && g.HX > 56 && g.HX <= 72 && g.HY <= 48) { g.RS = 72; g.RR = 48; }
    else if (p.rect_planes && g.HX <= 56 && g.HY > 56 && g.HY <= 64) { g.RS = 56; g.RR = 64; }
    g.padx = -g.n * g.lox;
    g.pady = -g.n * g.loy;
    const int ja = p.j0, jb = p.j0 + (g.n - 1) * p.dir;
    const int jlo = ja < jb ? ja : jb, jhi = ja < jb ? jb : ja;
    g.lv_layer0 = jlo >> 3;
    g.lv_layers = (jhi >> 3) - g.lv_layer0 + 1;
    return g;
}

constexpr int kOccRing = 3; // slices the occlusion operands are staged ahead of their use

// Workgroup barrier for LDS traffic only. __syncthreads() carries a workgroup-scope fence, which the compiler has to
// lower to s_waitcnt vmcnt(0): inside the chain's slice loop that would drain the asynchronous global->LDS copies
// issued for the slices AHEAD at every barrier and expose their full latency once per slice. Here only this wave's LDS
// operations are waited for; copy completion is tracked explicitly with s_waitcnt vmcnt(N) by the caller.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// asynchronous global -> LDS copies: lane l of the wave lands at lds_wave_base + size*l
__device__ __forceinline__ void dma_dword(const float* src, float* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                     (__attribute__((address_space(3))) void*) lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void dma_16(const void* src, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                     (__attribute__((address_space(3))) void*) lds_wave_base, 16, 0, 0);
}

constexpr int kMaxDevices = 64;
inline int current_device()
{
    int dev = 0;
    (void) hipGetDevice(&dev);
    return dev >= 0 && dev < kMaxDevices ? dev : 0;
}

// the dynamic-LDS ceiling of a kernel is a per-device attribute: raised once per device and instantiation (`done`: one
// bit per device, a function-local static of the launcher)
template <typename K>
inline hipError_t allow_big_lds(K kernel, std::atomic<uint64_t>& done, int bytes)
{
    const uint64_t bit = (uint64_t) 1 << current_device();
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// per-format halves of launch_light_chain (one translation unit each, tbrm_light_chain.hip compiled twice)
hipError_t launch_light_chain_u8(const ChunkParams& p, int mode, hipStream_t s);
hipError_t launch_light_chain_f32(const ChunkParams& p, int mode, hipStream_t s);

} // namespace tbrm
