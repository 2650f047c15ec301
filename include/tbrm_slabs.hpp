// tbrm_slabs.hpp — C++ host driver for slab-partitioned operation over the GPUs of one node, in ONE process (the shape a
// game engine has: the reference is a UE plugin, one process, and would own one tbrm handle per GPU).
//
// Header-only, on top of the C-ABI (tbrm.h: tbrm_slab_*, tbrm_resources_create_slab, tbrm_slab_light_halo,
// tbrm_raymarch_lit_slab_device) and the HIP runtime for the copies between GPUs (hipMemcpyPeerAsync: xGMI between the
// MI355X of a node, an ordinary device copy when two handles share a GPU). The same sequence over one process per GPU and
// RCCL is tbraymarcherplugin_amd/slabs.py; DESIGN.md 7 describes the decomposition. There is no reference counterpart
// (the reference is single-GPU); the operator names follow URaymarchUtils / ARaymarchVolume.
//
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -ltbrm -lamdhip64
#pragma once

#include "tbrm.h"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace tbrm_plugin {

class FSlabGroup {
public:
    // Handles[k] owns light-volume slices [Bounds[k], Bounds[k + 1]); whole-volume or slab-resident handles alike.
    FSlabGroup(std::vector<tbrm_resources*> InHandles, std::vector<int32_t> InBounds, std::vector<int> InDevices)
        : Handles(std::move(InHandles)), Bounds(std::move(InBounds)), Devices(std::move(InDevices))
    {
        if (Handles.empty() || Bounds.size() != Handles.size() + 1 || Devices.size() != Handles.size())
            throw std::invalid_argument("FSlabGroup: one device and one [begin, end) per handle");
        for (size_t k = 0; k < Handles.size(); ++k) {
            void* S = nullptr;
            Check(tbrm_stream(Handles[k], &S), "tbrm_stream");
            Streams.push_back((hipStream_t) S);
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            hipEvent_t A = nullptr, B = nullptr;
            Hip(hipEventCreateWithFlags(&A, hipEventDisableTiming), "hipEventCreateWithFlags");
            Hip(hipEventCreateWithFlags(&B, hipEventDisableTiming), "hipEventCreateWithFlags");
            Done.push_back(A);
            CopiesDone.push_back(B);
        }
    }
    ~FSlabGroup()
    {
        for (hipEvent_t E : Done) (void) hipEventDestroy(E);
        for (hipEvent_t E : CopiesDone) (void) hipEventDestroy(E);
    }
    FSlabGroup(const FSlabGroup&) = delete;
    FSlabGroup& operator=(const FSlabGroup&) = delete;

    // Ordering between the handles' streams: by default with events only (hipStreamWaitEvent) — the host never waits inside
    // an operation, a handle's stream runs ahead as far as its neighbours' data allows. true: drain every stream around
    // every exchange instead (the simple, slow form; kept for A/B checks).
    bool bHostSynchronise = false;

    int Num() const { return (int) Handles.size(); }
    tbrm_slab Slab(int k) const { return tbrm_slab{Bounds[k], Bounds[k + 1]}; }
    size_t BytesMoved = 0; // between handles, since construction

    // AddDirLightToSingleVolume over the slabs
    void AddDirLight(const tbrm_dir_light_params& Light, bool bAdded, const tbrm_world_params& World) { LightOperation(nullptr, Light, bAdded, World); }

    // ChangeDirLightInSingleVolume over the slabs, with the reference's fallback to remove + add when the major axes
    // differ (LightingShaders.cpp:192-198)
    void ChangeDirLight(const tbrm_dir_light_params& Old, const tbrm_dir_light_params& New, const tbrm_world_params& World)
    {
        if (LightOperation(&Old, New, true, World)) return;
        LightOperation(nullptr, Old, false, World);
        LightOperation(nullptr, New, true, World);
    }

    // ARaymarchVolume::ResetAllLights (RaymarchVolume.cpp:418-451)
    void ResetAllLights(const std::vector<tbrm_dir_light_params>& Lights, const tbrm_world_params& World)
    {
        for (tbrm_resources* H : Handles) Check(tbrm_clear_light_volume(H, 0.0f), "tbrm_clear_light_volume");
        for (const tbrm_dir_light_params& L : Lights) AddDirLight(L, true, World);
    }

    // Slab-resident handles: after light operations, before a frame — every handle receives its two neighbours' boundary
    // brick layers of the light volume (a ring: the light volume is sampled with wrap addressing).
    void ExchangeLightHalos()
    {
        const int N = Num();
        if (N == 1) return;
        if (bHostSynchronise) Drain();
        else
            for (int k = 0; k < N; ++k) Mark(k); // the light operations so far
        struct FSide { void* Send; void* Recv; size_t Bytes; };
        std::vector<FSide> Lower(N), Upper(N);
        for (int k = 0; k < N; ++k) {
            Check(tbrm_slab_light_halo(Handles[k], 0, &Lower[k].Send, &Lower[k].Recv, &Lower[k].Bytes), "tbrm_slab_light_halo");
            Check(tbrm_slab_light_halo(Handles[k], 1, &Upper[k].Send, &Upper[k].Recv, &Upper[k].Bytes), "tbrm_slab_light_halo");
        }
        for (int k = 0; k < N; ++k) {
            const int Up = this->Up(k);
            Copy(Lower[Up].Recv, Up, Upper[k].Send, k, Upper[k].Bytes);  // slab k's last layer -> the lower halo of the slab above
            Copy(Upper[k].Recv, k, Lower[Up].Send, Up, Lower[Up].Bytes); // that slab's first layer -> the upper halo of slab k
        }
        if (bHostSynchronise) Drain();
    }

    // The lit frame, slab by slab (tbrm_raymarch_lit_slab_device): the per-pixel state sweeps up through the slabs for the
    // rays that travel towards +z in volume space and down for the others; bit for bit the unpartitioned frame. The
    // result (tile.w * tile.h float4) is copied to HostOutRGBA.
    void RenderLit(const tbrm_camera& Camera, const tbrm_tile& Tile, const tbrm_raymarch_params& Params, const tbrm_world_params& World,
                   float* HostOutRGBA)
    {
        const int N = Num();
        const size_t Bytes = (size_t) Tile.w * Tile.h * 4 * sizeof(float);
        if (Bytes == 0) return;
        std::vector<void*> State(N, nullptr);
        for (int k = 0; k < N; ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            Hip(hipMalloc(&State[k], Bytes), "hipMalloc");
        }
        Hip(hipSetDevice(Devices[0]), "hipSetDevice");
        Hip(hipMemsetAsync(State[0], 0, Bytes, Streams[0]), "hipMemsetAsync");
        Mark(0);
        int Prev = 0;
        for (int Stage = 0; Stage < 2 * N; ++Stage) {
            const int k = Stage < N ? Stage : 2 * N - 1 - Stage;
            const int Direction = Stage < N ? +1 : -1;
            if (k != Prev) {
                if (bHostSynchronise) Hip(hipStreamSynchronize(Streams[Prev]), "hipStreamSynchronize");
                Copy(State[k], k, State[Prev], Prev, Bytes);
            }
            const tbrm_slab S = Slab(k);
            Check(tbrm_raymarch_lit_slab_device(Handles[k], &Camera, &Tile, &Params, &World, nullptr, (float*) State[k], &S, Direction),
                  "tbrm_raymarch_lit_slab_device");
            Mark(k);
            Prev = k;
        }
        Drain(); // the last stage depends on every stage and copy before it; drained here before the state buffers go
        Hip(hipSetDevice(Devices[0]), "hipSetDevice");
        Hip(hipMemcpy(HostOutRGBA, State[0], Bytes, hipMemcpyDeviceToHost), "hipMemcpy");
        for (int k = 0; k < N; ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            Hip(hipFree(State[k]), "hipFree");
        }
    }

private:
    std::vector<tbrm_resources*> Handles;
    std::vector<int32_t> Bounds;
    std::vector<int> Devices;
    std::vector<hipStream_t> Streams;
    std::vector<hipEvent_t> Done;       // on stream k: its latest chunk / stage / copy-in has been enqueued up to here
    std::vector<hipEvent_t> CopiesDone; // on stream k: the copies it performed (reading a neighbour's memory) up to here

    int Up(int k) const { return (k + 1) % Num(); }
    int Down(int k) const { return (k + Num() - 1) % Num(); }
    void Mark(int k)
    {
        if (!bHostSynchronise) Hip(hipEventRecord(Done[k], Streams[k]), "hipEventRecord");
    }
    // Before handle k overwrites memory its neighbours copy from (planes, buffers, boundary light-volume layers): their
    // copies so far must have been performed.
    void WaitReaders(int k)
    {
        if (bHostSynchronise || Num() == 1) return;
        Hip(hipStreamWaitEvent(Streams[k], CopiesDone[Down(k)], 0), "hipStreamWaitEvent");
        Hip(hipStreamWaitEvent(Streams[k], CopiesDone[Up(k)], 0), "hipStreamWaitEvent");
    }

    static void Check(int Code, const char* What)
    {
        if (Code != TBRM_OK) throw std::runtime_error(std::string(What) + ": " + tbrm_last_error());
    }
    static void Hip(hipError_t E, const char* What)
    {
        if (E != hipSuccess) throw std::runtime_error(std::string(What) + ": " + hipGetErrorString(E));
    }
    void Drain()
    {
        for (tbrm_resources* H : Handles) Check(tbrm_flush(H), "tbrm_flush");
    }
    // Dst on handle DstK <- Src on handle SrcK, enqueued on the destination's stream once the source handle's work up to
    // its last Mark() is done (host-synchronised mode: the caller has drained the streams).
    void Copy(void* Dst, int DstK, const void* Src, int SrcK, size_t Bytes)
    {
        if (!Dst || !Src || Bytes == 0) return;
        Hip(hipSetDevice(Devices[DstK]), "hipSetDevice");
        if (!bHostSynchronise) Hip(hipStreamWaitEvent(Streams[DstK], Done[SrcK], 0), "hipStreamWaitEvent");
        Hip(hipMemcpyPeerAsync(Dst, Devices[DstK], Src, Devices[SrcK], Bytes, Streams[DstK]), "hipMemcpyPeerAsync");
        if (!bHostSynchronise) Hip(hipEventRecord(CopiesDone[DstK], Streams[DstK]), "hipEventRecord");
        BytesMoved += Bytes;
    }

    // one AddDirLight (Removed == null) or fused ChangeDirLight; false: the Change has to run as remove + add
    bool LightOperation(const tbrm_dir_light_params* Removed, const tbrm_dir_light_params& Light, bool bAdded, const tbrm_world_params& World)
    {
        const int N = Num();
        int32_t NumPasses = 0;
        for (int k = 0; k < N; ++k) {
            const tbrm_slab S = Slab(k);
            const int Code = tbrm_slab_light_begin(Handles[k], Removed, &Light, bAdded ? 1 : 0, &World, &S, &NumPasses);
            if (Code == TBRM_ERR_AXES_DIFFER && Removed) return false;
            Check(Code, "tbrm_slab_light_begin");
        }
        for (int32_t Pass = 0; Pass < NumPasses; ++Pass) {
            std::vector<tbrm_slab_pass> Desc(N);
            for (int k = 0; k < N; ++k) {
                WaitReaders(k); // the set-up of a pass may already write its buffers
                Check(tbrm_slab_pass_begin(Handles[k], Pass, &Desc[k]), "tbrm_slab_pass_begin");
            }
            const size_t RowBytes = (size_t) Desc[0].plane_w * (size_t) Desc[0].plane_elem_bytes;
            if (Desc[0].lateral) { // every slab runs every chunk on its rows; halo rows cross the slab boundaries after each
                for (int32_t c = 0; c < Desc[0].n_chunks; ++c) {
                    for (int k = 0; k < N; ++k) {
                        WaitReaders(k); // chunk c overwrites the plane the neighbours fetched rows of two chunks ago
                        Check(tbrm_slab_pass_chunk(Handles[k], c), "tbrm_slab_pass_chunk");
                        Mark(k);
                    }
                    if (c + 1 == Desc[0].n_chunks) break;
                    if (bHostSynchronise) Drain();
                    const int32_t Halo = Desc[0].halo_rows;
                    for (int k = 0; k + 1 < N; ++k) {
                        const int32_t z = Bounds[k + 1];
                        for (int32_t s = 0; s < Desc[0].streams; ++s) {
                            char *Lo = Plane(k, c + 1, s), *Hi = Plane(k + 1, c + 1, s);
                            Copy(Hi + (size_t) (z - Halo) * RowBytes, k + 1, Lo + (size_t) (z - Halo) * RowBytes, k, (size_t) Halo * RowBytes);
                            Copy(Lo + (size_t) z * RowBytes, k, Hi + (size_t) z * RowBytes, k + 1, (size_t) Halo * RowBytes);
                        }
                    }
                    if (bHostSynchronise) Drain();
                }
            } else { // a pipeline along z, in propagation order
                for (int Pos = 0; Pos < N; ++Pos) {
                    const int k = Desc[0].dir > 0 ? Pos : N - 1 - Pos;
                    if (Pos > 0) {
                        const int Before = Desc[0].dir > 0 ? k - 1 : k + 1;
                        if (bHostSynchronise) Check(tbrm_flush(Handles[Before]), "tbrm_flush");
                        for (int32_t s = 0; s < Desc[0].streams; ++s)
                            Copy(Plane(k, 0, s), k, Plane(Before, Desc[Before].n_chunks, s), Before, (size_t) Desc[0].plane_h * RowBytes);
                    }
                    for (int32_t c = 0; c < Desc[k].n_chunks; ++c) Check(tbrm_slab_pass_chunk(Handles[k], c), "tbrm_slab_pass_chunk");
                    Mark(k);
                }
            }
        }
        return true;
    }
    char* Plane(int k, int32_t Boundary, int32_t Stream)
    {
        void* P = nullptr;
        Check(tbrm_slab_pass_plane(Handles[k], Boundary, Stream, &P), "tbrm_slab_pass_plane");
        return (char*) P;
    }
};

} // namespace tbrm_plugin
