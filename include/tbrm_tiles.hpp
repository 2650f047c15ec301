// tbrm_tiles.hpp — C++ host driver for image-tile rendering over the GPUs of one node, in ONE process: N whole-volume handles
// (volumes replicated, one per GPU), every light operator applied to each of them (no exchange: the update of ONE light is a
// serial slice sweep per axis), the lit frame split into interleaved groups of 8 rows (handle k renders every
// N-th group: an even share of the silhouette each) and gathered into one framebuffer —
//   * by hipMemcpy2DAsync between the handles' devices (xGMI between the MI355X of a node; the default), or
//   * by RCCL (-DTBRM_TILES_WITH_RCCL, link -lrccl): ncclAllGather of the tiles on the handles' own streams inside one
//     ncclGroupStart / ncclGroupEnd, one communicator per handle from ncclCommInitAll — what north_star words as "RCCL over
//     xGMI for the final framebuffer gather" — followed by a device-local interleave.
// The same decomposition over one process per GPU and torch.distributed is bench.py --gpus N (tbraymarcherplugin_amd/sharding.py);
// this is the shape a game-engine host has (the reference is a UE plugin: one process). No reference counterpart (the reference
// is single-GPU, RaymarchVolume.cpp:418-465); names follow URaymarchUtils / ARaymarchVolume.
//
// Header-only, on top of the C-ABI (tbrm.h) and the HIP runtime.
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -ltbrm -lamdhip64 [-DTBRM_TILES_WITH_RCCL -lrccl]
#pragma once

#include "tbrm.h"

#include <hip/hip_runtime_api.h>
#ifdef TBRM_TILES_WITH_RCCL
#include <rccl/rccl.h>
#endif

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace tbrm_plugin {

class FTileGroup {
public:
    enum class EGather { PeerCopy, Rccl };

    // One whole-volume handle per entry (created on Devices[k], volume / transfer function / window already set — or set through
    // ForEach); Width x Height: the framebuffer, Height a multiple of 8 * Num().
    FTileGroup(std::vector<tbrm_resources*> InHandles, std::vector<int> InDevices, int InWidth, int InHeight, EGather InGather = EGather::PeerCopy)
        : Handles(std::move(InHandles)), Devices(std::move(InDevices)), Width(InWidth), Height(InHeight), Gather(InGather)
    {
        const int N = (int) Handles.size();
        if (N == 0 || Devices.size() != Handles.size()) throw std::invalid_argument("FTileGroup: one device per handle");
        if (Width <= 0 || Height <= 0 || Height % (8 * N)) throw std::invalid_argument("FTileGroup: the height must split into groups of 8 rows per handle");
#ifndef TBRM_TILES_WITH_RCCL
        if (Gather == EGather::Rccl) throw std::invalid_argument("FTileGroup: built without TBRM_TILES_WITH_RCCL"); // (before anything is allocated)
#endif
        RowsPerHandle = Height / N;
        // every per-handle slot exists (null) before the first allocation: whatever throws below, Release() frees what was made
        Streams.assign(N, nullptr);
        Rendered.assign(N, nullptr);
        Gathered.assign(N, nullptr);
        Tiles.assign(N, nullptr);
        Frames.assign(N, nullptr);
        Staging.assign(N, nullptr);
        try {
            for (int k = 0; k < N; ++k) {
                void* S = nullptr;
                Check(tbrm_stream(Handles[k], &S), "tbrm_stream");
                Streams[k] = (hipStream_t) S;
                Hip(hipSetDevice(Devices[k]), "hipSetDevice");
                // direct copies between the handles' devices (xGMI); "already enabled" is fine, "not supported" leaves the runtime's staged path
                for (int j = 0; j < N; ++j)
                    if (Devices[j] != Devices[k]) {
                        int Can = 0;
                        if (hipDeviceCanAccessPeer(&Can, Devices[k], Devices[j]) == hipSuccess && Can) {
                            const hipError_t E = hipDeviceEnablePeerAccess(Devices[j], 0);
                            if (E != hipSuccess && E != hipErrorPeerAccessAlreadyEnabled) Hip(E, "hipDeviceEnablePeerAccess");
                            (void) hipGetLastError();
                            PeerDirect = true;
                        }
                    }
                Hip(hipEventCreateWithFlags(&Rendered[k], hipEventDisableTiming), "hipEventCreateWithFlags");
                Hip(hipEventCreateWithFlags(&Gathered[k], hipEventDisableTiming), "hipEventCreateWithFlags");
                Hip(hipMalloc((void**) &Tiles[k], TileBytes()), "hipMalloc");
                Hip(hipMalloc((void**) &Frames[k], FrameBytes()), "hipMalloc");
                if (Gather == EGather::Rccl) Hip(hipMalloc((void**) &Staging[k], FrameBytes()), "hipMalloc"); // the tiles of all handles, handle-major
            }
#ifdef TBRM_TILES_WITH_RCCL
            if (Gather == EGather::Rccl) {
                Comms.assign(N, nullptr);
                Nccl(ncclCommInitAll(Comms.data(), N, Devices.data()), "ncclCommInitAll");
            }
#endif
        } catch (...) {
            Release(); // (a destructor does not run for an object whose constructor threw)
            throw;
        }
    }
    ~FTileGroup() { Release(); }
    FTileGroup(const FTileGroup&) = delete;
    FTileGroup& operator=(const FTileGroup&) = delete;

    int Num() const { return (int) Handles.size(); }
    size_t TileBytes() const { return (size_t) RowsPerHandle * Width * 4 * sizeof(float); }
    size_t FrameBytes() const { return (size_t) Height * Width * 4 * sizeof(float); }
    size_t BytesMoved = 0; // between handles, since construction

    // the same call on every handle (uploads, tbrm_set_tf_lut, tbrm_set_windowing ...)
    template <class F>
    void ForEach(F&& Fn)
    {
        for (int k = 0; k < Num(); ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            Check(Fn(Handles[k], k), "FTileGroup::ForEach");
        }
    }

    // AddDirLightToSingleVolume / ChangeDirLightInSingleVolume / ClearResourceLightVolumes on every handle: the replicas stay
    // bit-identical because every handle runs the same operators in the same order.
    void AddDirLight(const tbrm_dir_light_params& Light, bool bAdded, const tbrm_world_params& World)
    {
        ForEach([&](tbrm_resources* H, int) { int Flag = 0; return tbrm_add_dir_light(H, &Light, bAdded ? 1 : 0, &World, &Flag, 0); });
    }
    void ChangeDirLight(const tbrm_dir_light_params& Old, const tbrm_dir_light_params& New, const tbrm_world_params& World)
    {
        ForEach([&](tbrm_resources* H, int) { int Flag = 0; return tbrm_change_dir_light(H, &Old, &New, &World, &Flag, 0); });
    }
    void ClearLightVolumes(float Value = 0.0f)
    {
        ForEach([&](tbrm_resources* H, int) { return tbrm_clear_light_volume(H, Value); });
    }
    // ARaymarchVolume::ResetAllLights (RaymarchVolume.cpp:418-451)
    void ResetAllLights(const std::vector<tbrm_dir_light_params>& Lights, const tbrm_world_params& World)
    {
        ClearLightVolumes(0.0f);
        for (const tbrm_dir_light_params& L : Lights) AddDirLight(L, true, World);
    }

    // The lit frame: every handle marches its rows (handle k: groups k, k + N, ... of 8 rows), then the tiles are gathered.
    // Nothing waits on the host: the copies / collectives are ordered behind the marches by events on the handles' streams.
    // bEveryHandle: every handle ends up with the whole frame (an all-gather); else only handle `Root` does (a gather).
    // Returns the device pointer of the assembled frame on handle Root (Width x Height x RGBA f32, premultiplied); it is complete
    // once that handle's stream has reached this point (tbrm_flush(Handle(Root)), or further work enqueued on its stream).
    const float* RenderLit(const tbrm_camera& Camera, const tbrm_raymarch_params& Params, const tbrm_world_params& World, int Root = 0,
                           bool bEveryHandle = false)
    {
        const int N = Num();
        if (Camera.width != Width || Camera.height != Height) throw std::invalid_argument("FTileGroup::RenderLit: the camera's framebuffer is not the group's");
        for (int k = 0; k < N; ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            // (a gather that still reads Tiles[k] for the frame before: the marches of this frame come behind it)
            for (int j = 0; j < N; ++j)
                if (j != k && Pending) Hip(hipStreamWaitEvent(Streams[k], Gathered[j], 0), "hipStreamWaitEvent");
            const tbrm_tile Tile{0, 8 * k, Width, RowsPerHandle, N};
            Check(tbrm_raymarch_lit_device(Handles[k], &Camera, &Tile, &Params, &World, nullptr, Tiles[k]), "tbrm_raymarch_lit_device");
            Hip(hipEventRecord(Rendered[k], Streams[k]), "hipEventRecord");
        }
        if (Gather == EGather::Rccl) GatherRccl();
        else GatherPeerCopies(Root, bEveryHandle);
        Pending = true;
        return Frames[Root];
    }
    tbrm_resources* Handle(int k) const { return Handles[k]; }
    const float* Frame(int k) const { return Frames[k]; }
    bool PeerDirect = false; // some pair of the handles' devices copies directly (hipDeviceEnablePeerAccess succeeded)

private:
    void Release() // idempotent; also the clean-up of a constructor that threw half way
    {
        for (tbrm_resources* H : Handles) (void) tbrm_flush(H);
#ifdef TBRM_TILES_WITH_RCCL
        for (ncclComm_t& C : Comms) {
            if (C) (void) ncclCommDestroy(C);
            C = nullptr;
        }
#endif
        for (size_t k = 0; k < Tiles.size(); ++k) {
            (void) hipSetDevice(Devices[k]);
            if (Rendered[k]) (void) hipEventDestroy(Rendered[k]);
            if (Gathered[k]) (void) hipEventDestroy(Gathered[k]);
            (void) hipFree(Tiles[k]);
            (void) hipFree(Frames[k]);
            (void) hipFree(Staging[k]);
            Rendered[k] = Gathered[k] = nullptr;
            Tiles[k] = Frames[k] = Staging[k] = nullptr;
        }
    }

    std::vector<tbrm_resources*> Handles;
    std::vector<int> Devices;
    int Width, Height, RowsPerHandle = 0;
    EGather Gather;
    std::vector<hipStream_t> Streams;
    std::vector<hipEvent_t> Rendered, Gathered;
    std::vector<float*> Tiles, Frames, Staging;
    bool Pending = false;
#ifdef TBRM_TILES_WITH_RCCL
    std::vector<ncclComm_t> Comms;
    static void Nccl(ncclResult_t R, const char* What)
    {
        if (R != ncclSuccess) throw std::runtime_error(std::string(What) + ": " + ncclGetErrorString(R));
    }
#endif

    static void Check(int Code, const char* What)
    {
        if (Code != TBRM_OK) throw std::runtime_error(std::string(What) + ": " + tbrm_last_error());
    }
    static void Hip(hipError_t E, const char* What)
    {
        if (E != hipSuccess) throw std::runtime_error(std::string(What) + ": " + hipGetErrorString(E));
    }

    // handle Src's tile into the frame of handle Dst: group g of the tile is rows (g N + Src) 8 ... + 8 of the frame — one strided copy
    void Interleave(int Dst, const float* TileOfSrc, int Src, hipStream_t Stream)
    {
        const size_t Group = (size_t) 8 * Width * 4 * sizeof(float);
        Hip(hipMemcpy2DAsync((char*) Frames[Dst] + (size_t) Src * Group, Group * Num(), TileOfSrc, Group, Group, RowsPerHandle / 8, hipMemcpyDefault, Stream),
            "hipMemcpy2DAsync");
    }

    // every destination pulls the tiles on its own stream, behind the events of the marches that made them
    void GatherPeerCopies(int Root, bool bEveryHandle)
    {
        const int N = Num();
        for (int d = 0; d < N; ++d) {
            if (!bEveryHandle && d != Root) { // (still marks the point up to which its tile may be read)
                Hip(hipSetDevice(Devices[d]), "hipSetDevice");
                continue;
            }
            Hip(hipSetDevice(Devices[d]), "hipSetDevice");
            for (int s = 0; s < N; ++s) {
                if (s != d) Hip(hipStreamWaitEvent(Streams[d], Rendered[s], 0), "hipStreamWaitEvent");
                Interleave(d, Tiles[s], s, Streams[d]);
                if (s != d) BytesMoved += TileBytes();
            }
        }
        for (int d = 0; d < N; ++d) { // the readers of every tile are done behind these
            Hip(hipSetDevice(Devices[d]), "hipSetDevice");
            Hip(hipEventRecord(Gathered[d], Streams[d]), "hipEventRecord");
        }
    }

    void GatherRccl()
    {
#ifdef TBRM_TILES_WITH_RCCL
        const int N = Num();
        Nccl(ncclGroupStart(), "ncclGroupStart");
        for (int k = 0; k < N; ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            Nccl(ncclAllGather(Tiles[k], Staging[k], TileBytes() / sizeof(float), ncclFloat, Comms[k], Streams[k]), "ncclAllGather");
        }
        Nccl(ncclGroupEnd(), "ncclGroupEnd");
        for (int k = 0; k < N; ++k) {
            Hip(hipSetDevice(Devices[k]), "hipSetDevice");
            for (int s = 0; s < N; ++s) Interleave(k, Staging[k] + (size_t) s * (TileBytes() / sizeof(float)), s, Streams[k]);
            Hip(hipEventRecord(Gathered[k], Streams[k]), "hipEventRecord");
            BytesMoved += (size_t) (N - 1) * TileBytes();
        }
#endif
    }
};

} // namespace tbrm_plugin
