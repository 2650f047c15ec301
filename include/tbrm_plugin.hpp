// tbrm_plugin.hpp — C++ host side above the C-ABI (include/tbrm.h), mirroring the plugin's operator surface so that
// code written against the reference reads the same here:
//
//   FDirLightParameters / FClippingPlaneParameters / FRaymarchWorldParameters / FBasicRaymarchRenderingResources
//       Source/Raymarcher/Public/Rendering/RaymarchTypes.h:20-153
//   FWindowingParameters                       Source/VolumeTextureToolkit/Public/VolumeAsset/VolumeInfo.h:31-53
//   URaymarchUtils (static operators)          Source/Raymarcher/Public/Util/RaymarchUtils.h:33-93
//   ARaymarchLight / ARaymarchClipPlane        Source/Raymarcher/Private/Actor/RaymarchLight.cpp:28-31, RaymarchClipPlane.cpp:32-35
//   ARaymarchVolume (Tick / ResetAllLights / UpdateSingleLight / setters)
//       Source/Raymarcher/Private/Actor/RaymarchVolume.cpp:327-465, :630-662, :746-800, :821-949
//
// Header-only, no engine types: FVector/FQuat/FTransform are the ABI PODs. The raymarch itself has no C++ entry point
// in the reference (it is a material on a cube mesh, RaymarchVolume.cpp:37-49); here ARaymarchVolume::RenderLit is the
// offscreen replacement. BASELINE.json spells two of the names URaymarchVolume / FRaymarchResources: both aliases exist.
#pragma once

#include "tbrm.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace tbrm_plugin {

using FVector = tbrm_vec3d;
using FQuat = tbrm_quatd;
using FTransform = tbrm_transform;

inline bool operator==(const FVector& a, const FVector& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const FVector& a, const FVector& b) { return !(a == b); }

// FTransform::Equals with the engine's default tolerances is not reproducible without the engine; exact comparison is
// the conservative choice (any change of the transform requests a recompute, RaymarchVolume.cpp:351).
inline bool TransformEquals(const FTransform& a, const FTransform& b)
{
    return a.rotation.x == b.rotation.x && a.rotation.y == b.rotation.y && a.rotation.z == b.rotation.z &&
           a.rotation.w == b.rotation.w && a.translation == b.translation && a.scale3d == b.scale3d;
}

struct FDirLightParameters { // RaymarchTypes.h:20-41
    FVector LightDirection{0, 0, 0};
    float LightIntensity = 0;
    FDirLightParameters() = default;
    FDirLightParameters(FVector LightDir, float LightInt) : LightDirection(LightDir), LightIntensity(LightInt) {}
    bool operator==(const FDirLightParameters& rhs) const { return LightDirection == rhs.LightDirection && LightIntensity == rhs.LightIntensity; }
    bool operator!=(const FDirLightParameters& rhs) const { return !(*this == rhs); }
    tbrm_dir_light_params abi() const { return tbrm_dir_light_params{LightDirection, LightIntensity, 0}; }
};

struct FClippingPlaneParameters { // RaymarchTypes.h:45-71
    FVector Center{0, 0, 0};
    FVector Direction{0, 0, 0};
    FClippingPlaneParameters() = default;
    FClippingPlaneParameters(FVector ClipCenter, FVector ClipDirection) : Center(ClipCenter), Direction(ClipDirection) {}
    friend bool operator==(const FClippingPlaneParameters& l, const FClippingPlaneParameters& r) { return l.Center == r.Center && l.Direction == r.Direction; }
    friend bool operator!=(const FClippingPlaneParameters& l, const FClippingPlaneParameters& r) { return !(l == r); }
};

struct FRaymarchWorldParameters { // RaymarchTypes.h:136-153
    FTransform VolumeTransform{{0, 0, 0, 1}, {0, 0, 0}, {1, 1, 1}};
    FClippingPlaneParameters ClippingPlaneParameters;
    friend bool operator==(const FRaymarchWorldParameters& l, const FRaymarchWorldParameters& r)
    {
        return TransformEquals(l.VolumeTransform, r.VolumeTransform) && l.ClippingPlaneParameters == r.ClippingPlaneParameters;
    }
    friend bool operator!=(const FRaymarchWorldParameters& l, const FRaymarchWorldParameters& r) { return !(l == r); }
    tbrm_world_params abi() const
    {
        return tbrm_world_params{VolumeTransform, {ClippingPlaneParameters.Center, ClippingPlaneParameters.Direction}};
    }
};

struct FWindowingParameters { // VolumeInfo.h:31-53
    float Center = 0.5f;
    float Width = 1.0f;
    bool LowCutoff = true;
    bool HighCutoff = true;
    tbrm_windowing_params abi() const { return tbrm_windowing_params{Center, Width, LowCutoff ? 1 : 0, HighCutoff ? 1 : 0}; }
};

// UCurveLinearColor restricted to what every shipped TF curve uses: RCIM_Linear keys per channel.
struct FColorCurve {
    std::vector<float> Times[4], Values[4]; // R, G, B, A
    void AddKey(float t, float r, float g, float b, float a)
    {
        const float v[4] = {r, g, b, a};
        for (int c = 0; c < 4; ++c) { Times[c].push_back(t); Values[c].push_back(v[c]); }
    }
};

enum class ERaymarchMaterial { Lit, Intensity, Octree }; // RaymarchVolume.h:24-29

// RaymarchTypes.h:87-129. The GPU objects the reference holds by pointer live behind one opaque handle.
struct FBasicRaymarchRenderingResources {
    bool bIsInitialized = false;
    tbrm_resources* Handle = nullptr; // DataVolumeTextureRef + TFTextureRef + LightVolumeRenderTarget + XYZReadWriteBuffers
    bool LightVolumeHalfResolution = false;
    FWindowingParameters WindowingParameters;
    int SizeX = 0, SizeY = 0, SizeZ = 0; // DataVolumeTextureRef->GetSizeX/Y/Z()
};
using FRaymarchResources = FBasicRaymarchRenderingResources;

struct URaymarchUtils { // RaymarchUtils.h:33-93; all static, like the Blueprint function library
    static void AddDirLightToSingleVolume(const FBasicRaymarchRenderingResources& Resources, const FDirLightParameters& LightParameters,
                                          const bool Added, const FRaymarchWorldParameters WorldParameters, bool& LightAdded, bool bGPUSync = false)
    {
        const tbrm_dir_light_params l = LightParameters.abi();
        const tbrm_world_params w = WorldParameters.abi();
        int flag = 0;
        tbrm_add_dir_light(Resources.Handle, &l, Added ? 1 : 0, &w, &flag, bGPUSync ? 1 : 0);
        LightAdded = flag != 0;
    }
    // Not in the reference: several AddDirLightToSingleVolume calls as one (tbrm_add_dir_lights: passes of different lights
    // that leave the same cube face share one slice loop). Returns the number of passes that ran paired.
    static int AddDirLightsToSingleVolume(const FBasicRaymarchRenderingResources& Resources, const std::vector<FDirLightParameters>& Lights,
                                          const bool Added, const FRaymarchWorldParameters WorldParameters, bool& LightsAdded)
    {
        std::vector<tbrm_dir_light_params> l;
        for (const FDirLightParameters& p : Lights) l.push_back(p.abi());
        const tbrm_world_params w = WorldParameters.abi();
        std::vector<int32_t> schedule(8 * l.size() + 8);
        int32_t entries = 0;
        LightsAdded = tbrm_add_dir_lights(Resources.Handle, l.data(), (int32_t) l.size(), Added ? 1 : 0, &w, schedule.data(), &entries) == TBRM_OK;
        int paired = 0;
        for (int32_t e = 0; e < entries; ++e) paired += schedule[4 * e + 2] >= 0 ? 2 : 0;
        return paired;
    }
    static void ChangeDirLightInSingleVolume(FBasicRaymarchRenderingResources& Resources, const FDirLightParameters OldLightParameters,
                                             const FDirLightParameters NewLightParameters, const FRaymarchWorldParameters WorldParameters,
                                             bool& LightAdded, bool bGPUSync = false)
    {
        const tbrm_dir_light_params o = OldLightParameters.abi(), n = NewLightParameters.abi();
        const tbrm_world_params w = WorldParameters.abi();
        int flag = 0;
        tbrm_change_dir_light(Resources.Handle, &o, &n, &w, &flag, bGPUSync ? 1 : 0);
        LightAdded = flag != 0;
    }
    // RaymarchUtils.cpp:94-102
    static bool GenerateOctree(FBasicRaymarchRenderingResources& Resources)
    {
        return Resources.Handle && tbrm_generate_octree(Resources.Handle) == TBRM_OK;
    }

    static void ClearResourceLightVolumes(FBasicRaymarchRenderingResources Resources, float ClearValue)
    {
        if (!Resources.Handle) return;
        tbrm_clear_light_volume(Resources.Handle, ClearValue);
    }
    // ColorCurveToTexture / MakeDefaultTFTexture produce the 256-sample LUT the handle stores as FFloat16.
    static void ColorCurveToTexture(const FColorCurve& Curve, std::vector<float>& OutTexture)
    {
        OutTexture.assign(256 * 4, 0.0f);
        const float* t[4] = {Curve.Times[0].data(), Curve.Times[1].data(), Curve.Times[2].data(), Curve.Times[3].data()};
        const float* v[4] = {Curve.Values[0].data(), Curve.Values[1].data(), Curve.Values[2].data(), Curve.Values[3].data()};
        const int32_t n[4] = {(int32_t) Curve.Times[0].size(), (int32_t) Curve.Times[1].size(), (int32_t) Curve.Times[2].size(), (int32_t) Curve.Times[3].size()};
        tbrm_color_curve_to_lut(t, v, n, OutTexture.data());
    }
    static void MakeDefaultTFTexture(std::vector<float>& OutTexture)
    {
        OutTexture.assign(256 * 4, 0.0f);
        tbrm_make_default_tf_lut(OutTexture.data());
    }
    // CreateBufferTextures / ReleaseOneAxisReadWriteBufferResources (RaymarchUtils.cpp:176-217): the four read/write
    // buffers per axis belong to the tbrm handle (created with it, released with it: tbrm_resources_create / _destroy), so
    // a host has nothing to create or release; the names are kept so that call sites compile unchanged.
    struct OneAxisReadWriteBufferResources {}; // RaymarchTypes.h:75-81
    static void CreateBufferTextures(int /*SizeX*/, int /*SizeY*/, int /*PixelFormat*/, OneAxisReadWriteBufferResources& /*RWBuffers*/) {}
    static void ReleaseOneAxisReadWriteBufferResources(OneAxisReadWriteBufferResources& /*Buffer*/) {}

    // The Blueprint-pure helpers (RaymarchUtils.cpp:219-252)
    static void GetVolumeTextureDimensions(const FBasicRaymarchRenderingResources* Resources, int32_t Dimensions[3])
    {
        Dimensions[0] = Resources ? Resources->SizeX : 0;
        Dimensions[1] = Resources ? Resources->SizeY : 0;
        Dimensions[2] = Resources ? Resources->SizeZ : 0;
    }
    // FTransform::ToMatrixWithScale / ToMatrixNoScale: row-vector convention, rows 0-2 = scaled rotation axes, row 3 = translation
    static void TransformToMatrix(const FTransform& Transform, double OutMatrix[4][4], bool WithScaling)
    {
        const FQuat& q = Transform.rotation;
        const double sx = WithScaling ? Transform.scale3d.x : 1.0, sy = WithScaling ? Transform.scale3d.y : 1.0, sz = WithScaling ? Transform.scale3d.z : 1.0;
        const double x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        const double xx = q.x * x2, yy = q.y * y2, zz = q.z * z2, xy = q.x * y2, xz = q.x * z2, yz = q.y * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        const double m[4][4] = {{(1.0 - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx, 0.0},
                                {(xy - wz) * sy, (1.0 - (xx + zz)) * sy, (yz + wx) * sy, 0.0},
                                {(xz + wy) * sz, (yz - wx) * sz, (1.0 - (xx + yy)) * sz, 0.0},
                                {Transform.translation.x, Transform.translation.y, Transform.translation.z, 1.0}};
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) OutMatrix[r][c] = m[r][c];
    }
    static void LocalToTextureCoords(FVector LocalCoords, FVector& TextureCoords) // local cube [-1,1]^3 -> texture [0,1]^3
    {
        TextureCoords = FVector{LocalCoords.x / 2.0 + 0.5, LocalCoords.y / 2.0 + 0.5, LocalCoords.z / 2.0 + 0.5};
    }
    static void TextureToLocalCoords(FVector TextureCoords, FVector& LocalCoords)
    {
        LocalCoords = FVector{(TextureCoords.x - 0.5) * 2.0, (TextureCoords.y - 0.5) * 2.0, (TextureCoords.z - 0.5) * 2.0};
    }
};

struct ARaymarchLight { // direction = actor forward vector (RaymarchLight.cpp:28-31)
    FVector ForwardVector{1, 0, 0};
    float LightIntensity = 1.0f;
    std::string Name = "RaymarchLight";
    FDirLightParameters GetCurrentParameters() const { return FDirLightParameters(ForwardVector, LightIntensity); }
};

struct ARaymarchClipPlane { // (location, -up) (RaymarchClipPlane.cpp:32-35)
    FVector Location{0, 0, 0};
    FVector UpVector{0, 0, 1};
    FClippingPlaneParameters GetCurrentParameters() const { return FClippingPlaneParameters(Location, FVector{-UpVector.x, -UpVector.y, -UpVector.z}); }
};

class ARaymarchVolume {
public:
    // ---- the reflected properties the hot path consumes (RaymarchVolume.h:60-266) ----
    FBasicRaymarchRenderingResources RaymarchResources;
    FRaymarchWorldParameters WorldParameters;
    std::vector<ARaymarchLight*> LightsArray;
    ARaymarchClipPlane* ClippingPlane = nullptr;
    FTransform ComponentTransform{{0, 0, 0, 1}, {0, 0, 0}, {100, 100, 100}}; // the cube mesh component (RaymarchVolume.cpp:47)
    bool bLightVolume32Bit = false;
    bool bFastShader = true; // accepted and ignored: the reference's GPUSync branch is a no-op (RaymarchUtils.cpp:51-59)
    bool bRequestedRecompute = false;
    bool bVisible = true;
    float RaymarchingSteps = 150;
    ERaymarchMaterial SelectRaymarchMaterial = ERaymarchMaterial::Lit;
    int Device = 0;
    int DataAddressMode = TBRM_ADDRESS_WRAP;
    bool bRecordLightsOnReset = false; // see ResetAllLights
    bool bBatchLightsOnReset = false;  // ResetAllLights adds all lights with one batched call (UNORM8 result may differ by one code at fp32 ties)

    struct FStats { int Resets = 0, LightAdds = 0, LightChanges = 0, Frames = 0; } Stats; // what Tick decided (test hook)

    ~ARaymarchVolume() { FreeRaymarchResources(); }

    // SetVolumeAsset (RaymarchVolume.cpp:467-560): UVolumeTexture-shaped buffer in, resources (re)initialised
    bool SetVolumeAsset(const void* Voxels, int SizeX, int SizeY, int SizeZ, int Format)
    {
        InitializeRaymarchResources(SizeX, SizeY, SizeZ, Format);
        if (!RaymarchResources.Handle) return false;
        const size_t bytes = (size_t) SizeX * SizeY * SizeZ * (Format == TBRM_FMT_G8 ? 1 : (Format == TBRM_FMT_G16 ? 2 : 4));
        if (tbrm_upload_volume(RaymarchResources.Handle, Voxels, bytes) != TBRM_OK) return false;
        bRequestedOctreeRebuild = true; // :554
        if (!bHasTF) {
            std::vector<float> lut;
            URaymarchUtils::MakeDefaultTFTexture(lut);
            tbrm_set_tf_lut(RaymarchResources.Handle, lut.data());
            bHasTF = true;
        }
        tbrm_windowing_params w = RaymarchResources.WindowingParameters.abi();
        tbrm_set_windowing(RaymarchResources.Handle, &w);
        RaymarchResources.bIsInitialized = tbrm_resources_is_initialized(RaymarchResources.Handle) != 0;
        // OnConstruction (RaymarchVolume.cpp:161-173): remember the lights' parameters
        LightParametersMap.clear();
        for (ARaymarchLight* Light : LightsArray)
            if (Light && Light->LightIntensity > 0.0f) LightParametersMap[Light] = Light->GetCurrentParameters();
        bRequestedRecompute = true; // :552
        return RaymarchResources.bIsInitialized;
    }

    void SetTFCurve(const FColorCurve& Curve) // :562-577
    {
        if (!RaymarchResources.Handle) return;
        std::vector<float> lut;
        URaymarchUtils::ColorCurveToTexture(Curve, lut);
        tbrm_set_tf_lut(RaymarchResources.Handle, lut.data());
        bHasTF = true;
        bRequestedRecompute = true;
    }

    // :746-784 — every windowing change requests a full recompute
    void SetWindowCenter(float Center) { if (Center != RaymarchResources.WindowingParameters.Center) { RaymarchResources.WindowingParameters.Center = Center; WindowingChanged(); } }
    void SetWindowWidth(float Width) { if (Width != RaymarchResources.WindowingParameters.Width) { RaymarchResources.WindowingParameters.Width = Width; WindowingChanged(); } }
    void SetLowCutoff(bool Cutoff) { if (Cutoff != RaymarchResources.WindowingParameters.LowCutoff) { RaymarchResources.WindowingParameters.LowCutoff = Cutoff; WindowingChanged(); } }
    void SetHighCutoff(bool Cutoff) { if (Cutoff != RaymarchResources.WindowingParameters.HighCutoff) { RaymarchResources.WindowingParameters.HighCutoff = Cutoff; WindowingChanged(); } }
    void SetRaymarchSteps(float InSteps) { RaymarchingSteps = InSteps; } // :802-819
    void SwitchRenderer(ERaymarchMaterial InSelectRaymarchMaterial) // :786-800, :304-307
    {
        SelectRaymarchMaterial = InSelectRaymarchMaterial;
        if (InSelectRaymarchMaterial == ERaymarchMaterial::Lit) bRequestedRecompute = true;
        if (InSelectRaymarchMaterial == ERaymarchMaterial::Octree) bRequestedOctreeRebuild = true;
    }

    FRaymarchWorldParameters GetWorldParameters() const // :630-646
    {
        FRaymarchWorldParameters r;
        if (ClippingPlane) r.ClippingPlaneParameters = ClippingPlane->GetCurrentParameters();
        else { r.ClippingPlaneParameters.Center = FVector{0, 0, 100000}; r.ClippingPlaneParameters.Direction = FVector{0, 0, -1}; }
        r.VolumeTransform = ComponentTransform;
        return r;
    }

    void Tick(float /*DeltaTime*/) // :327-416
    {
        if (!RaymarchResources.bIsInitialized || !bVisible) return;
        if (WorldParameters != GetWorldParameters()) { // volume transform changed or clipping plane moved
            bRequestedRecompute = true;
            WorldParameters = GetWorldParameters();
        }
        if (bRequestedOctreeRebuild && SelectRaymarchMaterial == ERaymarchMaterial::Octree) { // :358-363
            URaymarchUtils::GenerateOctree(RaymarchResources);
            bRequestedOctreeRebuild = false; // also when generation failed, as in the reference: no retry every tick
        }
        if (SelectRaymarchMaterial != ERaymarchMaterial::Lit) return;
        if (bRequestedRecompute) { ResetAllLights(); return; }
        std::vector<ARaymarchLight*> LightsToUpdate;
        for (ARaymarchLight* Light : LightsArray) {
            if (!Light) continue;
            auto it = LightParametersMap.find(Light);
            if (it == LightParametersMap.end()) { LightParametersMap[Light] = Light->GetCurrentParameters(); LightsToUpdate.push_back(Light); }
            else if (Light->GetCurrentParameters() != it->second) LightsToUpdate.push_back(Light);
        }
        // more than half of the lights need an update -> a full reset is quicker (:400-404)
        if (LightsToUpdate.size() > 1 && LightsToUpdate.size() >= LightsArray.size() / 2) ResetAllLights();
        else
            for (ARaymarchLight* UpdatedLight : LightsToUpdate) {
                UpdateSingleLight(UpdatedLight);
                LightParametersMap[UpdatedLight] = UpdatedLight->GetCurrentParameters();
            }
    }

    void ResetAllLights() // :418-451
    {
        if (!RaymarchResources.bIsInitialized) return;
        ReserveForLights(); // (lights added since the resources were initialised)
        URaymarchUtils::ClearResourceLightVolumes(RaymarchResources, 0);
        ++Stats.Resets;
        bool bResetWasSuccessful = true;
        if (bBatchLightsOnReset) {
            std::vector<FDirLightParameters> All;
            for (ARaymarchLight* Light : LightsArray)
                if (Light) All.push_back(Light->GetCurrentParameters());
            URaymarchUtils::AddDirLightsToSingleVolume(RaymarchResources, All, true, WorldParameters, bResetWasSuccessful);
            Stats.LightAdds += (int) All.size();
            if (!bResetWasSuccessful) { std::fprintf(stderr, "Error. Could not add the lights.\n"); return; }
            if (bRecordLightsOnReset)
                for (ARaymarchLight* Light : LightsArray)
                    if (Light) LightParametersMap[Light] = Light->GetCurrentParameters();
            bRequestedRecompute = false;
            return;
        }
        for (ARaymarchLight* Light : LightsArray) {
            if (!Light) continue;
            URaymarchUtils::AddDirLightToSingleVolume(RaymarchResources, Light->GetCurrentParameters(), true, WorldParameters, bResetWasSuccessful, bFastShader);
            ++Stats.LightAdds;
            if (!bResetWasSuccessful) { std::fprintf(stderr, "Error. Could not add/remove light %s.\n", Light->Name.c_str()); return; }
            // The reference leaves LightParametersMap untouched here (:418-451): a light that moved before a reset is
            // "changed" again on the next tick, from parameters that were never added to the volume. That behaviour is
            // kept by default; bRecordLightsOnReset makes the map follow the light volume's actual content instead.
            if (bRecordLightsOnReset) LightParametersMap[Light] = Light->GetCurrentParameters();
        }
        bRequestedRecompute = false;
    }

    void UpdateSingleLight(ARaymarchLight* UpdatedLight) // :453-465
    {
        bool bLightAddWasSuccessful = false;
        URaymarchUtils::ChangeDirLightInSingleVolume(RaymarchResources, LightParametersMap[UpdatedLight], UpdatedLight->GetCurrentParameters(), WorldParameters, bLightAddWasSuccessful);
        ++Stats.LightChanges;
        if (!bLightAddWasSuccessful) std::fprintf(stderr, "Error. Could not change light %s.\n", UpdatedLight->Name.c_str());
    }

    // Offscreen replacement of the M_Raymarch material pass: premultiplied RGBA float, Camera.width x Camera.height.
    bool RenderLit(const tbrm_camera& Camera, float* OutRGBA, int JitterFrame = -1, bool bSkipEmptySpace = true)
    {
        if (!RaymarchResources.bIsInitialized) return false;
        const tbrm_tile tile{0, 0, Camera.width, Camera.height, 1, 0};
        const tbrm_raymarch_params rp{RaymarchingSteps, JitterFrame, bSkipEmptySpace ? 1 : 0, 0};
        const tbrm_world_params w = WorldParameters.abi();
        ++Stats.Frames;
        return tbrm_raymarch_lit(RaymarchResources.Handle, &Camera, &tile, &rp, &w, OutRGBA) == TBRM_OK;
    }

    // Offscreen replacement of the M_Intensity_Raymarch material pass (the slice view, RaymarchVolume.cpp:72-73,:122-128).
    bool RenderIntensity(const tbrm_camera& Camera, float* OutRGBA, int JitterFrame = -1)
    {
        if (!RaymarchResources.Handle) return false;
        const tbrm_tile tile{0, 0, Camera.width, Camera.height, 1, 0};
        const tbrm_raymarch_params rp{RaymarchingSteps, JitterFrame, 0, 0};
        const tbrm_world_params w = WorldParameters.abi();
        ++Stats.Frames;
        return tbrm_raymarch_intensity(RaymarchResources.Handle, &Camera, &tile, &rp, &w, OutRGBA) == TBRM_OK;
    }

    // Offscreen replacement of the M_Octree_Raymarch material pass over level OctreeVolumeMip. Tick rebuilds the pyramid
    // when a rebuild was requested (new volume, SwitchRenderer(Octree); :358-363); a host that renders without ticking
    // gets the pending rebuild here.
    int OctreeVolumeMip = 0; // RaymarchVolume.h: the level the octree material samples
    bool bRequestedOctreeRebuild = true;
    bool RenderOctree(const tbrm_camera& Camera, float* OutRGBA, int JitterFrame = -1)
    {
        if (!RaymarchResources.Handle) return false;
        if (bRequestedOctreeRebuild) {
            if (!URaymarchUtils::GenerateOctree(RaymarchResources)) return false;
            bRequestedOctreeRebuild = false;
        }
        const tbrm_tile tile{0, 0, Camera.width, Camera.height, 1, 0};
        const tbrm_raymarch_params rp{RaymarchingSteps, JitterFrame, 0, 0};
        const tbrm_world_params w = WorldParameters.abi();
        ++Stats.Frames;
        return tbrm_raymarch_octree(RaymarchResources.Handle, &Camera, &tile, &rp, &w, OctreeVolumeMip, OutRGBA) == TBRM_OK;
    }

    // What the cube mesh would show with the currently selected material (SwitchRenderer, :786-800).
    bool Render(const tbrm_camera& Camera, float* OutRGBA, int JitterFrame = -1)
    {
        switch (SelectRaymarchMaterial) {
            case ERaymarchMaterial::Intensity: return RenderIntensity(Camera, OutRGBA, JitterFrame);
            case ERaymarchMaterial::Octree: return RenderOctree(Camera, OutRGBA, JitterFrame);
            default: return RenderLit(Camera, OutRGBA, JitterFrame);
        }
    }

    void FreeRaymarchResources() // :922-949
    {
        if (RaymarchResources.Handle) tbrm_resources_destroy(RaymarchResources.Handle);
        RaymarchResources.Handle = nullptr;
        RaymarchResources.bIsInitialized = false;
        bHasTF = false;
    }

private:
    std::map<ARaymarchLight*, FDirLightParameters> LightParametersMap;
    bool bHasTF = false;

    void InitializeRaymarchResources(int X, int Y, int Z, int Format) // :821-920
    {
        if (RaymarchResources.Handle) FreeRaymarchResources();
        tbrm_resources_desc d{};
        d.dim_x = X; d.dim_y = Y; d.dim_z = Z;
        d.data_format = Format;
        d.light_volume_32bit = bLightVolume32Bit ? 1 : 0;
        d.light_volume_half_resolution = RaymarchResources.LightVolumeHalfResolution ? 1 : 0;
        d.device = Device;
        d.data_address_mode = DataAddressMode;
        d.border_mode = TBRM_BORDER_ENGINE_8BIT;
        if (tbrm_resources_create(&d, &RaymarchResources.Handle) != TBRM_OK) {
            std::fprintf(stderr, "Tried to initialize Raymarch resources: %s\n", tbrm_last_error());
            RaymarchResources.Handle = nullptr;
            return;
        }
        RaymarchResources.SizeX = X; RaymarchResources.SizeY = Y; RaymarchResources.SizeZ = Z;
        ReservedLights = 0;
        ReserveForLights();
    }
    // Every buffer the light operators will need, now (:821-920 creates the read / write buffers and the light volume here, never
    // inside AddDirLightToSingleVolume): tbrm_resources_reserve for the lights the actor holds, four at least; again when more arrive.
    int ReservedLights = 0;
    void ReserveForLights()
    {
        if (!RaymarchResources.Handle) return;
        const int Want = std::max<int>(4, (int) LightsArray.size());
        if (Want <= ReservedLights) return;
        if (tbrm_resources_reserve(RaymarchResources.Handle, Want, 0) == TBRM_OK) ReservedLights = Want;
        else std::fprintf(stderr, "tbrm_resources_reserve: %s\n", tbrm_last_error());
    }
    void WindowingChanged()
    {
        if (RaymarchResources.Handle) { tbrm_windowing_params w = RaymarchResources.WindowingParameters.abi(); tbrm_set_windowing(RaymarchResources.Handle, &w); }
        bRequestedRecompute = true;
    }
};
using URaymarchVolume = ARaymarchVolume;

} // namespace tbrm_plugin
