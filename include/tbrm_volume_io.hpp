// tbrm_volume_io.hpp — the step before the hot path: MetaImage (.mhd + .raw / zlib-compressed .zraw) volumes in, the
// voxel array and the value range the windowing parameters are expressed in out. Header-only host C++ with the
// reference's names; link with -lz.
//
//   FVolumeInfo                      VolumeTextureToolkit/Public/VolumeAsset/VolumeInfo.h, Private/.../VolumeInfo.cpp:8-101
//   UMHDLoader::ParseVolumeInfoFromHeader   Private/VolumeAsset/Loaders/MHDLoader.cpp:18-181
//   IVolumeLoader::LoadRawDataFileFromInfo / ConvertData   Private/VolumeAsset/Loaders/VolumeLoader.cpp:16-128
//   ConvertArrayToNormalizedArray / ConvertArrayToFloat    Public/TextureUtilities.h:103-165, Private/TextureUtilities.cpp:304-350
//   ARaymarchVolume::LoadMHDFileIntoVolume{Normalized,TransientR32F}   Raymarcher/Private/Actor/RaymarchVolume.cpp:596-628
//
// Behaviour kept on purpose: the header is read as whitespace-separated words ("DimSize = 64 64 32": "DimSize=64" is not
// found); normalisation maps [min, max] of the file onto the full range of uint8 (8-bit inputs) or uint16 (everything
// else) with float arithmetic and truncation; for float inputs the running maximum starts at FLT_MIN (the smallest
// positive float, std::numeric_limits<float>::min(), TextureUtilities.h:113), so an all-negative float file keeps
// MaxValue = FLT_MIN. What the engine leaves undefined is pinned: a constant file (max == min) normalises to 0.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "tbrm_plugin.hpp"

namespace tbrm_plugin {

enum class EVolumeVoxelFormat { UnsignedChar, SignedChar, UnsignedShort, SignedShort, UnsignedInt, SignedInt, Float };

struct FVolumeInfo {
    bool bParseWasSuccessful = false;
    std::string DataFileName;
    EVolumeVoxelFormat OriginalFormat = EVolumeVoxelFormat::UnsignedChar;
    EVolumeVoxelFormat ActualFormat = EVolumeVoxelFormat::UnsignedChar;
    int Dimensions[3] = {0, 0, 0};
    double Spacing[3] = {0, 0, 0};
    double WorldDimensions[3] = {0, 0, 0};
    int BytesPerVoxel = 1;
    bool bIsSigned = false;
    bool bIsCompressed = false;
    bool bIsNormalized = false;
    long long CompressedByteSize = 0;
    float MinValue = 0.0f, MaxValue = 1.0f; // of the file, before normalisation: the unit of window centre / width

    long long GetTotalVoxels() const { return (long long) Dimensions[0] * Dimensions[1] * Dimensions[2]; }
    long long GetByteSize() const { return GetTotalVoxels() * BytesPerVoxel; }

    // window centre / width in file units <-> the normalised units the kernels see (VolumeInfo.cpp:18-55)
    float NormalizeValue(float v) const { return bIsNormalized ? (v - MinValue) / (MaxValue - MinValue) : v; }
    float DenormalizeValue(float v) const { return bIsNormalized ? (v * (MaxValue - MinValue)) + MinValue : v; }
    float NormalizeRange(float r) const { return bIsNormalized ? r / (MaxValue - MinValue) : r; }
    float DenormalizeRange(float r) const { return bIsNormalized ? r * (MaxValue - MinValue) : r; }

    static int VoxelFormatByteSize(EVolumeVoxelFormat f)
    {
        switch (f) {
            case EVolumeVoxelFormat::UnsignedChar: case EVolumeVoxelFormat::SignedChar: return 1;
            case EVolumeVoxelFormat::UnsignedShort: case EVolumeVoxelFormat::SignedShort: return 2;
            default: return 4;
        }
    }
    static bool IsVoxelFormatSigned(EVolumeVoxelFormat f)
    {
        return !(f == EVolumeVoxelFormat::UnsignedChar || f == EVolumeVoxelFormat::UnsignedShort || f == EVolumeVoxelFormat::UnsignedInt);
    }
    // VoxelFormatToPixelFormat (VolumeInfo.cpp:103-127) for the formats the raymarcher consumes; -1: no such data volume
    int TbrmFormat() const
    {
        switch (ActualFormat) {
            case EVolumeVoxelFormat::UnsignedChar: case EVolumeVoxelFormat::SignedChar: return TBRM_FMT_G8;
            case EVolumeVoxelFormat::UnsignedShort: case EVolumeVoxelFormat::SignedShort: return TBRM_FMT_G16;
            case EVolumeVoxelFormat::Float: return TBRM_FMT_R32_FLOAT;
            default: return -1; // PF_R32_SINT/UINT: "experimental" in the reference, not a raymarcher input
        }
    }
};

// How NormalizeArray seeds its running maximum. The reference starts it at std::numeric_limits<T>::min()
// (TextureUtilities.h:110) — for the integer formats the lowest value, for MET_FLOAT the smallest POSITIVE one, so a float
// volume without a positive voxel reports a maximum of 1.18e-38 instead of its own. true (the default): exactly that, results
// identical to the reference's on every input; false: lowest() for every type, which is what was meant. Process-wide, like the
// reference's behaviour.
inline bool& NormalizeSeedsMaximumLikeTheReference()
{
    static bool value = true;
    return value;
}

namespace detail {

// [min, max] of the array onto the full range of Out, with the reference's float arithmetic (TextureUtilities.h:103-149)
template <typename In, typename Out>
inline void NormalizeArray(const uint8_t* bytes, long long byte_size, std::vector<uint8_t>& out, float& out_min, float& out_max)
{
    const long long n = byte_size / (long long) sizeof(In);
    In lo = std::numeric_limits<In>::max();
    In hi = NormalizeSeedsMaximumLikeTheReference() ? std::numeric_limits<In>::min() : std::numeric_limits<In>::lowest(); // (TextureUtilities.h:110)
    for (long long i = 0; i < n; ++i) {
        In v;
        std::memcpy(&v, bytes + i * sizeof(In), sizeof(In));
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    out.resize((size_t) n * sizeof(Out));
    const Out out_lo = std::numeric_limits<Out>::min(), out_hi = std::numeric_limits<Out>::max();
    const float span = (float) hi - lo;
    for (long long i = 0; i < n; ++i) {
        In v;
        std::memcpy(&v, bytes + i * sizeof(In), sizeof(In));
        const float normalized = ((float) v - lo) / span;
        const float scaled = out_lo + (normalized * (out_hi - out_lo));
        const Out o = (scaled == scaled) ? (Out) scaled : (Out) 0; // max == min: 0/0 in the reference, 0 here
        std::memcpy(out.data() + i * sizeof(Out), &o, sizeof(Out));
    }
    out_min = (float) lo;
    out_max = (float) hi;
}

template <typename In>
inline void ToFloatArray(const uint8_t* bytes, long long voxels, std::vector<uint8_t>& out)
{
    out.resize((size_t) voxels * sizeof(float));
    for (long long i = 0; i < voxels; ++i) {
        In v;
        std::memcpy(&v, bytes + i * sizeof(In), sizeof(In));
        const float f = (float) v;
        std::memcpy(out.data() + i * sizeof(float), &f, sizeof(float));
    }
}

inline bool ReadFile(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    f.seekg(0);
    out.resize((size_t) n);
    return n == 0 || (bool) f.read((char*) out.data(), n);
}

inline std::string DirName(const std::string& path)
{
    const size_t k = path.find_last_of("/\\");
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

} // namespace detail

struct UMHDLoader {
    // MHDLoader.cpp:18-181
    static FVolumeInfo ParseVolumeInfoFromHeader(const std::string& FileName)
    {
        FVolumeInfo info;
        std::vector<uint8_t> raw;
        if (!detail::ReadFile(FileName, raw)) return info;
        const std::string text((const char*) raw.data(), raw.size());
        // positions the stream after `key` and its "=" (both whitespace-delimited words); false when the key is missing
        auto seek = [&](std::istringstream& in, const char* key, const char* alt = nullptr) {
            in.clear();
            in.str(text);
            std::string w;
            while (in >> w)
                if (w == key || (alt && w == alt)) return (bool) (in >> w);
            return false;
        };
        std::istringstream in;
        if (!seek(in, "DimSize") || !(in >> info.Dimensions[0] >> info.Dimensions[1] >> info.Dimensions[2])) return info;
        if (!seek(in, "ElementSpacing", "ElementSize") || !(in >> info.Spacing[0] >> info.Spacing[1] >> info.Spacing[2])) return info;
        for (int c = 0; c < 3; ++c) info.WorldDimensions[c] = info.Spacing[c] * info.Dimensions[c];
        std::string type;
        if (!seek(in, "ElementType") || !(in >> type)) return info;
        if (type == "MET_UCHAR") info.OriginalFormat = EVolumeVoxelFormat::UnsignedChar;
        else if (type == "MET_CHAR") info.OriginalFormat = EVolumeVoxelFormat::SignedChar;
        else if (type == "MET_USHORT") info.OriginalFormat = EVolumeVoxelFormat::UnsignedShort;
        else if (type == "MET_SHORT") info.OriginalFormat = EVolumeVoxelFormat::SignedShort;
        else if (type == "MET_UINT") info.OriginalFormat = EVolumeVoxelFormat::UnsignedInt;
        else if (type == "MET_INT") info.OriginalFormat = EVolumeVoxelFormat::SignedInt;
        else if (type == "MET_FLOAT") info.OriginalFormat = EVolumeVoxelFormat::Float;
        else return info;
        info.ActualFormat = info.OriginalFormat;
        info.BytesPerVoxel = FVolumeInfo::VoxelFormatByteSize(info.OriginalFormat);
        info.bIsSigned = FVolumeInfo::IsVoxelFormatSigned(info.OriginalFormat);
        if (seek(in, "CompressedDataSize")) {
            info.bIsCompressed = true;
            in >> info.CompressedByteSize;
        }
        if (!seek(in, "ElementDataFile") || !(in >> info.DataFileName)) return info;
        info.bParseWasSuccessful = info.Dimensions[0] > 0 && info.Dimensions[1] > 0 && info.Dimensions[2] > 0;
        return info;
    }

    // IVolumeLoader::LoadRawDataFileFromInfo (VolumeLoader.cpp:16-29): the data file next to the header, inflated if
    // the header carries CompressedDataSize (LoadZLibCompressedFileIntoArray, TextureUtilities.cpp:261-302)
    static bool LoadRawDataFileFromInfo(const std::string& FilePath, const FVolumeInfo& Info, std::vector<uint8_t>& Out)
    {
        std::vector<uint8_t> file;
        if (!detail::ReadFile(FilePath + "/" + Info.DataFileName, file)) return false;
        const long long want = Info.GetByteSize();
        if (!Info.bIsCompressed) {
            if ((long long) file.size() < want) return false; // "smaller than expected, cannot read volume"
            file.resize((size_t) want);                       // a larger file is a warning in the reference
            Out.swap(file);
            return true;
        }
        if ((long long) file.size() < Info.CompressedByteSize) return false;
        Out.resize((size_t) want);
        uLongf got = (uLongf) want;
        return uncompress(Out.data(), &got, file.data(), (uLong) Info.CompressedByteSize) == Z_OK && (long long) got == want;
    }

    // IVolumeLoader::ConvertData (VolumeLoader.cpp:97-128)
    static bool ConvertData(std::vector<uint8_t>& Voxels, FVolumeInfo& Info, bool bNormalize, bool bConvertToFloat)
    {
        std::vector<uint8_t> out;
        Info.bIsNormalized = bNormalize;
        const uint8_t* in = Voxels.data();
        if (bNormalize) {
            const long long bytes = Info.GetByteSize();
            switch (Info.OriginalFormat) { // NormalizeArrayByFormat (TextureUtilities.cpp:304-327)
                case EVolumeVoxelFormat::UnsignedChar: detail::NormalizeArray<uint8_t, uint8_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::SignedChar: detail::NormalizeArray<int8_t, uint8_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::UnsignedShort: detail::NormalizeArray<uint16_t, uint16_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::SignedShort: detail::NormalizeArray<int16_t, uint16_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::UnsignedInt: detail::NormalizeArray<uint32_t, uint16_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::SignedInt: detail::NormalizeArray<int32_t, uint16_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
                case EVolumeVoxelFormat::Float: detail::NormalizeArray<float, uint16_t>(in, bytes, out, Info.MinValue, Info.MaxValue); break;
            }
            if (Info.BytesPerVoxel > 1) { Info.BytesPerVoxel = 2; Info.ActualFormat = EVolumeVoxelFormat::UnsignedShort; }
            else Info.ActualFormat = EVolumeVoxelFormat::UnsignedChar;
            Voxels.swap(out);
        } else if (bConvertToFloat && Info.OriginalFormat != EVolumeVoxelFormat::Float) {
            const long long n = Info.GetTotalVoxels();
            switch (Info.OriginalFormat) { // ConvertArrayToFloat (TextureUtilities.cpp:329-350)
                case EVolumeVoxelFormat::UnsignedChar: detail::ToFloatArray<uint8_t>(in, n, out); break;
                case EVolumeVoxelFormat::SignedChar: detail::ToFloatArray<int8_t>(in, n, out); break;
                case EVolumeVoxelFormat::UnsignedShort: detail::ToFloatArray<uint16_t>(in, n, out); break;
                case EVolumeVoxelFormat::SignedShort: detail::ToFloatArray<int16_t>(in, n, out); break;
                case EVolumeVoxelFormat::UnsignedInt: detail::ToFloatArray<uint32_t>(in, n, out); break;
                case EVolumeVoxelFormat::SignedInt: detail::ToFloatArray<int32_t>(in, n, out); break;
                default: return false;
            }
            Info.BytesPerVoxel = 4;
            Info.ActualFormat = EVolumeVoxelFormat::Float;
            Voxels.swap(out);
        } else {
            Info.ActualFormat = Info.OriginalFormat;
        }
        return true;
    }

    // CreateVolumeFromFile (MHDLoader.cpp:183-222) without the engine's asset objects: header + voxels
    static bool LoadVolume(const std::string& FileName, bool bNormalize, bool bConvertToFloat, FVolumeInfo& OutInfo, std::vector<uint8_t>& OutVoxels)
    {
        OutInfo = ParseVolumeInfoFromHeader(FileName);
        if (!OutInfo.bParseWasSuccessful) return false;
        if (!LoadRawDataFileFromInfo(detail::DirName(FileName), OutInfo, OutVoxels)) return false;
        return ConvertData(OutVoxels, OutInfo, bNormalize, bConvertToFloat);
    }
};

// ARaymarchVolume::LoadMHDFileIntoVolumeNormalized / ...TransientR32F (RaymarchVolume.cpp:596-628): file -> SetVolumeAsset.
// OutInfo (optional) receives the header and the value range: window centre / width given in file units go through
// FVolumeInfo::NormalizeValue / NormalizeRange before ARaymarchVolume::SetWindowCenter / SetWindowWidth
// (TransferFuncMenu.cpp:68,:76).
inline bool LoadMHDFileIntoVolume(ARaymarchVolume& Volume, const std::string& FileName, bool bNormalize, bool bConvertToFloat, FVolumeInfo* OutInfo)
{
    FVolumeInfo info;
    std::vector<uint8_t> voxels;
    if (!UMHDLoader::LoadVolume(FileName, bNormalize, bConvertToFloat, info, voxels)) return false;
    if (OutInfo) *OutInfo = info;
    const int fmt = info.TbrmFormat();
    if (fmt < 0) return false;
    return Volume.SetVolumeAsset(voxels.data(), info.Dimensions[0], info.Dimensions[1], info.Dimensions[2], fmt);
}
inline bool LoadMHDFileIntoVolumeNormalized(ARaymarchVolume& Volume, const std::string& FileName, FVolumeInfo* OutInfo = nullptr)
{
    return LoadMHDFileIntoVolume(Volume, FileName, true, false, OutInfo);
}
inline bool LoadMHDFileIntoVolumeTransientR32F(ARaymarchVolume& Volume, const std::string& FileName, FVolumeInfo* OutInfo = nullptr)
{
    return LoadMHDFileIntoVolume(Volume, FileName, false, true, OutInfo);
}

} // namespace tbrm_plugin
