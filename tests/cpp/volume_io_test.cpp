// Driver for tests/test_volume_io.py: loads one .mhd with include/tbrm_volume_io.hpp and dumps what it read.
//   volume_io_test <file.mhd> <normalize 0|1> <to_float 0|1> <out.bin> [<maximum seeded like the reference 0|1, default 1>]
// stdout: key=value lines; out.bin: the converted voxel array.
#include <cstdio>
#include <cstdlib>

#include "tbrm_volume_io.hpp"

using namespace tbrm_plugin;

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    FVolumeInfo info;
    std::vector<uint8_t> voxels;
    if (argc > 5) NormalizeSeedsMaximumLikeTheReference() = std::atoi(argv[5]) != 0;
    const bool ok = UMHDLoader::LoadVolume(argv[1], std::atoi(argv[2]) != 0, std::atoi(argv[3]) != 0, info, voxels);
    std::printf("ok=%d\n", ok ? 1 : 0);
    std::printf("parsed=%d\n", info.bParseWasSuccessful ? 1 : 0);
    if (!info.bParseWasSuccessful) return 0;
    std::printf("dims=%d %d %d\n", info.Dimensions[0], info.Dimensions[1], info.Dimensions[2]);
    std::printf("spacing=%.9g %.9g %.9g\n", info.Spacing[0], info.Spacing[1], info.Spacing[2]);
    std::printf("world=%.9g %.9g %.9g\n", info.WorldDimensions[0], info.WorldDimensions[1], info.WorldDimensions[2]);
    std::printf("original_format=%d actual_format=%d bytes_per_voxel=%d signed=%d compressed=%d compressed_size=%lld normalized=%d\n",
                (int) info.OriginalFormat, (int) info.ActualFormat, info.BytesPerVoxel, info.bIsSigned ? 1 : 0, info.bIsCompressed ? 1 : 0,
                info.CompressedByteSize, info.bIsNormalized ? 1 : 0);
    std::printf("data_file=%s\n", info.DataFileName.c_str());
    std::printf("tbrm_format=%d\n", info.TbrmFormat());
    if (!ok) return 0;
    std::printf("min=%.9g max=%.9g\n", info.MinValue, info.MaxValue);
    std::printf("normalize_value_of_mid=%.9g normalize_range_of_span=%.9g\n",
                info.NormalizeValue(0.5f * (info.MinValue + info.MaxValue)), info.NormalizeRange(info.MaxValue - info.MinValue));
    if (FILE* f = std::fopen(argv[4], "wb")) {
        std::fwrite(voxels.data(), 1, voxels.size(), f);
        std::fclose(f);
    }
    return 0;
}
