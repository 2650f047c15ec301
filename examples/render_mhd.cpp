// render_mhd — the drop-in path end to end, as a host without the engine would use it: MetaImage volume in
// (include/tbrm_volume_io.hpp, the reference's UMHDLoader + normalisation), two directional lights and a clip-free frame
// through the reference-named C++ host side (include/tbrm_plugin.hpp: ARaymarchVolume, ARaymarchLight, Tick), image out as
// binary PPM (premultiplied RGBA composited over black).
//
//   g++ -std=c++17 -O2 -I include examples/render_mhd.cpp -o render_mhd -L tbraymarcherplugin_amd/lib -ltbrm -lz
//       (plus -Wl,-rpath,$PWD/tbraymarcherplugin_amd/lib -Wl,-rpath,/opt/rocm/lib to run it in place)
//   ./render_mhd volume.mhd out.ppm [width height steps]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tbrm_volume_io.hpp" // includes tbrm_plugin.hpp

using namespace tbrm_plugin;

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s volume.mhd out.ppm [width height steps]\n", argv[0]);
        return 2;
    }
    const int width = argc > 3 ? std::atoi(argv[3]) : 512, height = argc > 4 ? std::atoi(argv[4]) : 512;
    const float steps = argc > 5 ? (float) std::atof(argv[5]) : 256.0f;

    ARaymarchVolume volume;
    FVolumeInfo info;
    if (!LoadMHDFileIntoVolumeNormalized(volume, argv[1], &info)) { // RaymarchVolume.cpp:596-612
        std::fprintf(stderr, "could not load %s: %s\n", argv[1], tbrm_last_error());
        return 1;
    }
    std::printf("volume %d x %d x %d, values %g .. %g\n", info.Dimensions[0], info.Dimensions[1], info.Dimensions[2], info.MinValue, info.MaxValue);

    // window the upper half of the value range (window units are the file's units: FVolumeInfo::NormalizeValue / Range)
    volume.SetWindowCenter(info.NormalizeValue(info.MinValue + 0.6f * (info.MaxValue - info.MinValue)));
    volume.SetWindowWidth(info.NormalizeRange(0.8f * (info.MaxValue - info.MinValue)));
    volume.SetRaymarchSteps(steps);

    ARaymarchLight key, fill;
    key.ForwardVector = FVector{1, 0.35, -0.5};
    key.LightIntensity = 0.7f;
    fill.ForwardVector = FVector{-0.4, 1, -0.3};
    fill.LightIntensity = 0.3f;
    volume.LightsArray = {&key, &fill};
    volume.Tick(0.016f); // first tick: recompute requested -> ResetAllLights

    tbrm_camera cam{};
    cam.position = FVector{-145, -95, 80};
    const double fl = std::sqrt(145.0 * 145 + 95.0 * 95 + 80.0 * 80);
    cam.forward = FVector{145 / fl, 95 / fl, -80 / fl};
    const double rl = std::sqrt(cam.forward.x * cam.forward.x + cam.forward.y * cam.forward.y);
    cam.right = FVector{cam.forward.y / rl, -cam.forward.x / rl, 0};
    cam.up = FVector{cam.right.y * cam.forward.z - cam.right.z * cam.forward.y, cam.right.z * cam.forward.x - cam.right.x * cam.forward.z,
        cam.right.x * cam.forward.y - cam.right.y * cam.forward.x};
    cam.tan_half_fov_y = std::tan(25.0 * 3.14159265358979323846 / 180.0);
    cam.tan_half_fov_x = cam.tan_half_fov_y * width / height;
    cam.width = width;
    cam.height = height;

    std::vector<float> rgba((size_t) width * height * 4);
    if (!volume.RenderLit(cam, rgba.data())) {
        std::fprintf(stderr, "render failed: %s\n", tbrm_last_error());
        return 1;
    }
    double coverage = 0;
    std::vector<unsigned char> rgb((size_t) width * height * 3);
    for (size_t i = 0; i < (size_t) width * height; ++i) {
        coverage += rgba[4 * i + 3];
        for (int c = 0; c < 3; ++c) // premultiplied colour over black, display gamma 2.2
            rgb[3 * i + c] = (unsigned char) std::lround(255.0 * std::pow(std::min(std::max((double) rgba[4 * i + c], 0.0), 1.0), 1.0 / 2.2));
    }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::fwrite(rgb.data(), 1, rgb.size(), f);
    std::fclose(f);
    std::printf("wrote %s (%d x %d, mean alpha %.4f)\n", argv[2], width, height, coverage / ((double) width * height));
    return 0;
}
