// Drives include/tbrm_slabs.hpp (FSlabGroup, the single-process C++ driver of the slab-partitioned operators) with two
// slab-resident handles on one GPU and checks light volume and frame against one whole handle, bit for bit.
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/cpp/slabs_test.cpp -L <lib> -ltbrm -L /opt/rocm/lib -lamdhip64
#include "tbrm_slabs.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using tbrm_plugin::FSlabGroup;

static std::vector<uint16_t> make_volume(int nx, int ny, int nz)
{
    std::vector<uint16_t> v((size_t) nx * ny * nz);
    const double cx[3] = {0.35, 0.62, 0.5}, cy[3] = {0.4, 0.55, 0.7}, cz[3] = {0.3, 0.65, 0.5}, rad[3] = {0.22, 0.18, 0.12};
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const double u = (x + 0.5) / nx, w = (y + 0.5) / ny, t = (z + 0.5) / nz;
                double d = 0.05 + 0.02 * std::sin(37.0 * u + 11.0 * w) * std::cos(23.0 * t);
                for (int b = 0; b < 3; ++b) {
                    const double r2 = (u - cx[b]) * (u - cx[b]) + (w - cy[b]) * (w - cy[b]) + (t - cz[b]) * (t - cz[b]);
                    d += 0.8 * std::exp(-r2 / (rad[b] * rad[b]));
                }
                d = d < 0 ? 0 : (d > 1 ? 1 : d);
                v[((size_t) z * ny + y) * nx + x] = (uint16_t) (d * 65535.0 + 0.5);
            }
    return v;
}

#define TRY(expr)                                                               \
    do {                                                                        \
        if ((expr) != TBRM_OK) {                                                \
            std::printf("error at %s: %s\n", #expr, tbrm_last_error());         \
            return 2;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char** argv)
{
    const int n_slabs = argc > 1 ? std::atoi(argv[1]) : 2;
    const bool host_sync = argc > 2 && std::atoi(argv[2]) != 0;
    const int nx = 72, ny = 56, nz = 32 * n_slabs;
    const std::vector<uint16_t> vol = make_volume(nx, ny, nz);
    tbrm_resources_desc desc{};
    desc.dim_x = nx; desc.dim_y = ny; desc.dim_z = nz;
    desc.data_format = TBRM_FMT_G16;
    desc.data_address_mode = TBRM_ADDRESS_WRAP;
    desc.border_mode = TBRM_BORDER_ENGINE_8BIT;
    float lut[1024];
    TRY(tbrm_make_default_tf_lut(lut));
    const tbrm_windowing_params win{0.5f, 0.9f, 1, 0};

    tbrm_resources* whole = nullptr;
    TRY(tbrm_resources_create(&desc, &whole));
    TRY(tbrm_upload_volume(whole, vol.data(), vol.size() * 2));
    TRY(tbrm_set_tf_lut(whole, lut));
    TRY(tbrm_set_windowing(whole, &win));
    TRY(tbrm_clear_light_volume(whole, 0.0f));

    std::vector<tbrm_resources*> parts(n_slabs, nullptr);
    std::vector<int32_t> bounds;
    for (int k = 0; k <= n_slabs; ++k) bounds.push_back(k * nz / n_slabs);
    const size_t slice = (size_t) nx * ny;
    for (int k = 0; k < n_slabs; ++k) {
        const tbrm_slab owned{bounds[k], bounds[k + 1]};
        TRY(tbrm_resources_create_slab(&desc, &owned, &parts[k]));
        int32_t d[3], l[3];
        TRY(tbrm_slab_resident_slices(parts[k], d, l));
        TRY(tbrm_upload_volume_slices(parts[k], d[0], d[1] - d[0], vol.data() + (size_t) d[0] * slice, (size_t) (d[1] - d[0]) * slice * 2));
        if (d[2] >= 0) TRY(tbrm_upload_volume_slices(parts[k], d[2], 8, vol.data() + (size_t) d[2] * slice, 8 * slice * 2));
        TRY(tbrm_set_tf_lut(parts[k], lut));
        TRY(tbrm_set_windowing(parts[k], &win));
        TRY(tbrm_clear_light_volume(parts[k], 0.0f));
        std::printf("slab %d owns light slices [%d, %d), holds data slices [%d, %d) wrap copy of %d\n", k, owned.z_begin, owned.z_end, d[0], d[1], d[2]);
    }

    tbrm_world_params world{};
    world.volume_transform.rotation = tbrm_quatd{0, 0, 0, 1};
    world.volume_transform.translation = tbrm_vec3d{0, 0, 0};
    world.volume_transform.scale3d = tbrm_vec3d{100, 100, 100};
    world.clipping_plane.center = tbrm_vec3d{0, 0, 100000};
    world.clipping_plane.direction = tbrm_vec3d{0, 0, -1};

    const tbrm_dir_light_params lights[4] = {{{1, .35, -.5}, 0.5f, 0}, {{-.4, 1, -.3}, 0.4f, 0}, {{.2, -.3, -1}, 0.4f, 0}, {{.1, .45, 1}, 0.3f, 0}};
    try {
        FSlabGroup group(parts, bounds, std::vector<int>(n_slabs, 0));
        group.bHostSynchronise = host_sync;
        int flag = 0;
        std::vector<tbrm_dir_light_params> all(lights, lights + 4);
        group.ResetAllLights(all, world);
        for (const auto& l : all) TRY(tbrm_add_dir_light(whole, &l, 1, &world, &flag, 0));
        const tbrm_dir_light_params moved{{-.45, 1, -.25}, 0.4f, 0}, turned{{1, .1, -.2}, 0.4f, 0};
        group.ChangeDirLight(lights[1], moved, world);   // fused
        TRY(tbrm_change_dir_light(whole, &lights[1], &moved, &world, &flag, 0));
        group.ChangeDirLight(lights[2], turned, world);  // across faces: remove + add
        TRY(tbrm_change_dir_light(whole, &lights[2], &turned, &world, &flag, 0));
        for (int rep = 0; rep < 6; ++rep) { // back-to-back operations: the handles' streams run ahead of each other
            const tbrm_dir_light_params a{{0.3 + 0.1 * rep, -0.8, 0.4 - 0.15 * rep}, 0.2f, 0}, b{{0.35 + 0.1 * rep, -0.8, 0.45 - 0.15 * rep}, 0.25f, 0};
            group.AddDirLight(a, true, world);
            TRY(tbrm_add_dir_light(whole, &a, 1, &world, &flag, 0));
            group.ChangeDirLight(a, b, world);
            TRY(tbrm_change_dir_light(whole, &a, &b, &world, &flag, 0));
        }

        std::vector<uint8_t> ref((size_t) nx * ny * nz), got((size_t) nx * ny * nz / n_slabs);
        TRY(tbrm_download_light_volume(whole, ref.data(), ref.size()));
        size_t lit = 0, bad = 0;
        for (int k = 0; k < n_slabs; ++k) {
            TRY(tbrm_download_light_slices(parts[k], bounds[k], bounds[k + 1] - bounds[k], got.data(), got.size()));
            for (size_t i = 0; i < got.size(); ++i) {
                bad += got[i] != ref[(size_t) bounds[k] * slice + i];
                lit += got[i] != 0;
            }
        }
        std::printf("light volume: %zu voxels differ, %zu lit\n", bad, lit);

        group.ExchangeLightHalos();
        tbrm_camera cam{};
        cam.position = tbrm_vec3d{-145, -95, 80};
        const double fl = std::sqrt(145.0 * 145 + 95.0 * 95 + 80.0 * 80);
        cam.forward = tbrm_vec3d{145 / fl, 95 / fl, -80 / fl};
        const double rl = std::sqrt(cam.forward.y * cam.forward.y + cam.forward.x * cam.forward.x);
        cam.right = tbrm_vec3d{cam.forward.y / rl, -cam.forward.x / rl, 0};
        cam.up = tbrm_vec3d{cam.right.y * cam.forward.z - cam.right.z * cam.forward.y, cam.right.z * cam.forward.x - cam.right.x * cam.forward.z,
            cam.right.x * cam.forward.y - cam.right.y * cam.forward.x};
        cam.tan_half_fov_x = cam.tan_half_fov_y = std::tan(30.0 * 3.14159265358979323846 / 180.0);
        cam.width = cam.height = 64;
        const tbrm_tile tile{0, 0, 64, 64, 1, 0};
        const tbrm_raymarch_params rp{96.0f, -1, 1, 0};
        std::vector<float> want((size_t) 64 * 64 * 4), frame((size_t) 64 * 64 * 4);
        TRY(tbrm_raymarch_lit(whole, &cam, &tile, &rp, &world, want.data()));
        group.RenderLit(cam, tile, rp, world, frame.data());
        double sum_a = 0;
        for (size_t i = 3; i < want.size(); i += 4) sum_a += want[i];
        const bool same = std::memcmp(want.data(), frame.data(), want.size() * sizeof(float)) == 0;
        std::printf("frame: %s, mean alpha %.6f, %zu bytes moved between the handles\n", same ? "identical" : "DIFFERENT", sum_a / (64 * 64), group.BytesMoved);
        if (bad != 0 || lit == 0 || !same || !(sum_a > 0)) return 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
    for (tbrm_resources* p : parts) tbrm_resources_destroy(p);
    tbrm_resources_destroy(whole);
    std::printf("OK\n");
    return 0;
}
