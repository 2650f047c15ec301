// Drives include/tbrm_tiles.hpp (FTileGroup, the single-process C++ driver of image-tile rendering over several handles) with N
// whole-volume handles on one GPU: every light operator on each handle, the frame marched in interleaved 8-row groups and gathered
// — and checks the handles' light volumes and the assembled frame against one handle that does everything alone, bit for bit.
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/cpp/tiles_test.cpp -L <lib> -ltbrm -L /opt/rocm/lib -lamdhip64
//   (+ -DTBRM_TILES_WITH_RCCL -lrccl: the gather by ncclAllGather)
//   tiles_test <handles> <gather: 0 peer copies to handle 0, 1 peer copies to every handle, 2 RCCL>
// TBRM_TILES_DEVICES=0,1,2,3 puts handle k on the k-th listed device (wrapping around a shorter list; default: every handle on
// device 0, the one-GPU box's form) — the same binary runs over 2 - 8 real devices unchanged; the reference handle that does
// everything alone lives on the first listed device.
#include "tbrm_tiles.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using tbrm_plugin::FTileGroup;

static std::vector<uint16_t> make_volume(int n)
{
    std::vector<uint16_t> v((size_t) n * n * n);
    for (int z = 0; z < n; ++z)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const double u = (x + 0.5) / n - 0.5, w = (y + 0.5) / n - 0.5, t = (z + 0.5) / n - 0.5;
                const double r = std::sqrt(u * u + w * w + t * t);
                double d = 0.05 + 0.03 * std::sin(31.0 * u) * std::cos(17.0 * w + 5.0 * t);
                d += 0.55 * std::exp(-((r - 0.3) / 0.04) * ((r - 0.3) / 0.04)) + (r < 0.28 ? 0.3 : 0.0);
                d = d < 0 ? 0 : (d > 1 ? 1 : d);
                v[((size_t) z * n + y) * n + x] = (uint16_t) (d * 65535.0 + 0.5);
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
    const int n_handles = argc > 1 ? std::atoi(argv[1]) : 2;
    const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
    const int n = 64, fw = 136, fh = n_handles == 8 ? 128 : 96; // (96 = 8 x 12: splits over 1, 2, 3, 4, 6 handles; 128 over 8; a width that is no multiple of 8)
    std::vector<int> listed;
    if (const char* e = std::getenv("TBRM_TILES_DEVICES")) {
        for (const char* c = e; *c;) {
            char* end = nullptr;
            const long d = std::strtol(c, &end, 10);
            if (end == c) break;
            listed.push_back((int) d);
            c = *end == ',' ? end + 1 : end;
        }
    }
    if (listed.empty()) listed.push_back(0);
    int n_dev = 0;
    TRY(tbrm_device_count(&n_dev));
    for (int d : listed)
        if (d < 0 || d >= n_dev) { std::printf("TBRM_TILES_DEVICES names device %d, the box has %d\n", d, n_dev); return 2; }
    std::vector<int> devices((size_t) n_handles);
    for (int k = 0; k < n_handles; ++k) devices[(size_t) k] = listed[(size_t) k % listed.size()];
    const std::vector<uint16_t> vol = make_volume(n);
    tbrm_resources_desc desc{};
    desc.dim_x = desc.dim_y = desc.dim_z = n;
    desc.data_format = TBRM_FMT_G16;
    desc.data_address_mode = TBRM_ADDRESS_WRAP;
    desc.border_mode = TBRM_BORDER_ENGINE_8BIT;
    float lut[1024];
    TRY(tbrm_make_default_tf_lut(lut));
    const tbrm_windowing_params win{0.5f, 0.9f, 1, 0};

    std::vector<tbrm_resources*> handles(n_handles + 1, nullptr); // [n_handles]: the one that does everything alone
    for (size_t k = 0; k < handles.size(); ++k) {
        tbrm_resources*& h = handles[k];
        desc.device = k < devices.size() ? devices[k] : listed[0];
        if (hipSetDevice(desc.device) != hipSuccess) return 4;
        TRY(tbrm_resources_create(&desc, &h));
        TRY(tbrm_upload_volume(h, vol.data(), vol.size() * 2));
        TRY(tbrm_set_tf_lut(h, lut));
        TRY(tbrm_set_windowing(h, &win));
    }
    tbrm_resources* const alone = handles.back();
    handles.pop_back();

    tbrm_world_params world{};
    world.volume_transform.rotation = tbrm_quatd{0, 0, 0, 1};
    world.volume_transform.translation = tbrm_vec3d{0, 0, 0};
    world.volume_transform.scale3d = tbrm_vec3d{100, 100, 100};
    world.clipping_plane.center = tbrm_vec3d{0, 0, 100000};
    world.clipping_plane.direction = tbrm_vec3d{0, 0, -1};
    tbrm_camera cam{};
    cam.position = tbrm_vec3d{-145, -95, 80};
    const double fl = std::sqrt(145.0 * 145 + 95.0 * 95 + 80.0 * 80);
    cam.forward = tbrm_vec3d{145 / fl, 95 / fl, -80 / fl};
    const double rl = std::sqrt(cam.forward.y * cam.forward.y + cam.forward.x * cam.forward.x);
    cam.right = tbrm_vec3d{cam.forward.y / rl, -cam.forward.x / rl, 0};
    cam.up = tbrm_vec3d{cam.right.y * cam.forward.z - cam.right.z * cam.forward.y, cam.right.z * cam.forward.x - cam.right.x * cam.forward.z,
                        cam.right.x * cam.forward.y - cam.right.y * cam.forward.x};
    cam.tan_half_fov_y = std::tan(30.0 * 3.14159265358979323846 / 180.0);
    cam.tan_half_fov_x = cam.tan_half_fov_y * fw / fh;
    cam.width = fw;
    cam.height = fh;
    const tbrm_raymarch_params rp{128.0f, -1, 1, 0};
    const tbrm_tile full{0, 0, fw, fh, 1, 0};

    const std::vector<tbrm_dir_light_params> lights = {{{1, .35, -.5}, 0.5f, 0}, {{-.4, 1, -.3}, 0.4f, 0}, {{.2, -.3, -1}, 0.4f, 0}};
    try {
        FTileGroup group(handles, devices, fw, fh, mode == 2 ? FTileGroup::EGather::Rccl : FTileGroup::EGather::PeerCopy);
        int flag = 0;
        group.ResetAllLights(lights, world);
        if (hipSetDevice(listed[0]) != hipSuccess) return 4;
        TRY(tbrm_clear_light_volume(alone, 0.0f));
        for (const auto& l : lights) TRY(tbrm_add_dir_light(alone, &l, 1, &world, &flag, 0));
        size_t frames_bad = 0, light_bad = 0;
        double sum_a = 0;
        std::vector<float> want((size_t) fw * fh * 4), got((size_t) fw * fh * 4);
        std::vector<uint8_t> ref((size_t) n * n * n), lv((size_t) n * n * n);
        tbrm_dir_light_params cur = lights[1];
        for (int step = 0; step < 4; ++step) { // a light turns, a frame follows — back to back, nothing drained in between by the group
            const double a = 0.09 * (step + 1);
            const tbrm_dir_light_params next{{lights[1].light_direction.x * std::cos(a) - lights[1].light_direction.y * std::sin(a),
                                              lights[1].light_direction.x * std::sin(a) + lights[1].light_direction.y * std::cos(a), lights[1].light_direction.z},
                                             lights[1].light_intensity, 0};
            group.ChangeDirLight(cur, next, world);
            TRY(tbrm_change_dir_light(alone, &cur, &next, &world, &flag, 0));
            cur = next;
            const int root = step % n_handles;
            const float* frame = group.RenderLit(cam, rp, world, root, mode == 1);
            TRY(tbrm_raymarch_lit(alone, &cam, &full, &rp, &world, want.data()));
            TRY(tbrm_flush(group.Handle(root)));
            if (hipSetDevice(devices[(size_t) root]) != hipSuccess) return 4;
            if (hipMemcpy(got.data(), frame, got.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 4;
            if (hipSetDevice(listed[0]) != hipSuccess) return 4;
            frames_bad += std::memcmp(want.data(), got.data(), want.size() * sizeof(float)) != 0;
            if (mode != 0) // every handle holds the frame
                for (int k = 0; k < n_handles; ++k) {
                    TRY(tbrm_flush(group.Handle(k)));
                    if (hipSetDevice(devices[(size_t) k]) != hipSuccess) return 4;
                    if (hipMemcpy(got.data(), group.Frame(k), got.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 4;
                    frames_bad += std::memcmp(want.data(), got.data(), want.size() * sizeof(float)) != 0;
                }
        }
        for (size_t i = 3; i < want.size(); i += 4) sum_a += want[i];
        TRY(tbrm_download_light_volume(alone, ref.data(), ref.size()));
        for (int k = 0; k < n_handles; ++k) {
            TRY(tbrm_download_light_volume(group.Handle(k), lv.data(), lv.size()));
            light_bad += std::memcmp(ref.data(), lv.data(), ref.size()) != 0;
        }
        std::printf("devices:");
        for (int d : devices) std::printf(" %d", d);
        std::printf("\nlight volumes: %zu of %d replicas differ\n", light_bad, n_handles);
        std::printf("frames: %zu differ, mean alpha %.6f, %zu bytes moved between the handles\n", frames_bad, sum_a / ((double) fw * fh), group.BytesMoved);
        if (frames_bad != 0 || light_bad != 0 || !(sum_a > 0)) return 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
    for (tbrm_resources* h : handles) tbrm_resources_destroy(h);
    tbrm_resources_destroy(alone);
    std::printf("OK\n");
    return 0;
}
