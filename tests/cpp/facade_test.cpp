// Drives the C++ facade (include/tbrm_plugin.hpp) the way the reference's actor is driven: SetVolumeAsset, lights,
// Tick-based selective updates (RaymarchVolume.cpp:327-416), offscreen render. Prints one "key=value" line per check;
// tests/test_facade.py compiles it with g++ and, on a GPU box, runs it and compares against the oracle.
#include "tbrm_plugin.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace tbrm_plugin;

static uint32_t hash32(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t h = x * 73856093u ^ y * 19349663u ^ z * 83492791u ^ 0x5EED0002u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 32;
    std::vector<uint16_t> vol((size_t) n * n * n);
    for (int z = 0; z < n; ++z)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const double px = (x + 0.5) / n - 0.5, py = (y + 0.5) / n - 0.5, pz = (z + 0.5) / n - 0.5;
                const double r = std::sqrt(px * px + py * py + pz * pz);
                double v = r < 0.42 ? 0.35 : 0.0;
                v += 0.5 * std::exp(-((r - 0.30) / 0.03) * ((r - 0.30) / 0.03));
                v += 0.02 * (hash32(x, y, z) / 4294967296.0 - 0.5);
                v = v < 0 ? 0 : (v > 1 ? 1 : v);
                vol[((size_t) z * n + y) * n + x] = (uint16_t) (v * 65535.0 + 0.5);
            }

    ARaymarchLight l0, l1, l2;
    l0.ForwardVector = FVector{1, .35, -.5}; l0.LightIntensity = 0.5f;
    l1.ForwardVector = FVector{-.4, 1, -.3}; l1.LightIntensity = 0.4f;
    l2.ForwardVector = FVector{.2, -.3, -1}; l2.LightIntensity = 0.4f;

    URaymarchVolume volume; // alias of ARaymarchVolume
    volume.LightsArray = {&l0, &l1, &l2};
    if (!volume.SetVolumeAsset(vol.data(), n, n, n, TBRM_FMT_G16)) { std::printf("error=%s\n", tbrm_last_error()); return 2; }
    FColorCurve tf;
    tf.AddKey(0.0f, 0, 0, 0, 0); tf.AddKey(0.25f, .8f, .4f, .3f, 0); tf.AddKey(0.45f, .9f, .6f, .5f, .02f);
    tf.AddKey(0.70f, 1, 1, .9f, .15f); tf.AddKey(1.0f, 1, 1, 1, .40f);
    volume.SetTFCurve(tf);
    volume.SetWindowCenter(0.5f); volume.SetWindowWidth(0.9f); volume.SetHighCutoff(false);
    volume.SetRaymarchSteps(64);

    volume.Tick(0.016f); // first tick: recompute requested -> ResetAllLights
    std::printf("after_first_tick resets=%d adds=%d changes=%d\n", volume.Stats.Resets, volume.Stats.LightAdds, volume.Stats.LightChanges);
    volume.Tick(0.016f); // nothing moved
    std::printf("after_idle_tick resets=%d adds=%d changes=%d\n", volume.Stats.Resets, volume.Stats.LightAdds, volume.Stats.LightChanges);
    l1.ForwardVector = FVector{-.45, 1, -.3}; // one light of three moved -> selective update
    volume.Tick(0.016f);
    std::printf("after_one_moved resets=%d adds=%d changes=%d\n", volume.Stats.Resets, volume.Stats.LightAdds, volume.Stats.LightChanges);
    l0.LightIntensity = 0.45f; l2.ForwardVector = FVector{.25, -.3, -1}; // two of three moved -> reset is cheaper
    volume.Tick(0.016f);
    std::printf("after_two_moved resets=%d adds=%d changes=%d\n", volume.Stats.Resets, volume.Stats.LightAdds, volume.Stats.LightChanges);
    volume.SetWindowWidth(0.8f); // windowing change -> full recompute
    volume.Tick(0.016f);
    std::printf("after_window_change resets=%d adds=%d changes=%d\n", volume.Stats.Resets, volume.Stats.LightAdds, volume.Stats.LightChanges);

    tbrm_camera cam{};
    cam.position = FVector{-145, -95, 80};
    const double fl = std::sqrt(145.0 * 145 + 95.0 * 95 + 80.0 * 80);
    cam.forward = FVector{145 / fl, 95 / fl, -80 / fl};
    const double rl = std::sqrt(cam.forward.y * cam.forward.y + cam.forward.x * cam.forward.x);
    cam.right = FVector{cam.forward.y / rl, -cam.forward.x / rl, 0}; // forward x up(0,0,1)
    cam.up = FVector{cam.right.y * cam.forward.z - cam.right.z * cam.forward.y, cam.right.z * cam.forward.x - cam.right.x * cam.forward.z,
        cam.right.x * cam.forward.y - cam.right.y * cam.forward.x};
    cam.tan_half_fov_y = std::tan(30.0 * 3.14159265358979323846 / 180.0);
    cam.tan_half_fov_x = cam.tan_half_fov_y;
    cam.width = cam.height = 64;
    std::vector<float> img((size_t) 64 * 64 * 4);
    if (!volume.RenderLit(cam, img.data())) { std::printf("error=%s\n", tbrm_last_error()); return 3; }
    double sum_a = 0, max_a = 0;
    for (size_t i = 3; i < img.size(); i += 4) { sum_a += img[i]; max_a = img[i] > max_a ? img[i] : max_a; }
    std::printf("render mean_alpha=%.6f max_alpha=%.6f\n", sum_a / (64 * 64), max_a);
    // SwitchRenderer(Intensity): the slice view needs no lights; switching back to Lit requests a recompute (:786-800)
    volume.SwitchRenderer(tbrm_plugin::ERaymarchMaterial::Intensity);
    if (!volume.Render(cam, img.data())) { std::printf("error=%s\n", tbrm_last_error()); return 4; }
    double hit = 0, sum_i = 0;
    for (size_t i = 0; i < img.size(); i += 4) { hit += img[i + 3]; sum_i += img[i]; }
    std::printf("intensity hit_fraction=%.6f mean_intensity_of_hits=%.6f\n", hit / (64 * 64), hit > 0 ? sum_i / hit : 0.0);
    // SwitchRenderer(Octree): rebuilds the pyramid on the next frame, marches level OctreeVolumeMip unlit
    volume.SwitchRenderer(tbrm_plugin::ERaymarchMaterial::Octree);
    volume.OctreeVolumeMip = 1;
    if (!volume.Render(cam, img.data())) { std::printf("error=%s\n", tbrm_last_error()); return 5; }
    double oct_a = 0;
    for (size_t i = 3; i < img.size(); i += 4) oct_a += img[i];
    std::printf("octree mean_alpha=%.6f rebuild_pending=%d\n", oct_a / (64 * 64), volume.bRequestedOctreeRebuild ? 1 : 0);
    volume.SwitchRenderer(tbrm_plugin::ERaymarchMaterial::Lit);
    // batched reset: all lights through one tbrm_add_dir_lights call; the frame it lights is the same up to UNORM8 rounding ties
    volume.bBatchLightsOnReset = true;
    volume.Tick(0.016f); // SwitchRenderer(Lit) requested a recompute
    std::vector<float> img2((size_t) 64 * 64 * 4);
    if (!volume.RenderLit(cam, img2.data())) { std::printf("error=%s\n", tbrm_last_error()); return 6; }
    double sum_b = 0;
    for (size_t i = 3; i < img2.size(); i += 4) sum_b += img2[i];
    std::printf("batched_reset resets=%d adds=%d mean_alpha=%.6f\n", volume.Stats.Resets, volume.Stats.LightAdds, sum_b / (64 * 64));
    // Blueprint-pure helpers
    int32_t vd[3];
    tbrm_plugin::URaymarchUtils::GetVolumeTextureDimensions(&volume.RaymarchResources, vd);
    FVector tc, lc;
    tbrm_plugin::URaymarchUtils::LocalToTextureCoords(FVector{-1, 0, 0.5}, tc);
    tbrm_plugin::URaymarchUtils::TextureToLocalCoords(tc, lc);
    tbrm_transform tr{{0, 0, std::sin(0.25 * 3.14159265358979323846), std::cos(0.25 * 3.14159265358979323846)}, {1, 2, 3}, {2, 2, 2}}; // 90 deg about z
    double mtx[4][4];
    tbrm_plugin::URaymarchUtils::TransformToMatrix(tr, mtx, true);
    std::printf("helpers dims=%d,%d,%d tex=%.2f,%.2f,%.2f local=%.2f,%.2f,%.2f row0=%.3f,%.3f,%.3f row3=%.0f,%.0f,%.0f\n", vd[0], vd[1], vd[2], tc.x, tc.y, tc.z,
                lc.x, lc.y, lc.z, mtx[0][0], mtx[0][1], mtx[0][2], mtx[3][0], mtx[3][1], mtx[3][2]);
    uint64_t counters[3];
    tbrm_launch_counters(volume.RaymarchResources.Handle, counters);
    std::printf("launches chunk=%llu slice=%llu raymarch=%llu\n", (unsigned long long) counters[0], (unsigned long long) counters[1], (unsigned long long) counters[2]);
    std::printf("OK\n");
    return 0;
}
