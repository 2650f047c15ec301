"""GPU time of operators served from the contribution cache alone (k_apply_kept): ResetAllLights with every light's L kept,
and a ChangeDirLight back and forth between two directions. Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
lights = [S.light(i) for i in range(4)]


def reset(batched):
    res.clear_light_volume(0.0)
    total = 0.0
    if batched:
        res.add_dir_lights(lights, True, world)
        return res.last_gpu_time_ms(0)
    for l in lights:
        res.add_dir_light(l, True, world)
        total += res.last_gpu_time_ms(0)
    return total


for batched in (False, True):
    for rep in range(3):
        before = res.light_cache_stats()
        ms = reset(batched)
        st = res.light_cache_stats()
        print(f"reset_all_lights ({'one call' if batched else 'four calls'}) #{rep}: {ms:6.3f} ms  hits {st['hits'] - before['hits']} propagated {st['propagated'] - before['propagated']}", flush=True)
    ref = res.download_light_volume()
a, b = S.light(1), abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
for rep in range(6):
    before = res.light_cache_stats()
    res.change_dir_light(a, b, world)
    ms = res.last_gpu_time_ms(0)
    st = res.light_cache_stats()
    print(f"change back and forth #{rep}: {ms:6.3f} ms  hits {st['hits'] - before['hits']} propagated {st['propagated'] - before['propagated']}", flush=True)
    a, b = b, a
res.close()
