"""GPU time of a sequence of ChangeDirLight calls that turn ONE light by 5 degrees at a time (bench size), with what the
factor cache did for each: "cached" = only the new light's occlusion was computed, "both" = both lights', "from cache" = none,
"remove+add" = the major axes differed (LightingShaders.cpp:192-198). Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
li = int(os.environ.get("LIGHT", "1"))
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
for i in range(4):
    res.add_dir_light(S.light(i), True, world)
cur = S.light(li)
before = res.light_cache_stats()
for k in range(1, int(os.environ.get("STEPS", "14")) + 1):
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], 5.0 * k), S.LIGHTS[li][1])
    pa, _ = abi.host_light_passes(cur, world, (n, n, n))
    pb, _ = abi.host_light_passes(new, world, (n, n, n))
    fused = (pa[0].face, pa[1].face) == (pb[0].face, pb[1].face)
    res.change_dir_light(cur, new, world)
    ms = res.last_gpu_time_ms(0)
    st = res.light_cache_stats()
    kind = "remove+add" if not fused else ("cached" if st["hits"] - before["hits"] == 2 else ("from cache" if st["hits"] - before["hits"] == 4 else "both"))
    print(f"{5 * (k - 1):3d} -> {5 * k:3d} deg: {ms:6.3f} ms  {kind}", flush=True)
    before, cur = st, new
res.close()
