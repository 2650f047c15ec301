"""Timeline of ONE sweep launch at bench size: per hop distance from the upstream corner, when the tiles start, pass slice 63,
finish their last slice and end (tunable sweep_debug = 2; the library prints the stamps at tbrm_flush). Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
abi.set_tunable("sweep_debug", 2)
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]:
    abi.set_tunable(k, int(v))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
for i in range(2):
    res.add_dir_light(S.light(i), True, world)
res.flush()  # (warm-up; prints the last launch)
print("== Add of light 2 (its second pass is the last launch)", flush=True)
res.add_dir_light(S.light(2), True, world)
res.flush()
print("== Change of light 1 by 5 degrees (two streams; second pass)", flush=True)
res.change_dir_light(S.light(1), abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1]), world)
res.flush()
res.close()
