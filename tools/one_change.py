"""A handful of fused ChangeDirLight calls at bench size, for profiling under rocprofv3 (tunables come from the TBRM_*
environment variables). LIGHT = which config light (default 1), REPS = calls (default 6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
li = int(os.environ.get("LIGHT", "1"))
cfg = S.CONFIGS[3]
dev = torch.device("cuda", 0)
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
old = S.light(li)
new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], 5.0), S.LIGHTS[li][1])
res.add_dir_light(old, True, world)
best = 1e9
for rep in range(int(os.environ.get("REPS", "6"))):
    res.change_dir_light(old, new, world)
    best = min(best, res.last_gpu_time_ms(0))
    old, new = new, old
print(f"change L{li}: {best:.3f} ms (min)", {k: abi.get_tunable(k) for k in ("chunk_steps", "occ_slices")})
res.close()
