"""What follows a frame how soon: frame, frame, a cached Add (one sweep per pass), frame, a clear — back to back on the handle's
stream, for a rocprofv3 kernel trace (tools/trace_timeline.py). Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = 512
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
fb = int(os.environ.get("FB", cfg["fb"]))
cam = S.default_camera(fb, fb)
tile = abi.Tile(0, 0, fb, fb, 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((fb, fb, 4), dtype=torch.float32, device="cuda")
for i in range(4):
    res.add_dir_light(S.light(i), True, world)
res.add_dir_light(S.light(0), False, world)
res.add_dir_light(S.light(0), True, world)  # (from the cache from now on)
res.flush()
for _ in range(2):
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.add_dir_light(S.light(0), False, world)
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.add_dir_light(S.light(0), True, world)
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.clear_light_volume(0.0)
    for i in range(4):
        res.add_dir_light(S.light(i), True, world)
res.flush()
res.close()
