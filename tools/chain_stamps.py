#!/usr/bin/env python3
"""Timeline of ONE chained sweep launch (tunable sweep_debug = 2: every tile stamps its arrival, its first slice, its last slice and its
write-back; tbrm_flush prints them per pass): when do the tiles of pass i + 1 arrive, and when do they get to work? Diagnostics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
dims = (n, n, n)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0))
torch.cuda.synchronize()
world = S.default_world()
res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys("A")))
res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
res.reserve(4)
lights = [S.light(i) for i in range(4)]
res.clear_light_volume(0.0)
for l in lights:
    res.add_dir_light(l, True, world)
res.flush()
abi.set_tunable("sweep_debug", 2)
for i in range(4):
    print(f"-- light {i}: remove (cached factors: sweeps only)", file=sys.stderr, flush=True)
    res.add_dir_light(lights[i], False, world)
    res.flush()
    print(f"-- light {i}: add again", file=sys.stderr, flush=True)
    res.add_dir_light(lights[i], True, world)
    res.flush()
new = abi.DirLightParams(S.rotate_z(S.LIGHTS[0][0], 5.0), lights[0].light_intensity)
print("-- change light 0 by 5 degrees (fused, two streams)", file=sys.stderr, flush=True)
res.change_dir_light(lights[0], new, world)
res.flush()
res.close()
