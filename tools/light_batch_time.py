#!/usr/bin/env python3
"""ResetAllLights at 512^3 with 4 and 8 config lights: light by light vs tbrm_add_dir_lights (the pairing rule decides), and the reported schedule."""
import sys, numpy as np, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, synthetic as S
n = 512; dims = (n, n, n)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0)); torch.cuda.synchronize()
res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
res.upload_volume_device(vol.data_ptr(), vol.numel()*2); res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys("A"))); res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
world = S.default_world()
for nl in (4, 8):
    lights = [S.light(i) for i in range(nl)]
    for rep in range(3):
        res.clear_light_volume(0.0); res.flush()
        t = 0.0
        for l in lights:
            res.add_dir_light(l, True, world); res.flush(); t += res.last_gpu_time_ms(0)
        res.clear_light_volume(0.0); res.flush()
        sched = res.add_dir_lights(lights, True, world); res.flush(); tb = res.last_gpu_time_ms(0)
    print(f"{nl} lights: light by light {t:.3f} ms, batched {tb:.3f} ms ({len(sched)} entries, {sum(1 for s in sched if s[2] >= 0)} pairs): {sched}")
