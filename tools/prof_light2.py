import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tbraymarcherplugin_amd import abi, synthetic as S
n = 512
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize(); res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS)); res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
light = abi.DirLightParams((0.2, 0.3, 1), 0.3)   # single-axis? no: two passes Z then Y
light = abi.DirLightParams((0.05, 0.04, 1), 0.3)  # w0 > 0.99 -> one Z pass, generic fractional offsets (g = 1)
for dbg in [0, 2112]:
    os.environ["TBRM_DEBUG"] = str(dbg)
    for rep in range(2):
        res.add_dir_light(light, True, world); ms = res.last_gpu_time_ms(0)
    print("debug", dbg, round(ms, 3), "ms")
print(res.launch_counters()); res.close()
