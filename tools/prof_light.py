"""Times individual illumination operators at bench size (diagnostics; not part of the product)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402


def main():
    n = int(os.environ.get("N", "512"))
    cfg = S.CONFIGS[3]
    dev = torch.device("cuda", 0)
    vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize(); res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    world = S.default_world()
    cases = {
        "add +X only (1,0,0)": abi.DirLightParams((1, 0, 0), 0.3),
        "add +Y only (0,1,0)": abi.DirLightParams((0, 1, 0), 0.3),
        "add +Z only (0,0,1)": abi.DirLightParams((0, 0, 1), 0.3),
        "add L0 (X then Z, sheared)": S.light(0),
        "add L2 (Z then Y, sheared)": S.light(2),
    }
    best = {}
    for rep in range(5):
        for name, light in cases.items():
            res.add_dir_light(light, True, world)
            ms = res.last_gpu_time_ms(0)
            best[name] = min(best.get(name, 1e9), ms)
    for name, ms in best.items():
        print(f"{name:32s} {ms:8.3f} ms (min of 5)")
    old = S.light(1)
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
    res.add_dir_light(old, True, world)
    ms = 1e9
    for rep in range(6):
        res.change_dir_light(old, new, world)
        ms = min(ms, res.last_gpu_time_ms(0))
        old, new = new, old
    print(f"{'change L1 +-5deg (fused)':32s} {ms:8.3f} ms (min of 6)")
    print(res.launch_counters())
    res.close()


if __name__ == "__main__":
    main()
