"""GPU time of the light operators at bench size under the sweep kernel's tunables (cache off). VARIANTS = ';'-separated
lists of name=value pairs. Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
cfg = S.CONFIGS[3]
dev = torch.device("cuda", 0)
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
world = S.default_world()
variants = os.environ.get("VARIANTS", "light_sweep=0;light_sweep=1").split(";")
for var in variants:
    tun = dict(light_cache_mb=-1, force_slice_kernel=0, occ_overlap=4, light_sweep=1, sweep_prefetch=0, occ_slices=0, sweep_debug=0)
    for kv in var.split(","):
        if kv.strip():
            k, v = kv.split("=")
            tun[k.strip()] = int(v)
    for k, v in tun.items():
        abi.set_tunable(k, v)
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    adds, changes = [], []
    for rep in range(2):
        res.clear_light_volume(0.0)
        adds = []
        for i in range(4):
            res.add_dir_light(S.light(i), True, world)
            adds.append(res.last_gpu_time_ms(0))
    cur = [S.light(i) for i in range(4)]
    for k in range(1, 5):
        for i in range(4):
            new = abi.DirLightParams(S.rotate_z(S.LIGHTS[i][0], 5.0 * k), S.LIGHTS[i][1])
            res.change_dir_light(cur[i], new, world)
            changes.append(res.last_gpu_time_ms(0))
            cur[i] = new
    try:
        res.flush()
    except Exception as e:  # noqa: BLE001
        print("flush:", e)
    st = res.light_cache_stats()
    res.close()
    ch = np.array(changes[4:])
    print(f"{var:60s}: adds {' '.join('%.3f' % a for a in adds)} ms; change mean {ch.mean():.3f} min {ch.min():.3f} max {ch.max():.3f}; cache {st['entries']} entries {st['bytes'] / 2**20:.0f} MiB hits {st['hits']}", flush=True)
