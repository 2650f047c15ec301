"""Host time to enqueue one fused ChangeDirLight (about a hundred kernel launches) against its GPU time: python tools/host_enqueue_time.py [n]."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tbraymarcherplugin_amd import abi, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512; cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize(); res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS)); res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
old = S.light(1); new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
res.add_dir_light(old, True, world); res.flush()
for rep in range(3):
    res.change_dir_light(old, new, world); old, new = new, old
res.flush()
for k in (1, 4):
    t0 = time.perf_counter()
    for rep in range(k):
        res.change_dir_light(old, new, world); old, new = new, old
    t1 = time.perf_counter()
    res.flush()
    t2 = time.perf_counter()
    print(f"{k} change(s): host enqueue {1e3*(t1-t0)/k:.3f} ms per call, until flushed {1e3*(t2-t0)/k:.3f} ms per call, last event-timed {res.last_gpu_time_ms(0):.3f} ms")
res.close()
