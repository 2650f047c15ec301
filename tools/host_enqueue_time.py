"""Host time to enqueue the benchmark's step (a ChangeDirLight that propagates the new light — about 150 launches on two
streams — and the frame) against its GPU time: python tools/host_enqueue_time.py [n]. Tunables come from TBRM_* as usual."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
cam = S.default_camera(cfg["fb"], cfg["fb"])
tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")
for i in range(4):
    res.add_dir_light(S.light(i), True, world)
state = {"angle": 0.0, "cur": S.light(1)}


def change():
    state["angle"] += 1.0  # small turns: the same cube faces for all calls
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], state["angle"]), S.LIGHTS[1][1])
    res.change_dir_light(state["cur"], new, world)
    state["cur"] = new


for _ in range(3):
    change()
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
res.flush()
for what, fn in (("change", change), ("frame", lambda: res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())),
                 ("change + frame", lambda: (change(), res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())))):
    k = 6
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    t1 = time.perf_counter()
    res.flush()
    t2 = time.perf_counter()
    print(f"{what:15s}: host enqueue {1e3 * (t1 - t0) / k:.3f} ms per call, until flushed {1e3 * (t2 - t0) / k:.3f} ms per call; "
          f"event-timed: light {res.last_gpu_time_ms(0):.3f} ms, frame {res.last_gpu_time_ms(1):.3f} ms")
res.close()
