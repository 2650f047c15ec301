"""Would frames and light operators overlap if they ran on different streams? Two handles on one GPU hold the same scene: one
renders frames (its stream at default priority), the other removes and re-adds a light from the factor cache (two one-stream
sweeps per operator; its stream at the priority PRIO = 1 highest / 0 default). Wall time of N rounds of each alone and of both
enqueued together. Diagnostics for DESIGN.md 4.2c (not product)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = 512
cfg = S.CONFIGS[3]
dev = torch.device("cuda", 0)
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
world = S.default_world()


def make(prio):
    abi.set_tunable("stream_priority", prio)
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    for i in range(4):
        res.add_dir_light(S.light(i), True, world)
    res.flush()
    return res


prio = int(os.environ.get("PRIO", "1"))
frames = make(0)
lights = make(prio)
cam = S.default_camera(cfg["fb"], cfg["fb"])
tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")
N = 12


def run(do_frames, do_lights):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        if do_lights:
            lights.add_dir_light(S.light(k % 4), False, world)
            lights.add_dir_light(S.light(k % 4), True, world)
        if do_frames:
            frames.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    frames.flush()
    lights.flush()
    return 1e3 * (time.perf_counter() - t0) / N


for _ in range(2):
    run(True, True)
f = min(run(True, False) for _ in range(3))
l = min(run(False, True) for _ in range(3))
b = min(run(True, True) for _ in range(3))
print(f"light stream priority {prio}: frame alone {f:.3f} ms, remove + add from the cache alone {l:.3f} ms, both enqueued together {b:.3f} ms per round "
      f"(sum {f + l:.3f}, max {max(f, l):.3f})")
frames.close()
lights.close()
