"""How far ahead of the GPU the host runs in the benchmark's loop (a Change of one of four lights, then the frame): host time
of every call, and when the host was done enqueuing against when the GPU was done. Diagnostics."""
import os
import sys
import time

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
cam = S.default_camera(cfg["fb"], cfg["fb"])
tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")
lights = [S.light(i) for i in range(4)]
for l in lights:
    res.add_dir_light(l, True, world)
angle = [0.0] * 4


def step(k, log):
    li = k % 4
    angle[li] += 5.0
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], angle[li]), S.LIGHTS[li][1])
    t0 = time.perf_counter()
    res.change_dir_light(lights[li], new, world)
    t1 = time.perf_counter()
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    t2 = time.perf_counter()
    lights[li] = new
    if log is not None:
        log.append((t0, t1, t2))


for k in range(8):
    step(k, None)
res.flush()
log = []
t_start = time.perf_counter()
for k in range(8, 24):
    step(k, log)
t_host = time.perf_counter()
res.flush()
t_gpu = time.perf_counter()
print(f"16 steps: host done enqueuing after {1e3 * (t_host - t_start):.2f} ms, GPU done after {1e3 * (t_gpu - t_start):.2f} ms")
print("per step (us): start, change, frame:")
for t0, t1, t2 in log:
    print(f"  {1e6 * (t0 - t_start):9.0f} {1e6 * (t1 - t0):7.0f} {1e6 * (t2 - t1):7.0f}")
res.close()
