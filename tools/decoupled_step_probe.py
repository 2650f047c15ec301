"""Upper bound on what a frame that does not wait for the light update (a second light-volume buffer) could gain: two handles on
one GPU hold config 3's scene; one renders frames, the other turns a light 5 degrees per call (the benchmark's Change). Wall time
per round of each alone, of both enqueued together (no dependency between them), and of one handle doing both in turn (the
benchmark's step). Diagnostics, not product."""
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


class Scene:
    def __init__(self):
        self.res = abi.Resources((n, n, n), abi.FMT_G16)
        torch.cuda.synchronize()
        self.res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
        self.res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys(cfg["tf"])))
        self.res.set_windowing(abi.WindowingParams(*cfg["window"]))
        self.lights = [S.light(i) for i in cfg["lights"]]
        self.dirs = [S.LIGHTS[i][0] for i in cfg["lights"]]
        self.angle = [0.0] * len(self.lights)
        for l in self.lights:
            self.res.add_dir_light(l, True, world)
        self.res.flush()
        self.k = 0

    def change(self):
        li = self.k % len(self.lights)
        self.k += 1
        self.angle[li] += 5.0
        new = abi.DirLightParams(S.rotate_z(self.dirs[li], self.angle[li]), self.lights[li].light_intensity)
        self.res.change_dir_light(self.lights[li], new, world)
        self.lights[li] = new


a, b = Scene(), Scene()
cam = S.default_camera(cfg["fb"], cfg["fb"])
tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")
N = 20


def run(frames_on, lights_on):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        if lights_on is not None:
            lights_on.change()
        if frames_on is not None:
            frames_on.res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    a.res.flush()
    b.res.flush()
    return 1e3 * (time.perf_counter() - t0) / N


for _ in range(2):
    run(a, b)
    run(a, a)
f = min(run(a, None) for _ in range(3))
l = min(run(None, b) for _ in range(3))
both = min(run(a, b) for _ in range(3))
one = min(run(a, a) for _ in range(3))
print(f"frame alone {f:.3f} ms, Change alone (pipelined) {l:.3f} ms, two handles enqueued together {both:.3f} ms per round, one handle doing both in turn "
      f"(the benchmark's step) {one:.3f} ms")
a.res.close()
b.res.close()
