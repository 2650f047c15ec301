#!/usr/bin/env python3
"""ResetAllLights at 512^3 with 4 and 8 config lights: light by light (one tbrm_add_dir_light per light, enqueued back to back,
one sync at the end) vs tbrm_add_dir_lights (lights two at a time, passes that leave the same cube face in ONE sweep), cold
(nothing cached: a window nobody has used) and warm (the lights' factors kept), and the reported schedule. Diagnostics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = 512
dims = (n, n, n)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0))
torch.cuda.synchronize()
res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys("A")))
world = S.default_world()
stream = torch.cuda.ExternalStream(res.stream())
win_k = [0]


def stale_window():
    win_k[0] += 1
    c = np.float32(0.5)
    for _ in range(win_k[0]):
        c = np.nextafter(c, np.float32(2.0))
    res.set_windowing(abi.WindowingParams(float(c), 0.9, True, False))


def timed(fn):
    res.flush()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    out = fn()
    e1.record(stream)
    e1.synchronize()
    return float(e0.elapsed_time(e1)), out


for nl in (4, 8):
    lights = [S.light(i) for i in range(nl)]

    def one_by_one():
        res.clear_light_volume(0.0)
        for l in lights:
            res.add_dir_light(l, True, world)

    def batched():
        res.clear_light_volume(0.0)
        return res.add_dir_lights(lights, True, world)

    rows = {}
    for name, fn in (("light by light", one_by_one), ("batched", batched)):
        cold, warm = [], []
        for rep in range(3):
            stale_window()
            res.flush()
            t, sched = timed(fn)
            cold.append(t)
            t, sched = timed(fn)
            warm.append(t)
        rows[name] = (min(cold), min(warm), sched)
    s = rows["batched"][2]
    print(f"{nl} lights: light by light cold {rows['light by light'][0]:.3f} warm {rows['light by light'][1]:.3f} ms; batched cold {rows['batched'][0]:.3f} "
          f"warm {rows['batched'][1]:.3f} ms ({len(s)} sweeps, {sum(1 for e in s if e[2] >= 0)} pairs): {s}")
print(res.path_counters())
