#!/usr/bin/env python3
"""Chained sweeps (tunable sweep_chain: passes per k_light_sweep_chain launch) against one launch per pass, config 3's scene:
the benchmark's Changes (a light turned 5 degrees per call, synchronised per call and pipelined), ResetAllLights light by light
and as one tbrm_add_dir_lights call, warm and cold — GPU time by events on the library's stream — and the light volumes of the two
forms compared byte for byte after the same sequence. Diagnostics (profiles/r06_sweep_chain.txt)."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, sharding, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
dims = (n, n, n)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0))
torch.cuda.synchronize()
world = S.default_world()
variants = [int(v) for v in os.environ.get("CHAINS", "1,4").split(",")]
digests = {}
for chain in variants:
    abi.set_tunable("sweep_chain", chain)
    res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys("A")))
    res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
    res.reserve(4)
    stream = torch.cuda.ExternalStream(res.stream())
    lights = [S.light(i) for i in range(4)]
    dirs = [S.LIGHTS[i][0] for i in range(4)]
    angle = [0.0] * 4
    win_k = [0]

    def stale_window():
        win_k[0] += 1
        c = np.float32(0.5)
        for _ in range(win_k[0]):
            c = np.nextafter(c, np.float32(2.0))
        res.set_windowing(abi.WindowingParams(float(c), 0.9, True, False))

    def timed(fn, reps=3, before=None):
        best = []
        for _ in range(reps):
            if before:
                before()
            res.flush()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            e1.synchronize()
            best.append(float(e0.elapsed_time(e1)))
        return min(best), float(np.median(best))

    def reset():
        res.clear_light_volume(0.0)
        for l in lights:
            res.add_dir_light(l, True, world)

    def reset_batched():
        res.clear_light_volume(0.0)
        res.add_dir_lights(lights, True, world)

    def change(k):
        li = k % 4
        angle[li] += 5.0
        new = abi.DirLightParams(S.rotate_z(dirs[li], angle[li]), lights[li].light_intensity)
        res.change_dir_light(lights[li], new, world)
        lights[li] = new

    reset()
    res.flush()
    for k in range(8):
        change(k)
    res.flush()
    per_call = [timed(lambda: change(100 + k), reps=1)[0] for k in range(24)]
    res.flush()
    t0 = time.perf_counter()
    for k in range(40):
        change(200 + k)
    res.flush()
    pipelined = (time.perf_counter() - t0) / 40 * 1e3
    c0 = res.path_counters()
    rows = {
        "change, synchronised per call (mean / median of 24)": (float(np.mean(per_call)), float(np.median(per_call))),
        "changes back to back, wall per call": (pipelined, pipelined),
        "reset light by light, warm": timed(reset),
        "reset light by light, cold": timed(reset, before=stale_window),
        "reset batched, warm": timed(reset_batched),
        "reset batched, cold": timed(reset_batched, before=stale_window),
    }
    c1 = res.path_counters()
    res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
    reset()
    for k in range(12):
        change(300 + k)
    res.flush()
    lv = sharding.device_light_tensor(res).clone()
    torch.cuda.synchronize()
    digests[chain] = hashlib.sha1(lv.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"== sweep_chain = {chain}: light volume after the sequence sha1 {digests[chain]}; chain launches in the reset rows {c1['launches_sweep_chain'] - c0['launches_sweep_chain']}, "
          f"sweep launches {c1['launches_sweep'] - c0['launches_sweep']}")
    for name, (best, med) in rows.items():
        print(f"   {name:58s} {best:7.3f} ms (median {med:7.3f})")
    res.close()
print("light volumes identical across the variants:", len(set(digests.values())) == 1)
