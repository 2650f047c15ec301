#!/usr/bin/env python3
"""GPU time of ONE slab's share of a fused ChangeDirLight at BASELINE config 3's size (or argv[1]^3: 1024 = config 4), for 1/2/4/8 slabs, measured on one
GPU: the chunks of one member are enqueued back to back (no exchange — the planes' halo rows then hold stale values, which
changes no timing) and timed with HIP events on the handle's stream. Also prints the bytes a real run exchanges per
operation (from the emulated run with every slab in one process). Output: a table for DESIGN.md §6."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, slabs, synthetic as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dims = (n, n, n)
dev = torch.device("cuda", 0)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), dev)
torch.cuda.synchronize()
lut = abi.color_curve_to_lut(S.tf_keys("A"))
win = abi.WindowingParams(0.5, 0.9, True, False)
world = S.default_world()


def handle():
    r = abi.Resources(dims, abi.FMT_G16, False, False, 0)
    r.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    r.set_tf_lut(lut)
    r.set_windowing(win)
    r.clear_light_volume(0.0)
    for i in range(4):
        r.add_dir_light(S.light(i), True, world)
    r.flush()
    return r


res = handle()
cases = [("L1 5 deg about z (passes along y and x: both lateral)", 1), ("L2 5 deg about z (passes along z and y: pipeline + lateral)", 2)]
print(f"{'operation':62s} {'slabs':>5s} {'slab':>4s} {'ms of this slab':>16s} {'chunks':>7s}")
for name, li in cases:
    old = S.light(li)
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], 5.0), S.LIGHTS[li][1])
    for n_slabs in (1, 2, 4, 8):
        bounds = slabs.slab_bounds(n, n_slabs)
        for k in sorted({0, n_slabs // 2}):
            m = slabs.DeviceSlab(res, k, *bounds[k])
            times = []
            for rep in range(4):
                with m.stream_context():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n_passes = m.light_begin(old, new, True, world)
                    e0.record()
                    chunks = 0
                    for i in range(n_passes):
                        d = m.pass_begin(i)
                        for c in range(d.n_chunks):
                            m.pass_chunk(c)
                        chunks += d.n_chunks
                    e1.record()
                    e1.synchronize()
                    times.append(e0.elapsed_time(e1))
            print(f"{name:62s} {n_slabs:5d} {k:4d} {min(times[1:]):16.3f} {chunks:7d}", flush=True)
res.close()

# bytes exchanged per operation (all slabs in one process)
for n_slabs in (2, 8):
    parts = [handle() for _ in range(n_slabs)]
    bounds = slabs.slab_bounds(n, n_slabs)
    members = [slabs.DeviceSlab(r, k, *bounds[k]) for k, r in enumerate(parts)]
    for name, li in cases:
        fabric = slabs.make_fabric([b[0] for b in bounds] + [n])
        old = S.light(li)
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], 5.0), S.LIGHTS[li][1])
        slabs.change_dir_light(members, fabric, old, new, world)
        print(f"{name}: {n_slabs} slabs exchange {fabric.bytes_moved / 2**20:.1f} MiB in total per operation", flush=True)
    for r in parts:
        r.close()
