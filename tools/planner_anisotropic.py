#!/usr/bin/env python3
"""Which kernel do the axis passes of a light take on anisotropic CT shapes? The planner alone (tbrm_host_plan_light: no device), 1000
random light directions per shape (seeded), the volume as a cube in the world (anisotropic voxels, what a CT series with thick slices
is) and with its extent proportional to its dimensions (isotropic voxels). Output: profiles/r06_planner_anisotropic.txt."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

WHY = {0: "-", 1: "taps on both sides of the pixel", 2: "reach beyond 14 texels", 3: "more hand-off words than a lane carries", 4: "short ragged downward pass",
       5: "sweeps off", 6: "more than 1024 slices"}
PATH = {0: "sweep", 1: "chunked chain", 2: "slice per launch"}
rng = np.random.default_rng(0x5EED0600)
dirs = rng.normal(size=(1000, 3))
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
print(__doc__.strip().splitlines()[0])
for dims in ((512, 512, 512), (512, 512, 128), (512, 512, 300), (256, 256, 90), (1024, 1024, 200)):
    for shape in ("cube in the world (anisotropic voxels)", "extent proportional to the dimensions (isotropic voxels)"):
        scale = (100.0, 100.0, 100.0) if shape.startswith("cube") else tuple(100.0 * d / max(dims) for d in dims)
        world = S.default_world()
        world.volume_transform.scale3d = abi.Vec3d(*scale)
        paths = collections.Counter()
        why = collections.Counter()
        reach = collections.Counter()
        n_pass = 0
        lights_all_sweep = 0
        for d in dirs:
            light = abi.DirLightParams(tuple(float(x) for x in d), 1.0)
            plan = abi.host_plan_light(light, world, dims)
            n_pass += len(plan)
            lights_all_sweep += all(p[0] == 0 for p in plan)
            for k, (path, a, b, w) in enumerate(plan):
                paths[(k, path)] += 1
                if path != 0:
                    why[WHY[w]] += 1
                else:
                    reach[max(a, b)] += 1
        print(f"\n{dims[0]} x {dims[1]} x {dims[2]}, {shape}: {n_pass} passes of 1000 lights; every pass of the light on the sweep: {lights_all_sweep / 10:.1f} % of the lights")
        for k in (0, 1):
            tot = sum(v for (kk, _), v in paths.items() if kk == k)
            if tot:
                print(f"   pass {k} ({'the major axis' if k == 0 else 'the second axis'}): " + ", ".join(f"{PATH[p]} {100.0 * v / tot:.1f} %" for (kk, p), v in sorted(paths.items()) if kk == k))
        if why:
            print("   why the sweep declined: " + ", ".join(f"{w}: {v}" for w, v in why.most_common()))
        print("   reach of the swept passes (texels): " + ", ".join(f"{r}: {v}" for r, v in sorted(reach.items())))
