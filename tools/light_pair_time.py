#!/usr/bin/env python3
"""Batched multi-light add (tbrm_add_dir_lights), evidence for the pairing rule: GPU time of every pair of the 8 config lights at 512^3, light by light vs forced into one slice loop (tunable light_batching = 2), with each pass's cube face and previous-slice tap ranges."""
import sys, numpy as np, torch, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbraymarcherplugin_amd import abi, synthetic as S
n = 512; dims = (n, n, n)
vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0)); torch.cuda.synchronize()
res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
res.upload_volume_device(vol.data_ptr(), vol.numel()*2); res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys("A"))); res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
world = S.default_world()
lights = [S.light(i) for i in range(8)]
def rng(size, off):
    import math
    lo, hi = 10**9, -10**9
    for c in (0, size//2, size-1):
        u = np.float32(np.float32(np.float32(c)+np.float32(0.5))/np.float32(size)) + np.float32(off)
        x = np.float32(u*np.float32(size)) - np.float32(0.5)
        d = int(math.floor(x)) - c
        lo=min(lo,d); hi=max(hi,d+1)
    return lo,hi
def single(l):
    best = 1e9
    for rep in range(3):
        res.add_dir_light(l, True, world); res.flush(); best = min(best, res.last_gpu_time_ms(0))
    return best
def batch(ls):
    best = 1e9
    for rep in range(3):
        sched = res.add_dir_lights(ls, True, world); res.flush(); best = min(best, res.last_gpu_time_ms(0))
    return best, sched
res.clear_light_volume(0.0)
ts = [single(l) for l in lights]
info = []
for i, l in enumerate(lights):
    ps, k = abi.host_light_passes(l, world, dims)
    info.append([(p.face, rng(p.td[0], p.prev_pixel_offset[0]), rng(p.td[1], p.prev_pixel_offset[1])) for p in ps[:k]])
    print(i, f"{ts[i]:.3f} ms", info[-1])
abi.set_tunable("light_batching", 2)
for i, j in itertools.combinations(range(8), 2):
    tb, sched = batch([lights[i], lights[j]])
    pairs = [s for s in sched if s[2] >= 0]
    if pairs:
        print(f"lights {i},{j}: separate {ts[i]+ts[j]:.3f} paired {tb:.3f} gain {ts[i]+ts[j]-tb:+.3f} ms; pairs {pairs}")
