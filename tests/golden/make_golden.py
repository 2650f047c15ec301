"""Regenerates tests/golden/scene_*.npz with the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference has no golden vectors for this path, so these fixtures pin the oracle against itself
(regression) and give the GPU tests a second, committed target. Inputs are produced by the deterministic synthetic
generators; only the outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

SCENES = {
    # BASELINE config 1 in miniature: float data, float light volume, default window, one light
    "mini_config1": dict(dims=(32, 32, 32), dtype="float32", light_32bit=True, half_res=False, tf="A",
                         window=(0.5, 1.0, True, True), lights=[0], change=None, fb=(64, 64), steps=48.0, jitter=-1, seed=1),
    # configs 2/3 in miniature: UNORM16 data, UNORM8 light volume, windowing with one cutoff, 4 lights + 1 update
    "mini_config3": dict(dims=(40, 36, 44), dtype="uint16", light_32bit=False, half_res=False, tf="A",
                         window=(0.5, 0.9, True, False), lights=[0, 1, 2, 3], change=(1, 5.0), fb=(80, 64), steps=64.0, jitter=-1, seed=3),
    # config 5 in miniature: bone TF (early termination, empty space), half-resolution light volume, jitter on
    "mini_config5": dict(dims=(48, 48, 48), dtype="uint16", light_32bit=False, half_res=True, tf="B",
                         window=(0.5, 0.8, True, True), lights=[0, 2, 5], change=None, fb=(64, 64), steps=80.0, jitter=3, seed=5),
}


def build(name, make_scene):
    """make_scene(volume, cfg) -> object with the OracleScene/Resources operator methods."""
    cfg = SCENES[name]
    vol = S.make_volume_numpy(cfg["dims"], np.dtype(cfg["dtype"]), S.seed_for_config(cfg["seed"]))
    sc = make_scene(vol, cfg)
    sc.set_tf_lut(abi.color_curve_to_lut(S.tf_keys(cfg["tf"])))
    sc.set_windowing(abi.WindowingParams(*cfg["window"]))
    world = S.default_world()
    for i in cfg["lights"]:
        sc.add_dir_light(S.light(i), True, world)
    if cfg["change"]:
        i, deg = cfg["change"]
        sc.change_dir_light(S.light(i), abi.DirLightParams(S.rotate_z(S.LIGHTS[i][0], deg), S.LIGHTS[i][1]), world)
    cam = S.default_camera(*cfg["fb"])
    tile = abi.Tile(0, 0, cfg["fb"][0], cfg["fb"][1])
    rp = abi.RaymarchParams(cfg["steps"], cfg["jitter"], True)
    return sc, cam, tile, rp, world


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name in SCENES:
        sc, cam, tile, rp, world = build(name, lambda v, c: oracle.OracleScene(v, c["light_32bit"], c["half_res"]))
        img, n = sc.raymarch_lit(cam, tile, rp, world)
        np.savez_compressed(os.path.join(here, f"scene_{name}.npz"), light=sc.light, image=img, nominal_samples=np.int64(n))
        print(name, sc.light.shape, sc.light.dtype, img.shape, n, float(img[..., 3].mean()))


if __name__ == "__main__":
    main()
