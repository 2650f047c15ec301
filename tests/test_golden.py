"""Committed golden fixtures (tests/golden/scene_*.npz, written by tests/golden/make_golden.py with the oracle):
the oracle must keep reproducing them bit-for-bit, and the HIP path must match them on the GPU."""
import importlib.util
import os

import numpy as np
import pytest

from tbraymarcherplugin_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(spec)
spec.loader.exec_module(make_golden)


def load(name):
    return np.load(os.path.join(HERE, "golden", f"scene_{name}.npz"))


@pytest.mark.parametrize("name", list(make_golden.SCENES))
def test_oracle_reproduces_golden(oracle_mod, name):
    want = load(name)
    sc, cam, tile, rp, world = make_golden.build(name, lambda v, c: oracle_mod.OracleScene(v, c["light_32bit"], c["half_res"]))
    img, n = sc.raymarch_lit(cam, tile, rp, world)
    assert np.array_equal(sc.light, want["light"])
    assert np.array_equal(img, want["image"])
    assert n == int(want["nominal_samples"])
    assert want["image"][..., 3].max() > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(make_golden.SCENES))
def test_hip_path_matches_golden(gpu, name):
    want = load(name)

    def make(vol, cfg):
        res = abi.Resources(cfg["dims"], abi.DTYPE_FMT[vol.dtype], cfg["light_32bit"], cfg["half_res"])
        res.upload_volume(vol)
        return res

    res, cam, tile, rp, world = make_golden.build(name, make)
    with res:
        lv = res.download_light_volume()
        img = res.raymarch_lit(cam, tile, rp, world)
        n = res.count_nominal_samples(cam, tile, rp, world)
    if lv.dtype == np.uint8:
        assert np.array_equal(lv, want["light"]), f"{np.count_nonzero(lv != want['light'])} UNORM8 voxels differ"
    else:
        assert np.abs(lv - want["light"]).max() <= 2e-6
    assert np.abs(img - want["image"]).max() <= 2e-6  # north_star tolerance: 1e-4
    assert n == int(want["nominal_samples"])
