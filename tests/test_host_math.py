"""Host parameter math of the product (tbrm_host_*) against the oracle's independent restatement and against
hand-derived known answers (LightingShaderUtils.cpp:29-265, LightingShaders.cpp:100-131, RaymarchUtils.cpp:113-174)."""
import json
import os

import numpy as np
import pytest

from tbraymarcherplugin_amd import abi, synthetic as S

HERE = os.path.dirname(os.path.abspath(__file__))


def random_world(rng, clip=True):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    tr = abi.identity_transform(scale=tuple(rng.uniform(40, 250, 3)), translation=tuple(rng.uniform(-50, 50, 3)), rotation=tuple(q))
    if clip:
        return abi.make_world(tr, clip_center=tuple(rng.uniform(-30, 30, 3)), clip_direction=tuple(rng.normal(size=3)))
    return abi.make_world(tr)


def passes_equal(a, b):
    da, db = a.as_dict(), b.as_dict()
    for k in da:
        va, vb = np.asarray(da[k], dtype=np.float64), np.asarray(db[k], dtype=np.float64)
        if not np.array_equal(va, vb, equal_nan=True):
            return False, k
    return True, None


def test_light_passes_match_oracle_on_random_inputs(oracle_mod):
    rng = np.random.default_rng(1234)
    for trial in range(300):
        world = random_world(rng)
        light = abi.DirLightParams(tuple(rng.normal(size=3)), float(rng.uniform(0.05, 1.5)))
        dims = tuple(int(v) for v in rng.integers(5, 700, 3))
        mode = int(trial % 2)
        got, n_got = abi.host_light_passes(light, world, dims, mode)
        ref, n_ref = oracle_mod.light_passes(light, world, dims, mode)
        assert n_got == n_ref
        for i in range(2):
            ok, field = passes_equal(got[i], ref[i])
            assert ok, (trial, i, field, got[i].as_dict(), ref[i].as_dict())
        cg, dg = abi.host_local_clipping(world)
        cr, dr = oracle_mod.local_clipping(world)
        assert np.array_equal(cg, cr) and np.array_equal(dg, dr)
        assert np.array_equal(abi.host_world_to_local(world.volume_transform), oracle_mod.world_to_local(world.volume_transform))


def test_major_axes_known_answers():
    world = S.default_world()
    dims = (64, 64, 64)
    # light shining along +X comes from -X: one pass on face -X, weight exactly 1, iterating upwards
    p, n = abi.host_light_passes(abi.DirLightParams((1, 0, 0), 0.5), world, dims)
    assert n == 1 and p[0].face == 1 and p[0].axis == 0 and p[0].weight == 1.0 and p[0].light_alpha == 0.5
    assert (p[0].start, p[0].stop, p[0].dir) == (0, 64, 1) and list(p[0].td) == [64, 64, 64]
    assert list(p[0].prev_pixel_offset) == [0.0, 0.0] and p[0].step_size == pytest.approx(1 / 64)
    assert list(p[0].uvw_offset) == pytest.approx([-1 / 64, 0, 0])
    assert p[1].weight == 0.0 and p[1].face == 0  # ties resolved by ascending face index
    # towards -Z: face +Z (even face: iterates downwards from TD.Z-1)
    p, n = abi.host_light_passes(abi.DirLightParams((0, 0, -1), 1.0), world, (32, 48, 16))
    assert n == 1 and p[0].face == 4 and (p[0].start, p[0].stop, p[0].dir) == (15, -1, -1) and list(p[0].td) == [32, 48, 16]
    # 45 degrees: weights 0.5 / 0.5 (second = 1 - first), two passes
    p, n = abi.host_light_passes(abi.DirLightParams((1, 1, 0), 1.0), world, dims)
    assert n == 2 and {p[0].face, p[1].face} == {1, 3}
    assert p[0].weight == pytest.approx(0.5, abs=1e-6) and p[1].weight == np.float32(1) - np.float32(p[0].weight)
    # X pass of a (1,1,0) light (it sits at (-1,-1,0)): TD = (Y,Z,X); face -X divides by -x, so the previous-slice
    # offset is (y, z)/|x| / TD.Z = (-1, 0)/64: towards the light
    px = p[0] if p[0].face == 1 else p[1]
    assert list(px.prev_pixel_offset) == pytest.approx([-1 / 64, 0.0])
    assert px.step_size == pytest.approx(np.sqrt(2) / 64, rel=1e-6)
    # first-axis weight above 0.99 snaps to 1 and the second pass disappears
    p, n = abi.host_light_passes(abi.DirLightParams((1, 0.05, 0), 1.0), world, dims)
    assert n == 1 and p[0].weight == 1.0 and p[1].weight == 0.0
    # zero direction: nothing to do
    p, n = abi.host_light_passes(abi.DirLightParams((0, 0, 0), 1.0), world, dims)
    assert n == 0
    # non-cubic volume: both offset components are divided by TD.Z (the reference's "incorrect but consistent" rule)
    p, n = abi.host_light_passes(abi.DirLightParams((0.5, 0.25, -1), 1.0), world, (40, 20, 10))
    assert p[0].face == 4 and list(p[0].td) == [40, 20, 10]
    assert list(p[0].prev_pixel_offset) == pytest.approx([-0.5 / 10, -0.25 / 10])
    assert np.linalg.norm(p[0].uvw_offset) == pytest.approx(1 / 10)  # renormalised to 1/min(TD)


def test_border_colours_and_clipping_defaults():
    w = abi.WindowingParams(0.5, 0.9, True, False)
    assert abi.host_data_border(w, abi.BORDER_EXACT_FLOAT) == pytest.approx(0.05)
    assert abi.host_data_border(w, abi.BORDER_ENGINE_8BIT) == np.float32(13) / np.float32(255)  # round(0.05*255) = 13
    assert abi.host_data_border(abi.WindowingParams(0.2, 1.0), abi.BORDER_ENGINE_8BIT) == 0.0   # clamped below 0
    world = S.default_world()
    c, d = abi.host_local_clipping(world)  # "no clip plane" defaults, scale 100
    assert np.allclose(c, [0.5, 0.5, 1000.5]) and np.array_equal(d, [0, 0, -1])
    p, _ = abi.host_light_passes(abi.DirLightParams((1, 0, 0), 0.4), world, (8, 8, 8), abi.BORDER_ENGINE_8BIT)
    # sRGB-8-bit round trip of 0.4: encode 0.6652 -> 170/255 -> decode 0.4020
    assert p[0].border_light == pytest.approx(0.40198, abs=2e-5)
    p, _ = abi.host_light_passes(abi.DirLightParams((1, 0, 0), 0.4), world, (8, 8, 8), abi.BORDER_EXACT_FLOAT)
    assert p[0].border_light == np.float32(0.4)


def test_transfer_function_construction(oracle_mod):
    # default TF: grey ramp i/255, alpha 1 (MakeDefaultTFTexture)
    d = abi.make_default_tf_lut()
    assert np.array_equal(d, oracle_mod.make_default_tf_lut())
    assert d[0].tolist() == [0, 0, 0, 1] and d[255].tolist() == [1, 1, 1, 1] and d[51, 0] == np.float32(51) / np.float32(255)
    # FFloat16 storage == numpy.float16 rounding
    rng = np.random.default_rng(7)
    lut = rng.uniform(-0.2, 1.3, (256, 4)).astype(np.float32)
    lut[0] = [0, 1, 65504.0, 1e-8]
    baked = abi.host_bake_tf_lut(lut)
    assert np.array_equal(baked, lut.astype(np.float16).astype(np.float32))
    assert np.array_equal(baked, oracle_mod.bake_tf(lut))
    # curves recovered from the reference's own assets (tests/golden/tf_curves.json): piecewise-linear keys at i/255
    curves = json.load(open(os.path.join(HERE, "golden", "tf_curves.json")))
    assert "TF_CT-Bone" in curves and len(curves) >= 20
    for name, c in curves.items():
        assert c["interp_modes"] == [0], name  # every shipped key is RCIM_Linear
        keys = [(ch["times"], ch["values"]) for ch in c["channels"]]
        got = abi.color_curve_to_lut(keys)
        assert np.array_equal(got, oracle_mod.color_curve_to_lut(keys)), name
        for ch in range(4):  # independent check: numpy's piecewise-linear interpolation
            t = np.asarray(keys[ch][0], dtype=np.float64)
            v = np.asarray(keys[ch][1], dtype=np.float64)
            want = np.interp(np.arange(256) / 255.0, t, v)
            assert np.abs(got[:, ch] - want).max() < 1e-5, (name, ch)  # fp32 vs fp64 evaluation
    bone = abi.color_curve_to_lut([(ch["times"], ch["values"]) for ch in curves["TF_CT-Bone"]["channels"]])
    assert bone[:126, 3].max() == 0.0 and bone[200, 3] == pytest.approx(0.7112, abs=1e-3)  # alpha = 0 below 0.4934
    # the bench's TF-B keys are those of TF_CT-Bone
    tfb = abi.color_curve_to_lut(S.TF_B_KEYS)
    assert np.array_equal(tfb, bone)


def test_split_reciprocal_decode_is_exact():
    """The kernels decode UNORM codes as fma(c, r, c*r2) with r = RN(1/d), r2 = RN(1/d - r). In exact rational arithmetic
    with round-to-nearest-even to binary32 after each operation, that equals RN(c/d) for every 8- and 16-bit code — the
    constants in tbrm_device_math.h are checked here too."""
    from fractions import Fraction

    def rn(fr):  # Fraction -> nearest binary32 (normal range), as a Fraction
        if fr == 0:
            return Fraction(0)
        sign, fr = (-1, -fr) if fr < 0 else (1, fr)
        e = fr.numerator.bit_length() - fr.denominator.bit_length()
        if Fraction(2) ** e > fr:
            e -= 1
        q = fr / Fraction(2) ** (e - 23)
        n, rem = divmod(q.numerator, q.denominator)
        twice = 2 * rem
        if twice > q.denominator or (twice == q.denominator and n % 2 == 1):
            n += 1
        return sign * n * Fraction(2) ** (e - 23)

    for d, r_hex, r2_hex in ((255, "0x1.010102p-8", "-0x1.fdfdfep-33"), (65535, "0x1.0001p-16", "0x1.0001p-48")):
        r, r2 = Fraction(float.fromhex(r_hex)), Fraction(float.fromhex(r2_hex))
        assert r == rn(Fraction(1, d)) and r2 == rn(Fraction(1, d) - r)
        for c in range(d + 1):
            t = rn(c * r2)
            assert rn(c * r + t) == rn(Fraction(c, d)), (d, c)


def test_planner_alone_says_which_kernel_a_pass_takes():
    """tbrm_host_plan_light (no device): the passes of config 3's lights at 512^3 all take the pipelined sweep; a light that grazes a
    thin volume's large face reaches too far for it (the reason is reported)"""
    world = S.default_world()
    for i in range(4):
        plan = abi.host_plan_light(S.light(i), world, (512, 512, 512))
        assert len(plan) == 2 and all(p[0] == 0 and 1 <= max(p[1], p[2]) <= 14 and p[3] == 0 for p in plan), plan
    plan = abi.host_plan_light(abi.DirLightParams((1.0, 0.3, 0.9), 1.0), world, (512, 512, 64))
    assert plan[0][0] == 0, plan                      # the major axis sweeps (its taps lie within a texel or two)
    assert plan[1][0] in (1, 2) and plan[1][3] in (2, 3), plan  # the second axis: reach beyond 14 texels / too many hand-off words
    assert abi.host_plan_light(abi.DirLightParams((0.0, 0.0, 0.0), 1.0), world, (64, 64, 64)) == []  # zero direction: no pass
