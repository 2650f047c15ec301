"""HIP path vs CPU oracle on identical seeded inputs, through the C-ABI.

Bar: the UNORM8 light volume is bit-exact; float light volumes and RGBA within 1e-4 (BASELINE.json north_star).
The tolerance asserted for floats is tighter (2e-6) because both sides evaluate the same fp32 sequence.

Tolerance model: these gates show that the kernels evaluate the build's ONE arithmetic definition exactly as
the oracle does. The definition restates the reference's shaders; where those lean on hardware (D3D11 fixed-point filter
weights, the GPU's pow, sRGB border colours) nothing in this image can produce the hardware's output, so "bit-exact" is
self-consistency. A frame or light volume captured from the reference itself would be a separate fixture, compared with
a tolerance chosen for hardware filtering (about one UNORM8 code).
"""
import numpy as np
import pytest

from conftest import small_volume
from tbraymarcherplugin_amd import abi, synthetic as S

pytestmark = pytest.mark.gpu

RGBA_TOL = 1e-4      # north_star tolerance
TIGHT_TOL = 2e-6     # what the shared arithmetic spec actually delivers


def make_pair(gpu, oracle_mod, dims, dtype, light_32bit=False, half_res=False, addr=abi.ADDRESS_WRAP,
              border=abi.BORDER_ENGINE_8BIT, tf="A", window=(0.5, 0.9, True, False), seed=0x5EED0002):
    vol = small_volume(dims, dtype, seed)
    res = abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)], light_32bit, half_res, 0, addr, border)
    orc = oracle_mod.OracleScene(vol, light_32bit, half_res, addr, border)
    lut = abi.color_curve_to_lut(S.tf_keys(tf))
    w = abi.WindowingParams(*window)
    res.upload_volume(vol)
    res.set_tf_lut(lut)
    res.set_windowing(w)
    orc.set_tf_lut(lut)
    orc.set_windowing(w)
    return res, orc


def assert_light_equal(res, orc):
    got = res.download_light_volume()
    if got.dtype == np.uint8:
        diff = np.count_nonzero(got != orc.light)
        assert diff == 0, f"{diff} of {got.size} UNORM8 light voxels differ (max |d| = {np.abs(got.astype(int) - orc.light.astype(int)).max()})"
    else:
        np.testing.assert_allclose(got, orc.light, rtol=0, atol=TIGHT_TOL)


@pytest.fixture(params=["sweep", "chunk", "slice"])
def kernel_variant(request, tunables):
    """Runs a test with the production kernels (the pipelined sweep wherever it applies: UNORM8 light volumes, passes of whole
    brick layers; else the chunked chain), with the chunked chain alone, and with the one-slice-per-launch kernel."""
    tunables("force_slice_kernel", 1 if request.param == "slice" else 0)
    tunables("light_sweep", 1 if request.param == "sweep" else 0)
    return request.param


@pytest.fixture(params=["4", "8"])
def ray_lanes(request, tunables):
    """Runs a raymarch test with both lanes-per-ray variants of k_raymarch_lit (the launcher picks by frame size otherwise)."""
    tunables("ray_lanes", int(request.param))
    return request.param


FACE_LIGHTS = [((1, .35, -.5), 0.5), ((-1, .2, .4), 0.6), ((.3, 1, -.2), 0.5), ((.25, -1, .5), 0.7),
               ((.1, .45, 1), 0.5), ((-.35, .2, -1), 0.9), ((1, 0, 0), 0.5), ((0, 0, -1), 0.8), ((1, 1, 0), 0.6)]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("light_32bit", [False, True])
def test_add_dir_light_all_faces(gpu, oracle_mod, dtype, light_32bit, kernel_variant):
    res, orc = make_pair(gpu, oracle_mod, (40, 36, 44), dtype, light_32bit)
    world = S.default_world()
    with res:
        for d, inten in FACE_LIGHTS:
            light = abi.DirLightParams(d, inten)
            assert res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
            assert_light_equal(res, orc)
        # removing a light again (Added = false)
        light = abi.DirLightParams(*FACE_LIGHTS[1])
        res.add_dir_light(light, False, world)
        orc.add_dir_light(light, False, world)
        assert_light_equal(res, orc)
        counters = res.launch_counters()
        if kernel_variant != "slice":  # every one of these passes is within the chunk kernels' envelope
            assert counters["chunk"] > 0 and counters["slice"] == 0, counters
        else:
            assert counters["chunk"] == 0 and counters["slice"] > 0, counters


@pytest.mark.parametrize("light_32bit", [False, True])
def test_change_dir_light_fused_and_fallback(gpu, oracle_mod, light_32bit, kernel_variant):
    res, orc = make_pair(gpu, oracle_mod, (48, 48, 48), np.uint16, light_32bit)
    world = S.default_world()
    with res:
        lights = [S.light(i) for i in range(4)]
        for l in lights:
            res.add_dir_light(l, True, world)
            orc.add_dir_light(l, True, world)
        # small rotation: same major axes -> fused path
        old = lights[1]
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
        assert orc.change_dir_light(old, new, world) == 2
        res.change_dir_light(old, new, world)
        assert_light_equal(res, orc)
        # large rotation: axes differ -> remove + add
        old = lights[2]
        new = abi.DirLightParams((1.0, 0.1, -0.2), S.LIGHTS[2][1])
        assert orc.change_dir_light(old, new, world) == -1
        res.change_dir_light(old, new, world)
        assert_light_equal(res, orc)
        # intensity-only change, and an axis-aligned light (second axis weight 0 in the fused pass)
        old = abi.DirLightParams((0, 0, -1), 0.4)
        res.add_dir_light(old, True, world)
        orc.add_dir_light(old, True, world)
        new = abi.DirLightParams((0, 0, -1), 0.7)
        assert orc.change_dir_light(old, new, world) == 2
        res.change_dir_light(old, new, world)
        assert_light_equal(res, orc)


def test_half_resolution_and_clip_plane(gpu, oracle_mod, kernel_variant):
    res, orc = make_pair(gpu, oracle_mod, (45, 40, 37), np.uint16, False, half_res=True)
    # rotated, non-uniformly scaled volume and a clip plane through it
    tr = abi.identity_transform(scale=(100.0, 120.0, 80.0), translation=(10.0, -5.0, 3.0),
                                rotation=(0.1305262, 0.0, 0.0, 0.9914449))
    world = abi.make_world(tr, clip_center=(12.0, -2.0, 5.0), clip_direction=(0.3, -0.2, 0.93))
    with res:
        assert res.light_dims == (23, 20, 19)
        for i in (0, 3, 5):
            res.add_dir_light(S.light(i), True, world)
            orc.add_dir_light(S.light(i), True, world)
        assert_light_equal(res, orc)
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[3][0], -4.0), 0.35)
        res.change_dir_light(S.light(3), new, world)
        orc.change_dir_light(S.light(3), new, world)
        assert_light_equal(res, orc)


def test_clear_light_volume_and_flags(gpu, oracle_mod):
    res, orc = make_pair(gpu, oracle_mod, (16, 16, 16), np.uint8)
    with res:
        res.clear_light_volume(0.3)
        orc.clear_light_volume(0.3)
        assert_light_equal(res, orc)
        assert res.download_light_volume().flat[0] == 77  # trunc(0.3*255+0.5)
        # zero direction: no-op, LightAdded stays true (LightingShaders.cpp:41-46, RaymarchUtils.cpp:48)
        assert res.add_dir_light(abi.DirLightParams((0, 0, 0), 1.0), True, S.default_world()) is True
        assert_light_equal(res, orc)
    # uninitialised resources: LightAdded = false (RaymarchUtils.cpp:39-45)
    bare = abi.Resources((8, 8, 8), abi.FMT_G8)
    with bare:
        flag = abi.C.c_int(1)
        code = bare.lib.tbrm_add_dir_light(bare.handle, abi.C.byref(S.light(0)), 1, abi.C.byref(S.default_world()),
                                           abi.C.byref(flag), 0)
        assert code == abi.ERR_NOT_INITIALIZED and flag.value == 0


def lit_pair(gpu, oracle_mod, dims=(48, 48, 48), dtype=np.uint16, light_32bit=False, **kw):
    res, orc = make_pair(gpu, oracle_mod, dims, dtype, light_32bit, **kw)
    world = S.default_world()
    for i in (0, 2):
        res.add_dir_light(S.light(i), True, world)
        orc.add_dir_light(S.light(i), True, world)
    return res, orc, world


@pytest.mark.parametrize("dtype,light_32bit,addr", [(np.uint16, False, abi.ADDRESS_WRAP), (np.float32, True, abi.ADDRESS_WRAP),
                                                    (np.uint8, False, abi.ADDRESS_CLAMP)])
def test_raymarch_lit_matches_oracle(gpu, oracle_mod, dtype, light_32bit, addr, ray_lanes):
    res, orc, world = lit_pair(gpu, oracle_mod, (48, 40, 44), dtype, light_32bit, addr=addr)
    cam = S.default_camera(96, 80)
    tile = abi.Tile(0, 0, 96, 80)
    with res:
        for steps, jitter, skip in [(64.0, -1, False), (64.0, -1, True), (100.0, 3, True)]:
            rp = abi.RaymarchParams(steps, jitter, skip)
            got = res.raymarch_lit(cam, tile, rp, world)
            ref, n_ref = orc.raymarch_lit(cam, tile, rp, world)
            assert np.abs(got - ref).max() <= TIGHT_TOL <= RGBA_TOL
            assert ref[..., 3].max() > 0.5  # the scene is not empty
            assert res.count_nominal_samples(cam, tile, rp, world) == n_ref


def test_raymarch_skipping_is_exact_and_bone_tf(gpu, oracle_mod):
    res, orc, world = lit_pair(gpu, oracle_mod, (64, 64, 64), np.uint16, tf="B", window=(0.5, 0.8, True, True))
    cam = S.default_camera(128, 128)
    tile = abi.Tile(0, 0, 128, 128)
    with res:
        a = res.raymarch_lit(cam, tile, abi.RaymarchParams(96.0, -1, False), world)
        b = res.raymarch_lit(cam, tile, abi.RaymarchParams(96.0, -1, True), world)
        assert np.array_equal(a, b), "empty-space skipping changed the image"
        ref, _ = orc.raymarch_lit(cam, tile, abi.RaymarchParams(96.0, -1, True), world)
        assert np.abs(b - ref).max() <= TIGHT_TOL
        assert (ref[..., 3] == 1.0).any()  # early termination is exercised


def test_raymarch_clip_plane_tiles_and_depth(gpu, oracle_mod, ray_lanes):
    res, orc = make_pair(gpu, oracle_mod, (40, 40, 40), np.uint16)
    tr = abi.identity_transform(scale=(100.0, 100.0, 100.0), rotation=(0.0, 0.2588190, 0.0, 0.9659258))
    world = abi.make_world(tr, clip_center=(5.0, 0.0, 0.0), clip_direction=(0.5, 0.5, 0.7))
    cam = S.default_camera(64, 48)
    rp = abi.RaymarchParams(80.0, 1, True)
    with res:
        res.add_dir_light(S.light(1), True, world)
        orc.add_dir_light(S.light(1), True, world)
        full_ref, _ = orc.raymarch_lit(cam, abi.Tile(0, 0, 64, 48), rp, world)
        full = res.raymarch_lit(cam, abi.Tile(0, 0, 64, 48), rp, world)
        assert np.abs(full - full_ref).max() <= TIGHT_TOL
        # sub-rectangle and interleaved row groups reproduce the same pixels
        sub = res.raymarch_lit(cam, abi.Tile(16, 8, 24, 16), rp, world)
        assert np.array_equal(sub, full[8:24, 16:40])
        for rank in range(2):
            t = abi.Tile(0, 8 * rank, 64, 24, row_group_step=2)
            part = res.raymarch_lit(cam, t, rp, world)
            rows = [8 * rank + (j // 8) * 16 + (j % 8) for j in range(24)]
            assert np.array_equal(part, full[rows])
        # scene depth limits the exit point
        import torch
        depth = np.full((48, 64), 160.0, dtype=np.float32)
        d_dev = torch.from_numpy(depth).cuda()
        out = torch.empty((48, 64, 4), dtype=torch.float32, device="cuda")
        res.raymarch_lit_device(cam, abi.Tile(0, 0, 64, 48), rp, world, out.data_ptr(), d_dev.data_ptr())
        res.flush()
        ref_d, _ = orc.raymarch_lit(cam, abi.Tile(0, 0, 64, 48), rp, world, scene_depth=depth)
        assert np.abs(out.cpu().numpy() - ref_d).max() <= TIGHT_TOL
        assert not np.array_equal(ref_d, full_ref)


def test_raymarch_short_rays_and_odd_tiles(gpu, oracle_mod, ray_lanes):
    """The kernel evaluates 8 consecutive samples of a ray side by side and replays the accumulation in order: rays whose
    sample count falls on every side of a multiple of 8 (0, 1, 7, 8, 9, ...), fractional last steps, early exits in the
    middle of a group and tiles that are no multiple of the 8x4 pixel block must all match the serial oracle."""
    res, orc, world = lit_pair(gpu, oracle_mod, (32, 32, 32), np.uint16)
    with res:
        for w, h, x0, y0 in [(37, 23, 5, 3), (8, 4, 0, 0), (1, 1, 20, 17)]:
            cam = S.default_camera(64, 48)
            tile = abi.Tile(x0, y0, w, h)
            for steps in (0.75, 1.0, 4.3, 7.0, 7.999, 8.0, 8.5, 9.0, 15.9, 16.0, 17.25, 40.0):
                for jitter in (-1, 5):
                    rp = abi.RaymarchParams(steps, jitter, True)
                    got = res.raymarch_lit(cam, tile, rp, world)
                    ref, n_ref = orc.raymarch_lit(cam, tile, rp, world)
                    assert np.abs(got - ref).max() <= TIGHT_TOL, (w, h, steps, jitter)
                    assert res.count_nominal_samples(cam, tile, rp, world) == n_ref
        # a dense transfer function makes rays terminate early at many different sample indices
        dense = np.zeros((256, 4), dtype=np.float32)
        dense[:, :3] = np.linspace(0.2, 1.0, 256)[:, None]
        dense[:, 3] = np.linspace(0.0, 0.6, 256)
        res.set_tf_lut(dense)
        orc.set_tf_lut(dense)
        cam = S.default_camera(72, 56)
        tile = abi.Tile(0, 0, 72, 56)
        rp = abi.RaymarchParams(50.0, -1, True)
        got = res.raymarch_lit(cam, tile, rp, world)
        ref, _ = orc.raymarch_lit(cam, tile, rp, world)
        assert np.abs(got - ref).max() <= TIGHT_TOL
        assert (ref[..., 3] == 1.0).mean() > 0.05  # early exits are common among the rays that hit the cube


def test_raymarch_leaping_over_sparse_blobs(gpu, oracle_mod, ray_lanes):
    """Empty-space leaping (per-brick distance field): a volume that is empty except for a few small blobs — long leaps,
    blobs entered from empty space at every angle, wrap and clamp addressing — renders exactly as without skipping, and
    as the oracle renders it."""
    rng = np.random.default_rng(11)
    n = 96
    z, y, x = np.mgrid[0:n, 0:n, 0:n].astype(np.float32)
    vol = np.zeros((n, n, n), dtype=np.float32)
    for _ in range(9):
        c = rng.uniform(6, n - 6, 3)
        r = rng.uniform(2.5, 7.0)
        d2 = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2
        vol = np.maximum(vol, np.clip(1.2 - np.sqrt(d2) / r, 0.0, 1.0))
    vol[0:2, :, :] = 0.9  # something at the volume's face: leaps across the wrap seam must see it
    vol_u16 = (vol * 65535.0 + 0.5).astype(np.uint16)
    lut = abi.color_curve_to_lut(S.TF_A_KEYS)
    win = abi.WindowingParams(0.6, 0.7, True, False)
    world = S.default_world()
    for addr in (abi.ADDRESS_WRAP, abi.ADDRESS_CLAMP):
        res = abi.Resources((n, n, n), abi.FMT_G16, False, False, 0, addr, abi.BORDER_ENGINE_8BIT)
        orc = oracle_mod.OracleScene(vol_u16, False, False, addr, abi.BORDER_ENGINE_8BIT)
        with res:
            res.upload_volume(vol_u16)
            for o in (res, orc):
                o.set_tf_lut(lut)
                o.set_windowing(win)
                o.add_dir_light(S.light(0), True, world)
            cam = S.default_camera(160, 128)
            tile = abi.Tile(0, 0, 160, 128)
            for steps, jitter in ((96.0, -1), (333.0, 2)):
                a = res.raymarch_lit(cam, tile, abi.RaymarchParams(steps, jitter, False), world)
                b = res.raymarch_lit(cam, tile, abi.RaymarchParams(steps, jitter, True), world)
                assert np.array_equal(a, b), "leaping changed the image"
                assert b[..., 3].max() > 0.3
            ref, _ = orc.raymarch_lit(cam, tile, abi.RaymarchParams(96.0, -1, True), world)
            got = res.raymarch_lit(cam, tile, abi.RaymarchParams(96.0, -1, True), world)
            assert np.abs(got - ref).max() <= TIGHT_TOL


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_raymarch_intensity_matches_oracle(gpu, oracle_mod, dtype):
    """The Intensity render mode (PerformWindowedIntensityRaymarch): with and without a clip plane, jitter, tiles,
    scene depth; the clamp sampler is used whatever address mode the lit material has."""
    res, orc = make_pair(gpu, oracle_mod, (40, 36, 44), dtype, addr=abi.ADDRESS_WRAP, window=(0.45, 0.6, True, True))
    tr = abi.identity_transform(scale=(100.0, 120.0, 90.0), rotation=(0.0, 0.2588190, 0.0, 0.9659258))
    worlds = [S.default_world(), abi.make_world(tr, clip_center=(5.0, 0.0, 0.0), clip_direction=(0.5, 0.5, 0.7)),
              abi.make_world(tr, clip_center=(500.0, 0.0, 0.0), clip_direction=(1.0, 0.0, 0.0))]
    cam = S.default_camera(72, 56)
    full_tile = abi.Tile(0, 0, 72, 56)
    with res:
        for wi, world in enumerate(worlds):
            for steps, jitter in ((64.0, -1), (37.5, 4), (0.6, -1)):
                rp = abi.RaymarchParams(steps, jitter, True)
                got = res.raymarch_intensity(cam, full_tile, rp, world)
                ref = orc.raymarch_intensity(cam, full_tile, rp, world)
                assert np.abs(got - ref).max() <= TIGHT_TOL, (wi, steps, jitter)
                assert np.array_equal(got[..., 3], ref[..., 3])
        world = worlds[1]
        rp = abi.RaymarchParams(64.0, 2, True)
        full = res.raymarch_intensity(cam, full_tile, rp, world)
        assert (full[..., 3] == 1).any() and (full[..., 3] == 0).any()
        sub = res.raymarch_intensity(cam, abi.Tile(17, 9, 30, 21), rp, world)
        assert np.array_equal(sub, full[9:30, 17:47])
        import torch
        depth = np.full((56, 72), 150.0, dtype=np.float32)
        d_dev = torch.from_numpy(depth).cuda()
        out = torch.empty((56, 72, 4), dtype=torch.float32, device="cuda")
        res.raymarch_intensity_device(cam, full_tile, rp, world, out.data_ptr(), d_dev.data_ptr())
        res.flush()
        ref_d = orc.raymarch_intensity(cam, full_tile, rp, world, scene_depth=depth)
        assert np.abs(out.cpu().numpy() - ref_d).max() <= TIGHT_TOL


def test_random_lights_and_shapes_against_oracle(gpu, oracle_mod):
    """Seeded sweep over volume shapes (none a multiple of the 32-pixel tile, some smaller than a brick), light directions
    (axis-aligned, diagonal, every sign pattern) and operation sequences: every chain instantiation (plane sizes 40 / 56 /
    72, 1-3 halo slots, chunks of 16 / 8 / 4 slices, partial last chunks) meets the oracle bit for bit."""
    rng = np.random.default_rng(20240917)
    shapes = [(9, 17, 33), (33, 9, 17), (70, 45, 52), (64, 64, 64), (37, 96, 41), (100, 34, 66)]
    special = [(1, 0, 0), (0, -1, 0), (0, 0, 1), (1, 1, 0), (-1, 0, 1), (1, -1, 1), (0.999, 0.02, -0.03), (0.5, 0.5, 0.70710678)]
    for si, dims in enumerate(shapes):
        res, orc = make_pair(gpu, oracle_mod, dims, np.uint16, seed=0x5EED0100 + si)
        world = S.default_world() if si % 2 == 0 else abi.make_world(
            abi.identity_transform(scale=(100.0, 80.0, 120.0), rotation=(0.1830127, 0.1830127, 0.1830127, 0.9330127)),
            clip_center=(3.0, -2.0, 1.0), clip_direction=(0.3, 0.8, 0.52))
        with res:
            lights = []
            for k in range(5):
                d = special[(si * 3 + k) % len(special)] if k < 2 else tuple(rng.normal(size=3))
                lights.append(abi.DirLightParams(d, float(rng.uniform(0.2, 0.9))))
            for l in lights:
                assert bool(res.add_dir_light(l, True, world)) == bool(orc.add_dir_light(l, True, world))
            assert_light_equal(res, orc)
            for k in (0, 3, 1):
                new = abi.DirLightParams(tuple(rng.normal(size=3)), float(rng.uniform(0.2, 0.9)))
                res.change_dir_light(lights[k], new, world)
                orc.change_dir_light(lights[k], new, world)
                lights[k] = new
                assert_light_equal(res, orc)
            res.add_dir_light(lights[2], False, world)  # remove one again
            orc.add_dir_light(lights[2], False, world)
            assert_light_equal(res, orc)
            c = res.launch_counters()
            assert c["chunk"] > 0


@pytest.mark.parametrize("dtype,dims", [(np.uint16, (20, 12, 9)), (np.uint8, (33, 40, 17)), (np.float32, (16, 16, 16))])
def test_octree_pyramid_and_march_match_oracle(gpu, oracle_mod, dtype, dims):
    """The Octree render mode: the 4-level UNORM16 max pyramid is bit-exact, the unlit point-sampled march over every
    level matches the oracle (clip plane, jitter and fractional last step included)."""
    res, orc = make_pair(gpu, oracle_mod, dims, dtype, tf="A", window=(0.5, 0.9, True, False))
    world = abi.make_world(abi.identity_transform(scale=(100.0, 90.0, 110.0), rotation=(0.0, 0.2588190, 0.0, 0.9659258)),
                           clip_center=(8.0, 0.0, 0.0), clip_direction=(0.6, 0.3, 0.74))
    with res:
        with pytest.raises(abi.TbrmError):
            res.download_octree_mip(0)  # not generated yet
        res.generate_octree()
        ref_mips = orc.generate_octree()
        for m in range(4):
            assert res.octree_mip_dims(m) == ref_mips[m].shape[::-1]
            assert np.array_equal(res.download_octree_mip(m), ref_mips[m])
        cam = S.default_camera(56, 40)
        tile = abi.Tile(0, 0, 56, 40)
        for mip in range(4):
            for steps, jitter in ((48.0, -1), (21.5, 3)):
                rp = abi.RaymarchParams(steps, jitter, True)
                got = res.raymarch_octree(cam, tile, rp, world, mip)
                ref = orc.raymarch_octree(cam, tile, rp, world, mip)
                assert np.abs(got - ref).max() <= TIGHT_TOL, (mip, steps, jitter)
        assert ref[..., 3].max() > 0.2
        # a new volume invalidates the pyramid
        res.upload_volume(small_volume(dims, dtype, 0x5EED0777))
        with pytest.raises(abi.TbrmError):
            res.raymarch_octree(cam, tile, abi.RaymarchParams(16.0, -1, True), world, 0)


@pytest.mark.parametrize("addr", [abi.ADDRESS_WRAP, abi.ADDRESS_CLAMP])
def test_random_cameras_windows_and_modes_against_oracle(gpu, oracle_mod, ray_lanes, addr):
    """Seeded sweep of the three render modes over cameras the default view never produces — inside the cube, grazing a face,
    axis-parallel (1/dir = inf in the slab test), far away, narrow and wide fields of view — with random windows, cutoffs,
    clip planes, step counts and jitter frames, on a volume whose sides are no multiple of a brick. (This sweep caught the
    +1 tap of a clamped base tap of -1 being texel 1 instead of texel 0: samples in the outer half-texel shell.)"""
    rng = np.random.default_rng(77)
    dims = (45, 38, 52)
    res, orc = make_pair(gpu, oracle_mod, dims, np.uint16, seed=0x5EED0300, addr=addr)
    eyes = [(-145, -95, 80), (10, 5, -8), (0, 0, 0), (-49.9, 20, 10), (-300, 0, 0), (0, 0, 400), (55, -300, 0.001), (-60, -60, -60), (35, 260, -140)]
    with res:
        world0 = S.default_world()
        for o in (res, orc):
            o.add_dir_light(S.light(0), True, world0)
            o.add_dir_light(S.light(3), True, world0)
        orc.generate_octree()
        res.generate_octree()
        worst = 0.0
        for k, eye in enumerate(eyes):
            target = (0.0, 0.0, 0.0) if k != 2 else (30.0, 12.0, -7.0)
            up = (0.0, 0.0, 1.0) if abs(eye[0]) + abs(eye[1]) > 1e-3 else (0.0, 1.0, 0.0)
            cam = abi.look_at_camera(eye, target, up, float(rng.choice([20.0, 60.0, 110.0])), 48, 40)
            world = world0 if k % 3 else abi.make_world(
                abi.identity_transform(scale=(100.0, 100.0, 100.0)), clip_center=tuple(rng.uniform(-30, 30, 3)), clip_direction=tuple(rng.normal(size=3)))
            win = abi.WindowingParams(float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.2, 1.2)), bool(rng.integers(2)), bool(rng.integers(2)))
            res.set_windowing(win)
            orc.set_windowing(win)
            tile = abi.Tile(int(rng.integers(0, 8)), int(rng.integers(0, 8)), 37, 29)
            rp = abi.RaymarchParams(float(rng.uniform(3.0, 90.0)), int(rng.integers(-1, 8)), True)
            got, (ref, n_ref) = res.raymarch_lit(cam, tile, rp, world), orc.raymarch_lit(cam, tile, rp, world)
            assert res.count_nominal_samples(cam, tile, rp, world) == n_ref
            worst = max(worst, float(np.abs(got - ref).max()))
            assert np.array_equal(got, res.raymarch_lit(cam, tile, abi.RaymarchParams(rp.steps, rp.jitter_frame, False), world))
            worst = max(worst, float(np.abs(res.raymarch_intensity(cam, tile, rp, world) - orc.raymarch_intensity(cam, tile, rp, world)).max()))
            mip = int(rng.integers(0, 4))
            worst = max(worst, float(np.abs(res.raymarch_octree(cam, tile, rp, world, mip) - orc.raymarch_octree(cam, tile, rp, world, mip)).max()))
        assert worst <= TIGHT_TOL


def test_steep_secondary_passes(gpu, oracle_mod, kernel_variant):
    """Lights that barely leave their major axis: the second pass runs along an axis the light hardly travels, so its taps
    sit up to a dozen texels from the pixel — 2-slice chunks (Add), the slice kernel beyond that and for the fused Change."""
    res, orc = make_pair(gpu, oracle_mod, (64, 56, 72), np.uint16, False, seed=0x5EED0600)
    world = S.default_world()
    with res:
        for d, inten in [((1, 0.02, -0.09), 0.5), ((0.07, -1, 0.03), 0.4), ((-0.02, 0.12, 1), 0.6), ((1, 0.01, -0.03), 0.3)]:
            light = abi.DirLightParams(d, inten)
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
            assert_light_equal(res, orc)
        old = abi.DirLightParams((1, 0.02, -0.09), 0.5)
        new = abi.DirLightParams((1, 0.03, -0.1), 0.5)
        res.change_dir_light(old, new, world)
        orc.change_dir_light(old, new, world)
        assert_light_equal(res, orc)
        counters = res.launch_counters()
        if kernel_variant != "slice":
            assert counters["chunk"] > 0, counters


# ---- batched multi-light add -------------------------------------------------------------------

@pytest.mark.parametrize("light_32bit", [False, True])
@pytest.mark.parametrize("dims", [(48, 40, 56), (64, 64, 64)])
def test_batched_lights_match_oracle_replay(gpu, oracle_mod, light_32bit, dims, kernel_variant, tunables):
    """tbrm_add_dir_lights pairs passes of different lights that share a cube face; the oracle replays the reported pass
    order one pass at a time. UNORM8: bit-exact."""
    res, orc = make_pair(gpu, oracle_mod, dims, np.uint16, light_32bit, seed=0x5EED0500)
    world = S.default_world()
    # the 8 config lights, a bundle of nearly parallel lights (what pairs up: their taps fall inside one another's range),
    # a zero direction and an axis-aligned light
    lights = [S.light(i) for i in range(8)] + [abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], a), 0.1) for a in (2.0, -3.0, 4.0)]
    lights += [abi.DirLightParams((0, 0, 0), 0.3), abi.DirLightParams((0, 0, -1), 0.2), abi.DirLightParams((0.02, 0.01, -1), 0.2)]
    with res:
        res.clear_light_volume(0.0)
        orc.clear_light_volume(0.0)
        sched = res.add_dir_lights(lights, True, world)
        n_passes = sum(2 if b >= 0 else 1 for _, _, b, _ in sched)
        assert n_passes == sum(abi.host_light_passes(l, world, res.light_dims)[1] for l in lights)
        sweeps = kernel_variant == "sweep" and not light_32bit
        if kernel_variant == "sweep" and light_32bit:
            pass  # (float light volumes sweep pass by pass: the two-light sweep is built for UNORM8, and a pass that sweeps leaves the chain's pairing)
        elif kernel_variant != "slice":
            assert sum(b >= 0 for _, _, b, _ in sched) >= 2, f"fewer than two pairs: {sched}"
        else:  # the slice-per-launch kernel has no two-light form
            assert all(b < 0 for _, _, b, _ in sched)
        if sweeps:  # round 4: pairs on the production path are ONE sweep launch each (PASS_ADD2 of k_light_sweep)
            p = res.path_counters()
            assert p["pair_sweeps"] == sum(b >= 0 for _, _, b, _ in sched) and p["passes_chain"] == 0 and p["passes_slice"] == 0, (p, sched)
        for la, pa, lb, pb in sched:
            assert orc.add_dir_light_pass(lights[la], True, world, pa) == 1
            if lb >= 0:
                assert lb != la and orc.add_dir_light_pass(lights[lb], True, world, pb) == 1
        assert_light_equal(res, orc)
        batched = res.download_light_volume()
        # light by light (the reference's order): the same sum; UNORM8 may differ by one code at fp32 rounding ties
        res.clear_light_volume(0.0)
        for l in lights:
            res.add_dir_light(l, True, world)
        seq = res.download_light_volume()
        if light_32bit:
            np.testing.assert_allclose(batched, seq, rtol=0, atol=1e-5)
        else:
            d = np.abs(batched.astype(int) - seq.astype(int))
            assert d.max() <= 1 and np.count_nonzero(d) <= 2e-3 * d.size, (int(d.max()), int(np.count_nonzero(d)))
        # the switch that turns pairing off: the reference's order, bit for bit
        tunables("light_batching", 0)
        res.clear_light_volume(0.0)
        sched = res.add_dir_lights(lights, True, world)
        assert all(b < 0 for _, _, b, _ in sched)
        assert np.array_equal(res.download_light_volume(), seq)


def test_batched_light_removal_matches_oracle_replay(gpu, oracle_mod):
    res, orc = make_pair(gpu, oracle_mod, (56, 56, 56), np.uint16, False, seed=0x5EED0501)
    world = S.default_world()
    lights = [S.light(i) for i in range(6)]
    with res:
        for l in lights:
            res.add_dir_light(l, True, world)
            orc.add_dir_light(l, True, world)
        sched = res.add_dir_lights(lights[1:5], False, world)  # Added = false for a subset
        for la, pa, lb, pb in sched:
            orc.add_dir_light_pass(lights[1 + la], False, world, pa)
            if lb >= 0:
                orc.add_dir_light_pass(lights[1 + lb], False, world, pb)
        assert_light_equal(res, orc)


@pytest.mark.parametrize("light_32bit", [False, True])
def test_reset_all_lights_from_kept_factors_is_bit_exact(gpu, oracle_mod, light_32bit):
    """ResetAllLights (RaymarchVolume.cpp:418-451: clear, then every light again) when every light's occlusion factors are
    kept (the factor cache, UNORM8 light volumes): the batched call samples the data volume not once — every pass is a sweep
    over kept factors, none is paired — and the light volume follows the oracle's replay of the reported order; so does the
    removal of two of the lights. A float light volume is swept from the same kept factors (round 4): same results within 2e-6."""
    dims = (72, 48, 56)
    res, orc = make_pair(gpu, oracle_mod, dims, np.uint16, light_32bit, seed=0x5EED0502)
    world = S.default_world()
    lights = [S.light(i) for i in (0, 1, 2, 5)]
    axes = {abi.host_light_passes(l, world, dims)[0][k].axis for l in lights for k in range(abi.host_light_passes(l, world, dims)[1])}
    assert axes == {0, 1, 2}
    n_passes = sum(abi.host_light_passes(l, world, dims)[1] for l in lights)
    assert 4 < n_passes <= 8
    with res:
        for l in lights:  # one by one: every pass's occlusion computed and kept
            res.add_dir_light(l, True, world)
        res.clear_light_volume(0.0)
        orc.clear_light_volume(0.0)
        before = res.light_cache_stats()
        sched = res.add_dir_lights(lights, True, world)
        after = res.light_cache_stats()
        if not light_32bit:
            assert all(b < 0 for _, _, b, _ in sched) and len(sched) == n_passes, sched
            assert after["hits"] - before["hits"] == n_passes and after["propagated"] == before["propagated"], (before, after)
        else:  # (round 4: float light volumes sweep too, from the same kept factors; a pass the float sweep declines takes the chain)
            assert after["hits"] - before["hits"] >= n_passes - 2, (before, after)
        for la, pa, lb, pb in sched:
            orc.add_dir_light_pass(lights[la], True, world, pa)
            if lb >= 0:
                orc.add_dir_light_pass(lights[lb], True, world, pb)
        assert_light_equal(res, orc)
        # and the removal of two of them, again from what is kept
        sched = res.add_dir_lights(lights[1:3], False, world)
        for la, pa, lb, pb in sched:
            orc.add_dir_light_pass(lights[1 + la], False, world, pa)
            if lb >= 0:
                orc.add_dir_light_pass(lights[1 + lb], False, world, pb)
        assert_light_equal(res, orc)
        res.flush()


@pytest.mark.parametrize("opaque_shell", [False, True])
def test_add_and_change_share_kept_passes_only_when_the_shell_is_transparent(gpu, oracle_mod, opaque_shell):
    """The Add shader skips samples outside the unit cube (AddDirLightShader.usf:98), the Change shader takes them with the
    border colour (ChangeDirLightShader.usf:100-130). Where such a sample can be opaque the two compute different occlusion
    factors and the factor cache keeps them apart; where the volume's outer brick layer blended with the border colour maps
    to opacity 0 (air around the scan: k_shell_transparent) both compute the same factors and a Change is served from what
    the Add kept — and the removal across cube faces from what the Change kept. Either way the oracle's light volume, bit for
    bit."""
    dims = (72, 64, 56)
    vol = small_volume(dims, np.uint16, seed=0x5EED0504)
    vol = vol.copy()
    if not opaque_shell:  # air around the scan
        shell = np.ones(vol.shape, dtype=bool)
        shell[9:-9, 9:-9, 9:-9] = False
        vol[shell] = 0
    else:  # dense material up to the volume's faces
        vol[:, :, :3] = 52000
        vol[:, :2, :] = 48000
        vol[-2:, :, :] = 50000
    world = S.default_world()
    lut = abi.color_curve_to_lut(S.TF_A_KEYS)
    win = abi.WindowingParams(0.5, 0.9, True, False)
    orc = oracle_mod.OracleScene(vol, False)
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    l0, l1 = S.light(0), S.light(1)
    turned = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
    across = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 60.0), S.LIGHTS[1][1])
    pa, _ = abi.host_light_passes(turned, world, dims)
    pb, _ = abi.host_light_passes(across, world, dims)
    assert (pa[0].face, pa[1].face) != (pb[0].face, pb[1].face)
    with abi.Resources(dims, abi.FMT_G16, False) as res:
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(win)
        res.clear_light_volume(0.0)
        for l in (l0, l1):
            res.add_dir_light(l, True, world)
            orc.add_dir_light(l, True, world)
        assert_light_equal(res, orc)
        before = res.light_cache_stats()
        res.change_dir_light(l1, turned, world)  # the removed side: what the Add kept, if the shaders agree
        orc.change_dir_light(l1, turned, world)
        mid = res.light_cache_stats()
        assert_light_equal(res, orc)
        res.change_dir_light(turned, across, world)  # across faces: Add(turned, removed) + Add(across, added)
        orc.change_dir_light(turned, across, world)
        after = res.light_cache_stats()
        assert_light_equal(res, orc)
        n1 = abi.host_light_passes(l1, world, dims)[1]
        if opaque_shell:
            assert mid["hits"] == before["hits"] and after["hits"] == mid["hits"], (before, mid, after)
        else:
            assert mid["hits"] - before["hits"] == n1, (before, mid)
            assert after["hits"] - mid["hits"] == abi.host_light_passes(turned, world, dims)[1], (mid, after)


def test_occlusion_beside_the_chain_changes_nothing(gpu, tunables):
    """occ_overlap (tbrm.h): the occlusion of the next span — of the same pass or the first of the next pass — runs on a
    second stream beside the current span's propagation, into the other of two buffers. Several spans per pass (depth over
    128 slices), several passes and operators back to back: the light volume is the one of the serial schedule, bit for bit."""
    dims = (150, 96, 140)
    vol = small_volume(dims, np.uint16, seed=0x5EED0503)
    world = S.default_world()
    results = []
    for overlap, rect in ((0, 1), (2, 1), (1, 1), (2, 0), (0, 0)):  # and with / without the 72 x 48 LDS planes (16- instead of 8-slice chunks)
        tunables("occ_overlap", overlap)
        tunables("chain_rect_planes", rect)
        with abi.Resources(dims, abi.FMT_G16, False) as res:
            res.upload_volume(vol)
            res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
            res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
            res.clear_light_volume(0.0)
            for i in range(4):
                res.add_dir_light(S.light(i), True, world)
            cur = S.light(1)
            for k in range(1, 4):  # both lights, then twice the new light alone (the kept L of the old one)
                new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0 * k), S.LIGHTS[1][1])
                res.change_dir_light(cur, new, world)
                cur = new
            res.add_dir_lights([S.light(4), S.light(6)], True, world)
            res.add_dir_light(S.light(0), False, world)
            results.append(res.download_light_volume())
            assert res.launch_counters()["slice"] == 0
    assert results[0].max() > 60
    for other in results[1:]:
        assert np.array_equal(results[0], other)


# ---- randomized sweep of the light operators --------------------------------------------------------------------------

@pytest.mark.parametrize("seed", list(range(24)) + [100, 101, 102, 103, 129])  # 129: a cached Change whose ragged last chunk has windows 5 pixels wider than the tile
def test_random_light_operator_sequences_against_oracle(gpu, oracle_mod, seed):
    """Seeded random scenes — ragged dimensions, data / light formats, half-resolution light volume, border modes, transfer
    functions and windows, rotated and non-uniformly scaled volumes, clip planes — and random sequences of Add / Remove /
    Change / batched adds with random, near-axis and near-diagonal light directions. UNORM8 light volumes bit-exact."""
    rng = np.random.default_rng(0x5EED0700 + seed)
    big = seed >= 100  # several tiles per plane, several occlusion spans per pass (seconds of oracle time per operator)
    dims = tuple(int(v) for v in (rng.integers(130, 201, size=3) if big else rng.integers(17, 61, size=3)))
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    light_32bit = bool(rng.integers(0, 2)) if seed % 4 == 3 else False
    half_res = bool(seed % 5 == 2)
    border = abi.BORDER_EXACT_FLOAT if seed % 4 == 1 else abi.BORDER_ENGINE_8BIT
    window = (float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.4, 1.2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    res, orc = make_pair(gpu, oracle_mod, dims, dtype, light_32bit, half_res, abi.ADDRESS_WRAP, border, "AB"[seed % 2], window,
                         seed=0x5EED0710 + seed)
    if seed % 2:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        scale = tuple(float(v) for v in rng.uniform(60, 140, size=3))
        tr = abi.identity_transform(scale, tuple(float(v) for v in rng.uniform(-20, 20, size=3)), tuple(float(v) for v in q))
        cd = rng.normal(size=3)
        world = abi.make_world(tr, tuple(float(v) for v in rng.uniform(-30, 30, size=3)), tuple(float(v) for v in cd / np.linalg.norm(cd)))
    else:
        world = S.default_world()

    def random_light():
        kind = rng.integers(0, 4)
        if kind == 0:    # anywhere
            d = rng.normal(size=3)
        elif kind == 1:  # nearly along an axis
            d = rng.normal(size=3) * 0.08
            d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
        elif kind == 2:  # nearly on a face diagonal (two weights close to each other)
            d = rng.normal(size=3) * 0.05
            a, b = rng.choice(3, size=2, replace=False)
            d[a] = rng.choice([-1.0, 1.0])
            d[b] = d[a] * rng.choice([-1.0, 1.0]) * rng.uniform(0.97, 1.03)
        else:            # exactly axis-aligned
            d = np.zeros(3)
            d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
        return abi.DirLightParams(tuple(float(v) for v in d), float(rng.uniform(0.1, 0.7)))

    with res:
        present = []
        for step in range(4 if big else 7):
            op = rng.integers(0, 4) if present else 0
            if op == 0 or len(present) < 2:
                l = random_light()
                res.add_dir_light(l, True, world)
                orc.add_dir_light(l, True, world)
                present.append(l)
            elif op == 1:
                l = present.pop(int(rng.integers(0, len(present))))
                res.add_dir_light(l, False, world)
                orc.add_dir_light(l, False, world)
            elif op == 2:
                i = int(rng.integers(0, len(present)))
                old = present[i]
                d = np.array([old.light_direction.x, old.light_direction.y, old.light_direction.z])
                new_d = d + rng.normal(size=3) * (0.05 if rng.integers(0, 2) else 0.8) * max(np.linalg.norm(d), 1e-3)
                new = abi.DirLightParams(tuple(float(v) for v in new_d), float(rng.uniform(0.1, 0.7)))
                res.change_dir_light(old, new, world)
                orc.change_dir_light(old, new, world)
                present[i] = new
            else:
                batch = [random_light() for _ in range(3)]
                for la, pa, lb, pb in res.add_dir_lights(batch, True, world):
                    orc.add_dir_light_pass(batch[la], True, world, pa)
                    if lb >= 0:
                        orc.add_dir_light_pass(batch[lb], True, world, pb)
                present += batch
            assert_light_equal(res, orc)


# ---- degenerate sizes and values -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("dims", [(1, 1, 1), (3, 1, 9), (9, 17, 2), (8, 8, 8), (2, 33, 1)])
def test_tiny_and_flat_volumes(gpu, oracle_mod, dims):
    res, orc = make_pair(gpu, oracle_mod, dims, np.uint16, False, seed=0x5EED0900)
    world = S.default_world()
    with res:
        for d, inten in [((1, .35, -.5), 0.5), ((0, 0, -1), 0.4), ((-.4, 1, -.3), 0.3)]:
            light = abi.DirLightParams(d, inten)
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
            assert_light_equal(res, orc)
        old, new = abi.DirLightParams((-.4, 1, -.3), 0.3), abi.DirLightParams((-.45, 1, -.2), 0.35)
        res.change_dir_light(old, new, world)
        orc.change_dir_light(old, new, world)
        assert_light_equal(res, orc)
        cam = S.default_camera(24, 16)
        for steps in (0.4, 1.0, 7.5, 300.0):
            rp = abi.RaymarchParams(steps, -1, True)
            got = res.raymarch_lit(cam, abi.Tile(0, 0, 24, 16, 1), rp, world)
            want, _ = orc.raymarch_lit(cam, abi.Tile(0, 0, 24, 16, 1), rp, world)
            np.testing.assert_allclose(got, want, rtol=0, atol=TIGHT_TOL, err_msg=f"steps {steps}")
        assert res.raymarch_lit(cam, abi.Tile(0, 0, 0, 0, 1), rp, world).size == 0  # an empty tile is a no-op


def test_non_finite_voxels_and_degenerate_windows(gpu, oracle_mod):
    """float data with NaN / Inf voxels, and windows of width 0 and below: whatever the arithmetic spec yields, both sides
    yield it (NaN light values store as 0 in UNORM8, D3D's rule)."""
    dims = (20, 18, 22)
    vol = small_volume(dims, np.float32, 0x5EED0901)
    vol[3, 4, 5] = np.nan
    vol[10, 9, 8] = np.inf
    vol[15, 2, 17] = -np.inf
    vol[7, :, 3] = 2.5   # outside [0, 1]
    vol[8, 6, :] = -1.0
    world = S.default_world()
    cam = S.default_camera(32, 24)
    for light_32bit in (False, True):
        for window in [(0.5, 0.9, True, False), (0.5, 0.0, True, True), (0.4, -0.3, False, False), (0.5, 1e-30, False, True)]:
            res = abi.Resources(dims, abi.FMT_R32_FLOAT, light_32bit, False, 0)
            orc = oracle_mod.OracleScene(vol, light_32bit, False, abi.ADDRESS_WRAP, abi.BORDER_ENGINE_8BIT)
            lut = abi.color_curve_to_lut(S.tf_keys("A"))
            w = abi.WindowingParams(*window)
            with res:
                res.upload_volume(vol)
                res.set_tf_lut(lut)
                res.set_windowing(w)
                orc.set_tf_lut(lut)
                orc.set_windowing(w)
                res.clear_light_volume(0.0)
                for d, inten in [((1, .35, -.5), 0.5), ((.2, -.3, -1), 0.4)]:
                    light = abi.DirLightParams(d, inten)
                    res.add_dir_light(light, True, world)
                    orc.add_dir_light(light, True, world)
                got = res.download_light_volume()
                if light_32bit:
                    np.testing.assert_allclose(got, orc.light, rtol=0, atol=TIGHT_TOL, equal_nan=True, err_msg=str(window))
                else:
                    assert np.array_equal(got, orc.light), window
                rp = abi.RaymarchParams(48.0, -1, True)
                frame = res.raymarch_lit(cam, abi.Tile(0, 0, 32, 24, 1), rp, world)
                want, _ = orc.raymarch_lit(cam, abi.Tile(0, 0, 32, 24, 1), rp, world)
                np.testing.assert_allclose(frame, want, rtol=0, atol=TIGHT_TOL, equal_nan=True, err_msg=str(window))


# ---- randomized sweep of the render modes --------------------------------------------------------------------------------

@pytest.mark.parametrize("seed", range(16))
def test_random_render_scenes_against_oracle(gpu, oracle_mod, seed, tunables):
    """Seeded random scenes for the three render modes: ragged sizes, data / light formats, half-resolution light volume,
    wrap / clamp, transfer functions, windows, rotated and non-uniformly scaled volumes with clip planes, cameras anywhere
    (inside, far, grazing), fields of view, step counts from a fraction of a step to hundreds, jitter frames, odd tiles with
    interleaved row groups, skipping on / off (identical), both lane layouts."""
    rng = np.random.default_rng(0x5EED0B00 + seed)
    dims = tuple(int(v) for v in rng.integers(9, 70, size=3))
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    addr = abi.ADDRESS_CLAMP if seed % 4 == 1 else abi.ADDRESS_WRAP
    res, orc = make_pair(gpu, oracle_mod, dims, dtype, light_32bit=bool(seed % 5 == 3), half_res=bool(seed % 3 == 2), addr=addr, tf="AB"[seed % 2],
                         window=(float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.3, 1.2)), bool(rng.integers(2)), bool(rng.integers(2))),
                         seed=0x5EED0B10 + seed)
    if seed % 2:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        tr = abi.identity_transform(tuple(float(v) for v in rng.uniform(60, 140, size=3)), tuple(float(v) for v in rng.uniform(-20, 20, size=3)),
                                    tuple(float(v) for v in q))
        cd = rng.normal(size=3)
        world = abi.make_world(tr, tuple(float(v) for v in rng.uniform(-30, 30, size=3)), tuple(float(v) for v in cd / np.linalg.norm(cd)))
    else:
        world = S.default_world()
    with res:
        for _ in range(2):
            light = abi.DirLightParams(tuple(float(v) for v in rng.normal(size=3)), float(rng.uniform(0.2, 0.7)))
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
        orc.generate_octree()
        res.generate_octree()
        worst = 0.0
        for case in range(5):
            eye = rng.normal(size=3)
            eye = eye / np.linalg.norm(eye) * float(rng.choice([5.0, 45.0, 52.0, 130.0, 400.0]))
            target = tuple(float(v) for v in rng.uniform(-25, 25, size=3))
            w, h = int(rng.integers(17, 57)), int(rng.integers(16, 49))
            cam = abi.look_at_camera(eye, target, (0.0, 0.0, 1.0), float(rng.choice([15.0, 50.0, 100.0])), w, h)
            step = int(rng.choice([1, 1, 2]))
            th = int(rng.integers(1, max(2, (h - 8) // step)))
            tile = abi.Tile(int(rng.integers(0, 6)), int(rng.integers(0, 6)), int(rng.integers(1, w - 6)), th, step)
            steps = float(rng.choice([0.5, 3.0, 17.25, 64.0, 250.0]))
            rp = abi.RaymarchParams(steps, int(rng.integers(-1, 8)), True)
            tunables("ray_lanes", [4, 8][case % 2])
            got, (ref, n_ref) = res.raymarch_lit(cam, tile, rp, world), orc.raymarch_lit(cam, tile, rp, world)
            assert res.count_nominal_samples(cam, tile, rp, world) == n_ref
            worst = max(worst, float(np.abs(got - ref).max()))
            assert np.array_equal(got, res.raymarch_lit(cam, tile, abi.RaymarchParams(steps, rp.jitter_frame, False), world)), "skipping changed the image"
            worst = max(worst, float(np.abs(res.raymarch_intensity(cam, tile, rp, world) - orc.raymarch_intensity(cam, tile, rp, world)).max()))
            mip = int(rng.integers(0, 4))
            worst = max(worst, float(np.abs(res.raymarch_octree(cam, tile, rp, world, mip) - orc.raymarch_octree(cam, tile, rp, world, mip)).max()))
        assert worst <= TIGHT_TOL, worst


@pytest.mark.parametrize("light_32bit", [False, True])
def test_contribution_cache_hits_misses_and_invalidation(gpu, oracle_mod, light_32bit, tunables):
    """The factor cache (tbrm.h tbrm_light_cache_stats; UNORM8 light volumes) never shows in the results: a sequence of
    operators that hits it (the removed side of a ChangeDirLight was the added side of the previous one: only the new light's
    occlusion is computed; a light that oscillates between two directions: none is; the removal of a light that was added),
    misses it (first change after an add: the Add shader's guard differs from the Change shader's), and invalidates it (new
    window, new transfer function, new volume) follows the oracle step by step, and leaves the same light volume as the same
    sequence with the cache turned off. (A float light volume takes the chunked chain, which caches nothing.)"""
    dims = (104, 88, 72)
    world = S.default_world()
    vol = small_volume(dims, np.uint16)
    vol2 = small_volume(dims, np.uint16, seed=0x5EED0777)
    lut_a, lut_b = abi.color_curve_to_lut(S.TF_A_KEYS), abi.color_curve_to_lut(S.TF_B_KEYS)
    win_a, win_b = abi.WindowingParams(0.5, 0.9, True, False), abi.WindowingParams(0.45, 0.7, True, True)

    def rot(i, deg):
        return abi.DirLightParams(S.rotate_z(S.LIGHTS[i][0], deg), S.LIGHTS[i][1])

    l0, l1 = S.light(0), S.light(1)
    steps = [("add", l0), ("add", l1),
             ("change", l1, rot(1, 5)),           # miss + miss: the Add's stream carries the Add shader's guard
             ("change", rot(1, 5), rot(1, 10)),   # hit on the removed side
             ("change", rot(1, 10), rot(1, 5)),   # both sides cached: no occlusion launch at all
             ("change", rot(1, 5), rot(1, 10)),
             ("remove", l0),                      # hit: the stream the add computed
             ("window", win_b), ("change", rot(1, 10), rot(1, 15)),   # everything cached is stale
             ("change", rot(1, 15), rot(1, 20)),
             ("tf", lut_b), ("change", rot(1, 20), rot(1, 25)), ("add", l0),
             ("volume", vol2), ("change", rot(1, 25), rot(1, 30)), ("remove", l0), ("change", rot(1, 30), rot(1, 90))]  # last: across faces
    finals, stats = [], []
    for cache_mb in (-1, 0, 4):  # the default budget, off, room for a few entries (evictions)
        tunables("light_cache_mb", cache_mb)
        orc = oracle_mod.OracleScene(vol, light_32bit)
        orc.set_tf_lut(lut_a)
        orc.set_windowing(win_a)
        with abi.Resources(dims, abi.FMT_G16, light_32bit) as res:
            res.upload_volume(vol)
            res.set_tf_lut(lut_a)
            res.set_windowing(win_a)
            res.clear_light_volume(0.0)
            for i, step in enumerate(steps):
                kind = step[0]
                if kind in ("add", "remove"):
                    res.add_dir_light(step[1], kind == "add", world)
                    orc.add_dir_light(step[1], kind == "add", world)
                elif kind == "change":
                    res.change_dir_light(step[1], step[2], world)
                    orc.change_dir_light(step[1], step[2], world)
                elif kind == "window":
                    res.set_windowing(step[1])
                    orc.set_windowing(step[1])
                    continue
                elif kind == "tf":
                    res.set_tf_lut(step[1])
                    orc.set_tf_lut(step[1])
                    continue
                else:
                    res.upload_volume(step[1])
                    orc2 = oracle_mod.OracleScene(step[1], light_32bit)
                    orc2.set_tf_lut(lut_b)
                    orc2.set_windowing(win_b)
                    orc2.light[...] = orc.light
                    orc = orc2
                    continue
                assert_light_equal(res, orc)
                if i == 9:  # the host takes the cache's memory back between two operators
                    res.light_cache_clear()
                    assert res.light_cache_stats()["entries"] == 0
                if cache_mb < 0 and i == 4:
                    before = res.light_cache_stats()
                if cache_mb < 0 and i == 5 and not light_32bit:
                    after = res.light_cache_stats()
                    assert after["hits"] - before["hits"] == 4 and after["propagated"] == before["propagated"], (before, after)
            finals.append(res.download_light_volume())
            stats.append(res.light_cache_stats())
            assert res.launch_counters()["slice"] == 0
    if not light_32bit:
        assert stats[0]["hits"] >= 10 and stats[0]["entries"] > 3, stats
        assert 0 < stats[2]["entries"] and stats[2]["bytes"] <= 4 << 20, stats
    else:  # (round 4: float light volumes sweep, and their passes' factors are kept like any other)
        assert stats[0]["hits"] > 0 and stats[0]["entries"] > 0, stats
    assert stats[1]["hits"] == 0 and stats[1]["entries"] == 0, stats
    for other in finals[1:]:
        assert np.array_equal(finals[0], other) if not light_32bit else np.abs(finals[0] - other).max() == 0.0


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 7, 8, 11, 13])
def test_random_render_scenes_with_wave_wide_skipping(gpu, oracle_mod, seed, tunables):
    """The same random scenes with the lit march's wave-wide empty-space skipping forced on (ray_wave_skip = 1; by default
    it is on for volumes of at least 384 voxels a side only — the full-size tests): every trip it takes blind only performs the
    positions' additions, so every frame is the oracle's as before."""
    tunables("ray_wave_skip", 1)
    test_random_render_scenes_against_oracle(gpu, oracle_mod, seed, tunables)


def test_wave_wide_skipping_is_bit_identical_in_a_mostly_empty_volume(gpu, tunables):
    """A small opaque blob in a 160^3 volume of air, rays of 320 steps: long proven-empty ranges, most trips of a wave blind.
    Frames with ray_wave_skip on / off and with skipping off altogether are the same bits (lit, with a clip plane, jittered)."""
    n = 160
    vol = np.zeros((n, n, n), dtype=np.uint16)
    zz, yy, xx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    vol[(xx - 100) ** 2 + (yy - 70) ** 2 + (zz - 90) ** 2 < 14 ** 2] = 52000
    vol[(xx - 30) ** 2 + (yy - 120) ** 2 + (zz - 40) ** 2 < 9 ** 2] = 40000
    world = S.default_world()
    frames = []
    for ws, skipping in ((1, True), (0, True), (0, False)):
        tunables("ray_wave_skip", ws)
        with abi.Resources((n, n, n), abi.FMT_G16) as res:
            res.upload_volume(vol)
            res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
            res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
            res.clear_light_volume(0.0)
            res.add_dir_light(S.light(0), True, world)
            for jitter in (-1, 3):
                cam = S.default_camera(200, 152)
                frames.append(res.raymarch_lit(cam, abi.Tile(0, 0, 200, 152, 1), abi.RaymarchParams(320.0, jitter, skipping), world))
    for k in range(2):
        assert np.array_equal(frames[k], frames[2 + k]) and np.array_equal(frames[k], frames[4 + k]), k
    assert float(frames[0][..., 3].max()) > 0.5
