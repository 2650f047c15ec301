"""Size-independent properties of the HIP path at sizes the oracle would take too long for."""
import numpy as np
import pytest

from tbraymarcherplugin_amd import abi, sharding, synthetic as S

pytestmark = pytest.mark.gpu


def make(gpu, n, dtype=np.uint16, light_32bit=False, tf="A", window=(0.5, 0.9, True, False), seed=3):
    import torch

    vol = S.make_volume_torch((n, n, n), dtype, S.seed_for_config(seed), torch.device("cuda", 0))
    res = abi.Resources((n, n, n), abi.DTYPE_FMT[np.dtype(dtype)], light_32bit)
    torch.cuda.synchronize()  # the library reads the tensor on its own stream
    res.upload_volume_device(vol.data_ptr(), vol.numel() * vol.element_size())
    res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys(tf)))
    res.set_windowing(abi.WindowingParams(*window))
    return res


def test_chunk_kernels_equal_slice_kernel_at_256(gpu, tunables):
    """The production kernels (16 slices per launch pair) and the reference-structured kernel (one slice per launch)
    produce the same UNORM8 light volume, bit for bit, on a 256^3 volume with 4 lights and two selective updates."""
    world = S.default_world()
    results = []
    for variant in ("chunk", "slice"):
        tunables("force_slice_kernel", 1 if variant == "slice" else 0)
        with make(gpu, 256) as res:
            for i in range(4):
                res.add_dir_light(S.light(i), True, world)
            res.change_dir_light(S.light(1), abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1]), world)
            res.change_dir_light(S.light(2), abi.DirLightParams(S.rotate_z(S.LIGHTS[2][0], 60.0), S.LIGHTS[2][1]), world)
            results.append(res.download_light_volume())
            c = res.launch_counters()
            assert (c["chunk"] > 0 and c["slice"] == 0) if variant == "chunk" else (c["slice"] > 0 and c["chunk"] == 0)
    assert np.array_equal(results[0], results[1]), f"{np.count_nonzero(results[0] != results[1])} voxels differ"
    assert results[0].max() == 255 and results[0].min() < 60


def test_add_then_remove_restores_float_light_volume(gpu):
    world = S.default_world()
    with make(gpu, 192, light_32bit=True) as res:
        res.add_dir_light(S.light(0), True, world)
        base = res.download_light_volume()
        res.add_dir_light(S.light(3), True, world)
        assert np.abs(res.download_light_volume() - base).max() > 0.1
        res.add_dir_light(S.light(3), False, world)
        assert np.abs(res.download_light_volume() - base).max() < 1e-6
        res.clear_light_volume(0.0)
        assert not res.download_light_volume().any()


def test_transparent_volume_saturates(gpu):
    world = S.default_world()
    with make(gpu, 160) as res:
        lut = np.zeros((256, 4), dtype=np.float32)
        lut[:, :3] = 1.0
        res.set_tf_lut(lut)
        for d, inten, want in [((1, 0, 0), 0.5, 128), ((0, -1, 0), 0.4, 230), ((0, 0, 1), 0.3, 255)]:
            res.add_dir_light(abi.DirLightParams(d, inten), True, world)
            lv = res.download_light_volume()
            assert (lv[8:-8, 8:-8, 8:-8] == want).all()  # interior: away from the sRGB-quantised border colour
        img = res.raymarch_lit(S.default_camera(256, 256), abi.Tile(0, 0, 256, 256), abi.RaymarchParams(128.0, -1, True), world)
        assert not img.any()


def test_skipping_and_tiling_do_not_change_the_image_at_config2_size(gpu):
    world = S.default_world()
    cam = S.default_camera(512, 512)
    full_tile = abi.Tile(0, 0, 512, 512)
    for tf, window in (("A", (0.5, 0.9, True, False)), ("B", (0.5, 0.8, True, True))):
        with make(gpu, 256, tf=tf, window=window) as res:
            for i in range(2):
                res.add_dir_light(S.light(i), True, world)
            a = res.raymarch_lit(cam, full_tile, abi.RaymarchParams(256.0, -1, False), world)
            b = res.raymarch_lit(cam, full_tile, abi.RaymarchParams(256.0, -1, True), world)
            assert np.array_equal(a, b), "empty-space skipping changed pixels"
            assert a[..., 3].max() > 0.9 and np.isfinite(a).all()
            # 4-way interleaved tiles reassemble to the same frame
            parts = np.stack([res.raymarch_lit(cam, sharding.rank_tile(512, 512, r, 4), abi.RaymarchParams(256.0, -1, True), world) for r in range(4)])
            assert np.array_equal(sharding.assemble(parts, 512, 4), a)
            n = res.count_nominal_samples(cam, full_tile, abi.RaymarchParams(256.0, -1, True), world)
            n_parts = sum(res.count_nominal_samples(cam, sharding.rank_tile(512, 512, r, 4), abi.RaymarchParams(256.0, -1, True), world) for r in range(4))
            assert n == n_parts and n > 10_000_000


def test_the_order_pixel_blocks_go_to_the_xcds_in_does_not_change_the_image(gpu, tunables):
    """ray_xcd_rows (k_raymarch_lit): launch order, rows, bands of 2 and 4 rows; a frame whose rows of blocks do not divide by
    8 x rows takes the launch order whatever the tunable says."""
    world = S.default_world()
    with make(gpu, 128, tf="B", window=(0.5, 0.8, True, True)) as res:
        for i in range(2):
            res.add_dir_light(S.light(i), True, world)
        for w, h in ((512, 512), (256, 192), (136, 72)):  # 64, 24 and 9 rows of 8 x 8 blocks
            cam = S.default_camera(w, h)
            tile = abi.Tile(0, 0, w, h)
            rp = abi.RaymarchParams(128.0, -1, True)
            tunables("ray_xcd_rows", 0)
            want = res.raymarch_lit(cam, tile, rp, world)
            assert want[..., 3].max() > 0.5
            for rows in (1, 2, 4):
                tunables("ray_xcd_rows", rows)
                assert np.array_equal(res.raymarch_lit(cam, tile, rp, world), want), (w, h, rows)


def test_unorm_decode_is_exact_division(gpu):
    """The 2-instruction UNORM decode of the kernels (fma(c, r, c*r2), reciprocal split in two floats) equals IEEE
    c/255 and c/65535 for every code."""
    u8, u16 = abi.selftest_unorm_decode(0)
    assert np.array_equal(u8, np.arange(256, dtype=np.float32) / np.float32(255))
    assert np.array_equal(u16, np.arange(65536, dtype=np.float32) / np.float32(65535))


def test_unorm8_store_roundtrip_matches_d3d_rule(gpu):
    """The kernels' UNORM8 store (v_med3 clamp that also maps NaN to 0, x*255+0.5, floor) against the D3D11 rule evaluated
    in numpy, on special values, every rounding boundary and its float neighbours, and a dense random sweep."""
    rng = np.random.default_rng(7)
    edges = (np.arange(0, 256, dtype=np.float64) + 0.5) / 255.0
    e32 = edges.astype(np.float32)
    vals = np.concatenate([
        np.array([0.0, -0.0, 1.0, -1.0, 2.0, 0.5, 1e-30, -1e-30, np.inf, -np.inf, np.nan, -np.nan, 1.0 - 2**-24, 2**-149], dtype=np.float32),
        e32, np.nextafter(e32, np.float32(-1)), np.nextafter(e32, np.float32(2)),
        np.arange(256, dtype=np.float32) / np.float32(255),
        rng.uniform(-0.25, 1.25, 200_000).astype(np.float32),
    ])
    got = abi.selftest_unorm8_roundtrip(vals)
    x = np.where(np.isnan(vals), np.float32(0), vals)
    x = np.minimum(np.maximum(x, np.float32(0)), np.float32(1))
    code = np.trunc(x * np.float32(255) + np.float32(0.5)).astype(np.float32)  # float32 mul then add, like the shader
    want = code / np.float32(255)
    assert np.array_equal(got, want)


def test_light_parallel_reset_on_device(gpu):
    """Light-parallel ResetAllLights with two ranks emulated on one GPU: each "rank" owns a handle and
    propagates its share of 6 lights; the device-side combine (torch tensors aliasing the bricked light volumes, widened
    sum, saturation) must leave both handles with the saturating sum of the two private volumes — bit for bit — and within
    one UNORM8 code of the sequential single-handle reset."""
    import torch

    n, world_size = 128, 2
    world = S.default_world()
    lights = [S.light(i) for i in range(6)]
    handles = [make(gpu, n) for _ in range(world_size)]
    try:
        # private accumulators first (what each rank holds before the exchange), read back for the expected result
        private = []
        for r, res in enumerate(handles):
            res.clear_light_volume(0.0)
            for i in sharding.light_schedule(len(lights), r, world_size):
                res.add_dir_light(lights[i], True, world)
            private.append(res.download_light_volume().astype(np.int32))
        expected = np.minimum(private[0] + private[1], 255).astype(np.uint8)

        # the exchange, with the collectives emulated over the two local tensors
        res0_t = [sharding.device_light_tensor(res) for res in handles]
        for res in handles:
            res.flush()
        snapshot = [t.clone() for t in res0_t]

        def collectives(rank):
            def reduce_scatter_sum(t):
                total = snapshot[0].to(torch.int32) + snapshot[1].to(torch.int32)
                k = total.numel() // world_size
                return total[rank * k:(rank + 1) * k].clone()

            def all_gather(chunk):
                total = (snapshot[0].to(torch.int32) + snapshot[1].to(torch.int32)).clamp_(max=255).to(torch.uint8)
                k = total.numel() // world_size
                assert torch.equal(chunk, total[rank * k:(rank + 1) * k])
                return total

            return lambda t: sharding.combine_light_codes(t, world_size, reduce_scatter_sum, all_gather)

        for r, res in enumerate(handles):
            sharding.reset_all_lights_light_parallel(res, lights, world, r, world_size, collectives(r))
            assert np.array_equal(res.download_light_volume(), expected)

        with make(gpu, n) as seq:
            for l in lights:
                seq.add_dir_light(l, True, world)
            ref = seq.download_light_volume().astype(np.int32)
        diff = np.abs(expected.astype(np.int32) - ref)
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3 and ref.max() > 100
        # the combined volume is usable: a frame from either handle equals the other's
        cam, rp = S.default_camera(256, 256), abi.RaymarchParams(128.0, -1, True)
        a = handles[0].raymarch_lit(cam, abi.Tile(0, 0, 256, 256), rp, world)
        b = handles[1].raymarch_lit(cam, abi.Tile(0, 0, 256, 256), rp, world)
        assert np.array_equal(a, b)
    finally:
        for res in handles:
            res.close()


def test_full_size_properties_at_config3(gpu, tunables):
    """BASELINE config 3 at full size (512^3 UNORM16, 1024^2 frame, 512 steps), where the oracle would take minutes: the
    production chunk kernels and the reference-structured slice kernel leave the same light volume bit for bit after four
    Adds and a fused Change; empty-space skipping / leaping does not change a single pixel; interleaved row-group tiles
    reassemble the frame; an Add followed by its removal returns the light volume to within one UNORM8 code."""
    world = S.default_world()
    new1 = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
    volumes = []
    for variant in ("chunk", "slice"):
        tunables("force_slice_kernel", 1 if variant == "slice" else 0)
        with make(gpu, 512) as res:
            for i in range(4):
                res.add_dir_light(S.light(i), True, world)
            res.change_dir_light(S.light(1), new1, world)
            volumes.append(res.download_light_volume())
            if variant == "chunk":
                cam = S.default_camera(1024, 1024)
                full_tile = abi.Tile(0, 0, 1024, 1024)
                a = res.raymarch_lit(cam, full_tile, abi.RaymarchParams(512.0, -1, False), world)
                b = res.raymarch_lit(cam, full_tile, abi.RaymarchParams(512.0, -1, True), world)
                assert np.array_equal(a, b), "skipping / leaping changed the frame"
                assert (b[..., 3] == 1.0).any() and (b[..., 3] == 0.0).any()
                parts = np.stack([res.raymarch_lit(cam, sharding.rank_tile(1024, 1024, r, 8), abi.RaymarchParams(512.0, -1, True), world)
                                  for r in range(8)])
                assert np.array_equal(sharding.assemble(parts, 1024, 8), b)
                # add a fifth light and take it away again: the same stream is added and subtracted, so apart from voxels that
                # saturated the UNORM8 light volume returns to within one code (Q(Q(p + x) - x) = p except at fp32 near-ties)
                before = res.download_light_volume()
                extra = abi.DirLightParams((0.2, -0.7, 0.4), 0.15)
                res.add_dir_light(extra, True, world)
                lit = res.download_light_volume()
                res.add_dir_light(extra, False, world)
                after = res.download_light_volume()
                unsaturated = lit < 255
                assert (lit.astype(np.int32) >= before).all() and (lit != before).any()
                d = np.abs(after[unsaturated].astype(np.int32) - before[unsaturated])
                assert d.max() <= 1 and (d != 0).mean() < 1e-3
    tunables("force_slice_kernel", 0)
    assert np.array_equal(volumes[0], volumes[1]), f"{np.count_nonzero(volumes[0] != volumes[1])} voxels differ between the chunk and the slice kernels"


def test_error_behaviour_through_the_c_abi(gpu):
    """wrong sizes, calls in the wrong order and out-of-range arguments fail with a code and a message, and leave the handle usable"""
    import ctypes as C

    from conftest import small_volume
    from tbraymarcherplugin_amd import synthetic as S

    lib = abi.load()
    for bad in [dict(dims=(0, 8, 8)), dict(dims=(8, 8, 8), data_format=7), dict(dims=(8, 8, 8), device=99)]:
        with pytest.raises(abi.TbrmError) as e:
            abi.Resources(bad["dims"], bad.get("data_format", abi.FMT_G8), device=bad.get("device", 0))
        assert e.value.code == abi.ERR_INVALID_ARG, bad
    dims = (24, 20, 16)
    vol = small_volume(dims, np.uint8)
    world = S.default_world()
    cam = S.default_camera(16, 16)
    tile = abi.Tile(0, 0, 16, 16, 1)
    rp = abi.RaymarchParams(16.0, -1, True)
    with abi.Resources(dims, abi.FMT_G8) as res:
        # nothing uploaded yet: the reference's "resources not initialised" (RaymarchUtils.cpp:39-49)
        assert not res.is_initialized()
        flag = C.c_int(1)
        light = abi.DirLightParams((1, 0, 0), 0.5)
        assert lib.tbrm_add_dir_light(res.handle, C.byref(light), 1, C.byref(world), C.byref(flag), 0) == abi.ERR_NOT_INITIALIZED and flag.value == 0
        with pytest.raises(abi.TbrmError) as e:
            res.raymarch_lit(cam, tile, rp, world)
        assert e.value.code == abi.ERR_NOT_INITIALIZED
        # wrong byte counts
        assert lib.tbrm_upload_volume(res.handle, vol.ctypes.data, vol.nbytes - 1) == abi.ERR_INVALID_ARG and b"bytes" in lib.tbrm_last_error()
        res.upload_volume(vol)
        res.set_tf_lut(abi.make_default_tf_lut())
        assert res.is_initialized()
        out = np.empty(vol.shape, dtype=np.uint8)
        assert lib.tbrm_download_light_volume(res.handle, out.ctypes.data, out.nbytes + 3) == abi.ERR_INVALID_ARG
        # arguments out of range
        for bad_rp in (abi.RaymarchParams(0.0, -1, True), abi.RaymarchParams(-5.0, -1, True), abi.RaymarchParams(float("nan"), -1, True)):
            with pytest.raises(abi.TbrmError) as e:
                res.raymarch_lit(cam, tile, bad_rp, world)
            assert e.value.code == abi.ERR_INVALID_ARG
        out4 = np.empty((4, 4, 4), dtype=np.float32)
        bad_tile = abi.Tile(0, 0, -4, 4, 1)
        assert lib.tbrm_raymarch_lit(res.handle, C.byref(cam), C.byref(bad_tile), C.byref(rp), C.byref(world), out4.ctypes.data) == abi.ERR_INVALID_ARG
        with pytest.raises(abi.TbrmError):
            res.raymarch_octree(cam, tile, rp, world, 0)      # no pyramid yet
        res.generate_octree()
        with pytest.raises(abi.TbrmError):
            res.raymarch_octree(cam, tile, rp, world, 4)      # levels are 0..3
        with pytest.raises(abi.TbrmError):
            res.octree_mip_dims(-1)
        ms = C.c_float()
        assert lib.tbrm_last_gpu_time_ms(res.handle, 5, C.byref(ms)) == abi.ERR_INVALID_ARG
        # a zero light direction is the reference's silent no-op that still reports success (LightingShaders.cpp:41-46)
        before = res.download_light_volume()
        assert res.add_dir_light(abi.DirLightParams((0, 0, 0), 1.0), True, world)
        assert np.array_equal(res.download_light_volume(), before)
        # and the handle still works
        assert res.add_dir_light(light, True, world)
        frame = res.raymarch_lit(cam, tile, rp, world)
        assert np.isfinite(frame).all() and frame[..., 3].max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("center,width", [(0.5, 0.9), (0.5, 0.8), (0.5, 1.0), (0.3, 1.4), (0.123456, 0.0371), (0.75, 3.0000002), (-2.5, 7.25), (0.5, 1e-3),
                                          (0.5000001, 0.9), (0.5, -0.9)])
def test_division_free_window_position_is_the_ieee_quotient(gpu, center, width):
    """GetTransferFuncPosition (WindowedSampling.usf:14-17) divides by the window's width. For values filtered out of UNORM data the
    kernels compute the quotient with three fmas (q = RN(a / w) from y = RN(1 / w): Markstein's correction step) where the host
    vouches for the window; the device compares it with the IEEE division for EVERY float in [0, 1]: the same bits. Windows the
    host does not vouch for (a width whose significand is all ones, non-finite or extreme parameters) keep the division."""
    import ctypes as C

    lib = abi.load()
    bad, fast = C.c_uint64(123), C.c_int(-1)
    abi.check(lib.tbrm_selftest_window_division(0, center, width, C.byref(bad), C.byref(fast)))
    assert fast.value == 1 and bad.value == 0, (center, width, fast.value, bad.value)


@pytest.mark.gpu
@pytest.mark.parametrize("step0,step1", [(100.0 / 512.0, 100.0 / 724.0), (100.0 / 128.0, 100.0 / 128.0), (0.0, 1.0e-6), (0.37, 4.0), (100.0, 1.0e4), (3.0e38, 1.0e-30),
                                         (1.0e-45, 1.0)])
def test_short_opacity_correction_is_one_minus_pow(gpu, step0, step1):
    """The kernels evaluate 1 - pow(1 - a, step) (WindowedSampling.usf:35) without the parts of the exponential that cannot show in
    1 - r (a power below 2^-25 is gone there): the device compares the short form with 1 - pow for EVERY float 1 - a in [0, 1], as
    a single power and as the dual occlusion launch's pair — the same bits."""
    import ctypes as C

    lib = abi.load()
    bad = C.c_uint64(123)
    abi.check(lib.tbrm_selftest_opacity_correction(0, step0, step1, C.byref(bad)))
    assert bad.value == 0, (step0, step1, bad.value)
    assert lib.tbrm_selftest_opacity_correction(0, -1.0, 1.0, C.byref(bad)) == abi.ERR_INVALID_ARG


@pytest.mark.gpu
def test_windows_the_host_does_not_vouch_for_keep_the_division(gpu):
    import ctypes as C
    import struct

    lib = abi.load()
    all_ones = struct.unpack("<f", struct.pack("<I", 0x3f7fffff))[0]  # 0.99999994: significand all ones
    for center, width in ((0.5, all_ones), (0.5, float("inf")), (float("nan"), 0.9), (0.5, 1e-20), (1e20, 0.9)):
        bad, fast = C.c_uint64(0), C.c_int(-1)
        abi.check(lib.tbrm_selftest_window_division(0, center, width, C.byref(bad), C.byref(fast)))
        assert fast.value == 0, (center, width)
