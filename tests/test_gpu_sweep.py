"""The pipelined sweep kernel (k_light_sweep, tbrm_light_sweep.hip): that it is the kernel that runs where it applies, what it
declines, and the shapes the other suites do not reach — steep second passes (several hand-off words per lane), planes that are
not whole tiles, lights that pull two ways, more tiles than the device keeps resident, the hand-off under load (many operators
back to back, a frame beside them). Every result against the oracle, UNORM8 bit for bit."""
import numpy as np
import pytest

from tbraymarcherplugin_amd import abi, synthetic as S

pytestmark = pytest.mark.gpu


def scene(oracle_mod, dims, dtype=np.uint16, seed=0x5EED0900, tf=S.TF_A_KEYS, window=(0.5, 0.9, True, False)):
    vol = S.make_volume_numpy(dims, dtype, seed)
    lut = abi.color_curve_to_lut(tf)
    win = abi.WindowingParams(*window)
    orc = oracle_mod.OracleScene(vol, False)
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    res = abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)], False)
    res.upload_volume(vol)
    res.set_tf_lut(lut)
    res.set_windowing(win)
    res.clear_light_volume(0.0)
    return res, orc


def same(res, orc, what=""):
    res.flush()
    got = res.download_light_volume()
    assert np.array_equal(got, orc.light), f"{what}: {np.count_nonzero(got != orc.light)} of {got.size} UNORM8 light voxels differ"


def test_sweep_runs_where_it_applies_and_declines_the_rest(gpu, oracle_mod):
    world = S.default_world()
    # whole brick layers, UNORM8 light volume: one sweep launch per axis pass, no chunk of the chain
    res, orc = scene(oracle_mod, (64, 48, 56))
    with res:
        n = 0
        for i in (0, 1, 2):
            res.add_dir_light(S.light(i), True, world)
            orc.add_dir_light(S.light(i), True, world)
            n += abi.host_light_passes(S.light(i), world, (64, 48, 56))[1]
        c, pc = res.launch_counters(), res.path_counters()
        # every pass a sweep; a light's two passes share ONE launch (k_light_sweep_chain), a light with one pass takes one of its own
        assert pc["passes_sweep"] == n and pc["passes_chain"] == 0 and c["sweep"] == c["chunk"] and c["slice"] == 0, (c, pc)
        assert c["sweep"] == n - pc["launches_sweep_chain"] > 0, (c, pc)
        same(res, orc, "sweeps")
    # a depth that is not whole brick layers along one axis: the sweep runs over the padded volume (round 4; the chain before)
    dims = (64, 48, 52)
    res, orc = scene(oracle_mod, dims)
    with res:
        for i in (0, 2, 5):
            res.add_dir_light(S.light(i), True, world)
            orc.add_dir_light(S.light(i), True, world)
        c = res.launch_counters()
        assert c["sweep"] == c["chunk"] > 0 and c["slice"] == 0, c
        same(res, orc, "ragged depth")
    # what still declines: a downward pass over fewer than nine slices of a ragged depth (one padded layer: nothing behind it)
    dims = (40, 40, 5)
    res, orc = scene(oracle_mod, dims)
    with res:
        for dz in (1, -1):  # one of the two first passes runs downwards
            light = abi.DirLightParams((.1, .2, dz), 0.4)
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
        p = res.path_counters()
        assert p["passes_chain"] + p["passes_slice"] >= 1 and p["passes_sweep"] >= 1, p
        same(res, orc, "five slices")
    # a float light volume (bLightVolume32Bit): swept too since round 4 (float hand-off words, the light volume updated in place)
    vol = S.make_volume_numpy((48, 48, 48), np.uint16, 7)
    with abi.Resources((48, 48, 48), abi.FMT_G16, True) as res:
        res.upload_volume(vol)
        res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
        res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
        res.add_dir_light(S.light(0), True, world)
        c = res.launch_counters()
        assert c["sweep"] == c["chunk"] > 0, c


@pytest.mark.parametrize("direction", [(1, .12, -.05), (-.1, 1, .07), (.15, -.08, -1), (1, .3, .29), (-1, -.24, .9), (.5, .52, -1)])
def test_steep_second_passes_and_ragged_planes(gpu, oracle_mod, direction):
    """Second passes whose taps lie 2 to 9 texels away (up to six hand-off words per lane of the hand-off wave), on planes
    that are not whole 32 x 32 tiles (the overhanging pixels hold the border colour and hand nothing over)."""
    dims = (88, 72, 104)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0901)
    with res:
        light = abi.DirLightParams(direction, 0.7)
        other = abi.DirLightParams(S.rotate_z(direction, 4.0), 0.7)
        for op in ("add", "change", "remove"):
            if op == "add":
                res.add_dir_light(light, True, world)
                orc.add_dir_light(light, True, world)
            elif op == "change":
                res.change_dir_light(light, other, world)
                orc.change_dir_light(light, other, world)
            else:
                res.add_dir_light(other, False, world)
                orc.add_dir_light(other, False, world)
            same(res, orc, f"{direction} {op}")
        assert res.launch_counters()["slice"] == 0


TWO_WAY = [  # (pairs whose passes start from the same faces: the third-largest component changes sign)
    ((1, .3, .04), (1, .3, -.04)),
    ((1, .3, .2), (1, .32, -.2)),
    ((.3, 1, -.2), (.3, 1, .2)),         # y pass first
    ((.15, -.2, -1), (-.1, -.3, -1)),    # z pass first, downwards
    ((1, .2, .3), (1, -.2, .3)),
    ((.1, .45, 1), (-.1, .45, 1)),
]


@pytest.mark.parametrize("dims", [(64, 64, 64), (96, 72, 40)])
@pytest.mark.parametrize("cache", [0, -1])
def test_lights_that_pull_two_ways_are_swept_one_after_the_other(gpu, oracle_mod, tunables, dims, cache):
    """A fused Change whose two lights' minor components have opposite signs: the tiles of one stream depend on their upper
    neighbours, those of the other on the lower ones — no pipeline order serves both. The removed light's planes are swept
    first on their own (no light-volume update), and the fused sweep reads their hand-off records (SweepParams::
    r_from_records). Same results as the reference's slice loop; no pass falls back to the chain."""
    tunables("light_cache_mb", cache)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0902)
    with res:
        for k, (d_old, d_new) in enumerate(TWO_WAY):
            old, new = abi.DirLightParams(d_old, 0.6), abi.DirLightParams(d_new, 0.5 + 0.1 * (k % 3))
            pa, _ = abi.host_light_passes(old, world, dims)
            pb, _ = abi.host_light_passes(new, world, dims)
            assert (pa[0].face, pa[1].face) == (pb[0].face, pb[1].face), "meant to be a fused Change"
            res.add_dir_light(old, True, world)
            orc.add_dir_light(old, True, world)
            before = res.launch_counters()
            res.change_dir_light(old, new, world)
            orc.change_dir_light(old, new, world)
            after = res.launch_counters()
            # ("chunk" counts every propagation launch that covers more than a slice: all of them were sweeps)
            assert after["chunk"] - before["chunk"] == after["sweep"] - before["sweep"] and after["slice"] == before["slice"], (d_old, d_new, before, after)
            assert after["sweep"] - before["sweep"] >= 3, (d_old, d_new, before, after)  # at least one pass took two launches
            same(res, orc, f"two-way change {d_old} -> {d_new}")
        # and back again from what the cache kept (cache on), the other way round
        for d_old, d_new in TWO_WAY[:2]:
            res.change_dir_light(abi.DirLightParams(d_new, 0.5), abi.DirLightParams(d_old, 0.5), world)
            orc.change_dir_light(abi.DirLightParams(d_new, 0.5), abi.DirLightParams(d_old, 0.5), world)
        same(res, orc, "two-way changes back")


def test_two_way_changes_can_be_sent_to_the_chain(gpu, oracle_mod, tunables):
    """light_sweep = 2: passes whose lights pull two ways take the chained kernel, as they did before the two-launch form."""
    tunables("light_sweep", 2)
    dims = (64, 64, 64)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0902)
    with res:
        old, new = abi.DirLightParams((1, .3, .04), 0.6), abi.DirLightParams((1, .3, -.04), 0.6)
        res.add_dir_light(old, True, world)
        orc.add_dir_light(old, True, world)
        before = res.launch_counters()
        res.change_dir_light(old, new, world)
        orc.change_dir_light(old, new, world)
        after = res.launch_counters()
        assert after["chunk"] - before["chunk"] > after["sweep"] - before["sweep"], (before, after)
        same(res, orc, "two-way change on the chain")


def test_more_tiles_than_the_device_keeps_resident(gpu, oracle_mod, tunables):
    """A 1152 x 1152 plane is 1296 tiles: more workgroups than the device keeps resident at once (the tiles are dealt by
    ticket in upstream-first order, so a tile only ever waits for tiles that have started: late tiles simply start late).
    Against the one-slice-per-launch kernel on the same GPU (the oracle would take minutes)."""
    dims = (1152, 1152, 32)
    vol = S.make_volume_numpy(dims, np.uint8, 0x5EED0903)
    world = S.default_world()
    out = []
    for variant in ("sweep", "slice"):
        tunables("force_slice_kernel", 1 if variant == "slice" else 0)
        with abi.Resources(dims, abi.FMT_G8, False) as res:
            res.upload_volume(vol)
            res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
            res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
            res.clear_light_volume(0.0)
            # (the volume is flat: a direction this close to z still moves the taps one to two texels of the 1152-wide plane per slice)
            res.add_dir_light(abi.DirLightParams((.04, -.03, -1), 0.8), True, world)   # first pass along z: the big plane
            res.change_dir_light(abi.DirLightParams((.04, -.03, -1), 0.8), abi.DirLightParams((.045, -.025, -1), 0.8), world)
            res.flush()
            if variant == "sweep":
                assert res.launch_counters()["sweep"] >= 2
            out.append(res.download_light_volume())
    tunables("force_slice_kernel", 0)
    assert np.array_equal(out[0], out[1]), f"{np.count_nonzero(out[0] != out[1])} voxels differ"


def test_many_operators_back_to_back_with_frames_beside_them(gpu, oracle_mod):
    """The hand-off under load: 40 operators enqueued without a host synchronisation in between (the occlusion of each pass
    runs on the second stream beside the sweep of the pass before), a lit raymarch after every fourth; one comparison with the
    oracle at the end (its light volume after the same 40 operators) and the sweep's error word clean."""
    dims = (96, 96, 96)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0904)
    cam = S.default_camera(128, 128)
    tile = abi.Tile(0, 0, 128, 128, 1)
    rp = abi.RaymarchParams(96.0, -1, True)
    with res:
        cur = [S.light(i) for i in range(4)]
        for l in cur:
            res.add_dir_light(l, True, world)
            orc.add_dir_light(l, True, world)
        for k in range(36):
            i = k % 4
            new = abi.DirLightParams(S.rotate_z(S.LIGHTS[i][0], 7.0 * (k // 4 + 1)), S.LIGHTS[i][1])
            res.change_dir_light(cur[i], new, world)
            orc.change_dir_light(cur[i], new, world)
            cur[i] = new
            if k % 4 == 3:
                res.raymarch_lit(cam, tile, rp, world)
        same(res, orc, "40 operators")
        st = res.light_cache_stats()
        assert st["hits"] > 0 and st["entries"] > 0, st


def random_sweep_scene(oracle_mod, seed):
    """A random scene whose passes are whole brick layers (the sweep applies wherever the taps allow): dimensions, data type,
    window, volume transform with a clip plane, and a sequence of Adds, removals and Changes — small turns (fused, now and then
    with the third component changing sign: the two-launch form), large turns (remove + add), repeats (from the cache)."""
    rng = np.random.default_rng(0x5EED0A00 + seed)
    dims = tuple(int(8 * v) for v in rng.integers(3, 13 if seed < 100 else 24, size=3))
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    window = (float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.4, 1.2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    res, orc = scene(oracle_mod, dims, dtype, seed=0x5EED0A80 + seed, tf=[S.TF_A_KEYS, S.TF_B_KEYS][seed % 2], window=window)
    if seed % 3 == 1:
        q = rng.normal(size=4)
        tr = abi.identity_transform(tuple(float(v) for v in rng.uniform(60, 140, size=3)), tuple(float(v) for v in rng.uniform(-20, 20, size=3)),
                                    tuple(float(v) for v in q / np.linalg.norm(q)))
        cd = rng.normal(size=3)
        world = abi.make_world(tr, tuple(float(v) for v in rng.uniform(-30, 30, size=3)), tuple(float(v) for v in cd / np.linalg.norm(cd)))
    else:
        world = S.default_world()
    return res, orc, world, rng, dims


def run_random_sweep_scene(oracle_mod, seed, tunables=None):
    if tunables is not None:
        tunables("light_cache_mb", 0 if seed % 4 == 3 else -1)
    res, orc, world, rng, dims = random_sweep_scene(oracle_mod, seed)
    with res:
        lights = []
        for step in range(10):
            kind = int(rng.integers(0, 5)) if lights else 0
            if kind == 0 or len(lights) < 2:  # add
                d = rng.normal(size=3)
                if rng.integers(0, 3) == 0:
                    d[int(rng.integers(0, 3))] *= 0.05  # nearly in a coordinate plane: small offsets, signs that flip easily
                l = abi.DirLightParams(tuple(float(v) for v in d), float(rng.uniform(0.2, 0.9)))
                res.add_dir_light(l, True, world)
                orc.add_dir_light(l, True, world)
                lights.append(l)
            elif kind == 1:  # remove
                l = lights.pop(int(rng.integers(0, len(lights))))
                res.add_dir_light(l, False, world)
                orc.add_dir_light(l, False, world)
            else:  # change: a small turn, a flip of the smallest component, or anything
                i = int(rng.integers(0, len(lights)))
                old = lights[i]
                d = np.array([old.light_direction.x, old.light_direction.y, old.light_direction.z], dtype=np.float64)
                if kind == 2:
                    d = d + rng.normal(size=3) * 0.08 * np.linalg.norm(d)
                elif kind == 3:
                    k = int(np.argmin(np.abs(d)))
                    d[k] = -d[k] * float(rng.uniform(0.5, 1.5))
                else:
                    d = rng.normal(size=3)
                new = abi.DirLightParams(tuple(float(v) for v in d), float(rng.uniform(0.2, 0.9)))
                res.change_dir_light(old, new, world)
                orc.change_dir_light(old, new, world)
                lights[i] = new
            same(res, orc, f"seed {seed} dims {dims} step {step} kind {kind}")
        c = res.launch_counters()
        return c


@pytest.mark.parametrize("seed", list(range(16)) + [100, 101])
def test_random_operator_sequences_on_whole_brick_layers(gpu, oracle_mod, tunables, seed):
    """Seeded random scenes in which the sweep applies (tools/hunt_sweep_scenes.py runs the same beyond the pinned seeds):
    after every operator the UNORM8 light volume is the oracle's, bit for bit."""
    c = run_random_sweep_scene(oracle_mod, seed, tunables)
    assert c["slice"] == 0 or c["sweep"] > 0, c


@pytest.mark.parametrize("dims", [(40, 32, 8), (8, 48, 40), (16, 16, 16), (32, 8, 24)])
def test_passes_of_one_and_two_brick_layers(gpu, oracle_mod, dims):
    """The shortest passes the sweep takes (8 and 16 slices: the factor ring is longer than the pass, the loader's requests
    run past its end from the start): Adds, a fused Change, a removal."""
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, np.uint8, seed=0x5EED0905)
    with res:
        ls = [abi.DirLightParams(d, i) for d, i in [((1, .3, -.2), 0.5), ((-.2, 1, .3), 0.6), ((.25, -.3, -1), 0.7)]]
        for l in ls:
            res.add_dir_light(l, True, world)
            orc.add_dir_light(l, True, world)
        same(res, orc, f"{dims} adds")
        for l in ls:
            new = abi.DirLightParams(S.rotate_z((l.light_direction.x, l.light_direction.y, l.light_direction.z), 4.0), l.light_intensity)
            res.change_dir_light(l, new, world)
            orc.change_dir_light(l, new, world)
        same(res, orc, f"{dims} changes")
        res.add_dir_light(ls[0], False, world)
        orc.add_dir_light(ls[0], False, world)
        same(res, orc, f"{dims} removal")
        assert res.launch_counters()["sweep"] > 0


# ---- one occlusion launch for both axis passes of a light (tbrm_internal.h DualOcc, tunable occ_dual) ------------------------

DUAL_SCENES = [  # dims, light directions (every pair of pass axes, both orders), window
    ((64, 48, 56), [(1, .35, -.5), (-.4, 1, -.3), (.2, -.3, -1), (-1, -.6, .4), (.6, -1, -.2), (-.3, .2, 1)]),
    ((40, 72, 24), [(1, .1, .6), (.3, .2, -1), (-.2, 1, .5)]),           # planes that are not whole 16 x 16 blocks
    ((88, 40, 104), [(-1, .5, .1), (.1, -.7, 1), (.45, 1, -.2)]),
]


@pytest.mark.parametrize("dims,dirs", DUAL_SCENES)
@pytest.mark.parametrize("cache", [0, -1])
def test_both_passes_of_a_light_share_one_occlusion_launch(gpu, oracle_mod, tunables, dims, dirs, cache):
    """UVWOffset is the same for both axis passes of a light (LightingShaders.cpp:114-124), so one launch filters, windows and
    looks the transfer function up once per voxel and writes both passes' factors. Forced on and off: the light volumes are the
    oracle's — and therefore each other's — after every operator (adds, fused changes with one and with both lights sampled,
    a removal), and the counters say which form ran."""
    tunables("light_cache_mb", cache)
    world = S.default_world()
    results = {}
    for dual in (1, 0):
        tunables("occ_dual", dual)
        res, orc = scene(oracle_mod, dims, seed=0x5EED0A00)
        with res:
            cur = list(dirs)
            lights = [abi.DirLightParams(d, 0.45) for d in cur]
            for k, light in enumerate(lights):
                res.add_dir_light(light, True, world)
                orc.add_dir_light(light, True, world)
                same(res, orc, f"dual={dual} add {k}")
            for k, light in enumerate(lights):  # small turns: fused Changes (cache on: the new light alone is sampled)
                cur[k] = S.rotate_z(cur[k], 3.0)
                new = abi.DirLightParams(cur[k], 0.45)
                res.change_dir_light(light, new, world)
                orc.change_dir_light(light, new, world)
                same(res, orc, f"dual={dual} change {k}")
                lights[k] = new
            res.add_dir_light(lights[0], False, world)
            orc.add_dir_light(lights[0], False, world)
            same(res, orc, f"dual={dual} removal")
            c = res.path_counters()
            two_pass_ops = sum(1 for d in dirs if abi.host_light_passes(abi.DirLightParams(d, 0.45), world, dims)[1] == 2)
            if dual:
                assert c["occlusion_dual"] >= two_pass_ops, c
            else:
                assert c["occlusion_dual"] == 0 and c["occlusion_single"] > 0, c
            assert c["passes_chain"] == 0 and c["passes_slice"] == 0, c
            results[dual] = res.download_light_volume()
    assert np.array_equal(results[0], results[1])


def test_dual_occlusion_with_clip_plane_half_resolution_and_float_data(gpu, oracle_mod, tunables):
    """The dual launch under the conditions that change its sample loop: an active clip plane (per-voxel AlphaWeight), a
    half-resolution light volume (data : light = 2 : 1, more staged bricks per unit), float and 8-bit data, the Add shader's
    guard on samples outside the cube versus the Change shader's border colour (an opaque shell)."""
    for dtype, half, tf, window in ((np.float32, False, S.TF_A_KEYS, (0.5, 0.9, True, False)), (np.uint8, True, S.TF_B_KEYS, (0.5, 0.8, True, True)),
                                    (np.uint16, True, S.TF_A_KEYS, (0.3, 1.4, False, False))):
        dims = (96, 64, 80)
        vol = S.make_volume_numpy(dims, dtype, 0x5EED0A01)
        lut = abi.color_curve_to_lut(tf)
        win = abi.WindowingParams(*window)
        tr = abi.identity_transform(scale=(100.0, 120.0, 80.0), translation=(10.0, -5.0, 3.0), rotation=(0.1305262, 0.0, 0.0, 0.9914449))
        world = abi.make_world(tr, clip_center=(12.0, -2.0, 5.0), clip_direction=(0.3, -0.2, 0.93))
        for dual in (1, 0):
            tunables("occ_dual", dual)
            orc = oracle_mod.OracleScene(vol, False, half)
            orc.set_tf_lut(lut)
            orc.set_windowing(win)
            with abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)], False, half_res=half) as res:
                res.upload_volume(vol)
                res.set_tf_lut(lut)
                res.set_windowing(win)
                res.clear_light_volume(0.0)
                a, b = abi.DirLightParams((1, .4, -.55), 0.6), abi.DirLightParams((.97, .45, -.5), 0.6)
                res.add_dir_light(a, True, world)
                orc.add_dir_light(a, True, world)
                same(res, orc, f"{dtype} half={half} dual={dual} add")
                res.change_dir_light(a, b, world)
                orc.change_dir_light(a, b, world)
                same(res, orc, f"{dtype} half={half} dual={dual} change")
                if dual and res.path_counters()["passes_sweep"]:
                    assert res.path_counters()["occlusion_dual"] > 0


def test_a_failed_sweep_is_reported_by_every_join_until_the_light_volume_is_cleared(gpu, oracle_mod, tunables):
    """sweep_timeout_ms < 0: a tile that finds a neighbour's hand-off word missing gives up at once instead of polling, and the
    first tile of every launch reports so in any case — the failure a starved or reset device produces after its timeout. The
    handle must not hand out what that sweep left behind:
    flush, the light-volume download, the host-buffer frame, the operator timing and the next light operator all report it,
    until ClearResourceLightVolumes defines the light volume again — after which the sweeps run as before."""
    dims = (96, 96, 64)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0A02)
    cam, tile, rp = S.default_camera(64, 64), abi.Tile(0, 0, 64, 64), abi.RaymarchParams(48.0, -1, True)
    with res:
        tunables("sweep_timeout_ms", -1)
        res.add_dir_light(S.light(0), True, world)  # (the hook: the launch's first tile reports that it gave up)
        with pytest.raises(abi.TbrmError, match="undefined"):
            res.flush()
        for call in (res.flush, res.download_light_volume, lambda: res.raymarch_lit(cam, tile, rp, world), lambda: res.last_gpu_time_ms(0),
                     lambda: res.add_dir_light(S.light(0), True, world)):
            with pytest.raises(abi.TbrmError, match="undefined"):
                call()
        tunables("sweep_timeout_ms", 0)
        res.clear_light_volume(0.0)
        before = res.path_counters()
        for i in (0, 1):
            res.add_dir_light(S.light(i), True, world)
            orc.add_dir_light(S.light(i), True, world)
        same(res, orc, "after the clear")
        after = res.path_counters()
        assert after["passes_sweep"] > before["passes_sweep"] and after["passes_chain"] == before["passes_chain"], (before, after)


RAGGED = [  # light-volume dimensions that are no multiples of 8 (or of anything), lights through all six faces
    (67, 45, 53), (41, 90, 33), (100, 100, 57), (9, 70, 70), (75, 12, 130),
]


@pytest.mark.parametrize("dims", RAGGED)
@pytest.mark.parametrize("cache", [0, -1])
def test_pass_lengths_that_are_no_multiple_of_eight_are_swept(gpu, oracle_mod, tunables, dims, cache):
    """A scan's depth is whatever the scanner made it (512 x 512 x 373), and a half-resolution light volume rounds odd sizes
    up (RaymarchVolume.cpp:850-855): the sweep runs such passes over the volume padded to whole brick layers — upwards the extra
    slices come last, downwards they come first and the last of them hands the initial plane on. Adds through all six faces,
    fused changes, lights that pull two ways, removals: the oracle's light volume after every operator, no chain, no slice loop."""
    tunables("light_cache_mb", cache)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0B00 + dims[0])
    dirs = [(1, .35, -.5), (-.4, 1, -.3), (.2, -.3, -1), (-1, -.6, .4), (.6, -1, -.2), (-.3, .2, 1)]
    with res:
        for k, d in enumerate(dirs):
            light = abi.DirLightParams(d, 0.3)
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
            same(res, orc, f"{dims} add {d}")
        for k, d in enumerate(dirs):
            new_d = S.rotate_z(d, 4.0) if k % 2 == 0 else (d[0], d[1], -d[2] * 0.9) if abs(d[2]) < 0.45 else S.rotate_z(d, -3.0)
            res.change_dir_light(abi.DirLightParams(d, 0.3), abi.DirLightParams(new_d, 0.3), world)
            orc.change_dir_light(abi.DirLightParams(d, 0.3), abi.DirLightParams(new_d, 0.3), world)
            same(res, orc, f"{dims} change {d} -> {new_d}")
            dirs[k] = new_d
        res.add_dir_light(abi.DirLightParams(dirs[0], 0.3), False, world)
        orc.add_dir_light(abi.DirLightParams(dirs[0], 0.3), False, world)
        same(res, orc, f"{dims} removal")
        p = res.path_counters()
        assert p["passes_sweep"] > 0, p
        if max(dims) <= 3 * min(dims):  # (a flat volume's slanted passes reach further than the sweep's planes hold: those take the chain)
            assert p["passes_chain"] == 0 and p["passes_slice"] == 0, p


def test_half_resolution_light_volume_of_an_odd_scan_is_swept(gpu, oracle_mod):
    """187 = (373 + 1) / 2 slices: the case the N3 loader produces (LoadMHDFileIntoVolumeNormalized + half-resolution light volume)."""
    dims = (120, 96, 373)
    vol = S.make_volume_numpy(dims, np.uint16, 0x5EED0B10)
    lut = abi.color_curve_to_lut(S.TF_A_KEYS)
    win = abi.WindowingParams(0.5, 0.9, True, False)
    world = S.default_world()
    orc = oracle_mod.OracleScene(vol, False, True)
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    with abi.Resources(dims, abi.FMT_G16, False, half_res=True) as res:
        assert res.light_dims == (60, 48, 187)
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(win)
        res.clear_light_volume(0.0)
        for i in (2, 5):  # the two lights whose first pass runs along z, down and up
            res.add_dir_light(S.light(i), True, world)
            orc.add_dir_light(S.light(i), True, world)
            same(res, orc, f"light {i}")
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[2][0], 6.0), S.LIGHTS[2][1])
        res.change_dir_light(S.light(2), new, world)
        orc.change_dir_light(S.light(2), new, world)
        same(res, orc, "change")
        p = res.path_counters()
        # (the passes along z, 187 slices, are swept; the second passes of these lights run along y over a 60 x 187 plane 48 slices
        # deep: their taps lie up to 14 texels from the pixel — beyond the sweep's planes — and take the chain)
        assert p["passes_sweep"] >= 3, p


def test_launch_tags_start_over_after_65535_launches(gpu, oracle_mod, tunables):
    """The hand-off words carry a 16-bit launch tag and are never cleared between launches; after 65535 launches both record
    buffers are zeroed and the tags start over (a two-way Change reserves two consecutive tags). sweep_epoch_preset puts a
    fresh handle a few launches in front of that point: ordinary sweeps, fused and two-way Changes across the wrap leave the
    oracle's light volume — a stale word accepted as current would not."""
    tunables("sweep_epoch_preset", 0xFFFF - 9)
    dims = (96, 96, 64)
    world = S.default_world()
    res, orc = scene(oracle_mod, dims, seed=0x5EED0C00)
    with res:
        pairs = TWO_WAY[:3] * 3
        for k, (d_old, d_new) in enumerate(pairs):  # add (2 launches), two-way change (>= 3), removal (2): ~7 launches per round
            old, new = abi.DirLightParams(d_old, 0.5), abi.DirLightParams(d_new, 0.5)
            res.add_dir_light(old, True, world)
            orc.add_dir_light(old, True, world)
            res.change_dir_light(old, new, world)
            orc.change_dir_light(old, new, world)
            same(res, orc, f"round {k} after the change")
            res.add_dir_light(new, False, world)
            orc.add_dir_light(new, False, world)
            same(res, orc, f"round {k} after the removal")
        assert res.path_counters()["passes_sweep"] > 40


@pytest.mark.parametrize("dims", [(64, 48, 56), (67, 45, 53), (128, 128, 128)])
@pytest.mark.parametrize("cache", [0, -1])
def test_float_light_volumes_are_swept(gpu, oracle_mod, tunables, dims, cache):
    """bLightVolume32Bit (RaymarchVolume.cpp:857-866): planes and light volume are floats — nothing is re-quantised, a hand-off
    word is the float itself, the light volume takes every L as an fp32 atomic add. Adds through all six faces, fused changes,
    removals against the oracle within the float tolerance of the suite (2e-6; the UNORM8 sweep is bit-exact), no chain for
    one-way passes; a ragged volume and config 1's size included."""
    tunables("light_cache_mb", cache)
    world = S.default_world()
    vol = S.make_volume_numpy(dims, np.float32 if dims[0] == 128 else np.uint16, 0x5EED0D00)
    lut = abi.color_curve_to_lut(S.TF_A_KEYS)
    win = abi.WindowingParams(0.5, 0.9, True, False)
    orc = oracle_mod.OracleScene(vol, True)
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    dirs = [(1, .35, -.5), (-.4, 1, -.3), (.2, -.3, -1), (-1, -.6, .4), (.6, -1, -.2), (-.3, .2, 1)]

    def close(what):
        res.flush()
        got = res.download_light_volume()
        err = float(np.abs(got - orc.light).max())
        assert err <= 2e-6, f"{what}: max |diff| {err}"

    with abi.Resources(dims, abi.DTYPE_FMT[vol.dtype], True) as res:
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(win)
        res.clear_light_volume(0.0)
        for d in dirs:
            light = abi.DirLightParams(d, 0.3)
            res.add_dir_light(light, True, world)
            orc.add_dir_light(light, True, world)
            close(f"add {d}")
        for k, d in enumerate(dirs):
            new_d = S.rotate_z(d, 4.0)
            res.change_dir_light(abi.DirLightParams(d, 0.3), abi.DirLightParams(new_d, 0.3), world)
            orc.change_dir_light(abi.DirLightParams(d, 0.3), abi.DirLightParams(new_d, 0.3), world)
            close(f"change {d}")
            dirs[k] = new_d
        res.add_dir_light(abi.DirLightParams(dirs[0], 0.3), False, world)
        orc.add_dir_light(abi.DirLightParams(dirs[0], 0.3), False, world)
        close("removal")
        p = res.path_counters()
        assert p["passes_sweep"] > 0 and p["passes_slice"] == 0, p
        # (a float pass of more than three hand-off words per lane, or whose lights pull two ways, still takes the chain)
        assert p["passes_chain"] <= p["passes_sweep"] // 8, p


@pytest.mark.parametrize("light_32bit", [False, True])
def test_padding_voxels_of_a_ragged_light_volume_are_left_alone(gpu, oracle_mod, light_32bit):
    """The bricked light volume has voxels beyond the volume's own (whole 8 x 8 x 8 bricks), and a ragged pass is swept over whole
    brick layers: the slices it is padded with — whose occlusion factors nobody computed — must not write there. The raw tensor is
    what tbrm_light_volume_device_ptr and the multi-GPU combine hand out: garbage (NaNs, in a float volume) would travel."""
    from tbraymarcherplugin_amd import sharding

    dims = (67, 45, 53)
    vol = S.make_volume_numpy(dims, np.uint16, 0x5EED0C11)
    res = abi.Resources(dims, abi.FMT_G16, light_32bit)
    world = S.default_world()
    with res:
        res.upload_volume(vol)
        res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
        res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
        res.clear_light_volume(0.25)
        for d in [(1, .35, -.5), (-.4, 1, -.3), (.2, -.3, -1), (-1, -.6, .4), (.6, -1, -.2), (-.3, .2, 1)]:
            res.add_dir_light(abi.DirLightParams(d, 0.1), True, world)
        res.flush()
        p = res.path_counters()
        assert p["passes_sweep"] == 12 and p["passes_chain"] == 0 and p["passes_slice"] == 0, p
        raw = sharding.device_light_tensor(res).cpu().numpy()
    nx, ny, nz = dims  # (x fastest)
    bn = [(n + 7) // 8 for n in (nx, ny, nz)]
    bricks = raw.reshape(bn[2], bn[1], bn[0], 8, 8, 8)  # [bz][by][bx][z][y][x]
    dense = bricks.transpose(0, 3, 1, 4, 2, 5).reshape(bn[2] * 8, bn[1] * 8, bn[0] * 8)
    inside = np.zeros(dense.shape, dtype=bool)
    inside[:nz, :ny, :nx] = True
    clear = np.float32(0.25) if light_32bit else np.uint8(64)  # trunc(0.25 * 255 + 0.5)
    pad = dense[~inside]
    assert np.all(pad == clear), f"{np.count_nonzero(pad != clear)} of {pad.size} padding voxels changed (e.g. {pad[pad != clear][:4]})"
    assert np.any(dense[inside] != clear)


@pytest.mark.parametrize("dims", [(64, 64, 64), (96, 72, 40)])
def test_float_lights_that_pull_two_ways_are_swept(gpu, oracle_mod, dims):
    """bLightVolume32Bit with a fused Change whose lights pull opposite ways (round 5; until then: the chain): the removed light's
    planes first, as floats into their own records, then the fused sweep in the added light's order — within the suite's float
    tolerance of the oracle, every pass on the sweep."""
    world = S.default_world()
    vol = S.make_volume_numpy(dims, np.uint16, 0x5EED0902)
    lut = abi.color_curve_to_lut(S.TF_A_KEYS)
    win = abi.WindowingParams(0.5, 0.9, True, False)
    orc = oracle_mod.OracleScene(vol, True)
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    with abi.Resources(dims, abi.FMT_G16, True) as res:
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(win)
        res.clear_light_volume(0.0)
        two_launch_passes = 0
        for k, (d_old, d_new) in enumerate(TWO_WAY):
            old, new = abi.DirLightParams(d_old, 0.6), abi.DirLightParams(d_new, 0.5 + 0.1 * (k % 3))
            res.add_dir_light(old, True, world)
            orc.add_dir_light(old, True, world)
            before = res.path_counters()
            res.change_dir_light(old, new, world)
            orc.change_dir_light(old, new, world)
            after = res.path_counters()
            two_launch_passes += (after["launches_sweep"] - before["launches_sweep"]) - (after["passes_sweep"] - before["passes_sweep"])
            res.flush()
            err = float(np.abs(res.download_light_volume() - orc.light).max())
            assert err <= 2e-6, f"two-way float change {d_old} -> {d_new}: max |diff| {err}"
        p = res.path_counters()
        assert two_launch_passes >= len(TWO_WAY) // 2, (two_launch_passes, p)  # (planes + fused launch: the two-way form ran)
        # (a float pass of more than three hand-off words per lane still declines: the flat volume's steep second passes)
        assert p["passes_slice"] == 0 and p["passes_chain"] <= (2 if max(dims) == min(dims) else p["passes_sweep"] // 2), p


def test_block_lists_are_kept_with_the_handle(gpu, oracle_mod, tunables):
    """Round 5: the empty-block flags, work list and ranks of a pass depend on the light only through the integer range of data texels
    a block's samples touch; a light that turns by a few degrees finds the lists of the call before (tbrm_path_counters:
    block_lists_built does not move), a new window makes every list stale (rebuilt, results still the oracle's), and a factor cache
    entry keeps the ranks it was filled under."""
    for cache in (-1, 0):
        tunables("light_cache_mb", cache)
        world = S.default_world()
        res, orc = scene(oracle_mod, (96, 96, 96), seed=0x5EED0E00)
        with res:
            d = (1, .35, -.5)
            res.add_dir_light(abi.DirLightParams(d, 0.4), True, world)
            orc.add_dir_light(abi.DirLightParams(d, 0.4), True, world)
            built = []
            for k in range(6):
                new_d = S.rotate_z(d, 0.25)  # (small turns: the taps' integer ranges stay; a larger turn may cross a texel and build anew)
                res.change_dir_light(abi.DirLightParams(d, 0.4), abi.DirLightParams(new_d, 0.4), world)
                orc.change_dir_light(abi.DirLightParams(d, 0.4), abi.DirLightParams(new_d, 0.4), world)
                d = new_d
                built.append(res.path_counters()["block_lists_built"])
                same(res, orc, f"cache {cache}, turn {k}")
            assert built[-1] == built[1], built  # (the first Change may meet a new signature — two streams; after it nothing is built)
            win = abi.WindowingParams(0.45, 0.8, True, False)
            res.set_windowing(win)
            orc.set_windowing(win)
            res.clear_light_volume(0.0)
            orc.clear_light_volume(0.0)
            res.add_dir_light(abi.DirLightParams(d, 0.4), True, world)
            orc.add_dir_light(abi.DirLightParams(d, 0.4), True, world)
            assert res.path_counters()["block_lists_built"] > built[-1]  # a new window: new emptiness bits, new lists
            same(res, orc, f"cache {cache}, new window")
            res.change_dir_light(abi.DirLightParams(d, 0.4), abi.DirLightParams(S.rotate_z(d, -3.0), 0.4), world)
            orc.change_dir_light(abi.DirLightParams(d, 0.4), abi.DirLightParams(S.rotate_z(d, -3.0), 0.4), world)
            same(res, orc, f"cache {cache}, change under the new window")


def test_chained_sweeps_equal_one_launch_per_pass(gpu, oracle_mod, tunables):
    """k_light_sweep_chain (tunable sweep_chain: passes per launch): the next pass's tiles take their tickets behind this pass's and
    wait, brick layer by brick layer, for the tile of the pass before that owns the bricks — Adds (one stream), fused Changes (two),
    resets of four lights as one tbrm_add_dir_lights call (chains of four passes: sc1 loads from the third pass on), ragged depths,
    planes that are not whole tiles, upward and downward passes — against the oracle, and the launches counted."""
    world = S.default_world()
    for dims in ((96, 72, 64), (64, 100, 52)):
        for chain in (4, 2):
            tunables("sweep_chain", chain)
            res, orc = scene(oracle_mod, dims)
            with res:
                lights = [S.light(i) for i in (0, 1, 2, 5)]
                for la, pa, lb, pb in res.add_dir_lights(lights, True, world):
                    orc.add_dir_light_pass(lights[la], True, world, pa)
                    if lb >= 0:
                        orc.add_dir_light_pass(lights[lb], True, world, pb)
                same(res, orc, f"{dims} chain {chain}: batched reset")
                pc = res.path_counters()
                assert pc["launches_sweep_chain"] >= 2 and pc["passes_chain"] == 0, pc
                for k in range(6):
                    li = k % 4
                    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[(0, 1, 2, 5)[li]][0], 7.0 * (k + 1)), lights[li].light_intensity)
                    res.change_dir_light(lights[li], new, world)
                    orc.change_dir_light(lights[li], new, world)
                    lights[li] = new
                    same(res, orc, f"{dims} chain {chain}: change {k}")
                for l in lights[:2]:
                    res.add_dir_light(l, False, world)
                    orc.add_dir_light(l, False, world)
                same(res, orc, f"{dims} chain {chain}: removals")
                assert res.path_counters()["launches_sweep_chain"] > pc["launches_sweep_chain"]
    tunables("sweep_chain", 1)
    res, orc = scene(oracle_mod, (64, 64, 64))
    with res:
        res.add_dir_light(S.light(0), True, world)
        orc.add_dir_light(S.light(0), True, world)
        same(res, orc, "chain off")
        assert res.path_counters()["launches_sweep_chain"] == 0
