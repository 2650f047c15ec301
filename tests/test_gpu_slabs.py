"""Slab-partitioned illumination (tbrm_slab_*, slabs.py) on ONE GPU: several handles in one process stand in for the
ranks, planes move by device copies. The partitioned light volume must be bit-identical to the unpartitioned operator
(and through it to the oracle): per voxel the arithmetic and its inputs are the same.
"""
import numpy as np
import pytest

from conftest import small_volume
from tbraymarcherplugin_amd import abi, slabs, synthetic as S

pytestmark = pytest.mark.gpu


def make_handles(n_handles, dims, dtype, light_32bit=False, half_res=False, seed=0x5EED0400, tf="A", window=(0.5, 0.9, True, False)):
    vol = small_volume(dims, dtype, seed)
    lut = abi.color_curve_to_lut(S.tf_keys(tf))
    w = abi.WindowingParams(*window)
    out = []
    for _ in range(n_handles):
        res = abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)], light_32bit, half_res, 0)
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(w)
        res.clear_light_volume(0.0)
        out.append(res)
    return vol, lut, w, out


def slab_setup(handles, n_slabs):
    depth = handles[0].light_dims[2]
    bounds = slabs.slab_bounds(depth, n_slabs)
    members = [slabs.DeviceSlab(res, k, *bounds[k]) for k, res in enumerate(handles)]
    fabric = slabs.make_fabric([b[0] for b in bounds] + [depth])
    return members, fabric, bounds


def assert_slabs_equal(full, members, what):
    """every member's OWN slices equal the unpartitioned volume"""
    ref = full.download_light_volume()
    for m in members:
        got = m.res.download_light_volume()[m.z_begin:m.z_end]
        want = ref[m.z_begin:m.z_end]
        bad = np.count_nonzero(got != want)
        assert bad == 0, f"{what}: slab {m.slab_index} [{m.z_begin},{m.z_end}): {bad} of {got.size} voxels differ"


LIGHTS = [((1, .35, -.5), 0.5), ((-.4, 1, -.3), 0.4), ((.2, -.3, -1), 0.4), ((-1, -.6, .4), 0.3), ((.1, .45, 1), 0.5),
          ((0, 0, -1), 0.3), ((1, 0, 0), 0.2), ((.6, -1, -.2), 0.3), ((.7, .1, .69), 0.25)]


@pytest.mark.parametrize("n_slabs,dims,half_res", [(2, (72, 56, 64), False), (4, (128, 96, 128), False), (2, (80, 64, 128), True)])
@pytest.mark.parametrize("light_32bit", [False, True])
def test_slab_partitioned_lights_equal_single_handle(gpu, tunables, n_slabs, dims, half_res, light_32bit):
    tunables("slab_sweep", 1 if n_slabs == 2 else 0)  # (both forms of a slab's pass along z: one sweep / the chunked chain)
    _, _, _, handles = make_handles(n_slabs + 1, dims, np.uint16, light_32bit, half_res)
    full, parts = handles[0], handles[1:]
    members, fabric, _ = slab_setup(parts, n_slabs)
    world = S.default_world()
    try:
        for i, (d, inten) in enumerate(LIGHTS):
            light = abi.DirLightParams(d, inten)
            full.add_dir_light(light, True, world)
            slabs.add_dir_light(members, fabric, light, True, world)
            assert_slabs_equal(full, members, f"add {i} {d}")
        # remove one, change one within its faces (fused), change one across faces (remove + add)
        rem = abi.DirLightParams(*LIGHTS[3])
        full.add_dir_light(rem, False, world)
        slabs.add_dir_light(members, fabric, rem, False, world)
        assert_slabs_equal(full, members, "remove")
        old = abi.DirLightParams(*LIGHTS[1])
        new = abi.DirLightParams(S.rotate_z(LIGHTS[1][0], 5.0), LIGHTS[1][1])
        full.change_dir_light(old, new, world)
        slabs.change_dir_light(members, fabric, old, new, world)
        assert_slabs_equal(full, members, "fused change (lateral + lateral)")
        old = abi.DirLightParams(*LIGHTS[2])
        new = abi.DirLightParams((.25, -.2, -1), 0.45)
        full.change_dir_light(old, new, world)
        slabs.change_dir_light(members, fabric, old, new, world)
        assert_slabs_equal(full, members, "fused change (along z)")
        old = abi.DirLightParams(*LIGHTS[4])
        new = abi.DirLightParams((1.0, 0.1, -0.2), 0.5)
        full.change_dir_light(old, new, world)
        slabs.change_dir_light(members, fabric, old, new, world)
        assert_slabs_equal(full, members, "change across faces")
        assert fabric.bytes_moved > 0
        # round 4: a slab's share of a pass ALONG z is one pipelined sweep (started from the planes the slab before handed on);
        # lateral passes keep the chunked chain (their tiles would need a hand-off per slice across handles)
        for m in members:
            p = m.res.path_counters()
            assert (p["launches_sweep"] > 0) == (n_slabs == 2) and p["launches_chain"] > 0 and p["launches_slice"] == 0, (m.slab_index, p)
        # the gathered volume is the whole unpartitioned one, on every handle
        slabs.gather_light_volume(members, fabric)
        ref = full.download_light_volume()
        for m in members:
            assert np.array_equal(m.res.download_light_volume(), ref), f"gathered light volume of slab {m.slab_index}"
    finally:
        for h in handles:
            h.close()


@pytest.mark.parametrize("switch", ["occ_list=0", "sparse_occ=0", "chunk_steps=4"])
def test_slabs_with_the_diagnostic_kernel_paths(gpu, tunables, switch):
    """the occlusion launch without the work list / without the empty-block flags (block rows outside the slab's reach are
    then cut inside the kernel), and 4-slice chunks (four times the exchanges)"""
    name, _, value = switch.partition("=")
    tunables(name, int(value))
    _, _, _, handles = make_handles(3, (72, 56, 64), np.uint16)
    full, parts = handles[0], handles[1:]
    members, fabric, _ = slab_setup(parts, 2)
    world = S.default_world()
    try:
        for i, (d, inten) in enumerate(LIGHTS[:5]):
            light = abi.DirLightParams(d, inten)
            full.add_dir_light(light, True, world)
            slabs.add_dir_light(members, fabric, light, True, world)
        old = abi.DirLightParams(*LIGHTS[1])
        new = abi.DirLightParams(S.rotate_z(LIGHTS[1][0], 5.0), LIGHTS[1][1])
        full.change_dir_light(old, new, world)
        slabs.change_dir_light(members, fabric, old, new, world)
        assert_slabs_equal(full, members, switch)
    finally:
        for h in handles:
            h.close()


def test_slab_lights_equal_oracle_and_render(gpu, oracle_mod):
    """End to end as config 4 words it: slab-partitioned reset, light-volume gather, frame in image tiles — against the oracle."""
    from tbraymarcherplugin_amd import sharding

    dims = (64, 64, 64)
    vol, lut, w, handles = make_handles(2, dims, np.uint16, seed=0x5EED0401)
    orc = oracle_mod.OracleScene(vol, False, False, abi.ADDRESS_WRAP, abi.BORDER_ENGINE_8BIT)
    orc.set_tf_lut(lut)
    orc.set_windowing(w)
    members, fabric, _ = slab_setup(handles, 2)
    world = S.default_world()
    lights = [S.light(i) for i in range(4)]
    try:
        slabs.reset_all_lights(members, fabric, lights, world, lambda m: m.res.clear_light_volume(0.0))
        orc.clear_light_volume(0.0)
        for l in lights:
            orc.add_dir_light(l, True, world)
        slabs.gather_light_volume(members, fabric)
        for m in members:
            assert np.array_equal(m.res.download_light_volume(), orc.light)
        cam = S.default_camera(64, 64)
        rp = abi.RaymarchParams(64.0, -1, True)
        want, _ = orc.raymarch_lit(cam, abi.Tile(0, 0, 64, 64, 1), rp, world)
        parts = [handles[r].raymarch_lit(cam, sharding.rank_tile(64, 64, r, 2), rp, world) for r in range(2)]
        frame = sharding.assemble(np.stack(parts), 64, 2)
        np.testing.assert_allclose(frame, want, rtol=0, atol=1e-4)
    finally:
        for h in handles:
            h.close()


def test_slab_arguments_are_checked(gpu):
    _, _, _, (res,) = make_handles(1, (32, 32, 64), np.uint8)
    world = S.default_world()
    with res:
        light = abi.DirLightParams((1, .3, -.5), 0.5)
        assert res.slab_light_begin(None, light, True, world, abi.Slab(0, 48)) == 2  # bounds are checked when a pass is planned
        with pytest.raises(abi.TbrmError):
            res.slab_pass_begin(0)
        with pytest.raises(abi.TbrmError):
            res.slab_pass_chunk(0)
        assert res.slab_light_begin(None, light, True, world, abi.Slab(0, 32)) == 2
        with pytest.raises(abi.TbrmError):
            res.slab_pass_begin(2)
        d = res.slab_pass_begin(0)
        assert d.n_chunks >= 1 and d.streams == 1
        with pytest.raises(abi.TbrmError):
            res.slab_pass_chunk(d.n_chunks)
        with pytest.raises(abi.TbrmError):
            res.slab_pass_plane(0, 1)


def test_full_size_slabs_equal_single_handle(gpu):
    """BASELINE config 3's volume (512^3 UNORM16, UNORM8 light volume) in 2 and 8 slabs: several occlusion spans per pass,
    16- and 8-slice chunks, sparse work lists — bit-identical to the unpartitioned operator."""
    import torch

    dims = (512, 512, 512)
    vol = S.make_volume_torch(dims, np.uint16, S.seed_for_config(3), torch.device("cuda", 0))
    torch.cuda.synchronize()
    lut = abi.color_curve_to_lut(S.tf_keys("A"))
    w = abi.WindowingParams(0.5, 0.9, True, False)
    world = S.default_world()

    def handle():
        res = abi.Resources(dims, abi.FMT_G16, False, False, 0)
        res.upload_volume_device(vol.data_ptr(), vol.numel() * vol.element_size())
        res.set_tf_lut(lut)
        res.set_windowing(w)
        res.clear_light_volume(0.0)
        return res

    full = handle()
    ops = [("add", S.light(0)), ("add", S.light(2)), ("change", S.light(0), abi.DirLightParams(S.rotate_z(S.LIGHTS[0][0], 5.0), S.LIGHTS[0][1])),
           ("change", S.light(2), abi.DirLightParams(S.rotate_z(S.LIGHTS[2][0], 5.0), S.LIGHTS[2][1]))]
    try:
        for op in ops:
            if op[0] == "add":
                full.add_dir_light(op[1], True, world)
            else:
                full.change_dir_light(op[1], op[2], world)
        ref = full.download_light_volume()
        for n_slabs in (2, 8):
            parts = [handle() for _ in range(n_slabs)]
            try:
                members, fabric, _ = slab_setup(parts, n_slabs)
                for op in ops:
                    if op[0] == "add":
                        slabs.add_dir_light(members, fabric, op[1], True, world)
                    else:
                        slabs.change_dir_light(members, fabric, op[1], op[2], world)
                for m in members:
                    got = m.res.download_light_volume()[m.z_begin:m.z_end]
                    bad = np.count_nonzero(got != ref[m.z_begin:m.z_end])
                    assert bad == 0, f"{n_slabs} slabs, slab {m.slab_index}: {bad} voxels differ"
            finally:
                for p in parts:
                    p.close()
    finally:
        full.close()


# ---- the lit frame, slab by slab ---------------------------------------------------------------------------------------

def _lit_handles(n_handles, dims, dtype, addr, half_res=False, light_32bit=False):
    import torch

    vol = small_volume(dims, dtype, 0x5EED0402)
    lut = abi.color_curve_to_lut(S.tf_keys("A"))
    out = []
    for _ in range(n_handles):
        res = abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)], light_32bit, half_res, 0, addr)
        res.upload_volume(vol)
        res.set_tf_lut(lut)
        res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
        res.clear_light_volume(0.0)
        for i in range(3):
            res.add_dir_light(S.light(i), True, S.default_world())
        out.append(res)
    return out


@pytest.mark.parametrize("addr", [abi.ADDRESS_WRAP, abi.ADDRESS_CLAMP])
@pytest.mark.parametrize("n_slabs,dims,half_res", [(2, (72, 56, 64), False), (4, (48, 64, 128), False), (2, (64, 64, 128), True)])
def test_frame_marched_slab_by_slab_is_the_plain_frame(gpu, addr, n_slabs, dims, half_res, ray_lanes_env):
    """tbrm_raymarch_lit_slab_device: each handle accumulates only its slab's samples, the state travels up and down
    through the slabs — bit-identical to tbrm_raymarch_lit for cameras outside, inside and grazing the volume, with a clip
    plane, jitter, tiles, skipping on and off."""
    import torch

    handles = _lit_handles(n_slabs, dims, np.uint16, addr, half_res)
    members, fabric, _ = slab_setup(handles, n_slabs)
    dev = torch.device("cuda", 0)
    try:
        eyes = [(-145, -95, 80), (130, 40, -160), (10, -5, 190), (0.001, 300, 2), (20, -10, 15), (-60, -60, -60)]
        worlds = [S.default_world(), abi.make_world(abi.identity_transform(100.0, (5, -3, 2), (0.1, 0.2, -0.1, 0.97)), (10, 0, 5), (0.3, 0.2, -0.93))]
        case = 0
        for eye in eyes:
            for world in worlds:
                case += 1
                w, h = (72, 48) if case % 2 else (64, 64)
                cam = abi.look_at_camera(np.array(eye, dtype=float), (3.0, -2.0, 1.0), (0.0, 0.0, 1.0), 50.0, w, h)
                rp = abi.RaymarchParams(float(40 + 23 * case), (case % 3) - 1, bool(case % 2))
                tile = abi.Tile(0, 0, w, h, 1) if case % 3 else abi.Tile(8, 8, w - 16, 16, 2)
                want = handles[0].raymarch_lit(cam, tile, rp, world)
                got = slabs.render_lit(members, fabric, cam, tile, rp, world,
                                       lambda: torch.zeros((tile.h, tile.w, 4), dtype=torch.float32, device=dev))
                torch.cuda.synchronize()
                got = got.cpu().numpy()
                assert np.array_equal(got, want), f"eye {eye} case {case}: max |d| = {np.abs(got - want).max()}"
                assert want[..., 3].max() > 0.0 or case > 0
    finally:
        for hd in handles:
            hd.close()


@pytest.fixture(params=["4", "8"])
def ray_lanes_env(request, tunables):
    tunables("ray_lanes", int(request.param))
    return request.param


# ---- slab-resident handles: each GPU holds only its part of the two volumes ------------------------------------------------

@pytest.mark.parametrize("addr", [abi.ADDRESS_WRAP, abi.ADDRESS_CLAMP])
@pytest.mark.parametrize("n_slabs,dims,half_res,light_32bit", [(2, (72, 56, 64), False, False), (4, (48, 64, 128), False, False),
                                                              (2, (64, 64, 128), True, False), (2, (56, 72, 64), False, True)])
def test_slab_resident_handles_light_and_render_like_one_handle(gpu, addr, n_slabs, dims, half_res, light_32bit):
    """Handles that hold only their slab of the data and light volumes (plus halos): slab-partitioned light operators,
    light-volume halo exchange, frame marched slab by slab — light volume and frame bit-identical to one whole handle."""
    import torch

    vol = small_volume(dims, np.uint16, 0x5EED0403)
    lut = abi.color_curve_to_lut(S.tf_keys("A"))
    w = abi.WindowingParams(0.5, 0.9, True, False)
    fmt = abi.FMT_G16
    full = abi.Resources(dims, fmt, light_32bit, half_res, 0, addr)
    full.upload_volume(vol)
    depth = full.light_dims[2]
    bounds = slabs.slab_bounds(depth, n_slabs)
    parts = [abi.Resources(dims, fmt, light_32bit, half_res, 0, addr, owned=abi.Slab(*bounds[k])) for k in range(n_slabs)]
    handles = [full] + parts
    dev = torch.device("cuda", 0)
    try:
        for res in handles:
            res.set_tf_lut(lut)
            res.set_windowing(w)
        resident_fraction = []
        for res in parts:
            res.upload_resident_part(vol)
            (dlo, dhi, dwrap), (llo, lhi, lwrap) = res.resident_slices()
            resident_fraction.append((dhi - dlo) / dims[2])
            assert llo <= res.owned.z_begin and lhi >= res.owned.z_end
            res.clear_light_volume(0.0)
        full.clear_light_volume(0.0)
        if n_slabs == 4:
            assert min(resident_fraction) < 1.0  # a middle slab really holds less than the volume
        members = [slabs.DeviceSlab(res, k, *bounds[k]) for k, res in enumerate(parts)]
        fabric = slabs.make_fabric([b[0] for b in bounds] + [depth])
        world = S.default_world()
        for d, inten in LIGHTS[:6]:
            light = abi.DirLightParams(d, inten)
            full.add_dir_light(light, True, world)
            slabs.add_dir_light(members, fabric, light, True, world)
        old = abi.DirLightParams(*LIGHTS[1])
        new = abi.DirLightParams(S.rotate_z(LIGHTS[1][0], 5.0), LIGHTS[1][1])
        full.change_dir_light(old, new, world)
        slabs.change_dir_light(members, fabric, old, new, world)
        ref = full.download_light_volume()
        for m in members:
            got = m.res.download_light_slices(m.z_begin, m.z_end - m.z_begin)
            assert np.array_equal(got, ref[m.z_begin:m.z_end]), f"light volume of slab {m.slab_index}"
        slabs.exchange_light_halos(members, fabric)
        for case, eye in enumerate([(-145, -95, 80), (10, -5, 190), (0.001, 300, 2), (20, -10, 15), (30, 20, -170)]):
            cam = abi.look_at_camera(np.array(eye, dtype=float), (3.0, -2.0, 1.0), (0.0, 0.0, 1.0), 50.0, 64, 48)
            rp = abi.RaymarchParams(float(50 + 31 * case), (case % 3) - 1, True)
            tile = abi.Tile(0, 0, 64, 48, 1)
            want = full.raymarch_lit(cam, tile, rp, world)
            got = slabs.render_lit(members, fabric, cam, tile, rp, world, lambda: torch.zeros((48, 64, 4), dtype=torch.float32, device=dev))
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            assert np.array_equal(got, want), f"eye {eye}: max |d| = {np.abs(got - want).max()}"
        # the whole-volume operators refuse a handle that does not hold the whole volume
        with pytest.raises(abi.TbrmError):
            parts[0].add_dir_light(abi.DirLightParams((1, 0, 0), 0.1), True, world)
        with pytest.raises(abi.TbrmError):
            parts[0].raymarch_lit(cam, tile, rp, world)
        with pytest.raises(abi.TbrmError):
            parts[0].download_light_volume()
    finally:
        for res in handles:
            res.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_slab_resident_scenes_against_one_handle(gpu, seed):
    """Seeded random scenes on slab-resident handles: ragged x / y sizes, 2-4 slabs, formats, rotated and scaled volumes with
    clip planes, random light sequences, random cameras — light volume and frame bit-identical to one whole handle. A light
    whose pass only the slice kernel can run (no slab form) is skipped on both sides."""
    import torch

    rng = np.random.default_rng(0x5EED0800 + seed)
    depth = int(rng.choice([64, 96, 128]))
    n_slabs = int(rng.choice([s for s in (2, 3, 4) if depth % (32 * s) == 0]))
    half_res = seed % 5 == 4  # the light volume at half the data volume's resolution: `depth` is the LIGHT volume's
    dims = (int(rng.integers(40, 150)), int(rng.integers(40, 150)), 2 * depth if half_res else depth)
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    light_32bit = seed % 4 == 3
    addr = abi.ADDRESS_CLAMP if seed % 2 else abi.ADDRESS_WRAP
    vol = small_volume(dims, dtype, 0x5EED0810 + seed)
    lut = abi.color_curve_to_lut(S.tf_keys("AB"[seed % 2]))
    w = abi.WindowingParams(float(rng.uniform(0.35, 0.65)), float(rng.uniform(0.5, 1.1)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    fmt = abi.DTYPE_FMT[np.dtype(dtype)]
    full = abi.Resources(dims, fmt, light_32bit, half_res, 0, addr)
    full.upload_volume(vol)
    assert full.light_dims[2] == depth
    bounds = slabs.slab_bounds(depth, n_slabs)
    parts = [abi.Resources(dims, fmt, light_32bit, half_res, 0, addr, owned=abi.Slab(*bounds[k])) for k in range(n_slabs)]
    dev = torch.device("cuda", 0)
    if seed % 2:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        tr = abi.identity_transform(tuple(float(v) for v in rng.uniform(70, 130, size=3)), tuple(float(v) for v in rng.uniform(-15, 15, size=3)),
                                    tuple(float(v) for v in q))
        cd = rng.normal(size=3)
        world = abi.make_world(tr, tuple(float(v) for v in rng.uniform(-25, 25, size=3)), tuple(float(v) for v in cd / np.linalg.norm(cd)))
    else:
        world = S.default_world()
    try:
        for res in [full] + parts:
            res.set_tf_lut(lut)
            res.set_windowing(w)
            res.clear_light_volume(0.0)
        for res in parts:
            res.upload_resident_part(vol)
        members = [slabs.DeviceSlab(res, k, *bounds[k]) for k, res in enumerate(parts)]
        fabric = slabs.make_fabric([b[0] for b in bounds] + [depth])
        present, ran = [], 0
        for step in range(6):
            d = rng.normal(size=3)
            if step % 3 == 2:
                d *= 0.1
                d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
            light = abi.DirLightParams(tuple(float(v) for v in d), float(rng.uniform(0.15, 0.6)))
            def unsupported(fn):
                """runs fn; True when the slab form declined the operation (checked before anything is enqueued)"""
                try:
                    fn()
                    return False
                except abi.TbrmError as e:
                    assert e.code == abi.ERR_UNSUPPORTED, e
                    return True

            if present and step % 2:
                old = present[int(rng.integers(0, len(present)))]
                fused = [None]
                if unsupported(lambda: fused.__setitem__(0, slabs.light_operation(members, fabric, old, light, True, world))):
                    continue
                if fused[0]:
                    full.change_dir_light(old, light, world)
                    present[present.index(old)] = light
                else:  # across faces: remove, then add — each all or nothing
                    if unsupported(lambda: slabs.add_dir_light(members, fabric, old, False, world)):
                        continue
                    full.add_dir_light(old, False, world)
                    present.remove(old)
                    if not unsupported(lambda: slabs.add_dir_light(members, fabric, light, True, world)):
                        full.add_dir_light(light, True, world)
                        present.append(light)
            else:
                if unsupported(lambda: slabs.add_dir_light(members, fabric, light, True, world)):
                    continue
                full.add_dir_light(light, True, world)
                present.append(light)
            ran += 1
        assert ran > 0
        ref = full.download_light_volume()
        for m in members:
            got = m.res.download_light_slices(m.z_begin, m.z_end - m.z_begin)
            assert np.array_equal(got, ref[m.z_begin:m.z_end]), f"scene {seed}: light volume of slab {m.slab_index}"
        slabs.exchange_light_halos(members, fabric)
        for case in range(3):
            eye = rng.normal(size=3)
            eye = eye / np.linalg.norm(eye) * float(rng.choice([30.0, 120.0, 260.0]))
            cam = abi.look_at_camera(eye, tuple(float(v) for v in rng.uniform(-10, 10, size=3)), (0.0, 0.0, 1.0), 55.0, 56, 40)
            rp = abi.RaymarchParams(float(rng.integers(30, 200)), int(rng.integers(-1, 8)), bool(case % 2))
            tile = abi.Tile(0, 0, 56, 40, 1)
            want = full.raymarch_lit(cam, tile, rp, world)
            got = slabs.render_lit(members, fabric, cam, tile, rp, world, lambda: torch.zeros((40, 56, 4), dtype=torch.float32, device=dev))
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), want), f"scene {seed} camera {case}"
    finally:
        for res in [full] + parts:
            res.close()


@pytest.mark.parametrize("light_32bit", [False, True])
@pytest.mark.parametrize("dims,steep", [((32, 120, 128), (-0.15, 0.05, 1)), ((128, 120, 64), (1, 0.05, -0.12))])  # lateral / along z
def test_slabs_run_steep_passes_slice_by_slice(gpu, dims, steep, light_32bit):
    """Anisotropic volumes: a light close to an axis makes its second pass fetch the previous slice dozens of texels away —
    beyond the chunk kernels. In slab form such a pass runs one slice per step on the reference's read / write buffers
    (UNORM8 or float), halo rows = the taps' reach; bit-identical to the unpartitioned operator (which uses the same kernel)."""
    _, _, _, handles = make_handles(3, dims, np.uint16, light_32bit)
    full, parts = handles[0], handles[1:]
    members, fabric, _ = slab_setup(parts, 2)
    world = S.default_world()
    try:
        a = abi.DirLightParams(steep, 0.5)
        full.add_dir_light(a, True, world)
        slabs.add_dir_light(members, fabric, a, True, world)
        assert_slabs_equal(full, members, "steep add")
        assert full.launch_counters()["slice"] > 0 and parts[0].launch_counters()["slice"] > 0, "the test is meant to reach the slice kernel"
        b = abi.DirLightParams(tuple(v * (1.0 + 0.1 * i) for i, v in enumerate(steep)), 0.4)  # same faces: fused Change, slice by slice
        full.change_dir_light(a, b, world)
        slabs.change_dir_light(members, fabric, a, b, world)
        assert_slabs_equal(full, members, "steep fused change")
        c = abi.DirLightParams((0.3, 0.5, -0.8), 0.3)
        full.change_dir_light(b, c, world)
        slabs.change_dir_light(members, fabric, b, c, world)
        assert_slabs_equal(full, members, "change across faces")
        full.add_dir_light(c, False, world)
        slabs.add_dir_light(members, fabric, c, False, world)
        assert_slabs_equal(full, members, "remove")
    finally:
        for h in handles:
            h.close()


def test_dist_fabric_orders_rccl_transfers_with_the_handles_stream(gpu):
    """dist_fabric over RCCL (backend nccl) with device tensors, called WITHOUT an outer stream context: the fabric itself
    enqueues the point-to-point batch relative to the handle's stream, so a plane that a chunk kernel is still writing is
    sent after the kernel, and what is received is in place before the next operation on that stream reads it. One rank on
    this box's one GPU: the batch is a send to and a receive from rank 0 itself, which RCCL executes as a device copy."""
    import socket

    import torch
    import torch.distributed as dist

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        _, _, _, (res,) = make_handles(1, (128, 96, 64), np.uint16)
        depth = res.light_dims[2]
        member = slabs.DeviceSlab(res, 0, 0, depth)
        fabric = slabs.dist_fabric([0, depth], 0, 1, member=member)
        with pytest.raises(ValueError):  # device tensors and no member: refused instead of racing
            slabs.dist_fabric([0, depth], 0, 1)._p2p([("send", torch.zeros(4, device="cuda"), 0), ("recv", torch.zeros(4, device="cuda"), 0)])
        world = S.default_world()
        light = abi.DirLightParams((1, .35, -.5), 0.5)
        assert member.light_begin(None, light, True, world) >= 1
        desc = member.pass_begin(0)
        landing = torch.full((desc.plane_h, desc.plane_w), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for c in range(desc.n_chunks):
            member.pass_chunk(c)  # enqueued on the handle's stream, not waited for
        final = member.plane(desc.n_chunks, 0)  # what the last chunk writes
        fabric._p2p([("send", final, 0), ("recv", landing, 0)])  # no stream context around this call
        res.flush()
        torch.cuda.synchronize()
        assert torch.equal(landing, final) and float(landing.min()) >= 0.0 and float(landing.max()) > 0.0
        res.close()
    finally:
        dist.destroy_process_group()


def test_repartitioned_handles_keep_their_partitions_factors_apart(gpu, tunables):
    """The same handles partitioned in 2 slabs, then in 4 (what tests/test_gpu_full_size.py does at 1024^3, where this was found):
    a slab's share of a pass along z is swept from kept occlusion factors, and the factor cache's key has to hold WHICH slices
    the share covers — a downward pass of slab 0 starts at slice 63 in two slabs and at slice 31 in four; blocks and ranks of the
    one are not the other's."""
    tunables("light_cache_mb", -1)
    tunables("slab_sweep", 1)
    dims = (64, 64, 128)
    _, _, _, handles = make_handles(5, dims, np.uint16)
    full, parts = handles[0], handles[1:]
    world = S.default_world()
    lights = [abi.DirLightParams(d, 0.35) for d in ((.2, -.3, -1), (.1, .45, 1), (-.25, .1, 1))]  # passes along z, both directions
    try:
        for n_slabs in (2, 4, 2):
            members, fabric, _ = slab_setup(parts[:n_slabs], n_slabs)
            for h in [full] + parts[:n_slabs]:
                h.clear_light_volume(0.0)
            for light in lights:
                full.add_dir_light(light, True, world)
                slabs.add_dir_light(members, fabric, light, True, world)
            assert_slabs_equal(full, members, f"{n_slabs} slabs")
            for light in lights[:2]:  # removals: from the factors the adds kept
                full.add_dir_light(light, False, world)
                slabs.add_dir_light(members, fabric, light, False, world)
            assert_slabs_equal(full, members, f"{n_slabs} slabs, removals")
        assert parts[0].path_counters()["launches_sweep"] > 0 and parts[0].light_cache_stats()["hits"] > 0
    finally:
        for h in handles:
            h.close()
