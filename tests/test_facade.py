"""The C++ host side (include/tbrm_plugin.hpp: ARaymarchVolume / URaymarchUtils mirror) builds against the C-ABI with
plain g++, and on a GPU box reproduces the reference's Tick policy (RaymarchVolume.cpp:327-416)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
LIB_DIR = os.path.join(ROOT, "tbraymarcherplugin_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "facade_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L", LIB_DIR, "-ltbrm", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_facade_compiles_and_links_with_gxx(tmp_path, abi_mod):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_facade_tick_policy_on_gpu(tmp_path, gpu):
    out = subprocess.run([build(tmp_path), "32"], check=True, capture_output=True, text=True).stdout
    lines = dict(l.split(" ", 1) for l in out.strip().splitlines() if " " in l)
    assert out.strip().endswith("OK"), out
    assert lines["after_first_tick"] == "resets=1 adds=3 changes=0"      # recompute requested -> ResetAllLights
    assert lines["after_idle_tick"] == "resets=1 adds=3 changes=0"       # nothing moved: no work
    assert lines["after_one_moved"] == "resets=1 adds=3 changes=1"       # 1 of 3 lights -> ChangeDirLightInSingleVolume
    assert lines["after_two_moved"] == "resets=2 adds=6 changes=1"       # >1 and >= half -> full reset is quicker
    assert lines["after_window_change"] == "resets=3 adds=9 changes=1"   # windowing change -> bRequestedRecompute
    mean_a = float(lines["render"].split("mean_alpha=")[1].split()[0])
    assert 0.01 < mean_a < 0.9
    hit = float(lines["intensity"].split("hit_fraction=")[1].split()[0])
    grey = float(lines["intensity"].split("mean_intensity_of_hits=")[1].split()[0])
    assert 0.05 < hit < 1.0 and 0.0 <= grey <= 1.0                       # SwitchRenderer(Intensity): the slice view
    oct_a = float(lines["octree"].split("mean_alpha=")[1].split()[0])
    assert 0.01 < oct_a < 0.95 and "rebuild_pending=0" in lines["octree"]  # SwitchRenderer(Octree): pyramid built, level 1 marched
    # X axis of a 90 degree rotation about z, scaled by 2 -> (0, 2, 0); translation in row 3
    assert lines["helpers"] == "dims=32,32,32 tex=0.00,0.50,0.75 local=-1.00,0.00,0.50 row0=0.000,2.000,0.000 row3=1,2,3"
    assert lines["batched_reset"].startswith("resets=4 adds=12 ")          # bBatchLightsOnReset: one tbrm_add_dir_lights call
    assert abs(float(lines["batched_reset"].split("mean_alpha=")[1]) - mean_a) < 1e-6  # alpha does not depend on the light
    assert "slice=0" in lines["launches"] and "raymarch=4" in lines["launches"]  # two lit + one intensity + one octree frame


SLABS_SRC = os.path.join(ROOT, "tests", "cpp", "slabs_test.cpp")


def build_slabs(tmp_path):
    exe = str(tmp_path / "slabs_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    SLABS_SRC, "-o", exe, "-L", LIB_DIR, "-ltbrm", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIB_DIR}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_slab_driver_compiles_and_links_with_gxx(tmp_path, abi_mod):
    """include/tbrm_slabs.hpp (FSlabGroup: the single-process C++ driver of the slab-partitioned operators)"""
    assert os.path.exists(build_slabs(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("n_slabs,host_sync", [(2, 0), (4, 0), (3, 0), (4, 1)])
def test_slab_driver_on_gpu(tmp_path, gpu, n_slabs, host_sync):
    """slab-resident handles driven from C++, their streams ordered by events only (host_sync 0) or drained around every
    exchange (1): light volume and frame bit-identical to one whole handle"""
    p = subprocess.run([build_slabs(tmp_path), str(n_slabs), str(host_sync)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr
    assert "light volume: 0 voxels differ" in p.stdout and "frame: identical" in p.stdout


EXAMPLE_SRC = os.path.join(ROOT, "examples", "render_mhd.cpp")


def build_example(tmp_path):
    exe = str(tmp_path / "render_mhd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), EXAMPLE_SRC, "-o", exe, "-L", LIB_DIR, "-ltbrm",
                    "-lz", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_example_compiles(tmp_path, abi_mod):
    assert os.path.exists(build_example(tmp_path))


@pytest.mark.gpu
def test_example_renders_an_mhd_file(tmp_path, gpu):
    """examples/render_mhd.cpp: .mhd in (signed 16-bit, as CT data comes), frame out — loader, facade, Tick and the kernels together"""
    import numpy as np
    from tbraymarcherplugin_amd import synthetic as S

    vol = S.make_volume_numpy((48, 40, 36), np.float32, 0x5EED0A00)          # [z, y, x] in [0, 1]
    hu = (vol * 3000.0 - 1000.0).astype(np.int16)                            # Hounsfield-like units
    hu.tofile(tmp_path / "ct.raw")
    (tmp_path / "ct.mhd").write_text("ObjectType = Image\nNDims = 3\nDimSize = 48 40 36\nElementSpacing = 1 1 1.25\n"
                                     "ElementType = MET_SHORT\nElementDataFile = ct.raw\n")
    p = subprocess.run([build_example(tmp_path), str(tmp_path / "ct.mhd"), str(tmp_path / "out.ppm"), "96", "64", "80"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "volume 48 x 40 x 36" in p.stdout
    data = (tmp_path / "out.ppm").read_bytes()
    assert data.startswith(b"P6\n96 64\n255\n") and len(data) == len(b"P6\n96 64\n255\n") + 96 * 64 * 3
    mean_alpha = float(p.stdout.split("mean alpha")[1].split(")")[0])
    assert 0.01 < mean_alpha < 0.95
    assert np.frombuffer(data[-96 * 64 * 3:], dtype=np.uint8).max() > 20    # something lit is visible


TILES_SRC = os.path.join(ROOT, "tests", "cpp", "tiles_test.cpp")


def build_tiles(tmp_path, rccl=False):
    exe = str(tmp_path / ("tiles_test_rccl" if rccl else "tiles_test"))
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__"] + (["-DTBRM_TILES_WITH_RCCL"] if rccl else []) +
                   ["-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", TILES_SRC, "-o", exe, "-L", LIB_DIR, "-ltbrm", "-L", "/opt/rocm/lib",
                    "-lamdhip64"] + (["-lrccl"] if rccl else []) + [f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


@pytest.mark.parametrize("rccl", [False, True])
def test_tile_driver_compiles_and_links_with_gxx(tmp_path, abi_mod, rccl):
    """include/tbrm_tiles.hpp (FTileGroup: N handles in one process, interleaved 8-row tiles, gather by peer copies or — linked
    from C++ — RCCL's ncclAllGather)"""
    assert os.path.exists(build_tiles(tmp_path, rccl))


@pytest.mark.gpu
@pytest.mark.parametrize("n_handles,mode", [(2, 0), (4, 0), (3, 1), (4, 1), (1, 0)])
def test_tile_driver_on_gpu(tmp_path, gpu, n_handles, mode):
    """N whole-volume handles driven from C++ on one GPU: a light turns on every replica, each marches its interleaved rows, the tiles
    are gathered to one handle (mode 0) or to all (1) by copies ordered with events only — light volumes and frames bit-identical
    to one handle doing everything alone, four steps back to back"""
    p = subprocess.run([build_tiles(tmp_path), str(n_handles), str(mode)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr
    assert f"light volumes: 0 of {n_handles} replicas differ" in p.stdout and "frames: 0 differ" in p.stdout


@pytest.mark.gpu
def test_tile_driver_gathers_with_rccl_from_cpp(tmp_path, gpu):
    """the same with the gather as ncclAllGather on the handles' own streams (one communicator per handle, ncclCommInitAll): what
    a box with ONE GPU can run of it is a group of one — RCCL refuses two ranks on one device"""
    p = subprocess.run([build_tiles(tmp_path, True), "1", "2"], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr
    assert "frames: 0 differ" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tile_driver_over_every_visible_device(tmp_path, gpu, abi_mod, mode):
    """the same binary over the box's real devices (TBRM_TILES_DEVICES, one handle per device, up to 8): peer copies to one / to every
    handle, and RCCL's ncclAllGather across the devices. On a one-GPU box this is a group of one on device 0; on an 8-GPU node it is
    the first execution of the peer-access / ncclCommInitAll paths on distinct devices."""
    n = max(1, min(int(abi_mod.device_count()), 8))
    while n > 1 and (128 if n == 8 else 96) % (8 * n):
        n -= 1
    env = dict(os.environ, TBRM_TILES_DEVICES=",".join(str(d) for d in range(n)))
    p = subprocess.run([build_tiles(tmp_path, mode == 2), str(n), str(mode)], capture_output=True, text=True, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr
    assert "devices: " + " ".join(str(d) for d in range(n)) in p.stdout
    assert f"light volumes: 0 of {n} replicas differ" in p.stdout and "frames: 0 differ" in p.stdout

