"""The C-ABI library loads and exports every symbol include/tbrm.h declares; struct layouts match (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from tbraymarcherplugin_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tbrm.h")


def declared_symbols():
    text = open(HEADER).read()
    return re.findall(r"TBRM_API\s+[\w\s\*]+?\b(tbrm_\w+)\s*\(", text)


def test_header_symbols_are_exported_and_bound():
    lib = abi.load()
    declared = declared_symbols()
    assert len(declared) >= 30
    assert sorted(declared) == sorted(abi.SYMBOLS), set(declared) ^ set(abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in tbrm.h but not exported by libtbrm.so"
    assert b"gfx950" in lib.tbrm_version()
    header_version = int(re.search(r"#define\s+TBRM_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.tbrm_abi_version() == header_version == abi.ABI_VERSION  # (abi.load() refuses a library with another number)


def test_struct_layouts_match_the_header():
    structs = {"tbrm_vec3d": abi.Vec3d, "tbrm_quatd": abi.Quatd, "tbrm_transform": abi.Transform,
               "tbrm_dir_light_params": abi.DirLightParams, "tbrm_clipping_plane_params": abi.ClippingPlaneParams,
               "tbrm_world_params": abi.WorldParams, "tbrm_windowing_params": abi.WindowingParams,
               "tbrm_resources_desc": abi.ResourcesDesc, "tbrm_camera": abi.Camera, "tbrm_tile": abi.Tile,
               "tbrm_raymarch_params": abi.RaymarchParams, "tbrm_light_pass": abi.LightPass}
    src = '#include <stdio.h>\n#include "tbrm.h"\nint main(void){\n'
    for name in structs:
        src += f'  printf("{name} %zu\\n", sizeof({name}));\n'
    src += "  return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "sizes.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "sizes")
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for name, cls in structs.items():
        assert int(sizes[name]) == C.sizeof(cls), name


def test_argument_validation_without_touching_a_device():
    lib = abi.load()
    assert lib.tbrm_resources_create(None, None) == abi.ERR_INVALID_ARG
    assert lib.tbrm_add_dir_light(None, None, 1, None, None, 0) == abi.ERR_INVALID_ARG
    assert lib.tbrm_flush(None) == abi.ERR_INVALID_ARG
    assert lib.tbrm_resources_destroy(None) == abi.OK
    assert b"null" in lib.tbrm_last_error()


def test_every_handle_taking_entry_point_rejects_a_null_handle():
    """no entry point dereferences a null handle: TBRM_ERR_INVALID_ARG (or, for the two queries that return a value, 0) and a message"""
    lib = abi.load()
    z = C.c_void_p(None)
    buf = (C.c_float * 64)()
    i3 = (C.c_int32 * 3)()
    calls = {
        "tbrm_resources_light_volume_dims": lambda: lib.tbrm_resources_light_volume_dims(z, C.byref(i3)),
        "tbrm_upload_volume": lambda: lib.tbrm_upload_volume(z, buf, 4),
        "tbrm_upload_volume_device": lambda: lib.tbrm_upload_volume_device(z, buf, 4),
        "tbrm_upload_volume_slices": lambda: lib.tbrm_upload_volume_slices(z, 0, 8, buf, 4),
        "tbrm_resources_reserve": lambda: lib.tbrm_resources_reserve(z, 4, 0),
        "tbrm_set_tf_lut": lambda: lib.tbrm_set_tf_lut(z, buf),
        "tbrm_set_windowing": lambda: lib.tbrm_set_windowing(z, C.byref(abi.WindowingParams())),
        "tbrm_add_dir_lights": lambda: lib.tbrm_add_dir_lights(z, None, 0, 1, C.byref(abi.make_world()), None, None),
        "tbrm_change_dir_light": lambda: lib.tbrm_change_dir_light(z, None, None, None, None, 0),
        "tbrm_clear_light_volume": lambda: lib.tbrm_clear_light_volume(z, 0.0),
        "tbrm_raymarch_lit": lambda: lib.tbrm_raymarch_lit(z, None, None, None, None, buf),
        "tbrm_raymarch_lit_device": lambda: lib.tbrm_raymarch_lit_device(z, None, None, None, None, None, buf),
        "tbrm_raymarch_lit_slab_device": lambda: lib.tbrm_raymarch_lit_slab_device(z, None, None, None, None, None, buf, None, 0),
        "tbrm_raymarch_intensity": lambda: lib.tbrm_raymarch_intensity(z, None, None, None, None, buf),
        "tbrm_raymarch_octree": lambda: lib.tbrm_raymarch_octree(z, None, None, None, None, 0, buf),
        "tbrm_generate_octree": lambda: lib.tbrm_generate_octree(z),
        "tbrm_octree_mip_dims": lambda: lib.tbrm_octree_mip_dims(z, 0, C.byref(i3)),
        "tbrm_count_nominal_samples": lambda: lib.tbrm_count_nominal_samples(z, None, None, None, None, None),
        "tbrm_download_light_volume": lambda: lib.tbrm_download_light_volume(z, buf, 4),
        "tbrm_download_light_slices": lambda: lib.tbrm_download_light_slices(z, 0, 8, buf, 4),
        "tbrm_upload_light_volume": lambda: lib.tbrm_upload_light_volume(z, buf, 4),
        "tbrm_light_volume_device_ptr": lambda: lib.tbrm_light_volume_device_ptr(z, None, None),
        "tbrm_slab_light_begin": lambda: lib.tbrm_slab_light_begin(z, None, None, 1, None, None, None),
        "tbrm_slab_pass_begin": lambda: lib.tbrm_slab_pass_begin(z, 0, None),
        "tbrm_slab_pass_chunk": lambda: lib.tbrm_slab_pass_chunk(z, 0),
        "tbrm_slab_pass_plane": lambda: lib.tbrm_slab_pass_plane(z, 0, 0, None),
        "tbrm_slab_resident_slices": lambda: lib.tbrm_slab_resident_slices(z, C.byref(i3), C.byref(i3)),
        "tbrm_slab_light_halo": lambda: lib.tbrm_slab_light_halo(z, 0, None, None, None),
        "tbrm_launch_counters": lambda: lib.tbrm_launch_counters(z, None),
        "tbrm_light_cache_stats": lambda: lib.tbrm_light_cache_stats(z, None),
        "tbrm_sweep_launches": lambda: lib.tbrm_sweep_launches(z, None),
        "tbrm_light_cache_clear": lambda: lib.tbrm_light_cache_clear(z),
        "tbrm_path_counters": lambda: lib.tbrm_path_counters(z, None),
        "tbrm_stream": lambda: lib.tbrm_stream(z, None),
        "tbrm_last_gpu_time_ms": lambda: lib.tbrm_last_gpu_time_ms(z, 0, None),
    }
    for name, call in calls.items():
        assert call() == abi.ERR_INVALID_ARG, name
        assert lib.tbrm_last_error(), name
    assert lib.tbrm_resources_is_initialized(z) == 0
    assert lib.tbrm_resources_create_slab(None, None, None) == abi.ERR_INVALID_ARG
    # every symbol that takes the handle first is covered here or in the test above
    covered = set(calls) | {"tbrm_resources_destroy", "tbrm_resources_is_initialized", "tbrm_add_dir_light", "tbrm_flush",
                            "tbrm_raymarch_intensity_device", "tbrm_raymarch_octree_device", "tbrm_download_octree_mip"}
    free = {"tbrm_abi_version", "tbrm_version", "tbrm_last_error", "tbrm_device_count", "tbrm_set_tunable", "tbrm_get_tunable", "tbrm_resources_create", "tbrm_resources_create_slab", "tbrm_color_curve_to_lut",
            "tbrm_make_default_tf_lut", "tbrm_host_bake_tf_lut", "tbrm_selftest_unorm_decode", "tbrm_selftest_unorm8_roundtrip", "tbrm_selftest_window_division", "tbrm_selftest_opacity_correction",
            "tbrm_host_light_passes", "tbrm_host_plan_light", "tbrm_host_local_clipping", "tbrm_host_data_border", "tbrm_host_world_to_local"}
    assert set(abi.SYMBOLS) == covered | free, set(abi.SYMBOLS) ^ (covered | free)


@pytest.mark.skipif(abi.device_count() > 0, reason="only meaningful on a machine without a HIP device")
def test_no_cpu_fallback_without_a_device():
    """The product path fails loudly instead of computing on the host."""
    with pytest.raises(abi.TbrmError) as e:
        abi.Resources((8, 8, 8), abi.FMT_G8)
    assert e.value.code == abi.ERR_NO_DEVICE
