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


@pytest.mark.skipif(abi.device_count() > 0, reason="only meaningful on a machine without a HIP device")
def test_no_cpu_fallback_without_a_device():
    """The product path fails loudly instead of computing on the host."""
    with pytest.raises(abi.TbrmError) as e:
        abi.Resources((8, 8, 8), abi.FMT_G8)
    assert e.value.code == abi.ERR_NO_DEVICE
