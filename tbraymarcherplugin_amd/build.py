"""Builds libtbrm.so (C-ABI + gfx950 kernels) in-tree with hipcc. Used by __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtbrm.so")

SOURCES = ["tbrm_api.cpp", "tbrm_light_passes.cpp", "tbrm_host_math.cpp", "tbrm_kernels.hip", "tbrm_light_kernels.hip"]
HEADERS = ["tbrm_internal.h", "tbrm_resources.h", "tbrm_device_math.h", "tbrm_device_sampling.h", "tbrm_host_math.h", "../../include/tbrm.h"]

# -ffp-contract=off + explicit fma is the arithmetic contract with the oracle (DESIGN.md "Arithmetic spec").
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU path)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path()] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
