"""Builds libtbrm.so (C-ABI + gfx950 kernels) in-tree with hipcc. Used by __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtbrm.so")

SWEEP_TILE_ROWS = 16 if os.environ.get("TBRM_BUILD_VARIANTS") == "1" else 32

# (source, object name, extra flags): tbrm_light_chain.hip is compiled once per light-volume format so that the two halves
# of the chain kernel's instantiations build in parallel
UNITS = [
    ("tbrm_api.cpp", "tbrm_api", []), ("tbrm_api_render.cpp", "tbrm_api_render", []), ("tbrm_api_slabs.cpp", "tbrm_api_slabs", []), ("tbrm_light_plan.cpp", "tbrm_light_plan", []), ("tbrm_factor_cache.cpp", "tbrm_factor_cache", []), ("tbrm_light_enqueue.cpp", "tbrm_light_enqueue", []),
    ("tbrm_light_operators.cpp", "tbrm_light_operators", []), ("tbrm_block_lists.cpp", "tbrm_block_lists", []), ("tbrm_host_math.cpp", "tbrm_host_math", []),
    ("tbrm_kernels.hip", "tbrm_kernels", []), ("tbrm_volume_kernels.hip", "tbrm_volume_kernels", []), ("tbrm_light_kernels.hip", "tbrm_light_kernels", []),
    ("tbrm_light_chain.hip", "tbrm_light_chain_u8", ["-DTBRM_CHAIN_LFMT=0"]), ("tbrm_light_chain.hip", "tbrm_light_chain_f32", ["-DTBRM_CHAIN_LFMT=2"]),
    ("tbrm_light_sweep_dispatch.cpp", "tbrm_light_sweep_dispatch", []),
] + [
    # k_light_sweep: one unit per mode (PASS_ADD 0, PASS_CHANGE 1, PASS_ADD2 2, PASS_PLANES 5); tile height 32 — the 32 x 16 form
    # (measured slower) is the variant build: TBRM_BUILD_VARIANTS=1 compiles the library with 16-row tiles instead
    ("tbrm_light_sweep.hip", f"tbrm_light_sweep_m{m}_t{SWEEP_TILE_ROWS}", [f"-DTBRM_SWEEP_UNIT_MODE={m}", f"-DTBRM_SWEEP_UNIT_TH={SWEEP_TILE_ROWS}"])
    for m in (1, 0, 2, 5)
]
SOURCES = sorted({u[0] for u in UNITS})
HEADERS = ["tbrm_internal.h", "tbrm_resources.h", "tbrm_device_math.h", "tbrm_device_sampling.h", "tbrm_host_math.h", "tbrm_light_chain.h", "tbrm_light_sweep.h", "tbrm_light_passes.h",
           "../../include/tbrm.h"]

# -ffp-contract=off + explicit fma is the arithmetic contract with the oracle (DESIGN.md "Arithmetic spec").
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", f"-DTBRM_SWEEP_TILE_ROWS={SWEEP_TILE_ROWS}",
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
    """One hipcc -c per source, in parallel (the kernel files take most of a minute each), then one link.

    Safe against concurrent callers (N ranks of a torchrun launch that all find a stale library): one builder at a time
    holds a lock file, the others wait and then find the library fresh; objects go to a per-process directory and the
    library is linked under a temporary name and moved into place, so a reader never maps a half-written file."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(LIB_DIR, exist_ok=True)
    lock = open(os.path.join(LIB_DIR, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not needs_build():  # another process built it while this one waited
            return LIB_PATH
        return _build_locked(verbose, ThreadPoolExecutor)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose, ThreadPoolExecutor):
    obj_dir = os.path.join(LIB_DIR, "obj", str(os.getpid()))
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]

    def compile_one(unit):
        src, name, extra = unit
        obj = os.path.join(obj_dir, name + ".o")
        cmd = [hipcc_path()] + compile_flags + extra + ["-c", "-x", "hip", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), max(os.cpu_count() or 4, 4))) as pool:
        objs = list(pool.map(compile_one, UNITS))
    tmp_lib = LIB_PATH + f".{os.getpid()}.tmp"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", tmp_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp_lib, LIB_PATH)
    shutil.rmtree(obj_dir, ignore_errors=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
