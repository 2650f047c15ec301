"""A second libtbrm.so with extra -D flags on some translation units, for A/B timing on the GPU box (TBRM_LIB_PATH selects it).

    python tools/build_variant.py NAME [--only SUBSTRING] [--units SUBSTRING] -DFOO=1 ...

NAME -> tools/tmp/exp/libtbrm_NAME.so. The flags go to the units whose object name contains --only (default: "sweep"); the
other units' objects are compiled once into tools/tmp/obj_base/ and reused until a source or header changes. --units keeps
only the sweep units whose name contains the substring (the others are replaced by the product's: a faster build).
Diagnostics, never part of the product."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tbraymarcherplugin_amd import build as tb  # noqa: E402


def main():
    args = sys.argv[1:]
    name = args.pop(0)
    only = "sweep"
    while args and args[0] in ("--only",):
        args.pop(0)
        only = args.pop(0)
    extra = args
    base_dir = os.path.join(ROOT, "tools", "tmp", "obj_base")
    out_dir = os.path.join(ROOT, "tools", "tmp", "exp")
    os.makedirs(base_dir, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    compile_flags = [f for f in tb.FLAGS if f != "-shared"] + ["-Wno-unused-variable", "-Wno-unused-but-set-variable"]
    newest = max(os.path.getmtime(os.path.join(tb.CSRC, f)) for f in tb.SOURCES + tb.HEADERS)

    def compile_one(unit):
        src, oname, uflags = unit
        varied = only in oname
        flags = compile_flags + uflags + (extra if varied else [])
        tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
        obj = os.path.join(base_dir, f"{oname}_{tag}.o")
        if os.path.exists(obj) and os.path.getmtime(obj) > newest:
            return obj
        subprocess.run([tb.hipcc_path()] + flags + ["-c", "-x", "hip", os.path.join(tb.CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=max(os.cpu_count() or 4, 4)) as pool:
        objs = list(pool.map(compile_one, tb.UNITS))
    lib = os.path.join(out_dir, f"libtbrm_{name}.so")
    subprocess.run([tb.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
