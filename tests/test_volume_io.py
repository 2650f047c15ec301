"""MetaImage reader + normalisation (include/tbrm_volume_io.hpp): the reference ships no sample files, so
the fixtures are generated here (.mhd + .raw, and zlib-compressed .zraw) and the expected arrays are a numpy restatement
of ConvertArrayToNormalizedArray / ConvertArrayToFloat (TextureUtilities.h:103-165): float arithmetic, truncation."""
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "volume_io_test.cpp")

MET = {np.uint8: "MET_UCHAR", np.int8: "MET_CHAR", np.uint16: "MET_USHORT", np.int16: "MET_SHORT", np.uint32: "MET_UINT",
       np.int32: "MET_INT", np.float32: "MET_FLOAT"}
FMT_INDEX = {np.uint8: 0, np.int8: 1, np.uint16: 2, np.int16: 3, np.uint32: 4, np.int32: 5, np.float32: 6}


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("vio") / "volume_io_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-lz"], check=True)
    return exe


def write_mhd(folder, name, arr, spacing=(0.5, 0.75, 2.0), compressed=False, spacing_key="ElementSpacing", extra=""):
    z, y, x = arr.shape
    data_name = name + (".zraw" if compressed else ".raw")
    payload = arr.tobytes()
    lines = ["ObjectType = Image", "NDims = 3", f"DimSize = {x} {y} {z}", f"{spacing_key} = {spacing[0]} {spacing[1]} {spacing[2]}",
             f"ElementType = {MET[arr.dtype.type]}", "ElementByteOrderMSB = False"]
    if compressed:
        payload = zlib.compress(payload, 6)
        lines += ["CompressedData = True", f"CompressedDataSize = {len(payload)}"]
    lines += [extra] if extra else []
    lines.append(f"ElementDataFile = {data_name}")
    with open(os.path.join(folder, data_name), "wb") as f:
        f.write(payload)
    path = os.path.join(folder, name + ".mhd")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def run(driver, path, normalize, to_float, out_bin, reference_seed=True):
    out = subprocess.run([driver, path, "1" if normalize else "0", "1" if to_float else "0", out_bin, "1" if reference_seed else "0"],
                         check=True, capture_output=True, text=True).stdout
    kv = {}
    for line in out.strip().splitlines():
        key, _, rest = line.partition("=")
        kv[key] = rest
    return kv


def normalized_reference(arr, reference_seed=True):
    """ConvertArrayToNormalizedArray<In, Out> in numpy float32. The running maximum starts at numeric_limits<T>::min() like the
    reference's (TextureUtilities.h:110: for float the smallest POSITIVE value — a float volume without a positive voxel
    reports a maximum of 1.18e-38); reference_seed = False: at the lowest value of the type (the loader's switch)."""
    flat = arr.reshape(-1)
    lo, hi = flat.min(), flat.max()
    if reference_seed and arr.dtype == np.float32:
        hi = max(hi, np.finfo(np.float32).tiny)
    out_t = np.uint8 if arr.dtype.itemsize == 1 else np.uint16
    out_max = np.float32(np.iinfo(out_t).max)
    span = np.float32(hi) - np.float32(lo)
    with np.errstate(invalid="ignore", divide="ignore"):
        normalized = (flat.astype(np.float32) - np.float32(lo)) / span
        scaled = np.float32(0) + normalized * out_max
    scaled = np.where(np.isnan(scaled), np.float32(0), scaled)
    return np.trunc(scaled).astype(out_t).reshape(arr.shape), float(np.float32(lo)), float(np.float32(hi))


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32])
@pytest.mark.parametrize("compressed", [False, True])
def test_mhd_load_normalized_and_float(driver, tmp_path, dtype, compressed):
    rng = np.random.default_rng(hash((dtype.__name__, compressed)) % 2**32)
    shape = (5, 7, 9)  # z, y, x
    if dtype is np.float32:
        arr = rng.normal(100.0, 400.0, size=shape).astype(np.float32)
    else:
        info = np.iinfo(dtype)
        arr = rng.integers(max(info.min, -3000), min(info.max, 60000), size=shape, dtype=np.int64).astype(dtype)
    path = write_mhd(str(tmp_path), "vol", arr, compressed=compressed)
    out_bin = str(tmp_path / "out.bin")

    kv = run(driver, path, True, False, out_bin)
    assert kv["ok"] == "1" and kv["parsed"] == "1"
    assert kv["dims"] == "9 7 5" and kv["spacing"] == "0.5 0.75 2" and kv["world"] == "4.5 5.25 10"
    assert f"original_format={FMT_INDEX[dtype]} " in kv["original_format"] or kv["original_format"].startswith(str(FMT_INDEX[dtype]))
    want, lo, hi = normalized_reference(arr)
    got = np.fromfile(out_bin, dtype=want.dtype).reshape(shape)
    assert np.array_equal(got, want)
    gmin, gmax = (float(v.split("=")[-1]) for v in ("min=" + kv["min"]).replace(" max=", "|max=").split("|"))
    assert gmin == pytest.approx(lo, rel=1e-7) and gmax == pytest.approx(hi, rel=1e-7)
    assert ("actual_format=0" if arr.dtype.itemsize == 1 else "actual_format=2") in "original_format=" + kv["original_format"]
    assert kv["tbrm_format"] == ("0" if arr.dtype.itemsize == 1 else "1")  # TBRM_FMT_G8 / TBRM_FMT_G16
    assert f"compressed={1 if compressed else 0} " in "original_format=" + kv["original_format"]
    assert got.max() == np.iinfo(want.dtype).max and got.min() == 0  # the file's range fills the output type
    # window values given in file units map to [0, 1]
    mid, span = (float(v.split("=")[-1]) for v in ("normalize_value_of_mid=" + kv["normalize_value_of_mid"]).replace(" normalize_range_of_span=", "|x=").split("|"))
    assert mid == pytest.approx(0.5, abs=1e-6) and span == pytest.approx(1.0, abs=1e-6)

    if dtype is not np.float32:  # the R32F path: values kept, converted to float
        kv = run(driver, path, False, True, out_bin)
        assert kv["ok"] == "1" and kv["tbrm_format"] == "2"
        got = np.fromfile(out_bin, dtype=np.float32).reshape(shape)
        assert np.array_equal(got, arr.astype(np.float32))


def test_mhd_header_rules_and_failures(driver, tmp_path):
    arr = np.arange(2 * 3 * 4, dtype=np.uint16).reshape(2, 3, 4)
    out_bin = str(tmp_path / "o.bin")
    # ElementSize is accepted in place of ElementSpacing (MHDLoader.cpp:58)
    kv = run(driver, write_mhd(str(tmp_path), "a", arr, spacing_key="ElementSize"), True, False, out_bin)
    assert kv["ok"] == "1"
    # words, not characters: "DimSize=4 3 2" is not found
    path = write_mhd(str(tmp_path), "b", arr)
    text = open(path).read().replace("DimSize = 4 3 2", "DimSize=4 3 2")
    open(path, "w").write(text)
    assert run(driver, path, True, False, out_bin)["parsed"] == "0"
    # unknown element type
    path = write_mhd(str(tmp_path), "c", arr)
    open(path, "w").write(open(path).read().replace("MET_USHORT", "MET_DOUBLE"))
    assert run(driver, path, True, False, out_bin)["parsed"] == "0"
    # data file shorter than the header says
    path = write_mhd(str(tmp_path), "d", arr)
    open(os.path.join(str(tmp_path), "d.raw"), "wb").write(arr.tobytes()[:-3])
    kv = run(driver, path, True, False, out_bin)
    assert kv["parsed"] == "1" and kv["ok"] == "0"
    # corrupt zlib stream
    path = write_mhd(str(tmp_path), "e", arr, compressed=True)
    blob = bytearray(open(os.path.join(str(tmp_path), "e.zraw"), "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    open(os.path.join(str(tmp_path), "e.zraw"), "wb").write(bytes(blob))
    assert run(driver, path, True, False, out_bin)["ok"] == "0"
    # a constant file: max == min -> everything normalises to 0 (0/0 in the reference)
    const = np.full((2, 2, 2), 7, dtype=np.int16)
    kv = run(driver, write_mhd(str(tmp_path), "f", const), True, False, out_bin)
    assert kv["ok"] == "1" and not np.fromfile(out_bin, dtype=np.uint16).any()
    # all-negative float file. By default exactly the reference: its maximum starts at FLT_MIN (TextureUtilities.h:110), so the
    # file's largest voxel (-1) does not reach the top code; with the switch off: the true maximum (-1) and the full range
    neg = -np.arange(1, 9, dtype=np.float32).reshape(2, 2, 2)
    kv = run(driver, write_mhd(str(tmp_path), "g", neg), True, False, out_bin)
    want, lo, hi = normalized_reference(neg)
    got = np.fromfile(out_bin, dtype=np.uint16).reshape(2, 2, 2)
    assert np.array_equal(got, want) and lo == -8.0 and 0.0 < hi < 1e-37 and got.max() < 65535 and got.min() == 0
    assert float(kv["min"].split()[0].split("=")[-1]) == -8.0 and 0.0 < float(kv["min"].split("max=")[1]) < 1e-37
    kv = run(driver, write_mhd(str(tmp_path), "g", neg), True, False, out_bin, reference_seed=False)
    want, lo, hi = normalized_reference(neg, reference_seed=False)
    got = np.fromfile(out_bin, dtype=np.uint16).reshape(2, 2, 2)
    assert np.array_equal(got, want) and hi == -1.0 and lo == -8.0 and got.max() == 65535 and got.min() == 0


def test_actor_level_loaders_compile(tmp_path):
    """ARaymarchVolume-level entry points (LoadMHDFileIntoVolumeNormalized / ...TransientR32F, RaymarchVolume.cpp:596-628)
    are header-only glue over UMHDLoader + SetVolumeAsset: they must compile against the façade."""
    src = tmp_path / "use.cpp"
    src.write_text('#include "tbrm_volume_io.hpp"\n'
                   "bool use(tbrm_plugin::ARaymarchVolume& v, const char* p) {\n"
                   "    tbrm_plugin::FVolumeInfo info;\n"
                   "    const bool a = tbrm_plugin::LoadMHDFileIntoVolumeNormalized(v, p, &info);\n"
                   "    if (a) v.SetWindowCenter(info.NormalizeValue(300.0f));\n"
                   "    return a || tbrm_plugin::LoadMHDFileIntoVolumeTransientR32F(v, p);\n"
                   "}\n")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], check=True)
