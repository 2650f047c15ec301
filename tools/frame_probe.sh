#!/bin/bash
# The lit march with s_memtime stamps around the regions of a wave's trip (tools/diagnostics/frame_probe.patch, -DTBRM_RAY_PROBE):
#   tools/frame_probe.sh build   applies the patch, builds tools/tmp/exp/libtbrm_rayprobe.so, takes the patch out again
#   tools/frame_probe.sh run     (on a GPU box) prints where the cycles of configs 3 and 5's frames go
# PC sampling and thread trace are not available on this stack (profiles/r06_pcsamp_unavailable.txt): this is the fallback.
set -e
cd "$(dirname "$0")/.."
if [ "${1:-build}" = build ]; then
  git apply tools/diagnostics/frame_probe.patch
  python tools/build_variant.py rayprobe --only tbrm_kernels -DTBRM_RAY_PROBE || { git apply -R tools/diagnostics/frame_probe.patch; exit 1; }
  git apply -R tools/diagnostics/frame_probe.patch
else
  TBRM_LIB_PATH=$PWD/tools/tmp/exp/libtbrm_rayprobe.so python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from tbraymarcherplugin_amd import abi, synthetic as S
NAMES = ["positions (every addition of the ray), clip test", "texel split, offset tables (LDS), leap-distance byte landed, range arithmetic", "tap offsets, 16 taps issued and landed",
         "decode, 2 trilinear filters, window, TF, opacity correction", "exchange (LDS) + in-order accumulation", "empty trips taken in one go, range renewal", "loop control + what the trip before left in flight"]
for config in (3, 5):
    cfg = S.CONFIGS[config]
    n = cfg["n"]
    vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(config), torch.device("cuda", 0))
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.tf_keys(cfg["tf"])))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    world = S.default_world()
    for i in cfg["lights"]:
        res.add_dir_light(S.light(i), True, world)
    fb = cfg["fb"]
    cam = S.default_camera(fb, fb)
    out = torch.empty((fb, fb, 4), dtype=torch.float32, device="cuda")
    lib = abi.load()
    st = (C.c_ulonglong * 16)()
    tile, rp = abi.Tile(0, 0, fb, fb, 1), abi.RaymarchParams(float(cfg["steps"]), -1, True)
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.flush()
    lib.tbrm_debug_ray_probe(st, 1)
    res.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
    res.flush()
    ms = res.last_gpu_time_ms(1)
    lib.tbrm_debug_ray_probe(st, 0)
    v = [int(x) for x in st]
    total = sum(v[:7])
    waves, trips, sampling, accumulating = v[11], v[8], v[9], v[10]
    print(f"config {config}: frame {ms:.3f} ms with the probe ({waves} waves, {trips} wave trips: {sampling / max(trips, 1):.3f} sample, {accumulating / max(trips, 1):.3f} accumulate); "
          f"cycles per wave {total / max(waves, 1):.0f}, per sampling trip {total / max(sampling, 1):.0f} (s_memtime ticks, as lane 0 of each wave saw them)")
    for k, name in enumerate(NAMES):
        print(f"   {100.0 * v[k] / max(total, 1):5.1f} %  {v[k] / max(sampling, 1):7.1f} per sampling trip   {name}")
    res.close()
PY
fi
