#!/bin/bash
# Builds tools/tmp/exp/libtbrm_raystats.so (the lit march counting how full its waves are: -DTBRM_RAY_STATS) and, on a GPU box,
# prints the counts for the benchmark's frame: tools/ray_stats.sh build | run
set -e
cd "$(dirname "$0")/.."
OUT=tools/tmp/exp
if [ "${1:-build}" = build ]; then
  python tools/build_variant.py raystats --only tbrm_kernels -DTBRM_RAY_STATS
else
  TBRM_LIB_PATH=$PWD/$OUT/libtbrm_raystats.so python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from tbraymarcherplugin_amd import abi, synthetic as S
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
    st = (C.c_ulonglong * 6)()
    res.flush()
    lib.tbrm_debug_ray_stats(st, 1)
    res.raymarch_lit_device(cam, abi.Tile(0, 0, fb, fb, 1), abi.RaymarchParams(float(cfg["steps"]), -1, True), world, out.data_ptr())
    res.flush()
    lib.tbrm_debug_ray_stats(st, 0)
    trips, notdone, live, busy, contrib_trips, contrib = [int(v) for v in st]
    print(f"config {config}: wave trips {trips}, lanes not done {notdone / (64 * trips):.3f} of the lanes; trips in which any lane samples {busy / trips:.3f}, "
          f"lanes sampling in those {live / (64 * max(busy, 1)):.3f}; trips in which any lane has alpha != 0: {contrib_trips / max(busy, 1):.3f} of the sampling trips, "
          f"lanes with alpha != 0 in those {contrib / (64 * max(contrib_trips, 1)):.3f}, of all sampling lanes {contrib / max(live, 1):.3f}")
    res.close()
PY
fi
