"""Phase timing inside the chain kernel (s_memtime stamps, chain_stamps tunable): where a chunk's time goes.
Stamps: 0 entry, 1 staging pattern + flags done, 2 copies issued, 3 slots set up, 4 copies landed + sync, 5 border init + sync,
6 slice loop done, 7 light-volume tile stored."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(os.environ.get("N", "512"))
cfg = S.CONFIGS[3]
dev = torch.device("cuda", 0)
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
res = abi.Resources((n, n, n), abi.FMT_G16)
torch.cuda.synchronize()
res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
res.set_windowing(abi.WindowingParams(*cfg["window"]))
world = S.default_world()
abi.set_tunable("chain_stamps", 1)
abi.set_tunable("occ_prefetch", 0)
lib = abi.load()
lib.tbrm_debug_chain_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for combo in sys.argv[1:] or [""]:
    for k in ("tile_h", "chunk_steps"):
        abi.set_tunable(k, 0)
    for kv in filter(None, combo.split(",")):
        k, v = kv.split("=")
        abi.set_tunable(k, int(v))
    old = S.light(1)
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
    res.add_dir_light(old, True, world)
    for rep in range(3):
        res.change_dir_light(old, new, world)
        old, new = new, old
    ms = res.last_gpu_time_ms(0)
    nwg = 512 if abi.get_tunable("tile_h") == 16 else 256
    buf = np.zeros((nwg, 32), dtype=np.uint64)
    abi.check(lib.tbrm_debug_chain_stamps(res.handle, buf.ctypes.data, nwg))
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    rel = t - t0
    d = np.diff(t, axis=1)
    print(f"[{combo or 'defaults'}] change {ms:.3f} ms; last chain launch: {nwg} workgroups; entry spread {rel[:, 0].max()} ticks; "
          f"last exit {rel[:, 7].max()} ticks")
    if abi.get_tunable("tile_h") == 16:
        own, halo = t[:, 8:14], t[:, 16:22]
        print("   slice 3, owner wave (start, -, -, work done, lds drained, barrier passed): " + " ".join(f"{(own[:, i] - own[:, 0]).mean():.0f}" for i in (3, 4, 5)))
        print("   slice 3, halo wave (start, copies issued, -, work done, copies+lds waited, barrier passed): " + " ".join(f"{(halo[:, i] - halo[:, 0]).mean():.0f}" for i in (1, 3, 4, 5)))  # k_light_chain2: stamps 0-3 of an owner wave, 4-7 of a halo wave (entry, copies issued + slots set up, loop start, loop end)
        print("   owner wave: " + "  ".join(f"{i}->{i + 1}: {d[:, i].mean():.0f}/{d[:, i].max()}" for i in range(3))
              + "   halo wave: " + "  ".join(f"{i}->{i + 1}: {d[:, 4 + i].mean():.0f}/{d[:, 4 + i].max()}" for i in range(3)))
    else:
        print("   phase (ticks, mean / max over workgroups): " + "  ".join(f"{i}->{i + 1}: {d[:, i].mean():.0f}/{d[:, i].max()}" for i in range(7)))
res.close()
