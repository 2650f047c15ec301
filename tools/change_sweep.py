"""GPU time of the light operators at bench size under combinations of the library's tunables (tile height, chunk length,
occlusion prefetch). Diagnostics for DESIGN.md 4.2.

    N=512 python tools/change_sweep.py "tile_h=16" "tile_h=16,occ_prefetch=0" ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

NAMES = ["chunk_steps", "occ_slices", "sparse_occ", "occ_list", "light_cache_mb"]


def main():
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
    defaults = {k: abi.get_tunable(k) for k in NAMES}
    combos = sys.argv[1:] or [""]
    lights = [S.light(i) for i in range(4)]
    for l in lights:
        res.add_dir_light(l, True, world)
    ref = None
    for combo in combos:
        for k, v in defaults.items():
            abi.set_tunable(k, v)
        for kv in filter(None, combo.split(",")):
            k, v = kv.split("=")
            abi.set_tunable(k, int(v))
        out = []
        for li in (1, 0, 2):
            old = lights[li]
            new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], 5.0), S.LIGHTS[li][1])
            ms = 1e9
            for rep in range(6):
                res.change_dir_light(old, new, world)
                ms = min(ms, res.last_gpu_time_ms(0))
                old, new = new, old
            out.append(f"change L{li} {ms:.3f}")
        ms = 1e9
        for rep in range(3):
            res.add_dir_light(lights[0], True, world)
            ms = min(ms, res.last_gpu_time_ms(0))
            res.add_dir_light(lights[0], False, world)
        out.append(f"add L0 {ms:.3f}")
        res.clear_light_volume(0.0)
        for l in lights:
            res.add_dir_light(l, True, world)
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], 5.0), S.LIGHTS[1][1])
        res.change_dir_light(lights[1], new, world)
        lv = res.download_light_volume()
        res.change_dir_light(new, lights[1], world)
        if ref is None:
            ref = lv
        same = bool((lv == ref).all())
        print(f"{combo or 'defaults':44s} " + "  ".join(out) + f"  ms; light volume == first combo's: {same}", flush=True)
    res.close()


if __name__ == "__main__":
    main()
