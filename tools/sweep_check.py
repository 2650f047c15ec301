"""The pipelined sweep kernel (k_light_sweep) against the one-slice-per-launch kernel on the same GPU, then its timings at
bench size next to the chunked chain. Diagnostics: `SIZES=small` skips the 512^3 part, `TIMING=0` the timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

LIGHTS = [((1, .35, -.5), 0.5), ((-1, .2, .4), 0.6), ((.3, 1, -.2), 0.5), ((.25, -1, .5), 0.7), ((.1, .45, 1), 0.5),
          ((-.35, .2, -1), 0.9), ((1, 0, 0), 0.5), ((0, 0, -1), 0.8), ((1, 1, 0), 0.6), ((.9, .1, .05), 0.4)]


def run_ops(dims, dtype, variant, seed=3):
    for k, v in variant.items():
        abi.set_tunable(k, v)
    vol = S.make_volume_numpy(dims, dtype, S.seed_for_config(seed))
    world = S.default_world()
    out = []
    with abi.Resources(dims, abi.DTYPE_FMT[np.dtype(dtype)]) as res:
        res.upload_volume(vol)
        res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
        res.set_windowing(abi.WindowingParams(0.5, 0.9, True, False))
        res.clear_light_volume(0.0)
        ls = [abi.DirLightParams(d, i) for d, i in LIGHTS]
        for l in ls:
            res.add_dir_light(l, True, world)
        res.flush()
        out.append(res.download_light_volume())
        for i, l in enumerate(ls[:6]):
            new = abi.DirLightParams(S.rotate_z(LIGHTS[i][0], 7.0), LIGHTS[i][1])
            res.change_dir_light(l, new, world)
        res.flush()
        out.append(res.download_light_volume())
        res.add_dir_light(ls[7], False, world)
        res.add_dir_light(ls[8], False, world)
        res.flush()
        out.append(res.download_light_volume())
    return out


def main():
    abi.load()
    ok = True
    sizes = [((40, 32, 48), np.uint8), ((64, 64, 64), np.uint16), ((96, 80, 72), np.uint16), ((128, 128, 128), np.uint16), ((24, 72, 136), np.float32),
             ((160, 160, 160), np.uint16)]
    base = {"light_cache_mb": 0, "occ_overlap": 2}
    for dims, dtype in sizes:
        ref = run_ops(dims, dtype, dict(base, force_slice_kernel=1, light_sweep=0))
        for rows in (2,):
            for pf in (6, 3):
                t0 = time.time()
                try:
                    got = run_ops(dims, dtype, dict(base, force_slice_kernel=0, light_sweep=1, sweep_prefetch=pf))
                except Exception as e:  # noqa: BLE001
                    print(f"{dims} rows={rows} pf={pf}: EXCEPTION {e}", flush=True)
                    ok = False
                    continue
                diffs = [int(np.count_nonzero(a != b)) for a, b in zip(ref, got)]
                print(f"{dims} {np.dtype(dtype).name} rows={rows} pf={pf}: differing voxels after adds / changes / removes = {diffs}  ({time.time() - t0:.1f} s)", flush=True)
                ok = ok and not any(diffs)
    chain = run_ops((64, 64, 64), np.uint16, dict(base, force_slice_kernel=0, light_sweep=0))
    print("chain vs slice at 64^3:", [int(np.count_nonzero(a != b)) for a, b in zip(run_ops((64, 64, 64), np.uint16, dict(base, force_slice_kernel=1, light_sweep=0)), chain)])
    print("SWEEP PARITY", "OK" if ok else "FAILED", flush=True)
    if os.environ.get("TIMING", "1") == "0":
        return 0 if ok else 1

    import torch

    n = int(os.environ.get("N", "512"))
    cfg = S.CONFIGS[3]
    dev = torch.device("cuda", 0)
    vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
    world = S.default_world()
    variants = [("chain, cache off", dict(light_cache_mb=0, light_sweep=0)),
                ("chain, cache on ", dict(light_cache_mb=-1, light_sweep=0))]
    for pf in (3, 6):
        for ov in (0, 2):
            variants.append((f"sweep pf={pf} overlap={ov}", dict(light_cache_mb=0, light_sweep=1, sweep_prefetch=pf, occ_overlap=ov)))
    lvs = {}
    for name, tun in variants:
        abi.set_tunable("force_slice_kernel", 0)
        abi.set_tunable("occ_overlap", 2)
        for k, v in tun.items():
            abi.set_tunable(k, v)
        res = abi.Resources((n, n, n), abi.FMT_G16)
        torch.cuda.synchronize()
        res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
        res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
        res.set_windowing(abi.WindowingParams(*cfg["window"]))
        adds, changes = [], []
        for rep in range(2):
            res.clear_light_volume(0.0)
            adds = []
            for i in range(4):
                res.add_dir_light(S.light(i), True, world)
                adds.append(res.last_gpu_time_ms(0))
        cur = [S.light(i) for i in range(4)]
        for k in range(1, 9):
            for i in range(4):
                new = abi.DirLightParams(S.rotate_z(S.LIGHTS[i][0], 5.0 * k), S.LIGHTS[i][1])
                res.change_dir_light(cur[i], new, world)
                changes.append(res.last_gpu_time_ms(0))
                cur[i] = new
        res.flush()
        lvs[name] = res.download_light_volume()
        res.close()
        ch = np.array(changes[4:])
        print(f"{name}: adds {['%.3f' % a for a in adds]} ms; change mean {ch.mean():.3f} min {ch.min():.3f} max {ch.max():.3f} ms", flush=True)
    first = lvs[variants[0][0]]
    for name, lv in lvs.items():
        d = int(np.count_nonzero(lv != first))
        if d:
            ok = False
        print(f"  light volume of '{name}' vs '{variants[0][0]}': {d} voxels differ")
    print("ALL", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
