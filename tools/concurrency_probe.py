"""How well do two independent light operators overlap on one GPU? Two handles (each has its own HIP stream) run a fused
ChangeDirLight each; wall time of both enqueued together against one alone. Diagnostics for DESIGN.md 4.2 (not product).

    N=512 python tools/concurrency_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402


def make(n, dev):
    cfg = S.CONFIGS[3]
    vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    return res


def timed(handles, lights, world, reps=8):
    """ms per round; a round = one Change on every handle, enqueued back to back, then all flushed"""
    best = 1e9
    for rep in range(reps):
        for h in handles:
            h.flush()
        t0 = time.perf_counter()
        for h, (old, new) in zip(handles, lights):
            h.change_dir_light(old, new, world)
        for h in handles:
            h.flush()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        lights[:] = [(b, a) for a, b in lights]
    return best


def main():
    n = int(os.environ.get("N", "512"))
    dev = torch.device("cuda", 0)
    world = S.default_world()
    a, b = make(n, dev), make(n, dev)
    pairs = []
    for idx in (1, 2):
        old = S.light(idx)
        new = abi.DirLightParams(S.rotate_z(S.LIGHTS[idx][0], 5.0), S.LIGHTS[idx][1])
        pairs.append((old, new))
    for h, (old, _) in zip((a, b), pairs):
        h.add_dir_light(old, True, world)
    for env in ({}, {"chunk_steps": 8}):
        for k in ("chunk_steps",):
            abi.set_tunable(k, env.get(k, 0))
        one_a = timed([a], [pairs[0]], world)
        one_b = timed([b], [pairs[1]], world)
        both = timed([a, b], [pairs[0], pairs[1]], world)
        same = timed([a, b], [pairs[0], pairs[0]], world)
        print(f"{env or 'default'}: alone L1 {one_a:.3f} ms, alone L2 {one_b:.3f} ms, both handles {both:.3f} ms "
              f"(sum {one_a + one_b:.3f}), both L1 {same:.3f} ms", flush=True)
    a.close()
    b.close()


if __name__ == "__main__":
    main()
