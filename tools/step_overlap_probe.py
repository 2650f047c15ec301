"""What would the benchmark's step gain if the lit frame ran BESIDE the next light update instead of behind it (a second light-volume
buffer would make that legal: the first pass of an update could read one buffer and write the other while the frame still reads the
first)? Two handles on one GPU hold config 3's scene: one renders frames, the other turns a light 5 degrees per call (the benchmark's
fused, chained Change with its occlusion on the second stream). Wall time per round of each alone, of both enqueued together, and their
sum — the bound on the idea, before its cost. Diagnostics (profiles/EXPERIMENTS.md "Round 6")."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = 512
cfg = S.CONFIGS[3]
dev = torch.device("cuda", 0)
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), dev)
world = S.default_world()


def make(prio):
    abi.set_tunable("stream_priority", prio)
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    res.reserve(4)
    for i in range(4):
        res.add_dir_light(S.light(i), True, world)
    res.flush()
    return res


for prio in (0, 1):
    frames = make(0)
    lights = make(prio)
    cur = [S.light(i) for i in range(4)]
    angle = [0.0] * 4
    cam = S.default_camera(cfg["fb"], cfg["fb"])
    tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
    rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
    out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")
    N = 16
    step = [0]

    def run(do_frames, do_lights):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            if do_lights:
                k = step[0]
                step[0] += 1
                li = k % 4
                angle[li] += 5.0
                new = abi.DirLightParams(S.rotate_z(S.LIGHTS[li][0], angle[li]), cur[li].light_intensity)
                lights.change_dir_light(cur[li], new, world)
                cur[li] = new
            if do_frames:
                frames.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())
        frames.flush()
        lights.flush()
        return 1e3 * (time.perf_counter() - t0) / N

    for _ in range(2):
        run(True, True)
    f = min(run(True, False) for _ in range(3))
    l = min(run(False, True) for _ in range(3))
    b = min(run(True, True) for _ in range(3))
    print(f"light handle's stream priority {prio}: frame alone {f:.3f} ms, ChangeDirLight alone {l:.3f} ms, both enqueued together {b:.3f} ms per round "
          f"(sum {f + l:.3f}: the step as it is; max {max(f, l):.3f})", flush=True)
    frames.close()
    lights.close()
