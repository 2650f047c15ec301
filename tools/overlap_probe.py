"""What would it buy to run a frame beside the NEXT light update (a second light-volume buffer)? Probe with two handles (each has its
own streams): handle A takes ChangeDirLights, handle B renders frames of the same scene; both loops alone, then interleaved.
python tools/overlap_probe.py [n]. Diagnostics."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from tbraymarcherplugin_amd import abi, synthetic as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = S.CONFIGS[3]
vol = S.make_volume_torch((n, n, n), cfg["dtype"], S.seed_for_config(3), torch.device("cuda", 0))
world = S.default_world()
cam = S.default_camera(cfg["fb"], cfg["fb"])
tile = abi.Tile(0, 0, cfg["fb"], cfg["fb"], 1)
rp = abi.RaymarchParams(float(cfg["steps"]), -1, True)
out = torch.empty((cfg["fb"], cfg["fb"], 4), dtype=torch.float32, device="cuda")


def handle():
    res = abi.Resources((n, n, n), abi.FMT_G16)
    torch.cuda.synchronize()
    res.upload_volume_device(vol.data_ptr(), vol.numel() * 2)
    res.set_tf_lut(abi.color_curve_to_lut(S.TF_A_KEYS))
    res.set_windowing(abi.WindowingParams(*cfg["window"]))
    for i in range(4):
        res.add_dir_light(S.light(i), True, world)
    res.flush()
    return res


A, B = handle(), handle()
state = {"angle": 0.0, "cur": S.light(1)}


def change():
    state["angle"] += 1.0
    new = abi.DirLightParams(S.rotate_z(S.LIGHTS[1][0], state["angle"]), S.LIGHTS[1][1])
    A.change_dir_light(state["cur"], new, world)
    state["cur"] = new


def frame():
    B.raymarch_lit_device(cam, tile, rp, world, out.data_ptr())


def both():
    change()
    frame()


for _ in range(4):
    both()
A.flush(); B.flush()
k = 24
for rep in range(2):
    for what, fn in (("changes alone (handle A)", change), ("frames alone (handle B)", frame), ("interleaved, two handles", both)):
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        A.flush(); B.flush()
        t1 = time.perf_counter()
        print(f"{what:28s}: {1e3 * (t1 - t0) / k:.3f} ms per iteration", flush=True)
A.close(); B.close()
