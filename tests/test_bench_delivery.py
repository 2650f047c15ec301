"""bench.py's N > 1 record: the frame_delivery / tile_parallel blocks (wall clock of raymarch-only loops WITH the gather inside:
one GPU, tiles + all-gather, tiles + gather to rank 0) — the function that produces them run here on CPU tensors over gloo with
two ranks and a stub in place of the handle (a deterministic "render" of the rows a tile asks for), so that the tiling, both
gathers, the assembled frames and the block's keys are checked without a device. On GPUs bench.py runs the same function over RCCL."""
import ctypes as C
import json
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"framebuffer", "frames_timed", "nominal_samples_per_frame", "frame_bytes", "timing", "one_gpu", "tiles_all_gather", "tiles_gather_to_root"}


class StubHandle:
    """raymarch_lit_device(cam, tile, rp, world, ptr) fills tile.w x tile.h x 4 floats: pixel (x, y) of the framebuffer = f(x, y)"""

    def raymarch_lit_device(self, cam, tile, rp, world, ptr):
        out = np.ctypeslib.as_array((C.c_float * (tile.h * tile.w * 4)).from_address(ptr)).reshape(tile.h, tile.w, 4)
        j = np.arange(tile.h)
        step = max(int(tile.row_group_step), 1)
        rows = tile.y0 + (j // 8) * 8 * step + (j % 8)  # the C-ABI's tile row rule (tbrm.h)
        x = np.arange(tile.w)
        out[...] = (rows[:, None, None] * 1000.0 + x[None, :, None] + np.arange(4)[None, None, :] * 0.25).astype(np.float32)

    def flush(self):
        pass


def _worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench
    from tbraymarcherplugin_amd import abi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    block = bench.frame_delivery(torch, dist, abi, StubHandle(), None, 24, 32, None, None, rank, world_size, torch.device("cpu"), None, True, 3, 12345,
                                 sync=lambda: None)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(block, f)
    dist.destroy_process_group()


def test_frame_delivery_block_over_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        b = json.load(open(tmp_path / f"rank{r}.json"))
        assert set(b) == KEYS
        assert b["framebuffer"] == [24, 32] and b["frames_timed"] == 3 and b["frame_bytes"] == 24 * 32 * 16
        for form in ("one_gpu", "tiles_all_gather", "tiles_gather_to_root"):
            assert b[form]["ms_per_frame"] > 0 and b[form]["gsamples_per_s"] >= 0
        assert b["tiles_all_gather"]["frame_equals_one_gpu_render"] is True  # the all-gathered frame lands in every rank
        assert b["tiles_all_gather"]["bytes_received_per_gpu"] == 24 * 32 * 16 // 2
        assert b["tiles_gather_to_root"]["frame_equals_one_gpu_render"] is (True if r == 0 else None)  # the root alone holds it
        assert abs(b["tiles_all_gather"]["speedup_vs_one_gpu"] - b["one_gpu"]["ms_per_frame"] / b["tiles_all_gather"]["ms_per_frame"]) < 0.05 * b["tiles_all_gather"]["speedup_vs_one_gpu"] + 1e-3


def test_committed_dry_run_record_carries_the_three_blocks():
    """the N > 1 line as the GPU box produced it (2 ranks on one GPU, gloo: profiles/r06_bench_dry_run_2_ranks_on_one_gpu.json)"""
    path = os.path.join(ROOT, "profiles", "r06_bench_dry_run_2_ranks_on_one_gpu.json")
    if not os.path.exists(path):
        import pytest

        pytest.skip("no round-6 dry-run record committed yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["gather"] in ("all", "root")
    for key in ("frame_delivery", "tile_parallel"):
        assert set(d[key]) >= KEYS, key
    assert d["tile_parallel"]["framebuffer"] == [2048, 2048] and d["frame_delivery"]["framebuffer"] == [1024, 1024]
    assert d["frame_speedup_vs_one_gpu"] == d["frame_delivery"]["tiles_all_gather" if d["gather"] == "all" else "tiles_gather_to_root"]["speedup_vs_one_gpu"]
    assert "raymarch_kernel_speedup_vs_one_gpu" in d["scaling_detail"]  # the kernel-event ratio is still there, under its own name
