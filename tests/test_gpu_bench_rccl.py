"""bench.py's N > 1 code path over RCCL on the one GPU of the box (TBRM_BENCH_FORCE_DIST=1: a process group of ONE rank, backend
nccl = RCCL): the asynchronous, double-buffered all_gather_into_tensor of the tiles on the library's own stream
(torch.cuda.ExternalStream), and the slab-partitioned light update with its point-to-point plane exchange and the all-gather
of the light volume. What an 8-GPU node would run, minus the other seven ranks — the only way to put these calls through RCCL
where the driver has one GPU per box."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.update({"TBRM_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_tile_gather_over_rccl_equals_the_single_gpu_frame(gpu):
    r = run_bench(["--config", "5"])
    assert r["n_gpus"] == 1 and r["gathered_frame_equals_single_gpu_render"] is True, r
    assert r["value"] > 0 and r["config"]["framebuffer"] == [2048, 2048]


def test_slab_partitioned_light_update_over_rccl_equals_the_unpartitioned_operator(gpu):
    r = run_bench(["--config", "3", "--slab-illumination"])
    assert r["slab_light_volume_equals_unpartitioned"] is True, r
    assert r["gathered_frame_equals_single_gpu_render"] is True, r


def test_light_update_on_one_rank_and_broadcast_over_rccl(gpu):
    """--light-update broadcast: the ChangeDirLight on rank 0 and ncclBroadcast of the light volume on the library's stream (one rank
    here: the broadcast is RCCL's own no-op path, the stream ordering and the line's fields are what is exercised)."""
    r = run_bench(["--config", "5", "--light-update", "broadcast"])
    assert r["gathered_frame_equals_single_gpu_render"] is True, r
    d = r["distributed"]
    assert d["backend"] == "nccl" and d["world_size_seen"] == 1 and d["light_update"] == "broadcast" and len(d["ms_per_step_per_rank"]) == 1, d


def test_gather_to_root_and_delivery_blocks_over_rccl(gpu):
    """--gather root: the timed step's tiles go to rank 0 by RCCL's gather (send / recv), and the line's frame_delivery / tile_parallel
    blocks time one GPU / tiles + all-gather / tiles + gather-to-root loops by wall clock with the gather inside (one rank here)."""
    r = run_bench(["--config", "3", "--gather", "root"])
    assert r["gather"] == "root" and r["gathered_frame_equals_single_gpu_render"] is True, r
    for key, fb in (("frame_delivery", [1024, 1024]), ("tile_parallel", [2048, 2048])):
        b = r[key]
        assert b["framebuffer"] == fb, b
        for form in ("tiles_all_gather", "tiles_gather_to_root"):
            assert b[form]["frame_equals_one_gpu_render"] is True and b[form]["ms_per_frame"] > 0, b
    assert r["frame_speedup_vs_one_gpu"] == r["frame_delivery"]["tiles_gather_to_root"]["speedup_vs_one_gpu"]
