#!/usr/bin/env python3
"""bench.py — headline benchmark of the raymarch + illumination hot path on MI355X.

Metric (BASELINE.json): volume Msamples/s (rays x steps) at 512^3, 1024^2 view; % HBM roofline.
Workload at N=1: BASELINE config 3 — 512^3 UNORM16 volume, UNORM8 light volume, 1024^2 RGBA f32
framebuffer, 512 steps, lights L0-L3, TF-A, window C=0.5 W=0.9 (low cutoff on, high off), jitter off.

One "step" = one frame in which a light moved: ChangeDirLightInSingleVolume on one of the four lights (the
selective light-volume update; a 5 degree rotation about Z, fused path) followed by the lit raymarch of this
rank's share of the framebuffer, and (N>1) the RCCL all-gather of the tiles. Inputs are resident in HBM before
the timed region; the initial ResetAllLights (clear + 4 adds) is untimed setup and reported separately.

N>1 (one process per GPU, torch.distributed/RCCL; `python bench.py --gpus N` from a bare shell re-launches itself under
torch.distributed.run): THE SAME WORKLOAD as N=1 — config 3 — at every N, so that value(N) / value(1) is a speed-up: STRONG
scaling of one fixed frame, rank r renders every N-th group of 8 rows (load-balanced interleave), volumes are replicated,
the selective light update is computed redundantly on every GPU (no data-path collective: the update of ONE light is one
serial slice sweep per axis) and the only exchange is the gather of the tiles (`--gather all`: all_gather, the
default; `--gather root`: to rank 0 only). The line reports the full step, the same workload timed on one GPU in the same run
(`speedup_vs_one_gpu`), the Amdahl ceiling the redundant update puts on the step, and — second timed loops, raymarch only,
WALL CLOCK WITH THE GATHER INSIDE — `frame_delivery` (this workload's frame: one GPU / tiles + all-gather / tiles + gather to
root; `frame_speedup_vs_one_gpu` is taken from it) and `tile_parallel` (the same three for north_star's tile-scaling workload,
config 5: 512^3, ONE 2048^2 frame, 8 lights, TF-B, skipping on — at this N and on one GPU in the same run). `--config 5` makes
config 5 the whole step's workload; `--weak` keeps round 1's mode (the framebuffer grows to ~N x fb^2 pixels).

value = nominal samples of all ranks per step / step time, in Msamples/s (nominal sample = one loop iteration of
PerformWindowedLitRaymarch that geometry prescribes, independent of early termination and skipping).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
# the CPU oracle's OpenMP threads stay on their cores (cpu_baseline: the driver's run and a builder's run of the unpinned
# oracle differed 4x); set before anything loads libgomp
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")


def framebuffer_for(n_gpus, base):
    """~n_gpus*base^2 pixels, near-square, height a multiple of 8*n_gpus (interleaved 8-row groups)."""
    if n_gpus == 1:
        return base, base
    w = int(round(base * math.sqrt(n_gpus) / 16.0)) * 16
    unit = 8 * n_gpus
    h = int(round(n_gpus * base * base / w / unit)) * unit
    return w, h


def relaunch_under_torchrun(n_gpus):
    """`python bench.py --gpus N` from a bare shell: one process per GPU through torch.distributed.run on this node
    (rendezvous on 127.0.0.1, a free port), same arguments. Returns the launcher's exit code."""
    import socket
    import subprocess

    from tbraymarcherplugin_amd import build as tb

    tb.build()  # once, here: the N ranks must not all find a stale library and rebuild it side by side
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(cmd, env=env).returncode


def kernel_source_hash():
    """sha1 over the kernel sources: what the committed counter summaries under profiles/ were collected from (tools/pmc_traffic.py
    and tools/issue_roofline.py record it; a summary collected from other sources is not quoted)."""
    import hashlib

    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "tbraymarcherplugin_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def committed_counters(kind, per_step):
    """The newest profiles/rNN_<kind>.json — HBM traffic / instruction-issue counters of this very command, collected in separate
    rocprofv3 --pmc passes (tools/measure_round.sh) — if it still describes the kernels this run launched: same kernel sources
    (kernel_source_hash) and the same launches per step (per_step: this run's tbrm_path_counters over the timed loop). Returns
    (data or None, where it came from or why it is not quoted)."""
    import glob
    import re

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}.json")), key=lambda q: int(re.search(r"r(\d+)_", os.path.basename(q)).group(1)))
    if not paths:
        return None, f"no profiles/rNN_{kind}.json"
    path = paths[-1]
    rel = os.path.relpath(path, ROOT)
    try:
        with open(path) as f:
            data = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, f"{rel} unreadable: {e}"
    sig = data.get("_signature")
    if not sig:
        return None, f"{rel} carries no launch signature (collected before round 4): not quoted"
    if sig.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{rel} was collected from other kernel sources (hash {sig.get('kernel_source_hash')}, now {kernel_source_hash()}): stale, not quoted"
    for key, want in (sig.get("launches_per_step") or {}).items():
        have = per_step.get(key)
        if have is None or abs(have - want) > 0.15 * max(want, 1.0):
            return None, f"{rel} was collected with {want:.2f} {key} launches per step, this run made {have}: not quoted"
    return data, rel


def gpu_ms(torch, stream, fn):
    """GPU time of whatever fn() enqueues on the library's stream, by HIP events recorded on that stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    fn()
    e1.record(stream)
    e1.synchronize()
    return float(e0.elapsed_time(e1))


def frame_delivery(torch, dist, abi, res, cam, fb_w, fb_h, rp, world, rank, n_gpus, device, lib_stream, one_gpu_dry_run, frames, samples_per_frame, sync=None):
    """What tile-parallel rendering delivers, by WALL CLOCK (barrier + synchronize on both sides, max over ranks), for the frame of
    the handle's scene: (a) one GPU renders the whole frame, no exchange; (b) N GPUs render interleaved 8-row tiles and all-gather
    them (the frame lands in every GPU); (c) the same with a gather to rank 0 only (RCCL send / recv: the frame lands where it is
    presented). Tiles and gather buffers are double-buffered exactly like the timed step's. Speed-ups are (a) / (b), (a) / (c):
    they INCLUDE the gather — the ratio of raymarch kernel times does not and reads ~N by construction."""
    sync = sync or torch.cuda.synchronize  # (tests/test_bench_delivery.py runs this function on CPU tensors over gloo)
    rows = fb_h // n_gpus
    tile = abi.Tile(0, 8 * rank, fb_w, rows, n_gpus)
    full_tile = abi.Tile(0, 0, fb_w, fb_h, 1)
    red_device = torch.device("cpu") if one_gpu_dry_run else device
    outs = [torch.empty((rows, fb_w, 4), dtype=torch.float32, device=device) for _ in range(2)]
    alls = [torch.empty((n_gpus, rows, fb_w, 4), dtype=torch.float32, device=device) for _ in range(2)]
    roots = [torch.empty((n_gpus, rows, fb_w, 4), dtype=torch.float32, device=device) if rank == 0 else None for _ in range(2)]
    full = torch.empty((fb_h, fb_w, 4), dtype=torch.float32, device=device)

    def gather(kind, b):
        if kind == "none":
            return None
        if one_gpu_dry_run:  # gloo: staged through host memory, synchronous
            res.flush()
            host = outs[b].cpu()
            if kind == "all":
                parts = [torch.empty(host.shape, dtype=host.dtype) for _ in range(n_gpus)]
                dist.all_gather(parts, host)
                alls[b].copy_(torch.stack(parts))
            else:
                parts = [torch.empty(host.shape, dtype=host.dtype) for _ in range(n_gpus)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    roots[b].copy_(torch.stack(parts))
            return None
        with torch.cuda.stream(lib_stream):  # ordered behind the march on the library's stream; nothing waits on the host
            if kind == "all":
                return dist.all_gather_into_tensor(alls[b], outs[b], async_op=True)
            return dist.gather(outs[b], list(roots[b].unbind(0)) if rank == 0 else None, dst=0, async_op=True)

    def loop(kind, count):
        pend = [None, None]
        for k in range(count):
            b = k & 1
            if pend[b] is not None:
                with torch.cuda.stream(lib_stream):
                    pend[b].wait()
                pend[b] = None
            if kind == "none":
                res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())
            else:
                res.raymarch_lit_device(cam, tile, rp, world, outs[b].data_ptr())
                pend[b] = gather(kind, b)
        for h in pend:
            if h is not None:
                h.wait()

    def timed(kind):
        loop(kind, 2)
        res.flush()
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        loop(kind, frames)
        res.flush()
        dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / frames * 1e3

    one_ms = timed("none")
    all_ms = timed("all")
    root_ms = timed("root")
    # the frame that reached rank 0 through each gather must be this rank's own render of the whole framebuffer
    from tbraymarcherplugin_amd import sharding

    res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())
    res.flush()
    last = (frames - 1) & 1
    ok_all = bool(torch.equal(sharding.assemble(alls[last], fb_h, n_gpus), full))
    ok_root = bool(torch.equal(sharding.assemble(roots[last], fb_h, n_gpus), full)) if rank == 0 else None
    frame_bytes = fb_w * fb_h * 16

    def rate(ms):
        return round(samples_per_frame / (ms * 1e-3) / 1e9, 2)

    return {"framebuffer": [fb_w, fb_h], "frames_timed": frames, "nominal_samples_per_frame": int(samples_per_frame),
            "frame_bytes": frame_bytes, "timing": "wall clock around the loop, barrier + synchronize on both sides, max over ranks; gathers asynchronous and double-buffered",
            "one_gpu": {"ms_per_frame": round(one_ms, 4), "gsamples_per_s": rate(one_ms)},
            "tiles_all_gather": {"ms_per_frame": round(all_ms, 4), "gsamples_per_s": rate(all_ms), "speedup_vs_one_gpu": round(one_ms / all_ms, 3),
                                 "bytes_received_per_gpu": frame_bytes * (n_gpus - 1) // n_gpus, "frame_equals_one_gpu_render": ok_all},
            "tiles_gather_to_root": {"ms_per_frame": round(root_ms, 4), "gsamples_per_s": rate(root_ms), "speedup_vs_one_gpu": round(one_ms / root_ms, 3),
                                     "bytes_received_by_root": frame_bytes * (n_gpus - 1) // n_gpus, "frame_equals_one_gpu_render": ok_root}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None,
                    help="SURVEY.md §8d config number; default: 3 (the metric's config) at EVERY N — value(N) / value(1) is then a speed-up; "
                         "5 = north_star's tile-scaling workload (one 2048^2 frame)")
    ap.add_argument("--weak", action="store_true", help="N>1: grow the framebuffer with N (~N x fb^2 pixels) instead of splitting one frame")
    ap.add_argument("--fixed-frame", action="store_true", help="N>1: split ONE frame of the config's size over the GPUs (the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: leave out the extra operator timings (reset, fallback change, one-GPU reference), so that every "
                         "step of the process is one ChangeDirLight + one raymarch")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--no-skipping", action="store_true")
    ap.add_argument("--raymarch-only", action="store_true", help="diagnostic: leave the light update out of the step")
    ap.add_argument("--light-parallel-reset", action="store_true",
                    help="N>1: the untimed ResetAllLights deals the lights over the ranks and combines the light volumes with "
                         "reduce-scatter + all-gather (SURVEY.md §8e) instead of adding every light on every GPU")
    ap.add_argument("--light-update", choices=["redundant", "broadcast"], default="redundant",
                    help="N>1: how the step's ChangeDirLight reaches every GPU. redundant (default): every GPU computes it (no exchange). "
                         "broadcast: rank 0 computes it and broadcasts the light volume (SURVEY.md 8e's other option; on the critical path "
                         "Change + broadcast + frame / N cannot beat Change + frame / N: reported for comparison)")
    ap.add_argument("--gather", choices=["all", "root"], default="all",
                    help="N>1: how the timed step's tiles are assembled. all (default): all_gather_into_tensor, the frame lands in every GPU. "
                         "root: a gather to rank 0 only (RCCL send / recv), the frame lands where it is presented. Whatever this is, the line's "
                         "frame_delivery / tile_parallel blocks time BOTH forms in raymarch-only loops.")
    ap.add_argument("--slab-illumination", action="store_true",
                    help="N>1: the timed ChangeDirLight is partitioned over the ranks in light-volume z slabs (plane halo exchange "
                         "per chunk / z pipeline, slabs.py), followed by an all-gather of the light volume, instead of being "
                         "computed redundantly on every GPU (BASELINE config 4's decomposition)")
    ap.add_argument("--slab-resident", action="store_true",
                    help="N>1, BASELINE config 4's decomposition taken literally: every GPU holds only its z slab of the data and "
                         "light volumes (tbrm_resources_create_slab); a step = slab-partitioned ChangeDirLight + light-volume halo "
                         "exchange + the frame marched slab by slab (strong scaling: one frame of the config's size)")
    args = ap.parse_args()
    if args.config is None:
        args.config = 3
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    if os.environ.get("TBRM_BENCH_LAUNCH_CHECK") == "1":  # tests/test_bench_launcher.py: what did the launcher start?
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "world_size": int(os.environ.get("WORLD_SIZE", "1")), "master_addr": os.environ.get("MASTER_ADDR"),
                          "config": args.config, "gpus": args.gpus}), flush=True)
        return

    import torch

    from tbraymarcherplugin_amd import abi, synthetic as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world_size != n_gpus and not (n_gpus == 1 and world_size == 1):
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {n_gpus}")
    dist = None
    # Dry run of the N>1 code path on a box with ONE GPU (every rank on device 0, gloo instead of RCCL): only for checking
    # the sharding / gather / reporting logic where no multi-GPU node is available; never used by the driver.
    one_gpu_dry_run = os.environ.get("TBRM_BENCH_ONE_GPU_DRY_RUN") == "1"
    if one_gpu_dry_run:
        local_rank = 0
    # TBRM_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL collectives included) with a single rank: the only way to
    # exercise it over RCCL on a 1-GPU box.
    if n_gpus > 1 or os.environ.get("TBRM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_gpu_dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    abi.load()
    if args.slab_resident:
        if dist is None:
            raise SystemExit("--slab-resident needs --gpus N > 1 (or TBRM_BENCH_FORCE_DIST=1)")
        return slab_resident_bench(args, torch, dist, S, abi, rank, local_rank, n_gpus, device, one_gpu_dry_run)

    cfg = S.CONFIGS[args.config]
    n = cfg["n"]
    dims = (n, n, n)
    seed = S.seed_for_config(args.config)
    fixed_frame = n_gpus > 1 and not args.weak
    fb_w, fb_h = (cfg["fb"], cfg["fb"]) if fixed_frame else framebuffer_for(n_gpus, cfg["fb"])
    if fb_h % (8 * n_gpus):
        raise SystemExit(f"framebuffer height {fb_h} does not split into interleaved 8-row groups over {n_gpus} GPUs")
    rows_per_rank = fb_h // n_gpus
    steps = float(cfg["steps"])

    # ---- setup (untimed): inputs resident in HBM ---------------------------------------------------------
    vol_dev = S.make_volume_torch(dims, cfg["dtype"], seed, device)
    res = abi.Resources(dims, abi.DTYPE_FMT[np.dtype(cfg["dtype"])], cfg["light_32bit"], False, local_rank)
    torch.cuda.synchronize()  # the library reads the tensor on its own stream: torch's generator kernels must be done
    res.upload_volume_device(vol_dev.data_ptr(), vol_dev.numel() * vol_dev.element_size())
    lut = abi.color_curve_to_lut(S.tf_keys(cfg["tf"]))
    res.set_tf_lut(lut)
    win = abi.WindowingParams(*cfg["window"])
    res.set_windowing(win)
    world = S.default_world()
    cam = S.default_camera(fb_w, fb_h)
    tile = abi.Tile(0, 8 * rank, fb_w, rows_per_rank, n_gpus)
    rp = abi.RaymarchParams(steps, -1, not args.no_skipping)

    lights = [S.light(i) for i in cfg["lights"]]
    light_dirs = [S.LIGHTS[i][0] for i in cfg["lights"]]
    lib_stream = torch.cuda.ExternalStream(res.stream(), device=device)  # the library's HIP stream, as torch sees it

    def reset_all_lights():  # ARaymarchVolume::ResetAllLights (RaymarchVolume.cpp:418-451) with the lights' current parameters
        res.clear_light_volume(0.0)
        for l in lights:
            res.add_dir_light(l, True, world)

    # InitializeRaymarchResources (RaymarchVolume.cpp:821-920): every buffer the light operators of this scene will need, now —
    # the scratch stores, hand-off records, block lists and the factor cache's arena (tbrm_resources_reserve); after it no operator
    # allocates or waits for a stream (light_paths_per_step.operator_alloc_calls / operator_host_syncs stay 0)
    t_res = time.perf_counter()
    res.reserve(len(lights))
    res.flush()
    reserve_ms = (time.perf_counter() - t_res) * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.light_parallel_reset and dist is not None and not cfg["light_32bit"]:
        from tbraymarcherplugin_amd import sharding

        def combine_u8(t):
            def reduce_scatter_sum(acc):
                if one_gpu_dry_run:  # gloo: no reduce_scatter
                    c = acc.cpu()
                    dist.all_reduce(c)
                    k = c.numel() // n_gpus
                    return c[rank * k:(rank + 1) * k].to(device)
                mine = torch.empty(acc.numel() // n_gpus, dtype=acc.dtype, device=device)
                dist.reduce_scatter_tensor(mine, acc)
                return mine

            def all_gather(chunk):
                if one_gpu_dry_run:
                    parts = [torch.empty(chunk.shape, dtype=chunk.dtype) for _ in range(n_gpus)]
                    dist.all_gather(parts, chunk.cpu())
                    return torch.cat(parts).to(device)
                full = torch.empty(chunk.numel() * n_gpus, dtype=chunk.dtype, device=device)
                dist.all_gather_into_tensor(full, chunk)
                return full

            return sharding.combine_light_codes(t, n_gpus, reduce_scatter_sum, all_gather)

        sharding.reset_all_lights_light_parallel(res, lights, world, rank, n_gpus, combine_u8)
    else:
        reset_all_lights()  # first call: includes the brick metadata kernels (the buffers were reserved above)
        res.flush()
    reset_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    if not (args.light_parallel_reset and dist is not None and not cfg["light_32bit"]):
        reset_all_lights()
        res.flush()
    reset_warm_wall_ms = (time.perf_counter() - t0) * 1e3

    # Two output tiles / gather buffers: the all-gather of frame k runs (asynchronously, on RCCL's stream) while the light
    # update of frame k+1 is already executing on the library's stream; a buffer is reused only after its gather is done.
    outs = [torch.empty((rows_per_rank, fb_w, 4), dtype=torch.float32, device=device) for _ in range(2)]
    root_only = args.gather == "root" and dist is not None
    gathers = ([torch.empty((n_gpus, rows_per_rank, fb_w, 4), dtype=torch.float32, device=device) if (rank == 0 or not root_only) else None for _ in range(2)]
               if dist is not None else [None, None])
    pending = [None, None]
    out, gathered = outs[0], gathers[0]
    my_samples = res.count_nominal_samples(cam, tile, rp, world)
    total_samples = my_samples
    red_device = torch.device("cpu") if one_gpu_dry_run else device
    if dist is not None:
        t = torch.tensor([my_samples], dtype=torch.int64, device=red_device)
        dist.all_reduce(t)
        total_samples = int(t.item())

    # slab-partitioned light update: this rank owns light-volume slices [z_r, z_r+1)
    slab_member = slab_fabric = None
    if args.slab_illumination and dist is not None:
        from tbraymarcherplugin_amd import slabs

        depth = res.light_dims[2]
        bounds = slabs.slab_bounds(depth, n_gpus)
        z_bounds = [b[0] for b in bounds] + [depth]
        slab_member = slabs.DeviceSlab(res, rank, *bounds[rank])
        if one_gpu_dry_run:  # gloo: stage through host memory
            def p2p(ops):
                res.flush()
                work = []
                for kind, t, peer in ops:
                    if kind == "send":
                        host = t.cpu()  # kept alive until the send has completed
                        work.append((dist.isend(host, peer), None, host))
                    else:
                        buf = torch.empty(t.shape, dtype=t.dtype)
                        work.append((dist.irecv(buf, peer), t, buf))
                for w, t, buf in work:
                    w.wait()
                    if t is not None:
                        t.copy_(buf)
                torch.cuda.synchronize()

            slab_fabric = slabs.make_fabric(z_bounds, list(range(n_gpus)), rank, p2p, sync_local=False)
        else:
            slab_fabric = slabs.dist_fabric(z_bounds, rank, n_gpus, member=slab_member)

    def gather_light(full, part):
        if one_gpu_dry_run:
            parts = [torch.empty(part.shape, dtype=part.dtype) for _ in range(n_gpus)]
            dist.all_gather(parts, part.cpu())
            full.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(full, part)

    angle = [0.0] * len(lights)
    ms_illum, ms_ray = [], []
    last = [0]  # buffer index of the most recent frame

    def one_step(k, record):
        if not args.raymarch_only:
            li = k % len(lights)
            angle[li] += 5.0
            new = abi.DirLightParams(S.rotate_z(light_dirs[li], angle[li]), lights[li].light_intensity)
            if slab_member is not None:
                with slab_member.stream_context():  # RCCL point-to-point operations ordered with the library's stream
                    slabs.change_dir_light([slab_member], slab_fabric, lights[li], new, world)
                    slabs.gather_light_volume([slab_member], slab_fabric, gather_light)
            elif args.light_update == "broadcast" and dist is not None and not one_gpu_dry_run:
                from tbraymarcherplugin_amd import sharding

                def bcast(t, src):  # ordered behind the operator on the library's stream; the frame behind it waits for it there
                    with torch.cuda.stream(lib_stream):
                        dist.broadcast(t, src)

                sharding.change_dir_light_on_owner(lambda: res.change_dir_light(lights[li], new, world), lambda: sharding.device_light_tensor(res),
                                                   rank, 0, bcast)
            else:
                res.change_dir_light(lights[li], new, world)
            lights[li] = new
        b = k & 1
        # Device-side ordering only, no host synchronisation inside a step: with torch's current stream set to the library's
        # stream, RCCL waits for the tile through an event on that stream, and wait() makes that stream (not the host) wait
        # for the gather that last read outs[b] / wrote gathers[b] before the raymarch overwrites the tile.
        if pending[b] is not None:
            with torch.cuda.stream(lib_stream):
                pending[b].wait()
            pending[b] = None
        res.raymarch_lit_device(cam, tile, rp, world, outs[b].data_ptr())
        if dist is not None:
            if one_gpu_dry_run:
                res.flush()
                if root_only:
                    parts = [torch.empty(outs[b].shape, dtype=outs[b].dtype) for _ in range(n_gpus)] if rank == 0 else None
                    dist.gather(outs[b].cpu(), parts, dst=0)
                    if rank == 0:
                        gathers[b].copy_(torch.stack(parts))
                else:
                    parts = [torch.empty(outs[b].shape, dtype=outs[b].dtype) for _ in range(n_gpus)]
                    dist.all_gather(parts, outs[b].cpu())
                    gathers[b].copy_(torch.stack(parts))
            else:
                if os.environ.get("TBRM_BENCH_HOST_SYNC") == "1":  # A/B: drain the library's stream on the host before the gather
                    res.flush()
                with torch.cuda.stream(lib_stream):
                    if root_only:
                        pending[b] = dist.gather(outs[b], list(gathers[b].unbind(0)) if rank == 0 else None, dst=0, async_op=True)
                    else:
                        pending[b] = dist.all_gather_into_tensor(gathers[b], outs[b], async_op=True)
        last[0] = b
        if record:
            if not args.raymarch_only and slab_member is None:
                ms_illum.append(res.last_gpu_time_ms(0))
            ms_ray.append(res.last_gpu_time_ms(1))

    for k in range(args.warmup):
        one_step(k, False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    paths0 = res.path_counters()
    # the timed loop runs as a host that never asks for GPU times would: without the library's own event pairs around every
    # operator (tbrm_last_gpu_time_ms; four markers per step between dependent kernels, ~1 % of the step). The gpu_ms passes below
    # turn them on again. TBRM_BENCH_GPU_TIMING=1 keeps them on here too (A/B).
    timing_in_loop = os.environ.get("TBRM_BENCH_GPU_TIMING") == "1"
    if not timing_in_loop:
        abi.set_tunable("gpu_timing", 0)
    try:
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k, False)
        for h in pending:
            if h is not None:
                h.wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:  # (process-global: whatever happens in the loop, later handles record their timing events again)
        abi.set_tunable("gpu_timing", 1)
    paths1 = res.path_counters()
    # which kernels the timed steps launched (include/tbrm.h tbrm_path_counters), per step
    light_paths = {k: round((paths1[k] - paths0[k]) / max(args.steps, 1), 3) for k in paths1}
    pending = [None, None]
    out, gathered = outs[last[0]], gathers[last[0]]
    per_rank_ms = None
    if dist is not None:
        mine = torch.tensor([elapsed / max(args.steps, 1) * 1e3], dtype=torch.float64, device=red_device)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(x.item()), 4) for x in every]
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N>1: the assembled frame must equal this rank's own render of the whole framebuffer (untimed check) ----
    gather_ok = None
    if dist is not None:
        from tbraymarcherplugin_amd import sharding

        full = torch.empty((fb_h, fb_w, 4), dtype=torch.float32, device=device)
        res.raymarch_lit_device(cam, abi.Tile(0, 0, fb_w, fb_h, 1), rp, world, full.data_ptr())
        res.flush()
        frame = sharding.assemble(gathered, fb_h, n_gpus) if gathered is not None else None  # (--gather root: rank 0 alone holds it)
        gather_ok = bool(torch.equal(frame, full)) if frame is not None else None
        if gather_ok is False and os.environ.get("TBRM_BENCH_DEBUG"):
            rows = torch.from_numpy(sharding.rank_rows(fb_h, rank, n_gpus)).to(device)
            print(f"[rank {rank}] frame-full max|d| = {float((frame - full).abs().max())}, own tile vs own full: "
                  f"{bool(torch.equal(out, full[rows]))}, gathered[rank] vs out: {bool(torch.equal(gathered[rank], out))}", flush=True)

    # ---- per-kernel GPU time with HIP events on the library's stream (separate, untimed pass) -------------
    n_event = max(3, min(args.steps, 20))
    for k in range(n_event):
        one_step(args.warmup + args.steps + k, True)
    for h in pending:
        if h is not None:
            h.wait()
    torch.cuda.synchronize()
    # the plain MEAN of the event-timed calls (round 5 left out calls above 3x the median: the factor cache grew inside operators
    # then — hipMemGetInfo + hipMalloc between an operator's enqueues; round 6's operators allocate nothing, tbrm_resources_reserve).
    # gpu_ms_spread lists every call and says how many WOULD have been left out by that rule (0 expected). (Calls that take the
    # remove + add path — the light's major axis changes, every 6th - 7th step — are 30 - 50 % longer: they are the benchmark's operators.)
    def robust_mean(ms):
        if not ms:
            return 0.0, 0
        med = float(np.median(ms))
        return float(np.mean(ms)), len([t for t in ms if t > 3.0 * med])

    ray_ms, ray_dropped = robust_mean(ms_ray)
    illum_ms, illum_dropped = robust_mean(ms_illum)
    spread = {"raymarch": [round(float(t), 4) for t in ms_ray], "change_dir_light": [round(float(t), 4) for t in ms_illum],
              "left_out_above_3x_median": {"raymarch": ray_dropped, "change_dir_light": illum_dropped},
              "plain_mean": {"raymarch": round(float(np.mean(ms_ray)), 4) if ms_ray else None, "change_dir_light": round(float(np.mean(ms_illum)), 4) if ms_illum else None},
              "median": {"raymarch": round(float(np.median(ms_ray)), 4) if ms_ray else None, "change_dir_light": round(float(np.median(ms_illum)), 4) if ms_illum else None},
              "note": "every event-timed call of the gpu_ms pass; gpu_ms quotes their plain mean (left_out_above_3x_median: how many calls round 5's "
                      "trimmed mean would have dropped — none are dropped now)"}

    # ---- slab mode: the partitioned + gathered light volume must equal the unpartitioned operator's (untimed replay) ----
    slab_ok = None
    if slab_member is not None and not args.raymarch_only:
        from tbraymarcherplugin_amd import sharding

        res.flush()
        got = sharding.device_light_tensor(res).clone()
        torch.cuda.synchronize()  # the copy runs on torch's stream: it must be complete before the library's stream rewrites the volume
        replay = [S.light(i) for i in cfg["lights"]]
        ang = [0.0] * len(replay)
        res.clear_light_volume(0.0)
        for l in replay:
            res.add_dir_light(l, True, world)
        for k in range(args.warmup + args.steps + n_event):
            li = k % len(replay)
            ang[li] += 5.0
            new = abi.DirLightParams(S.rotate_z(light_dirs[li], ang[li]), replay[li].light_intensity)
            res.change_dir_light(replay[li], new, world)
            replay[li] = new
        res.flush()
        want = sharding.device_light_tensor(res)
        slab_ok = bool(torch.equal(got, want))
        if not slab_ok and os.environ.get("TBRM_BENCH_DEBUG"):
            layers = (res.light_dims[2] + 7) // 8
            bad = (got != want).view(layers, -1).sum(dim=1).tolist()
            import hashlib
            hg, hw = (hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12] for x in (got, want))
            print(f"[rank {rank}] light volume sha1: slabs {hg}, unpartitioned replay {hw}; mismatching bytes per 8-slice layer: "
                  f"{ {i: b for i, b in enumerate(bad) if b} }", flush=True)

    # ---- the rest of SURVEY.md 8d's operator sequence, GPU time by HIP events on the library's stream (untimed pass) ----
    # Every rank runs the same operators, so the replicated light volumes stay identical.
    ops_ms = {}
    one_gpu = None
    if not args.raymarch_only and slab_member is None and not args.timed_only:
        res.flush()
        full_tile = abi.Tile(0, 0, fb_w, fb_h, 1)
        full = torch.empty((fb_h, fb_w, 4), dtype=torch.float32, device=device)
        # SURVEY.md 8d's operators, cold and warm. "Warm" = the factor cache holds the lights' occlusion (nothing samples the
        # data volume again); "cold" = a window that differs by one ulp has just made everything cached stale — what a moved
        # clip plane or volume transform (RaymarchVolume.cpp:351-356 -> ResetAllLights), a new window or transfer function
        # (TransferFuncMenu.cpp:63-78) and APerformanceTest1's window sweep (PerformanceTest1.cpp:75-84) cost; "uncached" =
        # the same with the cache off (light_cache_mb = 0: nothing is kept either).
        win_k = [0]

        def stale_window():  # a window centre nobody has used yet: win_k ulps above the config's
            win_k[0] += 1
            c = np.float32(cfg["window"][0])
            for _ in range(win_k[0]):
                c = np.nextafter(c, np.float32(2.0))
            res.set_windowing(abi.WindowingParams(float(c), *cfg["window"][1:]))

        def timed(fn, before=None, reps=2):
            best = None
            for _ in range(reps):
                if before is not None:
                    before()
                    res.flush()
                t = gpu_ms(torch, lib_stream, fn)
                best = t if best is None else min(best, t)
            return best

        # the same reset as ONE tbrm_add_dir_lights call (the facade's bBatchLightsOnReset): every light's passes are
        # planned together — passes of different lights that leave the same cube face and pull the same way share a sweep (PASS_ADD2),
        # the others run in the lights' order in chained launches of up to four passes
        batch_info = {}

        def reset_all_lights_batched():
            res.clear_light_volume(0.0)
            sched = res.add_dir_lights(lights, True, world)
            batch_info["sweeps"] = len(sched)
            batch_info["paired_sweeps"] = sum(1 for e in sched if e[2] >= 0)

        cache_default = abi.get_tunable("light_cache_mb")
        ops_ms["reset_all_lights_warm"] = timed(reset_all_lights)
        ops_ms["reset_all_lights_batched_warm"] = timed(reset_all_lights_batched)
        ops_ms["reset_all_lights_batched_cold"] = timed(reset_all_lights_batched, before=stale_window)
        ops_ms["reset_all_lights_plus_frame_warm"] = timed(lambda: (reset_all_lights(), res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())), reps=1)
        ops_ms["reset_all_lights_cold"] = timed(reset_all_lights, before=stale_window)
        # one step of APerformanceTest1's window sweep: new window centre, every light again, the frame
        ops_ms["window_sweep_step"] = timed(lambda: (stale_window(), reset_all_lights(), res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())))
        abi.set_tunable("light_cache_mb", 0)
        ops_ms["reset_all_lights_uncached"] = timed(reset_all_lights)
        # every Change from an idle handle, timed on torch's view of the library's stream: the operator's occlusion runs on the
        # handle's second stream, and between back-to-back calls the NEXT call's occlusion starts beside this call's sweeps and
        # falls outside the library's own event pair (round 4 reported 0.79 ms that way — less than the cached figure)
        unc = []
        for li in range(len(lights)):
            new = abi.DirLightParams(S.rotate_z(light_dirs[li], angle[li] + 5.0), lights[li].light_intensity)
            best = None
            for _ in range(2):  # (the better of two, like timed(): the first call of a kernel instantiation loads its code object)
                res.flush()
                t = gpu_ms(torch, lib_stream, lambda: res.change_dir_light(lights[li], new, world))
                best = t if best is None else min(best, t)
                res.change_dir_light(new, lights[li], world)
            unc.append(best)
        ops_ms["change_dir_light_uncached"] = float(np.mean(unc))
        abi.set_tunable("light_cache_mb", cache_default)
        res.set_windowing(win)
        reset_all_lights()
        # ChangeDirLight whose old and new major axes differ: remove + add (LightingShaders.cpp:192-198). A quarter turn about z
        # moves the first light's major axis from x to y (cold: the new direction's occlusion is computed); the second call
        # turns it back (warm: both directions' factors are at hand).
        turned = abi.DirLightParams(S.rotate_z(light_dirs[0], angle[0] + 90.0), lights[0].light_intensity)
        fb_ms = []
        for old, new in ((lights[0], turned), (turned, lights[0])):
            res.change_dir_light(old, new, world)
            fb_ms.append(res.last_gpu_time_ms(0))
        ops_ms["change_dir_light_fallback_cold"] = fb_ms[0]
        ops_ms["change_dir_light_fallback_warm"] = fb_ms[1]
        if n_gpus > 1:
            # the same workload on ONE GPU, in the same run (every rank does it, rank 0 reports): the whole frame + the update
            def whole_step(k):
                li = k % len(lights)
                angle[li] += 5.0
                new = abi.DirLightParams(S.rotate_z(light_dirs[li], angle[li]), lights[li].light_intensity)
                res.change_dir_light(lights[li], new, world)
                lights[li] = new
                res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())

            for k in range(2):
                whole_step(k)
            res.flush()
            k1 = max(3, min(args.steps, 10))
            t1 = time.perf_counter()
            for k in range(k1):
                whole_step(2 + k)
            res.flush()
            one_ms = (time.perf_counter() - t1) / k1 * 1e3
            res.raymarch_lit_device(cam, full_tile, rp, world, full.data_ptr())
            one_ray_ms = res.last_gpu_time_ms(1)
            one_gpu = {"ms_per_step": round(one_ms, 4), "raymarch_ms": round(one_ray_ms, 4)}

    # ---- N > 1: what the tiles DELIVER, wall clock with the gather inside (second timed loops, raymarch only) -------------
    # frame_delivery: this run's workload (config 3 by default). tile_parallel: north_star's tile-scaling workload, config 5
    # (512^3, ONE 2048^2 frame, 8 lights, TF-B) at this N and on one GPU in the same run — the number "tile-parallel scaling" is about.
    delivery = tile_parallel = None
    if dist is not None and slab_member is None and not args.timed_only and not args.weak:
        k_frames = max(5, min(args.steps, 20))
        delivery = frame_delivery(torch, dist, abi, res, cam, fb_w, fb_h, rp, world, rank, n_gpus, device, lib_stream, one_gpu_dry_run, k_frames, total_samples)
        delivery["workload"] = f"config {args.config}"
        if args.config == 5:
            tile_parallel = delivery
        else:
            cfg5 = S.CONFIGS[5]
            n5, fb5 = cfg5["n"], cfg5["fb"]
            if fb5 % (8 * n_gpus) == 0:
                vol5 = S.make_volume_torch((n5, n5, n5), cfg5["dtype"], S.seed_for_config(5), device)
                res5 = abi.Resources((n5, n5, n5), abi.DTYPE_FMT[np.dtype(cfg5["dtype"])], cfg5["light_32bit"], False, local_rank)
                torch.cuda.synchronize()
                res5.upload_volume_device(vol5.data_ptr(), vol5.numel() * vol5.element_size())
                res5.set_tf_lut(abi.color_curve_to_lut(S.tf_keys(cfg5["tf"])))
                res5.set_windowing(abi.WindowingParams(*cfg5["window"]))
                res5.clear_light_volume(0.0)
                for i in cfg5["lights"]:
                    res5.add_dir_light(S.light(i), True, world)
                res5.flush()
                cam5 = S.default_camera(fb5, fb5)
                rp5 = abi.RaymarchParams(float(cfg5["steps"]), -1, not args.no_skipping)
                samples5 = res5.count_nominal_samples(cam5, abi.Tile(0, 0, fb5, fb5, 1), rp5, world)
                stream5 = torch.cuda.ExternalStream(res5.stream(), device=device)
                tile_parallel = frame_delivery(torch, dist, abi, res5, cam5, fb5, fb5, rp5, world, rank, n_gpus, device, stream5, one_gpu_dry_run, k_frames, samples5)
                tile_parallel["workload"] = "config 5: 512^3 uint16 volume, ONE 2048x2048 RGBA f32 frame, 512 steps, 8 dir lights, TF-B, empty-space skipping on; raymarch only"
                res5.close()
                del vol5

    # ---- roofline of the dominant kernel (algorithmic bytes) -------------------------------
    b_data = np.dtype(cfg["dtype"]).itemsize
    b_light = 4 if cfg["light_32bit"] else 1
    V = n ** 3
    ray_bytes = V * b_data + V * b_light + fb_w * rows_per_rank * 16 + 2048
    pass_bytes = V * b_data + 2 * V * b_light
    illum_bytes = 2 * pass_bytes  # fused Change = 2 axis passes
    if ray_ms >= illum_ms:
        dom = dict(kernel="k_raymarch_lit", achieved=ray_bytes / (ray_ms * 1e-3) / 1e9, launch_ms=ray_ms, alg_bytes=ray_bytes)
    else:
        dom = dict(kernel="k_light_occlusion+k_light_sweep (one ChangeDirLight = 2 axis passes)",
                   achieved=illum_bytes / (illum_ms * 1e-3) / 1e9, launch_ms=illum_ms, alg_bytes=illum_bytes)
    # HBM traffic from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    # summarised by tools/pmc_traffic.py into profiles/): per launch of the raymarch kernel, or summed over the launches
    # one ChangeDirLight makes (tools/pmc_traffic.py "_per_operator_call")
    traffic = None
    traffic_source = None
    per_step = {"k_light_sweep": light_paths["launches_sweep"], "k_light_occlusion": light_paths["occlusion_single"] + light_paths["occlusion_dual"],
                "k_raymarch_lit": light_paths["raymarch"]}
    if args.config == 3 and n_gpus == 1 and not args.raymarch_only:
        pmc, traffic_source = committed_counters("pmc_traffic", per_step)
        if pmc is not None:
            per_call = pmc["_per_operator_call"]
            traffic = int(per_call["raymarch_hbm_bytes"] if dom["kernel"].startswith("k_raymarch") else per_call["change_dir_light_hbm_bytes"])
            traffic_source += (": separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/measure_round.sh), NOT measured in "
                               "this run; checked against this run's kernel sources and launches per step")
    # why the fraction is what it is: the instruction-issue view of the hot kernels (tools/issue_roofline.py over two more
    # --pmc passes of this command; committed numbers, like the traffic, under the same check)
    issue = None
    if args.config == 3 and n_gpus == 1 and not args.raymarch_only:
        raw, where = committed_counters("issue", per_step)
        if raw is not None:
            issue = {k: {kk: vv for kk, vv in v.items() if kk != "per_launch"} for k, v in raw.items() if not k.startswith("_")}
            issue["source"] = (where + ": rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY / "
                               "SQ_ACTIVE_INST_LDS / GRBM_GUI_ACTIVE passes of this command (tools/measure_round.sh), NOT measured in this run; "
                               "valu_issue_frac = SQ_INSTS_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE)")
        else:
            issue = {"source": where}
    roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(dom["achieved"], 2), "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": round(dom["achieved"] * 1e9 / HBM_PEAK, 5), "traffic": traffic, "traffic_source": traffic_source,
                "alg_bytes_per_launch": int(dom["alg_bytes"]), "launch_ms": round(dom["launch_ms"], 4),
                "launch_ms_is": f"plain mean of {len(ms_ray)} event-timed calls (gpu_ms_spread lists every call)"}

    # ---- CPU baseline: the oracle on this host's cores, rank 0, N=1 only, bounded sample -------------------
    cpu = parity = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        moved = abi.DirLightParams(S.rotate_z(light_dirs[0], angle[0] + 5.0), lights[0].light_intensity)
        cpu, parity = cpu_baseline(args, cfg, res, vol_dev, lut, win, world, cam, rp, fb_w, fb_h, lights[0], moved)

    if rank == 0:
        value = total_samples * args.steps / elapsed / 1e6
        step_ms = elapsed / args.steps * 1e3
        scaling_note = None
        if n_gpus > 1 and one_gpu is not None:
            # what tile-parallel rendering can gain when every GPU repeats the light update (Amdahl): one GPU's step over
            # (update + 1/N of the frame); and what was measured
            ceiling = one_gpu["ms_per_step"] / (one_gpu["ms_per_step"] - one_gpu["raymarch_ms"] * (1.0 - 1.0 / n_gpus))
            scaling_note = {"one_gpu_same_workload": one_gpu, "speedup_vs_one_gpu": round(one_gpu["ms_per_step"] / step_ms, 3),
                            # ratio of raymarch KERNEL event times: compute only, the gather is not in it (reads ~N by construction)
                            "raymarch_kernel_speedup_vs_one_gpu": round(one_gpu["raymarch_ms"] / ray_ms, 3),
                            "amdahl_ceiling_with_redundant_light_update": round(ceiling, 3)}
        line = {
            "metric": "volume Msamples/s (rays x steps) at 512^3, 1024^2 view; % HBM roofline",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "strong" if fixed_frame else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {n}^3 {np.dtype(cfg['dtype']).name} volume, "
                                   f"{'f32' if cfg['light_32bit'] else 'u8'} light volume, {fb_w}x{fb_h} RGBA f32 framebuffer, "
                                   f"{int(steps)} steps, {len(lights)} dir lights, TF-{cfg['tf']}, 1 selective light update "
                                   f"(ChangeDirLight) + 1 lit raymarch per step",
                       "volume": [n, n, n], "framebuffer": [fb_w, fb_h], "steps": int(steps), "lights": len(lights),
                       "parallelism": (f"image tiles x{n_gpus} (interleaved 8-row groups) of "
                                       + ("ONE frame (strong scaling)" if fixed_frame else "a framebuffer that grows with N (weak scaling)")
                                       + ", volumes replicated, light update repeated on every GPU") if n_gpus > 1 else "single GPU",
                       "empty_space_skipping": not args.no_skipping, "raymarch_only": bool(args.raymarch_only),
                       "light_parallel_reset": bool(args.light_parallel_reset and dist is not None),
                       "slab_illumination": slab_member is not None},
            "nominal_samples_per_step": total_samples,
            "gathered_frame_equals_single_gpu_render": gather_ok,
            "slab_light_volume_equals_unpartitioned": slab_ok,
            "gpu_ms": dict({"raymarch": round(ray_ms, 4), "change_dir_light": round(illum_ms, 4)},
                           **{k: round(v, 4) for k, v in ops_ms.items()},
                           # host wall clock, call to flush: the handle's first ResetAllLights (skipping metadata kernels inside; buffers
                           # reserved before it), the second one, and tbrm_resources_reserve itself
                           first_reset_all_lights_host_wall=round(reset_ms, 2), second_reset_all_lights_host_wall=round(reset_warm_wall_ms, 2),
                           resources_reserve_host_wall=round(reserve_ms, 2)),
            "gpu_ms_spread": spread,
            "distributed": None if dist is None else {
                "backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "ms_per_step_per_rank": per_rank_ms,
                "light_update": args.light_update if slab_member is None else "slabs",
                # what the tiles alone gain (the part of a step that shards), next to the whole step's figure in scaling_detail
                "raymarch_ms_this_rank": round(ray_ms, 4)},
            "timed_loop_records_gpu_timing_events": bool(timing_in_loop),
            "light_paths_per_step": light_paths,  # tbrm_path_counters over the timed loop: sweep / chain / slice passes and launches, occlusion launches
            "raymarch_only_msamples_per_s": round(total_samples / (ray_ms * 1e-3) / 1e6, 2),
            "light_cache": res.light_cache_stats(),  # factor cache (include/tbrm.h tbrm_light_cache_stats)
            # gpu_ms.reset_all_lights_batched_*: what tbrm_add_dir_lights made of this scene's lights
            "reset_batching": None if not ops_ms else dict(batch_info, note=(
                "no two passes of different lights share a sweep: every same-face couple of this scene's lights pulls opposite ways along a plane axis "
                "(no tile order serves both); the lights' passes run in order, chained" if batch_info.get("paired_sweeps") == 0 else "PASS_ADD2 pairs")),
            # N > 1: this line's step against the SAME workload on one GPU in the same run (whole step; the frame alone)
            "speedup_vs_one_gpu": None if scaling_note is None else scaling_note["speedup_vs_one_gpu"],
            # the frame alone, WALL CLOCK WITH THE GATHER INSIDE (frame_delivery: the form of the timed step's --gather)
            "frame_speedup_vs_one_gpu": None if delivery is None else delivery["tiles_gather_to_root" if args.gather == "root" else "tiles_all_gather"]["speedup_vs_one_gpu"],
            "gather": None if dist is None else args.gather,
            "frame_delivery": delivery,
            "tile_parallel": tile_parallel,
            "scaling_detail": scaling_note,
            "roofline": roofline,
            "roofline_issue": issue,
            "cpu_baseline": cpu,
            "full_size_parity": parity,
        }
        print(json.dumps(line), flush=True)
    res.close()
    if dist is not None:
        dist.destroy_process_group()


def slab_resident_bench(args, torch, dist, S, abi, rank, local_rank, n_gpus, device, one_gpu_dry_run):
    """bench.py --slab-resident: no GPU holds the whole volume. See the flag's help and DESIGN.md 7."""
    from tbraymarcherplugin_amd import slabs

    cfg = S.CONFIGS[args.config]
    n = cfg["n"]
    dims = (n, n, n)
    fb = cfg["fb"]
    steps = float(cfg["steps"])
    fmt = abi.DTYPE_FMT[np.dtype(cfg["dtype"])]
    bounds = slabs.slab_bounds(n, n_gpus)
    z_bounds = [b[0] for b in bounds] + [n]
    res = abi.Resources(dims, fmt, cfg["light_32bit"], False, local_rank, owned=abi.Slab(*bounds[rank]))
    (dlo, dhi, dwrap), (llo, lhi, lwrap) = res.resident_slices()
    # synthetic volume: generated whole on the device (the generator has no slice form), only this handle's layers are kept
    vol_dev = S.make_volume_torch(dims, cfg["dtype"], S.seed_for_config(args.config), device)
    torch.cuda.synchronize()
    res.upload_volume_slices(dlo, vol_dev[dlo:dhi].cpu().numpy())
    if dwrap >= 0:
        res.upload_volume_slices(dwrap, vol_dev[dwrap:dwrap + 8].cpu().numpy())
    keep_whole = rank == 0  # rank 0 keeps the whole volume for the untimed check at the end
    if not keep_whole:
        del vol_dev
    lut = abi.color_curve_to_lut(S.tf_keys(cfg["tf"]))
    win = abi.WindowingParams(*cfg["window"])
    res.set_tf_lut(lut)
    res.set_windowing(win)
    world = S.default_world()
    cam = S.default_camera(fb, fb)
    tile = abi.Tile(0, 0, fb, fb, 1)
    rp = abi.RaymarchParams(steps, -1, False)
    member = slabs.DeviceSlab(res, rank, *bounds[rank])
    if one_gpu_dry_run:  # gloo: stage through host memory
        def p2p(ops):
            res.flush()
            work = []
            for kind, t, peer in ops:
                if kind == "send":
                    host = t.cpu()
                    work.append((dist.isend(host, peer), None, host))
                else:
                    buf = torch.empty(t.shape, dtype=t.dtype)
                    work.append((dist.irecv(buf, peer), t, buf))
            for w, t, buf in work:
                w.wait()
                if t is not None:
                    t.copy_(buf)
            torch.cuda.synchronize()

        fabric = slabs.make_fabric(z_bounds, list(range(n_gpus)), rank, p2p, sync_local=False)
    else:
        fabric = slabs.dist_fabric(z_bounds, rank, n_gpus, member=member)
    lights = [S.light(i) for i in cfg["lights"]]
    light_dirs = [S.LIGHTS[i][0] for i in cfg["lights"]]
    angle = [0.0] * len(lights)
    frame = [None]

    def new_state():
        return torch.zeros((fb, fb, 4), dtype=torch.float32, device=device)

    with member.stream_context():
        slabs.reset_all_lights([member], fabric, lights, world, lambda m: m.res.clear_light_volume(0.0))
    total_samples = res.count_nominal_samples(cam, tile, rp, world)

    def one_step(k):
        li = k % len(lights)
        angle[li] += 5.0
        new = abi.DirLightParams(S.rotate_z(light_dirs[li], angle[li]), lights[li].light_intensity)
        with member.stream_context():  # torch's zero fill and the point-to-point operations are ordered with the library's stream
            slabs.change_dir_light([member], fabric, lights[li], new, world)
            slabs.exchange_light_halos([member], fabric)
            frame[0] = slabs.render_lit([member], fabric, cam, tile, rp, world, new_state)
        lights[li] = new

    for k in range(args.warmup):
        one_step(k)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    res.flush()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    red_device = torch.device("cpu") if one_gpu_dry_run else device
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # untimed check on rank 0: one whole handle replays the lights and renders the last frame
    ok = None
    if rank == 0:
        whole = abi.Resources(dims, fmt, cfg["light_32bit"], False, local_rank)
        whole.upload_volume_device(vol_dev.data_ptr(), vol_dev.numel() * vol_dev.element_size())
        whole.set_tf_lut(lut)
        whole.set_windowing(win)
        replay = [S.light(i) for i in cfg["lights"]]
        ang = [0.0] * len(replay)
        whole.clear_light_volume(0.0)
        for l in replay:
            whole.add_dir_light(l, True, world)
        for k in range(args.warmup + args.steps):
            li = k % len(replay)
            ang[li] += 5.0
            new = abi.DirLightParams(S.rotate_z(light_dirs[li], ang[li]), replay[li].light_intensity)
            whole.change_dir_light(replay[li], new, world)
            replay[li] = new
        want = torch.empty((fb, fb, 4), dtype=torch.float32, device=device)
        whole.raymarch_lit_device(cam, tile, rp, world, want.data_ptr())
        whole.flush()
        own = res.download_light_slices(bounds[0][0], bounds[0][1] - bounds[0][0])
        ok = bool(torch.equal(frame[0], want)) and bool(np.array_equal(own, whole.download_light_volume()[bounds[0][0]:bounds[0][1]]))
        whole.close()
        value = total_samples * args.steps / elapsed / 1e6
        print(json.dumps({
            "metric": "volume Msamples/s (rays x steps) at 512^3, 1024^2 view; % HBM roofline",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {n}^3 {np.dtype(cfg['dtype']).name} volume in {n_gpus} z slabs (no GPU holds the "
                                   f"whole volume), {fb}x{fb} frame marched slab by slab, {int(steps)} steps, {len(lights)} dir lights, 1 "
                                   "slab-partitioned ChangeDirLight + light-volume halo exchange + 1 frame per step",
                       "volume": list(dims), "framebuffer": [fb, fb], "steps": int(steps), "lights": len(lights),
                       "parallelism": f"light-volume z slabs x{n_gpus}, slab-resident volumes",
                       "resident_data_slices": [dlo, dhi], "resident_light_slices": [llo, lhi]},
            "nominal_samples_per_step": total_samples,
            "slab_frame_and_light_volume_equal_one_whole_handle": ok,
            "roofline": None, "cpu_baseline": None}))
    res.close()


def cpu_baseline(args, cfg, res, vol_dev, lut, win, world, cam, rp, fb_w, fb_h, light_old, light_new):
    """Times the oracle (oracle/, test infrastructure: the checker, never the thing measured as `value`) on a bounded
    sample of the same workload — one step: the lit raymarch of every g-th group of 8 rows of the same frame (g chosen so
    that the sample takes about --cpu-seconds; g = 1, the whole frame, at config 3 on the GPU box), reading the light
    volume the GPU just produced, and one ChangeDirLight of the whole light volume. The reported rate is the step rate the
    sample implies: nominal samples of the frame / (frame time extrapolated from the rows + update time).

    The same oracle results are the FULL-SIZE PARITY check of the run: the oracle's rows against the GPU's frame of the
    same light volume (RGBA, tolerance 1e-4) and the oracle's light volume after the ChangeDirLight against the GPU's after
    the same operator (UNORM8: bit for bit). Returns (cpu_baseline, full_size_parity)."""
    from oracle import oracle
    from tbraymarcherplugin_amd import abi

    vol = vol_dev.cpu().numpy()
    orc = oracle.OracleScene(vol, cfg["light_32bit"])
    orc.set_tf_lut(lut)
    orc.set_windowing(win)
    res.flush()
    orc.light[...] = res.download_light_volume()
    gpu_frame = res.raymarch_lit(cam, abi.Tile(0, 0, fb_w, fb_h, 1), rp, world)  # the GPU's frame of this very light volume
    cores = oracle.load().orc_num_threads()
    groups = fb_h // 8
    # probe: 2 row groups from the middle of the frame
    probe = abi.Tile(0, 8 * (groups // 2), fb_w, 16, 1)
    t0 = time.perf_counter()
    _, n_probe = orc.raymarch_lit(cam, probe, rp, world)
    dt = max(time.perf_counter() - t0, 1e-4)
    rate = n_probe / dt
    _, n_full = orc.raymarch_lit(cam, abi.Tile(0, 0, fb_w, fb_h, 1), rp, world, count_only=True)
    g = 1
    while g < groups and n_full / g / rate > args.cpu_seconds:
        g *= 2
    sample = abi.Tile(0, 0, fb_w, (groups // g) * 8, g)
    t0 = time.perf_counter()
    cpu_rows, n_s = orc.raymarch_lit(cam, sample, rp, world)
    dt = time.perf_counter() - t0
    ray_rate = n_s / dt
    frame_s = n_full / ray_rate
    rows = (np.arange(sample.h) // 8) * 8 * g + np.arange(sample.h) % 8  # framebuffer row of every row of the sample tile
    rgba_max_abs = float(np.abs(cpu_rows - gpu_frame[rows]).max())
    parity = {"frame": f"{fb_w}x{fb_h}, every {g}-th 8-row group ({sample.h} rows) of the GPU frame against the oracle",
              "rgba_max_abs": rgba_max_abs, "rgba_tolerance": 1e-4, "rgba_ok": bool(rgba_max_abs <= 1e-4)}
    build = "gcc " + oracle.build_flags()
    if args.raymarch_only:
        return ({"value": round(ray_rate / 1e6, 3), "unit": "Msamples/s", "cores": int(cores), "kind": "port", "build": build,
                 "sample": f"oracle lit raymarch of every {g}-th 8-row group of the same {fb_w}x{fb_h} frame "
                           f"({n_s} nominal samples, {dt:.1f} s, OpenMP x{cores}); light volume taken from the GPU"}, parity)
    t0 = time.perf_counter()
    orc.change_dir_light(light_old, light_new, world)
    change_s = time.perf_counter() - t0
    res.change_dir_light(light_old, light_new, world)  # the same operator on the GPU, from the same light volume
    gpu_light = res.download_light_volume()
    if cfg["light_32bit"]:
        d = float(np.abs(gpu_light - orc.light).max())
        parity.update({"light_volume": "R32F light volume after one ChangeDirLight", "light_max_abs": d, "light_ok": bool(d <= 1e-5)})
    else:
        differ = int(np.count_nonzero(gpu_light != orc.light))
        parity.update({"light_volume": f"UNORM8 light volume ({gpu_light.size} voxels) after one ChangeDirLight, GPU against oracle",
                       "light_voxels_differ": differ, "light_ok": differ == 0})
    res.change_dir_light(light_new, light_old, world)
    # BASELINE.md 2 asks for the single-threaded figure too: the same probe rows and a ChangeDirLight of a 128^3 block of the
    # volume on ONE thread (the operator's cost is per voxel: scaled by the voxel count)
    lib = oracle.load()
    lib.orc_set_num_threads(1)
    t0 = time.perf_counter()
    _, n_1t = orc.raymarch_lit(cam, probe, rp, world)
    ray_1t = n_1t / max(time.perf_counter() - t0, 1e-6)
    m = min(128, vol.shape[0])
    o = (vol.shape[0] - m) // 4  # (a block that is part air, part object, like the volume)
    small = oracle.OracleScene(np.ascontiguousarray(vol[o:o + m, o:o + m, o:o + m]), cfg["light_32bit"])
    small.set_tf_lut(lut)
    small.set_windowing(win)
    t0 = time.perf_counter()
    small.change_dir_light(light_old, light_new, world)
    change_1t = (time.perf_counter() - t0) * (vol.size / float(m ** 3))
    lib.orc_set_num_threads(int(cores))
    single = {"cores": 1, "raymarch_msamples_per_s": round(ray_1t / 1e6, 3), "change_dir_light_s": round(change_1t, 2),
              "value": round(n_full / (n_full / ray_1t + change_1t) / 1e6, 4),
              "sample": f"one thread: the oracle's raymarch of 16 rows of the frame ({n_1t} nominal samples) and a ChangeDirLight of a {m}^3 "
                        f"block of the volume, scaled to the whole volume"}
    cpu = {"value": round(n_full / (frame_s + change_s) / 1e6, 3), "unit": "Msamples/s", "cores": int(cores), "kind": "port", "build": build,
           "sample": f"one step of the oracle, OpenMP x{cores}: ChangeDirLight over the whole light volume ({change_s:.2f} s) + lit "
                     f"raymarch of every {g}-th 8-row group of the same {fb_w}x{fb_h} frame ({n_s} nominal samples in {dt:.2f} s, "
                     f"i.e. {frame_s:.2f} s per frame); light volume taken from the GPU",
           "raymarch_only_msamples_per_s": round(ray_rate / 1e6, 3), "change_dir_light_s": round(change_s, 3), "single_thread": single,
           "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")}}
    return cpu, parity


if __name__ == "__main__":
    main()
