"""Multi-GPU decompositions of the hot path on one node.

1. Image-tile sharding of the lit raymarch (BASELINE config 5).

Rays are independent, so the framebuffer is split by rows: rank r renders every N-th group of 8 rows (a
load-balanced interleave: the cube's silhouette is spread evenly over the ranks), volumes are replicated, and the
only exchange is the all-gather of the tiles. The reference has no multi-GPU path; this is new host logic on top
of the single-GPU operator (tbrm_raymarch_lit with a tbrm_tile).

`render` is a callback (tile -> array/tensor of shape [tile.h, tile.w, 4]) so the same logic is exercised on CPU
(gloo) with the oracle as the renderer and on GPUs (RCCL) with the HIP path.

2. Light-parallel illumination (ResetAllLights, RaymarchVolume.cpp:418-451, with many lights).
Lights are independent and their contributions add: rank r propagates lights r, r+N, ... into its own zeroed light
volume, then the volumes are combined with ONE exchange — reduce-scatter of the UNORM8 codes widened to int32,
saturation at 255 (the render target's clamp), all-gather of the UNORM8 result. In real arithmetic this equals the
sequential reference for add-only sequences (every read-modify-write stays on the 1/255 grid and saturating addition
commutes); in fp32 it can differ from the sequential order at near-ties only, so the bit-exact gate of a multi-GPU
reset is the oracle run with the SAME light->rank schedule and combine rule (tests/test_sharding.py), and the
sequential single-GPU result is the reference it is compared to code by code. Float light volumes combine with a plain
sum (tolerance 1e-4). The callbacks make the same code run over gloo on CPU tensors and over RCCL on device tensors.

3. The selective update of ONE light on ONE GPU + a broadcast of the light volume (SURVEY.md §8e "Selective updates (Change) touch
one light -> run on one GPU + broadcast delta, or redundantly on all"). A ChangeDirLight is two serial sweeps: it does not shard
at 512^3, so at N > 1 every GPU either repeats it (no exchange: bench.py's default) or waits for the owner's
copy of the result (one broadcast of the light volume: 128 MiB at 512^3). On the critical path of a step the second form is
Change + broadcast + frame / N against Change + frame / N: it cannot be faster, it only leaves N - 1 GPUs free for other work
(and their factor caches cold). change_dir_light_on_owner implements it behind a callback so that it runs over gloo in the
tests and over RCCL in bench.py --light-update broadcast.
"""
import numpy as np

from . import abi

ROW_GROUP = 8


def rows_per_rank(height, world_size):
    unit = ROW_GROUP * world_size
    if height % unit != 0:
        raise ValueError(f"framebuffer height {height} must be a multiple of {unit} (8 rows x {world_size} ranks)")
    return height // world_size


def rank_tile(width, height, rank, world_size):
    """The tbrm_tile rank `rank` renders: rows 8*rank + 8*world_size*g + (0..7), g = 0, 1, ..."""
    return abi.Tile(0, ROW_GROUP * rank, width, rows_per_rank(height, world_size), world_size)


def rank_rows(height, rank, world_size):
    """Framebuffer row of every output row of rank_tile (the C-ABI's tile row rule, tbrm.h)."""
    h = rows_per_rank(height, world_size)
    j = np.arange(h)
    return ROW_GROUP * rank + (j // ROW_GROUP) * ROW_GROUP * world_size + (j % ROW_GROUP)


def assemble(gathered, height, world_size):
    """gathered: [world_size, rows_per_rank, width, 4] (numpy or torch) -> full frame [height, width, 4]."""
    h = rows_per_rank(height, world_size)
    groups = h // ROW_GROUP
    w = gathered.shape[2]
    # [rank, group, row-in-group, x, c] -> [group, rank, row-in-group, x, c]
    g = gathered.reshape(world_size, groups, ROW_GROUP, w, 4)
    if isinstance(g, np.ndarray):
        g = np.transpose(g, (1, 0, 2, 3, 4))
    else:
        g = g.permute(1, 0, 2, 3, 4)
    return g.reshape(height, w, 4)


def render_sharded(render, width, height, rank, world_size, all_gather):
    """Renders this rank's tile and gathers the frame. all_gather(local) -> [world_size, ...] stack."""
    tile = rank_tile(width, height, rank, world_size)
    local = render(tile)
    gathered = all_gather(local)
    return assemble(gathered, height, world_size)


# ---- light-parallel illumination ---------------------------------------------------------------------------------

def light_schedule(n_lights, rank, world_size):
    """Indices of the lights rank `rank` propagates: round-robin, in ascending order (the reference's order within a rank)."""
    return list(range(rank, n_lights, world_size))


def combine_light_codes(local_u8, world_size, reduce_scatter_sum, all_gather):
    """Saturating sum over ranks of UNORM8 light volumes (any layout, identical on every rank).

    local_u8: 1-D uint8 torch tensor, length divisible by world_size (the bricked device buffer is: bricks of 512 B).
    reduce_scatter_sum(int32 tensor [n]) -> int32 tensor [n / world_size], this rank's slice of the elementwise sum.
    all_gather(uint8 tensor [n / world_size]) -> uint8 tensor [n], the slices of all ranks in rank order.
    """
    import torch

    n = local_u8.numel()
    if n % world_size != 0:
        raise ValueError(f"light volume of {n} bytes does not split over {world_size} ranks")
    mine = reduce_scatter_sum(local_u8.to(torch.int32))
    return all_gather(mine.clamp_(max=255).to(torch.uint8))


def combine_light_float(local_f32, all_reduce_sum):
    """Float light volumes (bLightVolume32Bit): plain sum over ranks; all_reduce_sum(t) -> summed tensor."""
    return all_reduce_sum(local_f32)


def device_light_tensor(res):
    """A torch tensor aliasing the handle's bricked light volume in HBM (uint8 or float32, 1-D). The caller orders the
    library's stream and torch's stream (tbrm_flush before torch reads it, torch.cuda.synchronize before the library does)."""
    import torch

    ptr, nbytes = res.light_volume_device_ptr()
    is_u8 = res.light_dtype == np.uint8

    class _Alias:  # __cuda_array_interface__ v2: torch wraps the pointer without copying
        __cuda_array_interface__ = {"shape": (nbytes if is_u8 else nbytes // 4,), "typestr": "|u1" if is_u8 else "<f4",
                                    "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Alias(), device=torch.device("cuda", res.device))


def reset_all_lights_light_parallel(res, lights, world, rank, world_size, combine_u8, combine_f32=None):
    """ResetAllLights with the lights dealt over the ranks: clear, add this rank's lights, combine in place.

    combine_u8(t) / combine_f32(t) take this rank's light tensor and return the combined one (same shape); they wrap
    combine_light_codes / combine_light_float with the process group's collectives."""
    import torch

    res.clear_light_volume(0.0)
    for i in light_schedule(len(lights), rank, world_size):
        res.add_dir_light(lights[i], True, world)
    res.flush()
    t = device_light_tensor(res)
    out = combine_u8(t) if t.dtype == torch.uint8 else combine_f32(t)
    t.copy_(out)
    torch.cuda.synchronize(t.device)


# ---- selective update on one GPU + broadcast ---------------------------------------------------------------------------

def change_dir_light_on_owner(change, light_tensor, rank, owner, broadcast):
    """ChangeDirLight on rank `owner` only; every rank ends up with the owner's light volume.

    change(): runs the operator on this rank's volume (called on the owner only). light_tensor(): this rank's light volume as a
    tensor the collective can send / receive in place (sharding.device_light_tensor on GPUs). broadcast(t, src): the process
    group's broadcast, issued where it is ordered behind the operator (the library's stream on GPUs)."""
    if rank == owner:
        change()
    t = light_tensor()
    broadcast(t, owner)
    return t
