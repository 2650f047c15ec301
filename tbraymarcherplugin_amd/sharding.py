"""Image-tile sharding of the lit raymarch across the GPUs of one node (BASELINE config 5).

Rays are independent, so the framebuffer is split by rows: rank r renders every N-th group of 8 rows (a
load-balanced interleave: the cube's silhouette is spread evenly over the ranks), volumes are replicated, and the
only exchange is the all-gather of the tiles. The reference has no multi-GPU path; this is new host logic on top
of the single-GPU operator (tbrm_raymarch_lit with a tbrm_tile).

`render` is a callback (tile -> array/tensor of shape [tile.h, tile.w, 4]) so the same logic is exercised on CPU
(gloo) with the oracle as the renderer and on GPUs (RCCL) with the HIP path.
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
