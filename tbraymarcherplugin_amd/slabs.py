"""Slab-partitioned illumination over the GPUs of one node (BASELINE config 4): the host driver of the
C-ABI's tbrm_slab_* entry points.

The light volume's z range is dealt out in slabs; slab k computes and owns light-volume slices [z_k, z_k+1). Every GPU
keeps the whole (read-only) data volume — 288 GB of HBM hold any volume the plugin loads, so the data is replicated and
the WORK is what gets partitioned. An axis pass is a sequence of chunks (16 / 8 / 4 / 2 slices), and the only
thing one slab needs from another is propagated-light plane content at chunk boundaries:

  * pass along x or y ("lateral": z is the row axis of the slice plane) — all slabs run chunk c at the same time on
    their own rows; after the chunk each slab fetches `halo_rows` rows of the plane from both z neighbours (what the
    bilinear taps of the next chunk's slices can reach from its rows). N-way parallel, one small exchange per chunk.
  * pass along z — the slabs form a pipeline in propagation order; a slab imports the planes of the slab before it,
    runs its chunks and exports its final planes. One light is serial across slabs; consecutive operations overlap
    because a slab is free as soon as it has handed its planes on (every operation is enqueued asynchronously).

Per voxel the arithmetic and its inputs are exactly those of the unpartitioned operator, so the result is bit-identical
to the single-GPU light volume (tests/test_gpu_slabs.py) — no "same schedule" caveat as for the light-parallel reset.
Before a frame is rendered the slabs' light volumes are all-gathered (`gather_light_volume`: in the bricked layout a z
slab is one contiguous byte range) and the frame is rendered in image tiles (sharding.py), as config 4 words it: "halo
exchange for light volume + allgather of tiles".

The driver is written against two small interfaces so that the same code runs (a) on GPUs, one process per GPU, moving
planes with torch.distributed point-to-point operations (RCCL over xGMI), (b) in ONE process holding several handles on
one GPU (how the GPU tests check it on a single-GPU box) and (c) on CPU tensors over gloo with a stand-in backend
(tests/test_slabs.py: the exchange pattern itself):

  member    .slab_index, .light_begin(removed, light, added, world) -> n_passes, .pass_begin(i) -> tbrm_slab_pass fields,
            .pass_chunk(c), .plane(boundary, stream) -> 2-D tensor [plane_h, plane_w] aliasing the plane, .z_begin, .z_end,
            .sync()
  fabric    .owner(slab_index) -> rank, .rank, plus isend / irecv callables (torch.distributed's, or None when every
            slab is local)
"""
import numpy as np

from . import abi


def slab_bounds(depth, n_slabs, unit=32):
    """z ranges of n_slabs equal slabs of a light volume `depth` slices deep; bounds are multiples of `unit`."""
    if depth % (unit * n_slabs) != 0:
        raise ValueError(f"light-volume depth {depth} does not split into {n_slabs} slabs of a multiple of {unit} slices")
    step = depth // n_slabs
    return [(k * step, (k + 1) * step) for k in range(n_slabs)]


class DeviceSlab:
    """One slab on a GPU: an abi.Resources handle plus the z range it owns."""

    def __init__(self, res, slab_index, z_begin, z_end):
        self.res = res
        self.slab_index = slab_index
        self.z_begin, self.z_end = int(z_begin), int(z_end)
        self._slab = abi.Slab(self.z_begin, self.z_end)
        self._pass = None

    def light_begin(self, removed, light, added, world):
        return self.res.slab_light_begin(removed, light, added, world, self._slab)

    def pass_begin(self, index):
        self._pass = self.res.slab_pass_begin(index)
        return self._pass

    def pass_chunk(self, chunk):
        self.res.slab_pass_chunk(chunk)

    def plane(self, boundary, stream):
        import torch

        ptr = self.res.slab_pass_plane(boundary, stream)
        h, w = self._pass.plane_h, self._pass.plane_w
        typestr = "<f4" if self._pass.plane_elem_bytes == 4 else "|u1"  # float planes; UNORM8 read / write buffers of a sliced pass

        class _Alias:
            __cuda_array_interface__ = {"shape": (h, w), "typestr": typestr, "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Alias(), device=torch.device("cuda", self.res.device))

    def sync(self):
        self.res.flush()

    def render_stage(self, camera, tile, params, world, state, direction):
        """accumulates this slab's samples of every ray of the sweep into `state` (a [tile.h, tile.w, 4] float32 device tensor)"""
        self.res.raymarch_lit_slab_device(camera, tile, params, world, state.data_ptr(), self._slab, direction)

    def stream_context(self):
        """torch's current stream := the handle's stream, so that torch copies / RCCL operations are ordered with the
        library's kernels without a host synchronisation."""
        import torch

        return torch.cuda.stream(torch.cuda.ExternalStream(self.res.stream(), device=torch.device("cuda", self.res.device)))


class Fabric:
    """Moves tensors between slabs: a copy when both live in this process, point-to-point operations otherwise.

    p2p(ops) executes one batch: ops is a list of ("send" | "recv", tensor, peer rank) and the call returns when the
    operations are complete or ordered with the current stream (torch.distributed.batch_isend_irecv + wait: one RCCL
    group per exchange, so the sends and receives of a rank cannot block each other)."""

    def __init__(self, owner_of_slab, rank=0, p2p=None, sync_local=True):
        self.owner_of_slab = list(owner_of_slab)
        self.rank = rank
        self._p2p = p2p
        self._local, self._ops = [], []
        self.sync_local = sync_local
        self.bytes_moved = 0

    def owner(self, slab_index):
        return self.owner_of_slab[slab_index]

    def is_local(self, slab_index):
        return self.owner(slab_index) == self.rank

    def move(self, src_slab, src, dst_slab, dst):
        """Registers one transfer. src / dst are callables returning the tensor views (evaluated only on the side that
        holds them). Every rank registers the same transfers in the same order; each executes its part."""
        s_here, d_here = self.is_local(src_slab), self.is_local(dst_slab)
        if s_here and d_here:
            self._local.append((src(), dst()))
        elif s_here:
            t = src()
            self._ops.append(("send", t, self.owner(dst_slab)))
            self.bytes_moved += t.numel() * t.element_size()
        elif d_here:
            self._ops.append(("recv", dst(), self.owner(src_slab)))

    def complete(self, members):
        """Executes the registered transfers. Local copies between handles of one process are ordered by draining the
        members' streams (this is the single-GPU emulation, not a fast path)."""
        if self._local:
            if self.sync_local:
                for m in members:
                    m.sync()
            staged = [s.clone() for s, _ in self._local]  # all reads before any write: a transfer's source may be another's target
            for t, (_, d) in zip(staged, self._local):
                d.copy_(t)
                self.bytes_moved += t.numel() * t.element_size()
            if self.sync_local and staged and staged[0].is_cuda:
                import torch

                torch.cuda.synchronize(staged[0].device)
        if self._ops:
            self._p2p(self._ops)
        self._local, self._ops = [], []


def _run_pass(members, by_index, fabric, index):
    descs = {m.slab_index: m.pass_begin(index) for m in members}
    n_slabs = len(fabric.owner_of_slab)
    any_desc = next(iter(descs.values()))
    streams = any_desc.streams
    if any_desc.lateral:
        n_chunks, halo = any_desc.n_chunks, any_desc.halo_rows
        for c in range(n_chunks):
            for m in members:
                m.pass_chunk(c)
            if c + 1 == n_chunks:
                break
            b = c + 1  # the planes the next chunk reads
            for k in range(n_slabs - 1):  # boundary between slab k and k + 1, at z = bounds[k + 1]
                lo, hi = by_index.get(k), by_index.get(k + 1)
                z = fabric.z_bounds[k + 1]
                for si in range(streams):
                    # slab k's top rows become slab k+1's lower halo, slab k+1's bottom rows slab k's upper halo
                    fabric.move(k, lambda lo=lo, si=si: lo.plane(b, si)[z - halo:z], k + 1, lambda hi=hi, si=si: hi.plane(b, si)[z - halo:z])
                    fabric.move(k + 1, lambda hi=hi, si=si: hi.plane(b, si)[z:z + halo], k, lambda lo=lo, si=si: lo.plane(b, si)[z:z + halo])
            fabric.complete(members)
    else:
        # pipeline in propagation order: first_chunk is known for local slabs only, the order follows from the direction
        order = list(range(n_slabs)) if any_desc.dir > 0 else list(range(n_slabs - 1, -1, -1))
        for pos, k in enumerate(order):
            m = by_index.get(k)
            if pos > 0:
                prev = order[pos - 1]
                pm = by_index.get(prev)
                for si in range(streams):
                    fabric.move(prev, lambda pm=pm, si=si: pm.plane(descs[pm.slab_index].n_chunks, si), k, lambda m=m, si=si: m.plane(0, si))
                fabric.complete(members)
            if m is not None:
                for c in range(descs[k].n_chunks):
                    m.pass_chunk(c)


def light_operation(members, fabric, removed, light, added, world):
    """AddDirLight (removed is None) or ChangeDirLight (removed -> light) over the slabs. Returns False when a Change
    has to be run as remove + add (major axes differ; the caller does that: change_dir_light)."""
    by_index = {m.slab_index: m for m in members}
    try:
        counts = [m.light_begin(removed, light, added, world) for m in members]
    except abi.TbrmError as e:
        if e.code == abi.ERR_AXES_DIFFER and removed is not None:
            return False
        raise
    for i in range(counts[0] if counts else 0):
        _run_pass(members, by_index, fabric, i)
    return True


def add_dir_light(members, fabric, light, added, world):
    light_operation(members, fabric, None, light, added, world)


def change_dir_light(members, fabric, old, new, world):
    """ChangeDirLightInSingleVolume over the slabs, with the reference's fallback (LightingShaders.cpp:192-198)."""
    if not light_operation(members, fabric, old, new, True, world):
        add_dir_light(members, fabric, old, False, world)
        add_dir_light(members, fabric, new, True, world)


def reset_all_lights(members, fabric, lights, world, clear):
    """ResetAllLights (RaymarchVolume.cpp:418-451): clear, then add every light. clear(member) clears its light volume."""
    for m in members:
        clear(m)
    for light in lights:
        add_dir_light(members, fabric, light, True, world)


# ---- light-volume halo exchange (slab-resident handles) ------------------------------------------------------------------

def exchange_light_halos(members, fabric):
    """After light operations and before a frame: every slab-resident handle receives its two neighbours' boundary brick
    layers of the light volume (what the raymarch's taps reach beyond the slab). The neighbours form a ring — the light
    volume is sampled with wrap addressing, so slab 0's lower neighbour is the last slab. One exchange, two layers each."""
    import torch

    by_index = {m.slab_index: m for m in members}
    n = len(fabric.owner_of_slab)
    if n == 1:
        return

    def layer(m, side, which):
        snd, rcv, nbytes = m.res.slab_light_halo(side)
        ptr = snd if which == "send" else rcv

        class _Alias:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Alias(), device=torch.device("cuda", m.res.device))

    for m in members:
        m.sync()
    for k in range(n):
        up = (k + 1) % n  # slab k's last layer -> the lower halo of slab k+1; slab k+1's first layer -> the upper halo of slab k
        a, b = by_index.get(k), by_index.get(up)
        fabric.move(k, lambda a=a: layer(a, 1, "send"), up, lambda b=b: layer(b, 0, "recv"))
        fabric.move(up, lambda b=b: layer(b, 0, "send"), k, lambda a=a: layer(a, 1, "recv"))
    fabric.complete(members)


# ---- the lit frame, slab by slab ---------------------------------------------------------------------------------------

def render_lit(members, fabric, camera, tile, params, world, new_state):
    """A frame marched slab by slab (tbrm_raymarch_lit_slab_device): every ray's samples are accumulated by the slab they
    lie in, in ray order — one sweep up through the slabs for the rays that travel towards +z in volume space, one sweep
    down for the others — and the per-pixel LightEnergy state travels with the sweep (16 bytes per pixel per hop). The
    result is bit for bit the unpartitioned frame. new_state() -> a zeroed [tile.h, tile.w, 4] float32 tensor on the
    member's device. Returns the frame on the rank that owns slab 0, None elsewhere."""
    by_index = {m.slab_index: m for m in members}
    n = len(fabric.owner_of_slab)
    states = {m.slab_index: new_state() for m in members}
    order = [(k, +1) for k in range(n)] + [(k, -1) for k in range(n - 1, -1, -1)]
    prev = None
    for k, direction in order:
        if prev is not None and prev != k:
            fabric.move(prev, lambda prev=prev: states[prev], k, lambda k=k: states[k])
            fabric.complete(members)
        m = by_index.get(k)
        if m is not None:
            m.render_stage(camera, tile, params, world, states[k], direction)
        prev = k
    return states.get(0)


def make_fabric(z_bounds, owner_of_slab=None, rank=0, p2p=None, sync_local=True):
    """z_bounds: the n_slabs + 1 slab boundaries. owner_of_slab defaults to everything in this process."""
    n = len(z_bounds) - 1
    f = Fabric(owner_of_slab if owner_of_slab is not None else [rank] * n, rank, p2p, sync_local)
    f.z_bounds = list(z_bounds)
    return f


# ---- light volume: slabs -> everywhere -------------------------------------------------------------------------------

def slab_byte_range(res, z_begin, z_end):
    """Byte range of light-volume slices [z_begin, z_end) in the handle's bricked buffer: z brick layers are contiguous."""
    _, nbytes = res.light_volume_device_ptr()
    depth = res.light_dims[2]
    layers = (depth + 7) // 8
    if z_begin % 8 or (z_end % 8 and z_end != depth):
        raise ValueError("slab bounds must be multiples of 8")
    per_layer = nbytes // layers
    return (z_begin // 8) * per_layer, ((z_end + 7) // 8) * per_layer


def gather_light_volume(members, fabric, all_gather_into=None):
    """Completes every local handle's light volume with the other slabs' parts.

    One process per GPU, equal slabs: all_gather_into(full_tensor, my_part) is torch.distributed.all_gather_into_tensor
    (byte tensors over the bricked buffers). Several handles in one process: device copies."""
    import torch
    from .sharding import device_light_tensor

    for m in members:
        m.sync()
    tensors = {m.slab_index: device_light_tensor(m.res).view(torch.uint8) for m in members}
    ranges = {k: slab_byte_range(members[0].res, fabric.z_bounds[k], fabric.z_bounds[k + 1]) for k in range(len(fabric.z_bounds) - 1)}
    if all_gather_into is not None:
        (m,) = members  # one slab per process
        t = tensors[m.slab_index]
        lo, hi = ranges[m.slab_index]
        all_gather_into(t, t[lo:hi].clone())
        torch.cuda.synchronize(t.device)
        return
    for src in members:
        lo, hi = ranges[src.slab_index]
        for dst in members:
            if dst is not src:
                tensors[dst.slab_index][lo:hi].copy_(tensors[src.slab_index][lo:hi])
    torch.cuda.synchronize()


def dist_fabric(z_bounds, rank, world_size, group=None, member=None):
    """One slab per process: slab k lives on rank k, planes move with torch.distributed point-to-point operations (RCCL
    over xGMI on GPUs; gloo on CPU tensors in the tests).

    Device tensors alias buffers that the library's kernels read and write on the HANDLE's HIP stream, so the
    point-to-point operations have to be enqueued relative to that stream: every batch is issued with torch's current
    stream set to the handle's stream (`member`: the DeviceSlab of this rank — required for device tensors). A send is
    then ordered behind the chunk that wrote the plane and the next chunk behind the receive, without a host
    synchronisation and wherever the driver is called from."""
    import contextlib

    import torch.distributed as dist

    if len(z_bounds) - 1 != world_size:
        raise ValueError("one slab per rank")

    def p2p(ops):
        on_device = any(t.is_cuda for _, t, _ in ops)
        if on_device and member is None:
            raise ValueError("dist_fabric: device tensors need `member` (the rank's DeviceSlab) so that the transfers are "
                             "ordered with the handle's stream")
        with (member.stream_context() if on_device else contextlib.nullcontext()):
            batch = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, group) for kind, t, peer in ops]
            for w in dist.batch_isend_irecv(batch):
                w.wait()

    return make_fabric(z_bounds, list(range(world_size)), rank, p2p, sync_local=False)
