"""The slab driver's exchange pattern (slabs.py) on CPU: a stand-in member with the chunk structure of the real operator —
a chunk of a lateral pass reads `halo_rows` rows beyond its slab from the plane of the previous chunk boundary, a pass
along z imports the planes of the slab before it — whose planes start as NaN wherever the slab has not computed or
received them, so that a missing, misplaced or mis-ordered transfer poisons the result. Checked with every slab in one
process (device-copy path of the fabric) and with one slab per gloo rank (isend / irecv path). The GPU tests
(test_gpu_slabs.py) run the same driver on the HIP operator."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from tbraymarcherplugin_amd import slabs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (12, 10, 32)  # x, y, z of the toy light volume
M = 4                # slices per chunk
HALO = 4             # rows a chunk can reach (1 per slice)


class ToyPass:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def toy_occlusion(seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(0.5, 1.0, size=(DIMS[2], DIMS[1], DIMS[0])).astype(np.float32))


class ToyMember:
    """light[z, y, x] += plane after every slice; plane'[v, u] = occ * 0.5 * (plane[v + dv, u + du] + plane[v, u]) with a
    border value outside — the dependency structure of AddDirLightShader.usf:81-126 without its arithmetic."""

    def __init__(self, slab_index, z_begin, z_end, occ):
        self.slab_index, self.z_begin, self.z_end = slab_index, z_begin, z_end
        self.occ = occ
        self.light = torch.zeros(DIMS[2], DIMS[1], DIMS[0])
        self.passes = []

    def light_begin(self, removed, light, added, world):
        self.passes = light["passes"]
        self.streams = 2 if removed is not None else 1
        self.sign = 1.0 if added else -1.0
        return len(self.passes)

    def pass_begin(self, index):
        p = self.passes[index]
        self.axis, self.dir, self.du, self.dv = p["axis"], p["dir"], p["du"], p["dv"]
        dims3 = {0: (DIMS[1], DIMS[2], DIMS[0]), 1: (DIMS[0], DIMS[2], DIMS[1]), 2: (DIMS[0], DIMS[1], DIMS[2])}[self.axis]
        self.W, self.H, depth = dims3
        lateral = self.axis != 2
        if lateral:
            self.first, self.n_chunks = 0, depth // M
            self.start = 0 if self.dir > 0 else depth - 1
        else:
            self.n_chunks = (self.z_end - self.z_begin) // M
            self.first = (self.z_begin if self.dir > 0 else depth - self.z_end) // M
            self.start = self.z_begin if self.dir > 0 else self.z_end - 1
        self.planes = [[torch.full((self.H, self.W), float("nan")) for _ in range(self.streams)] for _ in range(2)]
        if self.first == 0:  # the pass starts here: the cleared buffers' value, known everywhere without an exchange
            for si in range(self.streams):
                self.planes[0][si][:] = 1.0 + si
        self.lateral = lateral
        return ToyPass(axis=self.axis, dir=self.dir, lateral=int(lateral), streams=self.streams, plane_w=self.W, plane_h=self.H,
                       chunk_slices=M, chunks_of_pass=depth // M, first_chunk=self.first, n_chunks=self.n_chunks,
                       halo_rows=HALO if lateral else 0)

    def plane(self, boundary, stream):
        return self.planes[boundary & 1][stream]

    def _voxels(self, j):
        if self.axis == 0:
            return (slice(None), slice(None), j), True   # [z, y, x=j] -> rows z, cols y
        if self.axis == 1:
            return (slice(None), j, slice(None)), True   # [z, y=j, x] -> rows z, cols x
        return (j, slice(None), slice(None)), False      # [z=j, y, x] -> rows y, cols x

    def pass_chunk(self, c):
        border = 0.25
        r0, r1 = (max(self.z_begin - HALO, 0), min(self.z_end + HALO, self.H)) if self.lateral else (0, self.H)
        for si in range(self.streams):
            win = self.planes[c & 1][si][r0:r1].clone()  # NaN where nobody delivered
            for s in range(M):
                j = self.start + (c * M + s) * self.dir
                idx, _ = self._voxels(j)
                occ = self.occ[idx][r0:r1]
                padded = torch.full((win.shape[0] + 2, win.shape[1] + 2), border)
                padded[1:-1, 1:-1] = win
                # rows outside the window that exist in the plane are unknown here (NaN), rows outside the plane are border
                if r0 > 0:
                    padded[0, :] = float("nan")
                if r1 < self.H:
                    padded[-1, :] = float("nan")
                shifted = padded[1 + self.dv:1 + self.dv + win.shape[0], 1 + self.du:1 + self.du + win.shape[1]]
                win = occ * 0.5 * (shifted + win)
                own = slice(self.z_begin - r0, self.z_end - r0) if self.lateral else slice(None)
                contrib = win[own] * (self.sign if self.streams == 1 else (1.0 if si == 0 else -1.0))
                if self.lateral:
                    full_idx = list(idx)
                    full_idx[0] = slice(self.z_begin, self.z_end)
                    self.light[tuple(full_idx)] += contrib
                else:
                    self.light[idx] += contrib
            out = self.planes[(c + 1) & 1][si]
            out[:] = float("nan")
            if self.lateral:
                out[self.z_begin:self.z_end] = win[self.z_begin - r0:self.z_end - r0]
            else:
                out[:] = win

    def sync(self):
        pass

    def render_stage(self, camera, tile, params, world, state, direction):
        """stand-in for a slab's stage of the lit march: even columns belong to the upward sweep, odd ones to the downward
        sweep; a stage appends its slab number as a decimal digit, so the final value spells the order of the stages"""
        cols = slice(0, None, 2) if direction > 0 else slice(1, None, 2)
        state[:, cols, :] = state[:, cols, :] * 10.0 + (self.slab_index + 1)


PASSES = {"passes": [{"axis": 0, "dir": 1, "du": -1, "dv": 1}, {"axis": 2, "dir": -1, "du": 1, "dv": -1},
                     {"axis": 1, "dir": -1, "du": 0, "dv": -1}, {"axis": 2, "dir": 1, "du": -1, "dv": 0}]}


def run_reference(occ):
    whole = ToyMember(0, 0, DIMS[2], occ)
    fabric = slabs.make_fabric([0, DIMS[2]])
    slabs.light_operation([whole], fabric, None, PASSES, True, None)
    slabs.light_operation([whole], fabric, PASSES, PASSES, True, None)  # two streams
    assert torch.isfinite(whole.light).all()
    return whole.light


@pytest.mark.parametrize("n_slabs", [2, 4])
def test_all_slabs_in_one_process(n_slabs):
    occ = toy_occlusion(7)
    want = run_reference(occ)
    bounds = slabs.slab_bounds(DIMS[2], n_slabs, unit=8)
    members = [ToyMember(k, *bounds[k], occ) for k in range(n_slabs)]
    fabric = slabs.make_fabric([b[0] for b in bounds] + [DIMS[2]])
    slabs.light_operation(members, fabric, None, PASSES, True, None)
    slabs.light_operation(members, fabric, PASSES, PASSES, True, None)
    for m in members:
        got = m.light[m.z_begin:m.z_end]
        assert torch.isfinite(got).all(), f"slab {m.slab_index} used plane rows nobody delivered"
        assert torch.equal(got, want[m.z_begin:m.z_end]), f"slab {m.slab_index}"
    assert fabric.bytes_moved > 0


def test_a_dropped_halo_is_noticed():
    """the stand-in really depends on the exchange: without it the result is poisoned"""
    occ = toy_occlusion(7)
    bounds = slabs.slab_bounds(DIMS[2], 2, unit=8)
    members = [ToyMember(k, *bounds[k], occ) for k in range(2)]
    fabric = slabs.make_fabric([b[0] for b in bounds] + [DIMS[2]])
    fabric.move = lambda *a, **k: None
    slabs.light_operation(members, fabric, None, PASSES, True, None)
    assert not all(torch.isfinite(m.light[m.z_begin:m.z_end]).all() for m in members)


def _worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import test_slabs as T

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    occ = T.toy_occlusion(7)
    want = T.run_reference(occ)
    bounds = slabs.slab_bounds(T.DIMS[2], world_size, unit=8)
    me = T.ToyMember(rank, *bounds[rank], occ)
    fabric = slabs.dist_fabric([b[0] for b in bounds] + [T.DIMS[2]], rank, world_size)
    slabs.light_operation([me], fabric, None, T.PASSES, True, None)
    slabs.light_operation([me], fabric, T.PASSES, T.PASSES, True, None)
    got = me.light[me.z_begin:me.z_end]
    ok = bool(torch.isfinite(got).all()) and torch.equal(got, want[me.z_begin:me.z_end]) and fabric.bytes_moved > 0
    frame = slabs.render_lit([me], fabric, None, None, None, None, lambda: torch.zeros(4, 6, 4))
    ok = ok and ((frame is not None and torch.equal(frame, T.expected_frame(world_size))) if rank == 0 else frame is None)
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "mismatch")


@pytest.mark.parametrize("world_size", [2, 4])
def test_one_slab_per_gloo_rank(tmp_path, world_size):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
    for r in range(world_size):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def expected_frame(n_slabs, h=4, w=6):
    up = float("".join(str(k + 1) for k in range(n_slabs)))
    down = float("".join(str(k + 1) for k in reversed(range(n_slabs))))
    f = torch.zeros(h, w, 4)
    f[:, 0::2, :] = up
    f[:, 1::2, :] = down
    return f


def test_frame_state_travels_up_then_down_in_one_process():
    occ = toy_occlusion(7)
    bounds = slabs.slab_bounds(DIMS[2], 4, unit=8)
    members = [ToyMember(k, *bounds[k], occ) for k in range(4)]
    fabric = slabs.make_fabric([b[0] for b in bounds] + [DIMS[2]])
    frame = slabs.render_lit(members, fabric, None, None, None, None, lambda: torch.zeros(4, 6, 4))
    assert torch.equal(frame, expected_frame(4))


def test_slab_bounds():
    assert slabs.slab_bounds(128, 4) == [(0, 32), (32, 64), (64, 96), (96, 128)]
    with pytest.raises(ValueError):
        slabs.slab_bounds(96, 4)
