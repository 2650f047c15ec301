"""Deterministic synthetic workloads of SURVEY.md §8d (volume, transfer functions, camera, lights, configs).

Not part of the hot path: input generators for tests and bench.py. numpy builds the small parity-test volumes;
the torch variant builds the 512^3 bench volume directly in HBM.
"""
import numpy as np

from . import abi

MASK64 = (1 << 64) - 1


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def blob_params(seed, n_blobs=6):
    """Blob centres in [-0.3,0.3]^3 and widths in [0.03,0.08] from splitmix64(seed)."""
    st = seed & MASK64
    out = []
    for _ in range(n_blobs):
        vals = []
        for _ in range(4):
            st, z = _splitmix64(st)
            vals.append((z >> 11) / float(1 << 53))
        out.append((tuple(-0.3 + 0.6 * v for v in vals[:3]), 0.03 + 0.05 * vals[3]))
    return out


def seed_for_config(cfg):
    return 0x5EED0000 + int(cfg)


def _hash32_np(x, y, z, seed):
    h = (x.astype(np.uint32) * np.uint32(73856093)) ^ (y.astype(np.uint32) * np.uint32(19349663)) ^ (
        z.astype(np.uint32) * np.uint32(83492791)) ^ np.uint32(seed & 0xFFFFFFFF)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def make_volume_numpy(dims, dtype, seed):
    """dims = (nx, ny, nz); returns an array indexed [z, y, x]."""
    nx, ny, nz = dims
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    px = (x + 0.5) / nx - 0.5
    py = (y + 0.5) / ny - 0.5
    pz = (z + 0.5) / nz - 0.5
    r = np.sqrt(px * px + py * py + pz * pz)
    t = np.clip((r - 0.45) / (0.40 - 0.45), 0.0, 1.0)
    v = 0.35 * (t * t * (3.0 - 2.0 * t)) + 0.50 * np.exp(-(((r - 0.30) / 0.02) ** 2))
    for (cx, cy, cz), sigma in blob_params(seed):
        d2 = (px - cx) ** 2 + (py - cy) ** 2 + (pz - cz) ** 2
        v += 0.4 * np.exp(-d2 / (sigma * sigma))
    with np.errstate(over="ignore"):
        h = _hash32_np(x, y, z, seed)
    v += 0.02 * (h.astype(np.float64) / 4294967296.0 - 0.5)
    v = np.clip(v, 0.0, 1.0)
    return _quantise_np(v, dtype)


def _quantise_np(v, dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return v.astype(np.float32)
    scale = 255.0 if dtype == np.uint8 else 65535.0
    return np.floor(v * scale + 0.5).astype(dtype)


def make_volume_torch(dims, dtype, seed, device):
    """Same field generated on `device` with torch (bench-size volumes). Returns a [z,y,x] tensor."""
    import torch

    nx, ny, nz = dims
    f64 = torch.float64
    xs = torch.arange(nx, device=device)
    ys = torch.arange(ny, device=device)
    zs = torch.arange(nz, device=device)
    px = ((xs.to(f64) + 0.5) / nx - 0.5).view(1, 1, nx)
    py = ((ys.to(f64) + 0.5) / ny - 0.5).view(1, ny, 1)
    out = torch.empty((nz, ny, nx), dtype={np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16,
                                           np.dtype(np.float32): torch.float32}[np.dtype(dtype)], device=device)
    blobs = blob_params(seed)
    M32 = 0xFFFFFFFF
    slab = max(1, min(nz, (1 << 24) // max(1, nx * ny)))  # bound temporaries to ~16M voxels
    for z0 in range(0, nz, slab):
        z1 = min(nz, z0 + slab)
        zi = zs[z0:z1]
        pz = ((zi.to(f64) + 0.5) / nz - 0.5).view(-1, 1, 1)
        r = torch.sqrt(px * px + py * py + pz * pz)
        t = torch.clamp((r - 0.45) / (0.40 - 0.45), 0.0, 1.0)
        v = 0.35 * (t * t * (3.0 - 2.0 * t)) + 0.50 * torch.exp(-(((r - 0.30) / 0.02) ** 2))
        for (cx, cy, cz), sigma in blobs:
            d2 = (px - cx) ** 2 + (py - cy) ** 2 + (pz - cz) ** 2
            v = v + 0.4 * torch.exp(-d2 / (sigma * sigma))
        h = ((xs.view(1, 1, nx) * 73856093) & M32) ^ ((ys.view(1, ny, 1) * 19349663) & M32) ^ (
            (zi.view(-1, 1, 1) * 83492791) & M32) ^ (seed & M32)
        h = h ^ (h >> 16)
        h = (h * 0x85EBCA6B) & M32
        h = h ^ (h >> 13)
        h = (h * 0xC2B2AE35) & M32
        h = h ^ (h >> 16)
        v = v + 0.02 * (h.to(f64) / 4294967296.0 - 0.5)
        v = torch.clamp(v, 0.0, 1.0)
        if np.dtype(dtype) == np.float32:
            out[z0:z1] = v.to(torch.float32)
        else:
            scale = 255.0 if np.dtype(dtype) == np.uint8 else 65535.0
            q = torch.floor(v * scale + 0.5)
            out[z0:z1] = q.to(torch.int32).to(out.dtype)
    return out


# piecewise-linear colour-curve keys: (times, values) per channel R,G,B,A
def _keys(rows):
    t = [r[0] for r in rows]
    return [(t, [r[1 + c] for r in rows]) for c in range(4)]


# TF-A "dense": low opacity, most rays traverse the whole cube
TF_A_KEYS = _keys([(0.0, 0, 0, 0, 0), (0.25, .8, .4, .3, 0), (0.45, .9, .6, .5, .02), (0.70, 1, 1, .9, .15), (1.0, 1, 1, 1, .40)])
# TF-B "bone": keys of the reference's Content/Curves/TF_CT-Bone.uasset
TF_B_KEYS = _keys([(0.0, 0, 0, 0, 0),
                   (0.49344614148139954, 0.7294120192527771, 0.2549020051956177, 0.3019610047340393, 0),
                   (0.6013756990432739, 0.9058820009231567, 0.8156859874725342, 0.5529410243034363, 0.715686023235321),
                   (1.0, 1, 1, 1, 0.7058820128440857)])

# (world direction, intensity) L0..L7
LIGHTS = [((1, .35, -.5), 0.5), ((-.4, 1, -.3), 0.4), ((.2, -.3, -1), 0.4), ((-1, -.6, .4), 0.3),
          ((.6, -1, -.2), 0.3), ((-.3, .2, 1), 0.3), ((1, -.1, .9), 0.2), ((-.8, .9, -.6), 0.2)]


def light(i):
    d, inten = LIGHTS[i]
    return abi.DirLightParams(d, inten)


def rotate_z(direction, degrees):
    a = np.deg2rad(degrees)
    x, y, z = direction
    return (x * np.cos(a) - y * np.sin(a), x * np.sin(a) + y * np.cos(a), z)


VOLUME_SCALE = 100.0  # cube mesh scale (RaymarchVolume.cpp:47)


def default_world():
    return abi.make_world(abi.identity_transform(VOLUME_SCALE))


def default_camera(width, height, vfov_deg=60.0):
    """Eye at volume-local (-1.45,-0.95,0.80), looking at the cube centre, up +Z."""
    eye = np.array([-1.45, -0.95, 0.80]) * VOLUME_SCALE
    return abi.look_at_camera(eye, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), vfov_deg, width, height)


CONFIGS = {
    1: dict(n=128, dtype=np.float32, light_32bit=True, fb=256, steps=128, lights=[0], tf="A",
            window=(0.5, 1.0, True, True)),
    2: dict(n=256, dtype=np.uint16, light_32bit=False, fb=512, steps=256, lights=[0], tf="A",
            window=(0.5, 0.9, True, False)),
    3: dict(n=512, dtype=np.uint16, light_32bit=False, fb=1024, steps=512, lights=[0, 1, 2, 3], tf="A",
            window=(0.5, 0.9, True, False)),
    4: dict(n=1024, dtype=np.uint16, light_32bit=False, fb=1024, steps=1024, lights=[0, 1, 2, 3], tf="A",
            window=(0.5, 0.9, True, False)),
    5: dict(n=512, dtype=np.uint16, light_32bit=False, fb=2048, steps=512, lights=list(range(8)), tf="B",
            window=(0.5, 0.8, True, True)),
}


def tf_keys(name):
    return TF_A_KEYS if name == "A" else TF_B_KEYS
