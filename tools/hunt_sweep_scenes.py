#!/usr/bin/env python3
"""Runs tests/test_gpu_sweep.py's random operator sequences (scenes whose passes are whole brick layers: the pipelined sweep, its
two-launch form for lights that pull opposite ways, the factor cache) for seeds outside the ones the suite pins (seeds >= 100
are the larger scenes): python tools/hunt_sweep_scenes.py <first> <end>. Needs a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402

oracle.build()
oracle.load()
from tbraymarcherplugin_amd import abi  # noqa: E402
import test_gpu_sweep as T  # noqa: E402

abi.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
sweeps = chunks = 0
for seed in range(lo, hi):
    abi.set_tunable("light_cache_mb", 0 if seed % 4 == 3 else -1)
    try:
        c = T.run_random_sweep_scene(oracle, seed)
        sweeps += c["sweep"]
        chunks += c["chunk"] - c["sweep"]
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "FAILED:", str(e)[:300], flush=True)
    if (seed - lo) % 50 == 49:
        print("... through seed", seed, "failures", bad, flush=True)
print("sweep-scene hunt seeds", lo, hi, "failures", bad, "sweep launches", sweeps, "chain launches", chunks, flush=True)
