#!/usr/bin/env python3
"""Runs tests/test_gpu_slabs.py::test_random_slab_resident_scenes_against_one_handle for a range of seeds outside the ones the
suite pins: python tools/hunt_slab_scenes.py <first> <end>. Needs a GPU."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tbraymarcherplugin_amd import abi
import test_gpu_slabs as T
abi.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        T.test_random_slab_resident_scenes_against_one_handle(abi, seed)
    except Exception as e:
        bad += 1
        print("SEED", seed, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
print("slab-resident scene sweep seeds", lo, hi, "failures", bad, flush=True)
