#!/usr/bin/env python3
"""Runs tests/test_gpu_parity.py::test_random_light_operator_sequences_against_oracle for a range of seeds outside the ones the
suite pins (seeds >= 100 are the larger scenes): python tools/hunt_light_operators.py <first> <end>. Needs a GPU."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle
oracle.build(); oracle.load()
from tbraymarcherplugin_amd import abi
import test_gpu_parity as T
abi.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        T.test_random_light_operator_sequences_against_oracle(abi, oracle, seed)
    except Exception as e:
        bad += 1
        print("SEED", seed, "FAILED:", str(e)[:300], flush=True)
    if (seed - lo) % 50 == 49:
        print("... through seed", seed, "failures", bad, flush=True)
print("light-operator sweep seeds", lo, hi, "failures", bad, flush=True)
