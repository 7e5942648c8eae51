#!/usr/bin/env python3
"""Runs the random-scene sweep of tests/test_gpu_fuzz.py over a range of seeds without stopping at the first failure and
prints the seeds that differ from the oracle (and in what): python tools/fuzz_sweep.py FIRST LAST"""
import os
import sys

import torch  # noqa: F401  (before the library: one HIP runtime for the process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_fuzz as T

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last + 1):
    try:
        T.test_random_scene(seed)
    except AssertionError as e:
        bad.append(seed)
        print("seed %d: %s" % (seed, str(e).splitlines()[0][:200] if str(e) else "assert"), flush=True)
    if seed % 250 == 0:
        print("... %d done, %d differ" % (seed - first + 1, len(bad)), flush=True)
print("differing seeds:", bad)
sys.exit(1 if bad else 0)
