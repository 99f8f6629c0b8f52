#!/usr/bin/env python3
"""One-off soak: the differential fuzz of tests/test_gpu_fuzz.py over many more seeds than the suite carries.
usage: python tools/soak_fuzz.py [first_seed] [count]   (ADAC_TUNING=knob=value,... sets tuning knobs first)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402

adac = importlib.import_module("duckdb-adaptive-compression_amd")
adac.build()
orc.build()
for kv in filter(None, os.environ.get("ADAC_TUNING", "").split(",")):   # e.g. ADAC_TUNING=single_pass_encode=2
    k, v = kv.split("=")
    adac.set_tuning(k, int(v))
ctx = adac.Context(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fn = getattr(fz.test_random_columns, "__wrapped__", fz.test_random_columns)
bad = []
for seed in range(first, first + count):
    try:
        fn(adac, orc, ctx, seed)
    except Exception as e:  # noqa: BLE001
        bad.append((seed, repr(e)[:300]))
        if len(bad) >= 5:
            break
    if (seed - first) % 200 == 199:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done: %d seeds, %d failures" % (count, len(bad)))
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
