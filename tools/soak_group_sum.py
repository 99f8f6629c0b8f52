"""Soak of adac_scan_group_sum: the cases of tests/test_gpu_group_sum.py over many seeds, with the 32-bit fast path
on and off (tuning knob "group_sum_wide").  usage: python3 tools/soak_group_sum.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
adac = importlib.import_module("duckdb-adaptive-compression_amd")
import test_gpu_group_sum as T
ctx = adac.Context(0)
counts = np.array([2048, 32767, 1, 0, 5000, 70001, 63, 4096], dtype=np.uint32)
for wide in (0, 1):
    adac.set_tuning("group_sum_wide", wide)
    bad = 0
    for rep in range(40):
        for vd in (np.uint16, np.uint8, np.int32, np.uint64):
            for kd, g, top in ((np.uint8, 6, 6), (np.uint16, 40, 50)):
                for vb in (1, 6, 9, 16):
                    rng = np.random.default_rng(5 + rep)
                    try:
                        T.run_case(adac, ctx, rng, vd, kd, counts, g, vb, top)
                    except AssertionError as e:
                        bad += 1
                        print("wide", wide, "FAIL", np.dtype(vd), np.dtype(kd), g, vb, str(e)[:80].replace("\n", " "))
    print("wide", wide, "failures", bad)
