#!/usr/bin/env python3
"""A/B of the fused scans between library builds, on ONE box in ONE call (box-to-box spread is 3-5 %, more than most
kernel changes are worth).  Every library runs in a child process of its own (a process maps one libadacodec.so), the
children are interleaved A B A B ... and the best of `--rounds` is kept per cell.

  python tools/ab_scan.py --libs ab/libadacodec_r02i.so duckdb-adaptive-compression_amd/libadacodec.so \
         --cells u64:8 u64:13 u32:8 u32:13 u8:3 --ops sum count select --out gpurun_out/ab.json
"""
import argparse
import importlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "duckdb-adaptive-compression_amd"


def child(lib_path, cells, ops, steps):
    adac = importlib.import_module(PKG)
    adac.LIB_PATH = os.path.abspath(lib_path)  # before the first lib() call
    ctx = adac.Context(0)
    rng = np.random.default_rng(7)
    res = {}
    for cell in cells:
        t, w = cell.split(":")
        w = int(w)
        dtype = np.dtype({"u64": np.uint64, "u32": np.uint32, "u16": np.uint16, "u8": np.uint8}[t])
        rows = int(400e6 * 8 / w)  # packed bytes beyond the 256 MiB Infinity Cache
        counts = adac.appender_segment_counts(rows, dtype.itemsize)
        vals = rng.integers(0, 2 ** w, size=rows, dtype=np.uint32).astype(dtype)
        lay = adac.Layout(ctx, dtype, counts)
        d_vals = ctx.upload(vals)
        d_words = ctx.alloc(lay.max_arena_words * 8 + 128).zero()
        lay.encode(d_vals, d_words)
        ctx.sync()
        del d_vals
        descs = lay.get_descs()
        rd = int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum())
        d_res = ctx.alloc(len(counts) * 8)
        d_bm = ctx.alloc((rows + 63) // 64 * 8 + 8)
        d_out = ctx.alloc(rows * dtype.itemsize + 64) if "decode" in ops else None
        fns = {"decode": lambda: lay.unpack(d_words, d_out),   # GB/s of packed bytes READ, like the others
               "sum": lambda: lay.scan_sum(d_words, d_res),
               "count": lambda: lay.scan_count_between(d_words, 0, 2 ** (w - 1), d_res),
               "select": lambda: lay.scan_select_between(d_words, 0, 2 ** (w - 1), d_bm, d_res)}
        for op in ops:
            fn = fns[op]
            fn()
            ctx.sync()
            ctx.timer_start()
            for _ in range(steps):
                fn()
            ms = ctx.timer_stop() / steps
            res["%s:%s" % (cell, op)] = rd / (ms * 1e-3) / 1e9
        if "sum" in ops:
            got = int(d_res.download(np.uint64, len(counts)).sum(dtype=np.uint64)) if ops[-1] == "sum" else None
            if got is not None:
                assert got == int(vals.astype(np.uint64).sum(dtype=np.uint64)), "SUM parity"
        del d_words, d_res, d_bm, d_out, lay
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--cells", nargs="+", default=["u64:8", "u64:13", "u32:8", "u32:13", "u8:3"])
    ap.add_argument("--ops", nargs="+", default=["select", "count", "sum"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.cells, a.ops, a.steps)
        return
    best = {lib: {} for lib in a.libs}
    for r in range(a.rounds):
        for lib in a.libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--cells", *a.cells, "--ops",
                                  *a.ops, "--steps", str(a.steps), "--libs", "x"], capture_output=True, text=True)
            if out.returncode != 0:
                print(out.stderr[-2000:], file=sys.stderr)
                sys.exit(1)
            res = json.loads(out.stdout.strip().splitlines()[-1])
            for k, v in res.items():
                best[lib][k] = max(best[lib].get(k, 0.0), v)
            print("round", r, lib, {k: round(v) for k, v in res.items()}, file=sys.stderr, flush=True)
    table = {"unit": "GB/s of packed bytes read, best of %d rounds x %d launches" % (a.rounds, a.steps), "libs": best}
    txt = json.dumps(table, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    print(txt)


if __name__ == "__main__":
    main()
