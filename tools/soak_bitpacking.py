#!/usr/bin/env python3
"""One-off soak of the BITPACKING path: random columns (type, length, per-group value patterns, NULLs) compressed by
the oracle's restatement of the reference and by the device; block images must be byte-identical, the device decode
(full, ranged, point fetch) must return the original rows.
usage: python tools/soak_bitpacking.py [first_seed] [count]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import bitpacking as bp  # noqa: E402
import test_gpu_bitpacking as tb  # noqa: E402

adac = importlib.import_module("duckdb-adaptive-compression_amd")
adac.build()
ctx = adac.Context(0)
ALL = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]


def random_column(rng):
    dtype = np.dtype(ALL[int(rng.integers(0, len(ALL)))])
    info = np.iinfo(dtype)
    half = int(info.max) // 2
    n = int(rng.choice([1, 2047, 2048, 2049, int(rng.integers(1, 9000)), int(rng.integers(9000, 40000))]))
    parts, left = [], n
    while left > 0:
        m = min(left, int(rng.choice([2048, 2048, 4096, int(rng.integers(1, 5000))])))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            p = np.full(m, int(rng.integers(0, half + 1)), dtype=np.int64)
        elif kind == 1:
            step = int(rng.integers(0, 7))
            p = int(rng.integers(0, 100)) + (step * np.arange(m, dtype=np.int64)) % max(half - 100, 1)
        elif kind == 2:
            p = np.sort(half // 2 + np.cumsum(rng.integers(0, 4, size=m)) % (half // 4 + 1))
        elif kind == 3:
            span = int(rng.integers(1, max(2, 8 * dtype.itemsize - 2)))
            p = half // 3 + rng.integers(0, 1 << min(span, 40), size=m)
        elif kind == 4:
            lo = int(info.min) // 2 if dtype.kind == "i" else 0
            p = rng.integers(lo, half, size=m, dtype=np.int64)
        elif kind == 5 and dtype.kind == "i":
            p = -np.abs(rng.integers(0, half // 2 + 1, size=m, dtype=np.int64)) - 1
        else:
            p = np.where(rng.random(m) < 0.5, 0, int(rng.integers(0, half + 1))).astype(np.int64)
        parts.append(np.asarray(p, dtype=np.int64))
        left -= m
    return np.concatenate(parts)[:n].astype(dtype)


def one(seed):
    rng = np.random.default_rng(seed)
    v = random_column(rng)
    valid = None
    if rng.random() < 0.3:
        valid = rng.random(len(v)) > rng.random() * 0.7
    try:
        comp = bp.Compressed(v, valid, null_zero=valid is not None)
    except ValueError:
        plan, d_blocks, _ = tb.gpu_compress(adac, ctx, v, valid)
        assert not plan.encodable
        return
    plan, d_blocks, _ = tb.gpu_compress(adac, ctx, v, valid)
    tb.assert_blocks_equal_oracle(plan, d_blocks, comp)
    counts = np.array([plan.segment(i)[1] for i in range(plan.nseg)], dtype=np.uint32)
    lay = adac.BitpackingLayout(ctx, v.dtype, np.arange(plan.nseg, dtype=np.uint64) * plan.BLOCK_STRIDE, counts)
    d_out = ctx.alloc(len(v) * v.dtype.itemsize + 64)
    lay.unpack(d_blocks, d_out)
    got = d_out.download(v.dtype, len(v))
    ok = np.ones(len(v), bool) if valid is None else valid
    assert np.array_equal(got[ok], v[ok]), "full scan"
    seg = int(rng.integers(0, plan.nseg))
    c = int(counts[seg])
    row0 = int(counts[:seg].sum())
    for _ in range(4):
        s = int(rng.integers(0, c))
        k = int(rng.integers(1, c - s + 1))
        shift = int(rng.integers(0, 5))
        d_r = ctx.alloc((k + shift) * v.dtype.itemsize + 64)
        lay.unpack_range(d_blocks, seg, s, k, d_r, shift)
        r = d_r.download(v.dtype, k + shift)[shift:]
        m = ok[row0 + s:row0 + s + k]
        assert np.array_equal(r[m], v[row0 + s:row0 + s + k][m]), ("range", seg, s, k)
    kf = 32
    rows = rng.integers(0, c, size=kf).astype(np.uint32)
    d_f = ctx.alloc(kf * v.dtype.itemsize + 16)
    lay.fetch_rows(d_blocks, ctx.upload(np.full(kf, seg, dtype=np.uint32)), ctx.upload(rows), kf, d_f)
    f = d_f.download(v.dtype, kf)
    m = ok[row0 + rows]
    assert np.array_equal(f[m], v[row0 + rows][m]), "fetch"


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
bad = []
for seed in range(first, first + count):
    try:
        one(seed)
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
