#!/usr/bin/env python3
"""One-off concurrency stress of the host mirror: N host threads scan / fetch random ranges of the same segments
(ctypes releases the GIL, so the C++ paths really run in parallel) while the background policy thread flips
representations every few milliseconds and one thread keeps calling Compact / Uncompact by hand.  Every returned
row must equal the loaded value; the decoded-segment cache is on in half of the runs.
usage: python tools/soak_threads.py [seconds] [threads]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
adac = importlib.import_module("duckdb-adaptive-compression_amd")
adac.build()
host = importlib.import_module("duckdb-adaptive-compression_amd.host")

if os.environ.get("SOAK_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SOAK_SWITCH_INTERVAL"]))
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
errors = []
for run, cache in enumerate((0, 16 << 20)):
    rng = np.random.default_rng(100 + run)
    pools = int(os.environ.get("SOAK_POOLS", "1"))   # several segment pools (all on the one GPU of the box)
    db = host.Database([0] * pools, adaptive=True, arena_bytes=128 << 20, decoded_cache_bytes=cache)
    cols = []
    for i in range(24):
        dtype = (np.uint32, np.int64, np.uint16)[i % 3]
        n = int(rng.integers(3000, 30000))
        v = (np.int64(i * 1000) + rng.integers(0, 1 << (5 + i % 11), size=n)).astype(dtype)
        s = db.create_segment(dtype, start=i * 100000)
        for off in range(0, n, 2048):
            s.append(v, offset=off, count=min(2048, n - off))
        cols.append((s, v))
    stop = threading.Event()
    counts = [0] * (nthreads + 1)

    def reader(tid):
        r = np.random.default_rng(1000 + tid)
        try:
            while not stop.is_set():
                if r.random() < 0.15 and grow:
                    s, full, have = grow[int(r.integers(0, len(grow)))]
                    a = int(r.integers(0, have - 64))
                    if not np.array_equal(s.scan(a, 64), full[a:a + 64]):
                        errors.append((run, tid, "scan of a growing segment", a))
                        return
                    counts[tid] += 1
                    continue
                s, v = cols[int(r.integers(0, len(cols)))]
                if r.random() < 0.2:
                    row = int(r.integers(0, len(v)))
                    if s.fetch_row(row) != v[row]:
                        errors.append((run, tid, "fetch", row))
                        return
                else:
                    a = int(r.integers(0, len(v)))
                    k = int(r.integers(1, min(2048, len(v) - a) + 1))
                    if not np.array_equal(s.scan(a, k), v[a:a + k]):
                        errors.append((run, tid, "scan", a, k))
                        return
                counts[tid] += 1
        except Exception as e:  # noqa: BLE001
            errors.append((run, tid, repr(e)[:200]))

    def flipper():
        r = np.random.default_rng(7)
        try:
            while not stop.is_set():
                s, _ = cols[int(r.integers(0, len(cols)))]
                (s.compact if r.random() < 0.5 else s.uncompact)()
                counts[nthreads] += 1
        except Exception as e:  # noqa: BLE001
            errors.append((run, "flipper", repr(e)[:200]))

    # one writer: appends 2048-row vectors to segments of its own (also read by the readers) while they flip
    grow = []
    for i in range(4):
        dtype = (np.uint32, np.int64)[i % 2]
        full = (np.int64(77000 + i) + rng.integers(0, 1 << 9, size=60000)).astype(dtype)
        s = db.create_segment(dtype, start=10_000_000 + i * 100000)
        s.append(full, offset=0, count=2048)
        grow.append([s, full, 2048])

    def writer():
        try:
            k = 0
            while not stop.is_set():
                g = grow[k % len(grow)]
                k += 1
                s, full, have = g
                if have + 2048 > len(full) or have + 2048 > 262136 // full.dtype.itemsize:
                    continue
                a = s.append(full, offset=have, count=2048)
                if a != 2048:
                    errors.append((run, "writer", "short append", a))
                    return
                g[2] = have + 2048
                lo = int(np.random.default_rng(k).integers(0, g[2] - 100))
                if not np.array_equal(s.scan(lo, 100), full[lo:lo + 100]):
                    errors.append((run, "writer", "readback", lo))
                    return
                counts[nthreads] += 1
                time.sleep(0.0005)
        except Exception as e:  # noqa: BLE001
            errors.append((run, "writer", repr(e)[:200]))

    db.enable_background(3)
    threads = [threading.Thread(target=reader, args=(t,)) for t in range(nthreads)] + [threading.Thread(target=flipper),
                                                                                      threading.Thread(target=writer)]
    for t in threads:
        t.start()
    t_end = time.time() + seconds / 2
    while time.time() < t_end and not errors:
        time.sleep(0.5)
        print("run", run, "ops", sum(counts), "errors", len(errors), flush=True)
    stop.set()
    for t in threads:
        t.join()
    db.disable_background()
    for s, full, have in grow:
        if not np.array_equal(np.concatenate([s.scan(a, min(2048, have - a)) for a in range(0, have, 2048)]), full[:have]):
            errors.append((run, "final readback of a grown segment"))
    for s, v in cols:  # final state must still read back
        if not np.array_equal(np.concatenate([s.scan(a, min(2048, len(v) - a)) for a in range(0, len(v), 2048)]), v):
            errors.append((run, "final readback"))
    db.close()
print("done: errors", errors[:5])
sys.exit(1 if errors else 0)
