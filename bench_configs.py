#!/usr/bin/env python3
"""Secondary harness (NOT the driver's bench.py): the reference's other workloads through the drop-in path,
one JSON object on stdout.  Numbers go to profiles/ and DESIGN.md.

  plugin_scan  benchmark/micro/succinct/sequential.cpp (SELECT * full scans): a column loaded and scanned through
               the C++ host mirror in ColumnData::ScanVector's call pattern (2048-row Scan calls that return
               rows to HOST memory), with and without the decoded-segment cache — the PCIe-inclusive path.
  adaptive     zipf_over_time.cpp / zipf_distribution_diff_skews.cpp: segment-access traces with Zipf skew
               0.5 / 1.0 / 2.0, one policy round per period: resident bytes, flips and re-encode rate.
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "duckdb-adaptive-compression_amd"


def load(db, lay, dtype, values):
    dtype = np.dtype(dtype)
    row = 0
    for count, cap in lay.appender_segments(len(values), dtype.itemsize):
        s = db.create_segment(dtype, start=row, segment_size=cap * dtype.itemsize)
        v = values[row:row + count]
        for off in range(0, count, 2048):
            s.append(v, offset=off, count=min(2048, count - off))
        row += count


PCIE_GBS = 63.0  # PCIe Gen5 x16, one direction (spec)


def plugin_scan(host, lay, n=int(os.environ.get("PLUGIN_SCAN_ROWS", 200_000_000))):
    values = np.arange(n, dtype=np.uint32)  # SuccinctSequentialInsert / C1 column
    out = {"rows": n, "dtype": "u32", "vector_size": 2048, "pcie_spec_GBps": PCIE_GBS, "variants": {}}
    ncpu = len(os.sched_getaffinity(0))
    variants = [("per_vector_device_decode", dict(decoded_cache_bytes=0), (1,), min(n, 4_000_000)),
                ("decoded_segment_cache", dict(decoded_cache_bytes=192 << 20, scan_lanes=8, prefetch_segments=8),
                 (1, 2, 4, 8, 16), n),
                ("decoded_segment_cache_2_pools_on_one_gpu",
                 dict(device=[0, 0], decoded_cache_bytes=96 << 20, scan_lanes=4, prefetch_segments=8), (1, 8), n)]
    for name, kw, thread_counts, rows_here in variants:
        kw = dict(kw)
        device = kw.pop("device", 0)
        db = host.Database(device, arena_bytes=1 << 30, **kw)
        t0 = time.perf_counter()
        load(db, lay, np.uint32, values[:rows_here])
        db.compact_all()
        t_load = time.perf_counter() - t0
        rec = {"rows": rows_here, "load_and_compact_s": t_load, "total_data_size": db.total_data_size, "scans": {}}
        for th in thread_counts:
            if th > ncpu:
                continue
            # cold: the cache (192 MiB) is smaller than the column (800 MB) and LRU, so every pass decodes and copies
            # every segment again; warm is measured on a prefix that fits
            cs, sec_cold, rows = db.full_scan(threads=th)
            assert cs == rows_here * (rows_here - 1) // 2
            cs, sec_cold2, rows = db.full_scan(threads=th)
            sec = min(sec_cold, sec_cold2)
            rec["scans"]["threads_%d" % th] = {
                "cold_scan_rows_per_s": rows / sec, "d2h_GBps": rows * 4 / sec / 1e9,
                "fraction_of_pcie_spec": rows * 4 / sec / 1e9 / PCIE_GBS}
        if kw.get("decoded_cache_bytes"):
            fit = [s for s in db.segments][:400]   # ~100 MB: stays resident
            db.full_scan(fit)
            for th in (1, 8):
                cs, sec_warm, rows = db.full_scan(fit, threads=th)
                rec["scans"]["warm_threads_%d" % th] = {"rows_per_s": rows / sec_warm}
        rec["cache"] = db.cache_stats()
        rec["prefetch"] = db.prefetch_stats()
        out["variants"][name] = rec
        db.close()
    return out


def adaptive(host, wl, nseg=400, rows=32767, periods=4):
    rng = np.random.default_rng(3)
    out = {"segments": nseg, "rows_per_segment": rows, "skews": {}}
    data = [((i << 34) + rng.integers(0, 1 << (10 + i % 12), size=rows)).astype(np.uint64) for i in range(nseg)]
    for skew, cache in ((0.5, 0), (1.0, 0), (2.0, 0), (0.5, 128 << 20)):
        # the last variant keeps decoded images of the packed segments it touches in the page-locked cache: a
        # look-up into a packed segment is then a memcpy, not a device round trip
        db = host.Database(0, adaptive=True, arena_bytes=512 << 20, decoded_cache_bytes=cache, prefetch_segments=1)
        db.reserve_staging(nseg * rows * 8 + (1 << 20))   # one-time page-locking kept out of the first policy round
        for i in range(nseg):
            s = db.create_segment(np.uint64, start=i * rows)
            for off in range(0, rows, 2048):
                s.append(data[i], offset=off, count=min(2048, rows - off))
        rec = {"raw_bytes": db.total_data_size, "periods": []}
        for p in range(periods):
            trace = wl.zipf_column(20000, np.uint32, domain=nseg, skew=skew, seed=500 + p, threads=1) - 1
            t0 = time.perf_counter()
            scanned = 0
            for i in trace:
                db.segments[int(i)].scan(0, 2048)
                scanned += 2048
            t_scan = time.perf_counter() - t0
            before = [s.compacted for s in db.segments]
            t0 = time.perf_counter()
            db.policy_step(0.90)
            t_policy = time.perf_counter() - t0
            after = [s.compacted for s in db.segments]
            packed = sum(1 for a, b in zip(before, after) if b and not a)
            expanded = sum(1 for a, b in zip(before, after) if a and not b)
            rec["periods"].append({
                "lookups_per_s": len(trace) / t_scan, "scanned_rows_per_s": scanned / t_scan,
                "resident_bytes": db.total_data_size, "arena_bytes": db.arena_used_bytes,
                "policy_step_s": t_policy, "segments_packed": packed, "segments_expanded": expanded,
                "reencode_raw_GBps": (packed + expanded) * rows * 8 / t_policy / 1e9 if t_policy > 0 else None,
            })
        out["skews"][str(skew) + ("_decoded_cache" if cache else "")] = rec
        db.close()
    return out


def bitpacking_scan(adac, n=50_000_000):
    """Full scan of on-disk BITPACKING segments: blocks written by the oracle's restatement of the
    reference's compress (CPU, untimed), decoded on the device; rates from HIP events on the codec stream."""
    from oracle import bitpacking as bp
    ctx = adac.Context(0)
    rng = np.random.default_rng(11)
    out = {"rows": n, "cases": []}
    cols = {
        "for_u32_w20": (5_000_000 + rng.integers(0, 1 << 20, size=n)).astype(np.uint32),
        "delta_for_i64_sorted": (10 ** 12 + np.cumsum(rng.integers(0, 1 << 9, size=n))).astype(np.int64),
        "for_u64_w32": (rng.integers(0, 1 << 32, size=n, dtype=np.uint64) + np.uint64(1 << 40)).astype(np.uint64),
    }
    stride = 262144
    for name, v in cols.items():
        t0 = time.perf_counter()
        comp = bp.Compressed(v)
        t_cpu = time.perf_counter() - t0
        nseg = comp.nseg
        buf = np.zeros(nseg * stride + 64, dtype=np.uint8)
        counts = np.zeros(nseg, dtype=np.uint32)
        used = 0
        for i in range(nseg):
            buf[i * stride:i * stride + bp.BLOCK_SIZE] = comp.block(i)
            counts[i] = comp.count(i)
            used += comp.size(i)
        d_blocks = ctx.upload(buf)
        lay = adac.BitpackingLayout(ctx, v.dtype, np.arange(nseg, dtype=np.uint64) * stride, counts)
        d_out = ctx.alloc(n * v.dtype.itemsize + 64)
        lay.unpack(d_blocks, d_out)
        ctx.sync()
        assert np.array_equal(d_out.download(v.dtype, n), v)
        ctx.timer_start()
        reps = 20
        for _ in range(reps):
            lay.unpack(d_blocks, d_out)
        ms = ctx.timer_stop() / reps
        # GPU compress of the same column: statistics + host decisions + group writes (wall time, host included)
        d_vals = ctx.upload(v)
        t0 = time.perf_counter()
        plan = adac.BitpackingPlan(ctx, v.dtype, d_vals, n)
        d_new = ctx.alloc(plan.nseg * plan.BLOCK_STRIDE + 64)
        plan.write(d_vals, d_new)
        ctx.sync()
        t_gpu_compress = time.perf_counter() - t0
        assert plan.nseg == nseg and plan.groups_by_mode() == comp.groups_by_mode()
        img = d_new.download(np.uint8, nseg * plan.BLOCK_STRIDE)
        assert all(np.array_equal(img[i * stride:i * stride + comp.size(i)], comp.block(i)[:comp.size(i)])
                   for i in range(0, nseg, max(1, nseg // 16)))
        del d_new, d_vals, plan
        out["cases"].append({
            "gpu_compress_values_per_s": n / t_gpu_compress, "gpu_compress_s": t_gpu_compress,
            "name": name, "dtype": str(v.dtype), "segments": nseg, "compressed_bytes": used,
            "modes": comp.groups_by_mode(), "cpu_compress_values_per_s": n / t_cpu, "decode_ms": ms,
            "decode_values_per_s": n / (ms * 1e-3),
            "algorithmic_GBps": (used + n * v.dtype.itemsize) / (ms * 1e-3) / 1e9,
        })
        del lay, d_blocks, d_out, comp
    ctx.close()
    return out


def q6_packed(adac, n=59_986_052):
    """C3's Q6 shape on packed columns: WHERE l_shipdate in a year AND l_discount BETWEEN
    5 AND 7 AND l_quantity < 24 -> SUM(l_extendedprice), on four int32 columns of TPC-H SF10 size that share
    their segment layout.  Three filter scans chain their selection bitmaps, the fourth scan aggregates under the
    final bitmap; nothing is decoded to HBM.  (Q6 proper sums price * discount — a two-column product is outside
    this codec's single-column scans.)  Beside it: the materialising plan (decode the four columns, then filter)
    counted at its decode cost alone."""
    ctx = adac.Context(0)
    rng = np.random.default_rng(1994)
    cols = {"l_shipdate": rng.integers(8036, 10562, size=n).astype(np.int32),       # days since 1970: 1992..1998
            "l_discount": rng.integers(0, 11, size=n).astype(np.int32),               # percent
            "l_quantity": rng.integers(1, 51, size=n).astype(np.int32),
            "l_extendedprice": rng.integers(90_000, 10_495_000, size=n).astype(np.int32)}  # cents
    counts = adac.appender_segment_counts(n, 4)
    enc, packed_bytes = {}, 0
    for name, v in cols.items():
        lay = adac.Layout(ctx, np.int32, counts)
        d_vals = ctx.upload(v)
        d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
        lay.encode(d_vals, d_words)
        ctx.sync()
        descs = lay.get_descs()
        packed_bytes += int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum())
        enc[name] = (lay, d_words, sorted(set(descs["width"].tolist())))
        del d_vals
    nw = (n + 63) // 64
    bm = [ctx.alloc(nw * 8 + 8) for _ in range(3)]
    d_cnt = ctx.alloc(len(counts) * 8)
    d_sum = ctx.alloc(len(counts) * 8)
    int_min = int(np.array([np.iinfo(np.int32).min]).view(np.uint32)[0])

    def q6():
        lay, w, _ = enc["l_shipdate"]
        lay.scan_select_between(w, 8766, 9130, bm[0], d_cnt)              # 1994-01-01 .. 1994-12-31
        lay, w, _ = enc["l_discount"]
        lay.scan_select_between(w, 5, 7, bm[1], d_cnt, bm[0])
        lay, w, _ = enc["l_quantity"]
        lay.scan_select_between(w, int_min, 23, bm[2], d_cnt, bm[1])
        lay, w, _ = enc["l_extendedprice"]
        lay.scan_sum(w, d_sum, bm[2])

    q6()
    ctx.sync()
    m = ((cols["l_shipdate"] >= 8766) & (cols["l_shipdate"] <= 9130) & (cols["l_discount"] >= 5) &
         (cols["l_discount"] <= 7) & (cols["l_quantity"] < 24))
    got = int(d_sum.download(np.uint64, len(counts)).sum(dtype=np.uint64))
    assert got == int(cols["l_extendedprice"][m].astype(np.int64).sum()), "Q6 parity"
    assert int(d_cnt.download(np.uint64, len(counts)).sum()) == int(m.sum())
    reps = 20
    by_group = {}
    for group in (2, 4, 8, 16):
        adac.set_tuning("scan_tiles_per_wg", group)
        q6()
        ctx.timer_start()
        for _ in range(reps):
            q6()
        by_group[group] = ctx.timer_stop() / reps
    adac.set_tuning("scan_tiles_per_wg", 0)
    q6()
    ctx.timer_start()
    for _ in range(reps):
        q6()
    ms = ctx.timer_stop() / reps
    steps = {}
    for name, fn in (("select_shipdate", lambda: enc["l_shipdate"][0].scan_select_between(enc["l_shipdate"][1], 8766, 9130, bm[0], d_cnt)),
                     ("select_discount_masked", lambda: enc["l_discount"][0].scan_select_between(enc["l_discount"][1], 5, 7, bm[1], d_cnt, bm[0])),
                     ("sum_price_masked", lambda: enc["l_extendedprice"][0].scan_sum(enc["l_extendedprice"][1], d_sum, bm[2])),
                     ("count_shipdate", lambda: enc["l_shipdate"][0].scan_count_between(enc["l_shipdate"][1], 8766, 9130, d_cnt))):
        fn()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        steps[name] = ctx.timer_stop() / reps
    # filter then project: only the surviving rows of l_extendedprice are decoded
    d_sel = ctx.alloc(int(m.sum()) * 4 + 64)
    lay, w, _ = enc["l_extendedprice"]
    got = lay.unpack_selected(w, bm[2], d_sel)
    assert got == int(m.sum()) and np.array_equal(d_sel.download(np.int32, got), cols["l_extendedprice"][m])
    ctx.timer_start()
    for _ in range(reps):
        lay.unpack_selected(w, bm[2], d_sel, None, False)
    steps["project_price_selected"] = ctx.timer_stop() / reps
    d_out = ctx.alloc(n * 4 + 64)
    ctx.timer_start()
    for _ in range(reps):
        for name in cols:
            lay, w, _ = enc[name]
            lay.unpack(w, d_out)
    ms_dec = ctx.timer_stop() / reps
    out = {"rows": n, "selected_rows": int(m.sum()), "widths": {k: v[2] for k, v in enc.items()},
           "packed_bytes": packed_bytes, "q6_on_packed_ms": ms, "q6_rows_per_s": n / (ms * 1e-3),
           "q6_packed_read_GBps": packed_bytes / (ms * 1e-3) / 1e9,
           "decode_four_columns_ms": ms_dec,
           "decode_four_columns_total_GBps": (packed_bytes + 4 * n * 4) / (ms_dec * 1e-3) / 1e9, "q6_ms_by_scan_tiles_per_wg": by_group, "step_ms": steps,
           "note": "q6_on_packed = 3 chained filter scans (selection bitmaps) + 1 masked SUM; decode_four_columns is "
                   "only the materialisation a decode-then-filter plan would pay before filtering"}
    ctx.close()
    return out


def q1_packed(adac, n=59_986_052):
    """C3's Q1 shape on packed columns ("Q1 = group-by sum"; benchmark log TPCH_runtime.txt:2-6):
    SUM(l_quantity), SUM(l_extendedprice), SUM(l_partkey), COUNT(*) GROUP BY (l_returnflag, l_linestatus) on int32
    columns of TPC-H SF10 size that share their segment layout; the group code is a 6-valued uint8 column (the
    engine's dictionary code of the two flags).  One adac_scan_group_sum per aggregated column, nothing decoded to
    HBM; checked against numpy's GROUP BY.  Beside it: decoding the same columns (what the reference's engine needs
    before its hash aggregate can start)."""
    ctx = adac.Context(0)
    rng = np.random.default_rng(1992)
    code = rng.choice(6, size=n, p=[.2466, .2534, .0004, .2500, .2490, .0006]).astype(np.uint8)
    cols = {"l_quantity": rng.integers(1, 51, size=n).astype(np.int32),
            "l_extendedprice": rng.integers(90_000, 10_495_000, size=n).astype(np.int32),
            "l_partkey": rng.integers(1, 2_000_001, size=n).astype(np.int32)}
    counts = adac.appender_segment_counts(n, 4)

    def enc_col(v):
        lay = adac.Layout(ctx, v.dtype, counts)
        d_vals = ctx.upload(v)
        d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
        lay.encode(d_vals, d_words)
        ctx.sync()
        descs = lay.get_descs()
        nbytes = int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum())
        return lay, d_words, nbytes, sorted(set(descs["width"].tolist()))

    klay, kwords, kbytes, kwidths = enc_col(code)
    d_sums = ctx.alloc(7 * 8)
    d_cnts = ctx.alloc(7 * 8)
    out = {"rows": n, "groups": 6, "key_widths": kwidths, "key_packed_bytes": kbytes, "columns": []}
    total_ms, total_bytes = 0.0, 0
    reps = 20
    for name, v in cols.items():
        lay, words, nbytes, widths = enc_col(v)
        lay.scan_group_sum(words, klay, kwords, 6, d_sums, d_cnts)
        ctx.sync()
        got_s = d_sums.download(np.uint64, 7).tolist()
        got_c = d_cnts.download(np.uint64, 7).tolist()
        for g in range(6):
            m = code == g
            assert got_c[g] == int(m.sum()) and got_s[g] == int(v[m].astype(np.int64).sum()), "Q1 parity"
        assert got_c[6] == 0
        ctx.timer_start()
        for _ in range(reps):
            lay.scan_group_sum(words, klay, kwords, 6, d_sums, d_cnts)
        ms = ctx.timer_stop() / reps
        # the same aggregate through the staged-LDS kernel alone (round 2's form), for the record
        adac.set_tuning("group_sum_rw", 0)
        lay.scan_group_sum(words, klay, kwords, 6, d_sums, d_cnts)
        ctx.timer_start()
        for _ in range(reps):
            lay.scan_group_sum(words, klay, kwords, 6, d_sums, d_cnts)
        ms_lds = ctx.timer_stop() / reps
        assert d_sums.download(np.uint64, 7).tolist() == got_s and d_cnts.download(np.uint64, 7).tolist() == got_c
        adac.set_tuning("group_sum_rw", 1)
        d_out = ctx.alloc(n * 4 + 64)
        lay.unpack(words, d_out)
        ctx.timer_start()
        for _ in range(reps):
            lay.unpack(words, d_out)
        ms_dec = ctx.timer_stop() / reps
        del d_out
        total_ms += ms
        total_bytes += nbytes + kbytes
        out["columns"].append({"column": name, "widths": widths, "packed_bytes": nbytes, "group_sum_ms": ms,
                               "rows_per_s": n / (ms * 1e-3), "packed_read_GBps": (nbytes + kbytes) / (ms * 1e-3) / 1e9,
                               "staged_lds_kernel_ms": ms_lds, "decode_only_ms": ms_dec})
        del lay, words
    out["q1_three_aggregates_ms"] = total_ms
    out["q1_rows_per_s"] = n / (total_ms * 1e-3)
    out["q1_packed_read_GBps"] = total_bytes / (total_ms * 1e-3) / 1e9
    out["note"] = ("one grouped scan per aggregated column over (value, group code); per-thread LDS bins (7 bins), one "
                   "partial per workgroup, k_group_final adds them; the reference's engine decodes every column first "
                   "(decode_only_ms per column) and then hashes 60 M rows on the CPU")
    ctx.close()
    return out


def c1_lookups(adac, wl, n=10_000_000, nlookups=10_000):
    """C1 (benchmark/micro/succinct/zipf_distribution.cpp:13-48): t1(i UINTEGER) with i = 0..N-1, compacted, then
    `SELECT i FROM t1 WHERE i == k` for Zipf(N, 1.0) keys (mt19937, seed 42).  Each look-up is one fused
    COUNT(== k) over the packed column: segments whose [min, min + 2^w) cannot hold k are skipped by the kernel
    (one 64-byte record read each), so a look-up costs a launch plus one segment's scan."""
    ctx = adac.Context(0)
    vals = np.arange(n, dtype=np.uint32)
    counts = adac.appender_segment_counts(n, 4)
    lay = adac.Layout(ctx, np.uint32, counts)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
    lay.encode(ctx.upload(vals), d_words)
    ctx.sync()
    descs = lay.get_descs()
    keys = (wl.zipf_column(nlookups, np.uint32, domain=n, skew=1.0, seed=42, threads=1).astype(np.int64) - 1) % n
    d_cnt = ctx.alloc(len(counts) * 8)
    lay.scan_count_eq(d_words, int(keys[0]), d_cnt)
    ctx.sync()
    t0 = time.perf_counter()
    for k in keys:
        lay.scan_count_eq(d_words, int(k), d_cnt)
    ctx.sync()
    wall = time.perf_counter() - t0
    hits = int(d_cnt.download(np.uint64, len(counts)).sum())
    assert hits == 1
    ctx.timer_start()
    for k in keys[:2000]:
        lay.scan_count_eq(d_words, int(k), d_cnt)
    dev_ms = ctx.timer_stop() / 2000
    out = {"rows": n, "segments": int(len(counts)), "max_width": int(descs["width"].max()),
           "packed_bytes": int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum()),
           "lookups": nlookups, "lookups_per_s_wall": nlookups / wall, "device_us_per_lookup": dev_ms * 1e3,
           "note": "one fused COUNT(== k) launch per look-up, results stay on the device; wall = Python loop included"}
    ctx.close()
    return out


def main():
    adac = importlib.import_module(PKG)
    adac.build()
    host = importlib.import_module(PKG + ".host")
    lay = importlib.import_module(PKG + ".layout")
    wl = importlib.import_module(PKG + ".workload")
    only = sys.argv[1:]
    jobs = {"plugin_scan": lambda: plugin_scan(host, lay), "adaptive": lambda: adaptive(host, wl),
            "bitpacking_scan": lambda: bitpacking_scan(adac), "q6_packed": lambda: q6_packed(adac), "q1_packed": lambda: q1_packed(adac),
            "c1_lookups": lambda: c1_lookups(adac, wl)}
    res = {k: f() for k, f in jobs.items() if not only or k in only}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
