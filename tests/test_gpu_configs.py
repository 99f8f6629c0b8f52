"""BASELINE.json's other configurations as parity cases at their full sizes, checked through size-independent
properties (round trips, checksums, known end-to-end sizes) because the oracle would take minutes there.
  C1  10 M uint32 sequential + 10 000 Zipf(1.0) point look-ups   (benchmark/micro/succinct/zipf_distribution.cpp)
  C3  TPC-H SF10 lineitem integer columns (l_orderkey / l_partkey / l_quantity as INTEGER)
  C5  adaptive hot/cold re-compaction under a Zipf segment-access trace, s in {0.5, 1.0, 2.0}
(C2 and C4 are bench.py's workload: its full-size round trip + checksum run inside every bench invocation.)"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods(adac):
    return (importlib.import_module(adac.__name__ + ".workload"), importlib.import_module(adac.__name__ + ".host"))


def test_c1_sequential_column_and_zipf_point_lookups(adac, golden, gpu_ctx):
    wl, host = _mods(adac)
    e5 = next(x for x in golden["end_to_end"] if x["id"] == "E5")
    n = e5["n"]
    vals = np.arange(n, dtype=np.uint32)
    counts = adac.appender_segment_counts(n, 4)
    assert len(counts) == e5["segments"]
    lay = adac.Layout(gpu_ctx, np.uint32, counts)
    d_vals = gpu_ctx.upload(vals)
    d_words = gpu_ctx.alloc(lay.max_arena_words * 8).zero()
    lay.encode(d_vals, d_words)
    descs = lay.get_descs()
    # GetTotalDataSize of the real reference after CompactAllSegments: 19 841 948 B (SURVEY.md §8d / BASELINE.md §2)
    assert sum(adac.size_in_bytes(int(d["count"]), int(d["width"])) for d in descs) == e5["total_data_size"]
    assert int(descs["width"].max()) <= 16
    # 10 000 look-ups `SELECT i FROM t1 WHERE i == k`, k ~ Zipf(n, 1.0), mt19937 seed 42
    keys = wl.zipf_column(10000, np.uint32, domain=n, skew=1.0, seed=42, threads=1)
    keys = np.minimum(keys, n - 1)
    # zonemap step (RowGroup::CheckZonemapSegments): the one segment whose [min, max] holds k, then its row
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    seg_min = descs["min"]
    segs = (np.searchsorted(seg_min, keys, side="right") - 1).astype(np.uint32)
    rows = (keys - starts[segs]).astype(np.uint32)
    d_out = gpu_ctx.alloc(len(keys) * 4)
    lay.fetch_rows(d_words, gpu_ctx.upload(segs), gpu_ctx.upload(rows), len(keys), d_out)
    assert np.array_equal(d_out.download(np.uint32, len(keys)), keys)
    # the same predicate as a fused scan over the whole column (no zonemap): exactly one hit, in that segment
    d_cnt = gpu_ctx.alloc(len(counts) * 8)
    for k, s in list(zip(keys[:40].tolist(), segs[:40].tolist())) + [(0, 0), (n - 1, len(counts) - 1)]:
        lay.scan_count_eq(d_words, k, d_cnt)
        c = d_cnt.download(np.uint64, len(counts))
        assert int(c.sum()) == 1 and int(c[s]) == 1
    # full scan round trip + checksum
    d_dec = gpu_ctx.alloc(n * 4)
    lay.unpack(d_words, d_dec)
    assert np.array_equal(d_dec.download(np.uint32, n), vals)
    lay.scan_sum(d_words, d_cnt)
    assert int(d_cnt.download(np.uint64, len(counts)).sum(dtype=np.uint64)) == n * (n - 1) // 2


def tpch_lineitem_int_columns(sf=10, seed=7):
    """Deterministic stand-in for dbgen's lineitem integer columns (extension/tpch/dbgen/dbgen.cpp:406-412):
    orders carry 1..7 line items; order keys are sparse (8 used of every 32); l_partkey uniform in
    [1, 200 000*SF]; l_quantity uniform in [1, 50]."""
    rng = np.random.default_rng(seed)
    n_orders = 1_500_000 * sf
    per = rng.integers(1, 8, size=n_orders)
    order_idx = np.arange(n_orders, dtype=np.int64)
    okey = (order_idx // 8) * 32 + order_idx % 8 + 1
    l_orderkey = np.repeat(okey, per).astype(np.int32)
    n = len(l_orderkey)
    l_partkey = rng.integers(1, 200_000 * sf + 1, size=n).astype(np.int32)
    l_quantity = rng.integers(1, 51, size=n).astype(np.int32)
    return l_orderkey, l_partkey, l_quantity


def test_c3_tpch_sf10_lineitem_integer_columns(adac, gpu_ctx):
    cols = dict(zip(("l_orderkey", "l_partkey", "l_quantity"), tpch_lineitem_int_columns(10)))
    n = len(cols["l_quantity"])
    assert 59_000_000 < n < 61_000_000
    counts = adac.appender_segment_counts(n, 4)
    expect_w = {"l_orderkey": (15, 20), "l_partkey": (21, 21), "l_quantity": (6, 6)}
    for name, v in cols.items():
        lay = adac.Layout(gpu_ctx, np.int32, counts)
        d_vals = gpu_ctx.upload(v)
        d_words = gpu_ctx.alloc(lay.max_arena_words * 8).zero()
        lay.encode(d_vals, d_words)
        descs = lay.get_descs()
        big = descs["count"] > 30000
        lo, hi = expect_w[name]
        assert lo <= int(descs["width"][big].min()) and int(descs["width"][big].max()) <= hi, name
        d_out = gpu_ctx.alloc(n * 4)
        lay.unpack(d_words, d_out)
        assert np.array_equal(d_out.download(np.int32, n), v), name
        d_res = gpu_ctx.alloc(len(counts) * 8)
        lay.scan_sum(d_words, d_res)  # Q1/Q6-style SUM over the packed column
        assert int(d_res.download(np.uint64, len(counts)).sum(dtype=np.uint64)) == int(v.astype(np.int64).sum()) & 0xFFFFFFFFFFFFFFFF
        if name == "l_quantity":      # Q6-style predicate count on the packed column
            lay.scan_count_eq(d_words, 24, d_res)
            assert int(d_res.download(np.uint64, len(counts)).sum()) == int((v == 24).sum())
        del lay, d_vals, d_words, d_out, d_res


@pytest.mark.parametrize("skew", [0.5, 1.0, 2.0])
def test_c5_adaptive_recompaction_under_zipf_trace(adac, skew):
    wl, host = _mods(adac)
    rng = np.random.default_rng(int(skew * 10))
    nseg, rows = 120, 32767
    db = host.Database(0, adaptive=True, arena_bytes=256 << 20)
    try:
        segs, data = [], []
        for i in range(nseg):
            v = ((i << 34) + rng.integers(0, 1 << (10 + i % 12), size=rows)).astype(np.uint64)
            s = db.create_segment(np.uint64, start=i * rows)
            for off in range(0, rows, 2048):
                s.append(v, offset=off, count=min(2048, rows - off))
            segs.append(s)
            data.append(v)
        raw_bytes = db.total_data_size
        assert raw_bytes == nseg * 262136
        sizes = []
        for period in range(3):
            # segment-access trace: ids ~ Zipf(nseg, skew) (the reference draws look-up KEYS from Zipf,
            # zipf_over_time.cpp:32-44; the hot segments are the ones holding small keys)
            trace = wl.zipf_column(4000, np.uint32, domain=nseg, skew=skew, seed=100 + period, threads=1) - 1
            hits = np.bincount(trace, minlength=nseg)
            for i in np.nonzero(hits)[0]:
                for _ in range(int(min(hits[i], 40))):  # reads are counted per scan call
                    r = int(rng.integers(0, rows - 64))
                    assert np.array_equal(segs[i].scan(r, 64), data[i][r:r + 64])
            reads = np.minimum(hits, 40)
            db.policy_step(0.90)
            order = sorted(range(nseg), key=lambda i: (int(reads[i]), i))
            # `cum_sum / curr_counter < compression_rate` is float / idx_t compared as double (column_segment_catalog.cpp:92-95)
            expect_hot = sorted(i for rank, i in enumerate(order)
                                if not (float(np.float32(rank + 1) / np.float32(nseg)) < 0.90))
            hot = [i for i, s in enumerate(segs) if not s.compacted]
            assert hot == expect_hot
            assert len(hot) == 12  # ranks 109..120 of 120: 108/120 is 0.9f = 0.89999998 < 0.90
            sizes.append(db.total_data_size)
            assert sizes[-1] < 0.45 * raw_bytes  # 90 % of the segments at 10..21 bits instead of 64
        for i in (0, 1, nseg // 2, nseg - 1):
            assert np.array_equal(segs[i].scan(0, rows), data[i])
        assert db.arena_used_bytes > 0
    finally:
        db.close()
