"""The C++ host mirror (ColumnSegment / CompressionFunction / ColumnSegmentCatalog over one GPU pool) against
the oracle's segment model, driven the way the reference's bulk load, scans and adaptive policy drive it."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host(adac):
    adac.build()
    return importlib.import_module(adac.__name__ + ".host")


def lay_mod(adac):
    return importlib.import_module(adac.__name__ + ".layout")


def load_column(db, oracle, adac, dtype, values, padded=False, adaptive=False, validity_prob=None, rng=None):
    """Appender-style bulk load into mirror segments and oracle segments side by side."""
    dtype = np.dtype(dtype)
    segs, orcs = [], []
    row = 0
    for count, cap in lay_mod(adac).appender_segments(len(values), dtype.itemsize):
        s = db.create_segment(dtype, start=row, segment_size=cap * dtype.itemsize)
        o = oracle.Segment(dtype, segment_size=cap * dtype.itemsize, adaptive=adaptive, padded=padded, store_min=True)
        v = values[row:row + count]
        off = 0
        while off < count:
            c = min(2048, count - off)
            validity = None
            if validity_prob is not None:
                bits = np.zeros(len(v) + 64, dtype=bool)
                bits[:len(v)] = rng.random(len(v)) > validity_prob
                validity = np.packbits(bits, bitorder="little")
                validity = np.concatenate([validity, np.zeros((-len(validity)) % 8, np.uint8)]).view(np.uint64)
            a = s.append(v, validity, offset=off, count=c)
            b = o.append(v, validity, offset=off, count=c)
            assert a == b == c
            off += c
        segs.append(s)
        orcs.append(o)
        row += count
    return segs, orcs


def assert_same_state(segs, orcs):
    for i, (s, o) in enumerate(zip(segs, orcs)):
        assert s.count == o.count, i
        assert s.compacted == o.compacted, i
        assert s.function == o.function, i
        assert s.data_size == o.data_size, i
        if s.compacted:
            assert s.width == o.width, i
            if o.width < 8 * s.dtype.itemsize:
                assert s.min_factor == o.min_factor, i


@pytest.mark.parametrize("eid", ["E1", "E2", "E3", "E4"])
def test_bulk_load_reproduces_reference_data_sizes(adac, oracle, host, golden, eid):
    """GetTotalDataSize of the real reference binary through the mirror."""
    e = next(x for x in golden["end_to_end"] if x["id"] == eid)
    dtype = np.dtype(e["dtype"])
    values = (e["base"] + np.arange(e["n"], dtype=np.int64)).astype(dtype)
    db = host.Database(0, padded=e["padded"], arena_bytes=64 << 20)
    try:
        segs, orcs = load_column(db, oracle, adac, dtype, values, padded=e["padded"])
        assert db.num_segments == len(segs)
        if "initial_data_size" in e:
            assert db.total_data_size == e["initial_data_size"]
        assert_same_state(segs, orcs)
        db.compact_all()  # CompactAllSegments (zipf_distribution.cpp:38-39)
        for o in orcs:
            o.compact()
        assert db.total_data_size == e["total_data_size"]
        assert_same_state(segs, orcs)
        # full scan, 2048-row vectors with a ragged tail per segment (ColumnData::ScanVector call pattern)
        row = 0
        for s in segs:
            n = s.count
            got = np.concatenate([s.scan(r, min(2048, n - r)) for r in range(0, n, 2048)])
            assert np.array_equal(got, values[row:row + n])
            row += n
        assert db.arena_used_bytes > 0
    finally:
        db.close()


def test_lazy_compaction_on_first_scan_and_fetch(adac, oracle, host):
    rng = np.random.default_rng(2)
    db = host.Database(0, arena_bytes=16 << 20)
    try:
        s = db.create_segment(np.uint32, start=1000, segment_size=262136)
        o = oracle.Segment(np.uint32, segment_size=262136, store_min=True)
        v = (70000 + rng.integers(0, 1 << 13, size=30000)).astype(np.uint32)
        v[7], v[8] = 70000, 70000 + (1 << 13) - 1
        for off in range(0, len(v), 2048):
            c = min(2048, len(v) - off)
            assert s.append(v, offset=off, count=c) == o.append(v, offset=off, count=c)
        assert not s.compacted and s.data_size == o.data_size == 9 + 8 * ((65534 * 32 + 63) // 64)
        assert np.array_equal(s.scan(123, 2048), v[123:123 + 2048])   # first scan compacts (column_segment.cpp:157)
        o.scan(123, 2048)
        assert s.compacted and s.width == o.width == 13 and s.min_factor == o.min_factor
        assert s.data_size == o.data_size
        for r in (0, 1, 29999, 4242):
            assert s.fetch_row(r) == v[r] == o.fetch_row(r)
        # scan_partial into the middle of a result vector
        res = np.full(3000, 7, dtype=np.uint32)
        s.scan(29000, 1000, result=res, result_offset=500)
        assert np.array_equal(res[500:1500], v[29000:30000]) and res[499] == 7 and res[1500] == 7
        with pytest.raises(host.HostError):
            s.scan(29990, 100)
    finally:
        db.close()


def test_append_after_compaction_uncompacts_and_recompacts(adac, oracle, host):
    """column_segment.cpp:254-268: an append to a bit-compressed segment uncompacts it, appends through the
    uncompressed function and recompacts through BitCompressFromUncompressed."""
    db = host.Database(0, arena_bytes=16 << 20)
    try:
        for dtype in (np.uint32, np.int32, np.uint64):
            s = db.create_segment(dtype, segment_size=8192 * np.dtype(dtype).itemsize)
            o = oracle.Segment(dtype, segment_size=8192 * np.dtype(dtype).itemsize, store_min=True)
            a = (5000 + np.arange(3000)).astype(dtype)
            b = (4000 + np.arange(2048) * 3).astype(dtype)
            for vals in (a[:2048], a[2048:]):
                assert s.append(vals) == o.append(vals)
            s.scan(0, 10)
            o.scan(0, 10)
            assert s.compacted and o.compacted and s.function == o.function == host.FN_SUCCINCT
            assert s.append(b) == o.append(b) == 2048
            assert_same_state([s], [o])
            assert s.function == host.FN_SUCCINCT and s.compacted
            got = np.concatenate([s.scan(r, min(2048, s.count - r)) for r in range(0, s.count, 2048)])
            assert np.array_equal(got, np.concatenate([a, b]))
            assert np.array_equal(got, o.scan(0, o.count))
    finally:
        db.close()


def test_nulls_through_the_append_slot(adac, oracle, host):
    rng = np.random.default_rng(4)
    db = host.Database(0, arena_bytes=16 << 20)
    try:
        for dtype in (np.int32, np.uint64):
            values = (90000 + rng.integers(0, 1000, size=70000)).astype(dtype)
            segs, orcs = load_column(db, oracle, adac, dtype, values, validity_prob=0.2, rng=rng)
            db.compact_all()
            for o in orcs:
                o.compact()
            assert_same_state(segs, orcs)
            for s, o in zip(segs, orcs):
                assert np.array_equal(s.scan(0, s.count), o.scan(0, o.count))  # NULL slots included, bit for bit
    finally:
        db.close()


def test_unpackable_and_unsupported(adac, oracle, host):
    db = host.Database(0, arena_bytes=16 << 20)
    try:
        s = db.create_segment(np.int32, segment_size=8192)
        o = oracle.Segment(np.int32, segment_size=8192, store_min=True)
        v = np.array([-1, 5, 7] * 100, dtype=np.int32)  # mixed sign: range spans 2^64, cannot shrink
        s.append(v), o.append(v)
        s.compact(), o.compact()
        assert s.compacted and s.width == o.width == 32 and s.data_size == o.data_size
        assert np.array_equal(s.scan(0, 300), v)   # the reference returns {4,10,12,...} here
        with pytest.raises(adac.AdacError):
            db.create_segment(np.float64)
        assert host.hlib().adach_type_is_supported(13) == 0
    finally:
        db.close()


def test_adaptive_policy_round(adac, oracle, host):
    """CompressLowestKSegments (column_segment_catalog.cpp:64-116): the 90 % least-read segments are compacted,
    the hottest 10 % expanded, counters reset; sizes follow the oracle's model of the same decisions."""
    rng = np.random.default_rng(6)
    n = 10 * 32767 + 5000
    values = (1 << 33) + rng.integers(0, 1 << 17, size=n).astype(np.uint64)
    db = host.Database(0, adaptive=True, arena_bytes=64 << 20)
    try:
        # adaptive mode: segments start UNCOMPRESSED, appended through FixedSizeAppend (column_segment.cpp:55-76)
        segs, orcs = [], []
        row = 0
        while row < n:
            c = min(32767, n - row)
            s = db.create_segment(np.uint64, start=row)
            o = oracle.Segment(np.uint64, adaptive=True, store_min=True)
            for off in range(0, c, 2048):
                k = min(2048, c - off)
                assert s.append(values[row:row + c], offset=off, count=k) == k
                o.append(values[row:row + c], offset=off, count=k)
            segs.append(s)
            orcs.append(o)
            row += c
        assert all(s.function == host.FN_UNCOMPRESSED and not s.compacted for s in segs)
        assert db.total_data_size == len(segs) * 262136
        # skewed access trace over segments: segment i read (len - i)^2 times
        reads = [(len(segs) - i) ** 2 for i in range(len(segs))]
        for s, o, r in zip(segs, orcs, reads):
            for _ in range(r):
                s.scan(0, 16)
                o.scan(0, 16)
        assert not any(s.compacted for s in segs)  # scans do not compact in adaptive mode
        db.policy_step(0.90)
        order = sorted(range(len(segs)), key=lambda i: (orcs[i].num_reads, i))
        for rank, i in enumerate(order):
            if float(np.float32(rank + 1) / np.float32(len(order))) < 0.90:
                orcs[i].compact()
            else:
                orcs[i].uncompact()
            orcs[i].reset_reads()
        hot = [i for i, s in enumerate(segs) if not s.compacted]
        assert hot == [0, 1]  # the two most-read of 11 segments stay expanded
        assert_same_state(segs, orcs)
        assert db.total_data_size == sum(o.data_size for o in orcs)
        row = 0
        for s in segs:
            assert np.array_equal(s.scan(0, s.count), values[row:row + s.count])
            row += s.count
        # next round with the access pattern reversed: hot segments cool down and get packed, cold ones expand
        for s, r in zip(segs, reversed(reads)):
            for _ in range(r):
                s.scan(5, 3)
        used_before = db.arena_used_bytes
        db.policy_step(0.90)
        hot = [i for i, s in enumerate(segs) if not s.compacted]
        assert hot == [len(segs) - 2, len(segs) - 1]
        assert db.arena_used_bytes != used_before
        row = 0
        for s in segs:
            assert np.array_equal(s.scan(0, s.count), values[row:row + s.count])
            row += s.count
    finally:
        db.close()


@pytest.mark.timeout(120)
def test_background_thread_compacts_while_scanning(adac, host):
    """The detached policy thread of the reference (column_segment_catalog.cpp:13-22), with a short period:
    scans keep returning correct rows while representations flip underneath."""
    import time
    rng = np.random.default_rng(9)
    db = host.Database(0, adaptive=True, arena_bytes=64 << 20)
    try:
        cols = []
        for i in range(8):
            v = (i * 1000 + rng.integers(0, 1 << 10, size=20000)).astype(np.uint32)
            s = db.create_segment(np.uint32, start=i * 20000)
            for off in range(0, len(v), 2048):
                s.append(v, offset=off, count=min(2048, len(v) - off))
            cols.append((s, v))
        db.enable_background(20)
        t_end = time.time() + 1.0
        flips = 0
        last = [s.compacted for s, _ in cols]
        while time.time() < t_end:
            for k, (s, v) in enumerate(cols):
                r = int(rng.integers(0, 20000 - 2048))
                reps = 1 + 3 * (k % 3)
                for _ in range(reps):
                    assert np.array_equal(s.scan(r, 2048), v[r:r + 2048])
            now = [s.compacted for s, _ in cols]
            flips += sum(a != b for a, b in zip(last, now))
            last = now
        db.disable_background()
        assert flips > 0
        assert any(s.compacted for s, _ in cols)
    finally:
        db.close()


def test_decoded_segment_cache_serves_vectors(adac, host):
    """with the vector-serving cache on, a full scan in the engine's call pattern decodes each
    segment once on the device (one miss per segment) and serves the other vectors from pinned host memory;
    results are identical with the cache off, and representation flips invalidate it."""
    rng = np.random.default_rng(12)
    n = 400_000
    values = (1 << 20) + rng.integers(0, 1 << 14, size=n).astype(np.uint32)
    results = {}
    for cache_bytes in (0, 8 << 20):
        db = host.Database(0, arena_bytes=32 << 20, decoded_cache_bytes=cache_bytes)
        try:
            row = 0
            for count, cap in lay_mod(adac).appender_segments(n, 4):
                s = db.create_segment(np.uint32, start=row, segment_size=cap * 4)
                for off in range(0, count, 2048):
                    s.append(values[row:row + count], offset=off, count=min(2048, count - off))
                row += count
            db.compact_all()
            cs, sec, rows = db.full_scan()
            assert rows == n and cs == int(values.astype(np.uint64).sum())
            cs2, _, _ = db.full_scan()
            assert cs2 == cs
            st = db.cache_stats()
            results[cache_bytes] = st
            if cache_bytes:
                nseg = len(db.segments)
                assert st["misses"] == nseg            # one device decode per segment, second pass all hits
                assert st["hits"] >= 2 * (n // 2048) - nseg
                assert 0 < st["bytes"] <= cache_bytes
                # a flip drops the cached image: uncompact + new values + recompact must be visible
                s = db.segments[1]
                before = s.scan(0, 2048).copy()
                s.uncompact()
                assert not s.compacted
                s.compact()
                assert np.array_equal(s.scan(0, 2048), before)
                # eviction: a cache smaller than the working set still returns correct rows
        finally:
            db.close()
    assert results[0] == {"hits": 0, "misses": 0, "bytes": 0}
    small = host.Database(0, arena_bytes=32 << 20, decoded_cache_bytes=300_000)  # holds one segment at a time
    try:
        row = 0
        for count, cap in lay_mod(adac).appender_segments(n, 4):
            s = small.create_segment(np.uint32, start=row, segment_size=cap * 4)
            for off in range(0, count, 2048):
                s.append(values[row:row + count], offset=off, count=min(2048, count - off))
            row += count
        small.compact_all()
        cs, _, rows = small.full_scan()
        assert rows == n and cs == int(values.astype(np.uint64).sum())
        assert small.cache_stats()["bytes"] <= 300_000
    finally:
        small.close()


def test_all_sixteen_plugin_slots_and_the_checkpoint_side_pipeline(adac, oracle, host):
    """succinct.cpp:335-343: FixedSize analyze slots, InitCompression / Compress / FinalizeCompress, init_scan,
    scans, fetch_row, EmptySkip, init_append / append / finalize_append; init_segment and revert_append are null.
    The compress slots fill BLOCK_SIZE segments (CreateEmptySegment -> Append until full -> FlushSegment) and each
    full segment compacts itself as it fills (column_segment.cpp:266-268)."""
    db = host.Database(0, succinct_enabled=True, adaptive=False, arena_bytes=64 << 20)
    for ctype in (10, 1):
        slots = db.function_slots(np.uint32, ctype)
        assert [n for n, p in slots.items() if not p] == ["init_segment", "revert_append"]
    with pytest.raises(adac.AdacError):
        db.function_slots(np.float32)   # InternalException("Unsupported type ...") territory (succinct.cpp:364)
    n = 200_000
    vals = (np.arange(n, dtype=np.uint32) * 3 + 1000)
    segs, sizes, score = db.compress_column(vals, row_group_start=122_880)
    assert score == 4 * n                                     # FixedSizeFinalAnalyze<T>: sizeof(T) * count
    per = 262_136 // 4
    counts = [per, per, per, n - 3 * per]
    assert [s.count for s in segs] == counts and sizes == [4 * c for c in counts]
    assert [s.start for s in segs] == [122_880 + sum(counts[:i]) for i in range(4)]
    row = 0
    for s, c in zip(segs, counts):
        o = oracle.Segment(np.uint32, segment_size=262_136, store_min=True)
        off = 0
        while off < c:   # the oracle segment fed the same vectors
            k = min(2048 - (row + off) % 2048, c - off)
            o.append(vals[row:row + c], None, offset=off, count=k)
            off += k
        if c == per:
            assert s.compacted and s.width == o.width and s.min_factor == o.min_factor
            assert s.data_size == o.data_size
        got = np.concatenate([s.scan(r, min(2048, c - r)) for r in range(0, c, 2048)])
        assert np.array_equal(got, vals[row:row + c])
        row += c
    # NULLs through the compress slot (Vector::ToUnifiedFormat hands the vector's own validity mask on)
    valid = np.ones(n, dtype=bool)
    valid[5::7] = False
    vm = np.packbits(valid, bitorder="little")
    vm = np.concatenate([vm, np.zeros((-len(vm)) % 8 + 8, np.uint8)]).view(np.uint64)
    segs2, _, _ = db.compress_column(vals.astype(np.int32) - 500_000, validity=vm)
    row = 0
    for s in segs2:
        got = np.concatenate([s.scan(r, min(2048, s.count - r)) for r in range(0, s.count, 2048)])
        ok = valid[row:row + s.count]
        assert np.array_equal(got[ok], (vals.astype(np.int32) - 500_000)[row:row + s.count][ok])
        row += s.count
    db.close()


@pytest.mark.timeout(120)
def test_concurrent_flips_appends_and_scans(adac, host):
    """Regression guard for two races found by tools/soak_threads.py: two representation flips of one segment
    overlapping (background policy thread + a second flipper), and the policy thread compacting a segment between
    an Append's Uncompact and its write.  Readers on six threads must always see the loaded rows."""
    import threading
    import time
    rng = np.random.default_rng(33)
    db = host.Database(0, adaptive=True, arena_bytes=64 << 20, decoded_cache_bytes=4 << 20)
    errors = []
    try:
        cols = []
        for i in range(10):
            dtype = (np.uint32, np.int64)[i % 2]
            v = (np.int64(i * 500) + rng.integers(0, 1 << (6 + i), size=12000)).astype(dtype)
            s = db.create_segment(dtype, start=i * 50000)
            for off in range(0, len(v), 2048):
                s.append(v, offset=off, count=min(2048, len(v) - off))
            cols.append((s, v))
        full = (900_000 + rng.integers(0, 1 << 8, size=40000)).astype(np.uint32)
        grow = db.create_segment(np.uint32, start=9_000_000)
        grow.append(full, offset=0, count=2048)
        have = [2048]
        stop = threading.Event()

        def reader(tid):
            r = np.random.default_rng(tid)
            try:
                while not stop.is_set():
                    if r.random() < 0.2:
                        a = int(r.integers(0, have[0] - 32))
                        ok = np.array_equal(grow.scan(a, 32), full[a:a + 32])
                    else:
                        s, v = cols[int(r.integers(0, len(cols)))]
                        a = int(r.integers(0, len(v) - 1))
                        k = int(r.integers(1, min(2048, len(v) - a) + 1))
                        ok = np.array_equal(s.scan(a, k), v[a:a + k])
                    if not ok:
                        errors.append(("mismatch", tid))
                        return
            except Exception as e:  # noqa: BLE001
                errors.append((tid, repr(e)[:200]))

        def flipper():
            r = np.random.default_rng(99)
            try:
                while not stop.is_set():
                    s, _ = cols[int(r.integers(0, len(cols)))]
                    (s.compact if r.random() < 0.5 else s.uncompact)()
            except Exception as e:  # noqa: BLE001
                errors.append(("flipper", repr(e)[:200]))

        def writer():
            try:
                while not stop.is_set() and have[0] + 2048 <= len(full):
                    assert grow.append(full, offset=have[0], count=2048) == 2048
                    have[0] += 2048
                    time.sleep(0.002)
            except Exception as e:  # noqa: BLE001
                errors.append(("writer", repr(e)[:200]))

        db.enable_background(3)
        threads = [threading.Thread(target=reader, args=(t,)) for t in range(6)]
        threads += [threading.Thread(target=flipper), threading.Thread(target=writer)]
        for t in threads:
            t.start()
        time.sleep(1.5)
        stop.set()
        for t in threads:
            t.join()
        db.disable_background()
        assert not errors, errors[:3]
        assert np.array_equal(np.concatenate([grow.scan(a, min(2048, have[0] - a)) for a in range(0, have[0], 2048)]),
                              full[:have[0]])
    finally:
        db.close()
