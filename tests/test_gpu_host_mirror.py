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
                pf = db.prefetch_stats()
                # ONE device decode per segment: the first segment misses and takes the next ones of the chain along
                # in its batch, every later segment was decoded ahead of its first touch; the second pass only hits.
                # A scan state pins the decoded block, so a look-up happens once per segment and scan, not per vector
                assert st["misses"] + pf["prefetched"] == nseg and st["misses"] == 1
                assert st["hits"] == 2 * nseg - 1
                assert pf["batches"] < nseg                       # several segments per launch and copy
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


def test_per_gpu_segment_pools_behind_the_plugin_surface(adac, oracle, host):
    """north star: "per-GPU segment pools" — a database with several pools (here all mapped to the one GPU of the test
    box) spreads its segments by id, every pool compacts / scans / persists its own, and the engine-visible results
    equal the single-pool database's: scans, sizes, the policy's hot set."""
    rng = np.random.default_rng(21)
    n = 700_000
    values = (5_000_000 + rng.integers(0, 1 << 13, size=n)).astype(np.uint32)
    outcome = []
    for devices in (0, [0, 0, 0]):
        db = host.Database(devices, adaptive=True, arena_bytes=32 << 20, decoded_cache_bytes=4 << 20)
        try:
            npools = db.num_pools
            assert npools == (1 if devices == 0 else 3)
            segs, orcs = load_column(db, oracle, adac, np.uint32, values, adaptive=True)
            assert [s.pool for s in segs] == [i % npools for i in range(len(segs))]   # segment id mod pools
            raw_size = db.total_data_size
            # reads: segment i is read i times (ColumnSegmentCatalog::AddReadAccess per scan call)
            for i, s in enumerate(segs):
                for _ in range(i):
                    s.scan(0, 64)
            db.policy_step(0.90)
            hot = [i for i, s in enumerate(segs) if not s.compacted]
            used = [db.pool_arena_used_bytes(p) for p in range(npools)]
            assert all(u > 0 for u in used) and sum(used) == db.arena_used_bytes
            cs, _, rows = db.full_scan(threads=1)
            cs4, _, rows4 = db.full_scan(threads=4)
            assert rows == rows4 == n and cs == cs4 == int(values.astype(np.uint64).sum())
            row = 0
            for s in segs:
                assert np.array_equal(s.scan(0, s.count), values[row:row + s.count])
                row += s.count
            images = db.persist([s for s in segs if s.compacted])
            outcome.append((hot, db.total_data_size, raw_size, [s.width for s in segs], images))
        finally:
            db.close()
    assert outcome[0][:4] == outcome[1][:4]
    assert outcome[0][4] == outcome[1][4]      # the block images do not depend on where a segment lived
    assert len(outcome[0][0]) >= 1             # some segments stayed hot / unpacked


def test_checkpoint_writes_block_images_and_they_load_back(adac, oracle, host):
    """f-3 through the plugin table: compress / compress_finalize flush the segments AND yield their block images
    (the ConvertToPersistent the reference leaves empty, column_segment.cpp:529-533); the images are the SDSL
    serialisation of the packed vectors; loading them into another database (other pool count) gives the same column."""
    rng = np.random.default_rng(8)
    for dtype, bits in ((np.uint32, 11), (np.int64, 29), (np.uint16, 16), (np.int8, 3)):
        dtype = np.dtype(dtype)
        n = 150_000
        info = np.iinfo(dtype)
        lo = int(info.min) // 2
        values = (lo + rng.integers(0, min(1 << bits, int(info.max) - lo), size=n)).astype(dtype)
        validity = np.full((n + 63) // 64 + 1, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        for e in rng.choice(n, 500, replace=False):
            validity[e >> 6] &= np.uint64(~(1 << (int(e) & 63)) & 0xFFFFFFFFFFFFFFFF)
        db = host.Database([0, 0], arena_bytes=16 << 20)
        db2 = host.Database(0, arena_bytes=16 << 20, decoded_cache_bytes=2 << 20)
        try:
            segs, sizes, score, images = db.checkpoint_column(values, validity)
            assert score == n * dtype.itemsize and len(images) == len(segs)
            valid = np.unpackbits(validity.view(np.uint8), bitorder="little")[:n].astype(bool)
            row = 0
            for s, img in zip(segs, images):
                c = s.count
                assert s.persistent
                d, t, words = adac.block_read(img)
                assert t == dtype and int(d["count"]) == c
                if c == 0:
                    continue
                assert s.compacted and int(d["width"]) == s.width
                # the image's vector is what SDSL would serialise, and decodes (oracle) to the column's valid rows
                assert img[:len(img) - 16] == oracle.serialize(words, c * int(d["width"]), int(d["width"]))
                packed = bool(d["flags"] & 1) and int(d["min"]) != 0xFFFFFFFFFFFFFFFF
                dec = oracle.unpack_flat(words, 0, c, int(d["width"]), int(d["min"]) if packed else 0, dtype)
                v = valid[row:row + c]
                assert np.array_equal(dec[v], values[row:row + c][v])
                row += c
            assert row == n
            starts = np.concatenate([[0], np.cumsum([s.count for s in segs])[:-1]]).astype(np.uint64)
            back = db2.load(images, dtype, starts)
            assert sum(s.data_size for s in back) == sum(s.data_size for s in segs)
            row = 0
            for s, b in zip(segs, back):
                assert b.persistent and b.count == s.count and b.width == s.width and b.compacted == (s.count > 0 or b.compacted)
                if s.count:
                    got = b.scan(0, b.count)
                    v = valid[row:row + s.count]
                    assert np.array_equal(got[v], values[row:row + s.count][v])
                    assert np.array_equal(got, s.scan(0, s.count))     # NULL slots too: same stored bits
                    r = int(rng.integers(0, s.count))
                    assert b.fetch_row(r) == got[r]
                row += s.count
            cs, _, rows = db2.full_scan(back)
            cs0, _, rows0 = db.full_scan(segs)
            assert (cs, rows) == (cs0, rows0)
            # a loaded segment is an ordinary one: append -> uncompact -> recompact
            tail = back[-2] if len(back) > 1 else back[-1]
            before = tail.scan(0, tail.count).copy()
            tail.uncompact()
            assert not tail.compacted and not tail.persistent
            tail.compact()
            assert np.array_equal(tail.scan(0, tail.count), before)
        finally:
            db.close()
            db2.close()


def test_arena_exhaustion_is_not_fatal_and_the_policy_thread_survives(adac, host):
    """a full arena used to throw out of the background thread (std::terminate) and leaked the blocks the
    batch had already taken.  Now the segments that do not fit stay unpacked, the round completes, nothing leaks,
    and the thread keeps running."""
    rng = np.random.default_rng(3)
    rows = 32767
    nseg = 24
    # 24 segments of 32767 x 14-bit values need ~57 KiB each packed: the arena holds about 8 of them
    db = host.Database(0, adaptive=True, arena_bytes=480 << 10)
    try:
        data = []
        for i in range(nseg):
            v = ((i << 20) + rng.integers(0, 1 << 14, size=rows)).astype(np.uint64)
            s = db.create_segment(np.uint64, start=i * rows)
            for off in range(0, rows, 2048):
                s.append(v, offset=off, count=min(2048, rows - off))
            data.append(v)
        db.policy_step(0.90)
        packed = [s for s in db.segments if s.compacted]
        st = db.prefetch_stats()
        assert 1 <= len(packed) < 21 and st["arena_exhausted"] >= 21 - len(packed)
        used = db.arena_used_bytes
        assert used <= 480 << 10
        for s, v in zip(db.segments, data):
            assert np.array_equal(s.scan(0, rows), v)
        db.enable_background(5)
        import time
        deadline = time.time() + 20
        while db.background_stats()["rounds"] < 6 and time.time() < deadline:
            db.segments[int(rng.integers(0, nseg))].scan(0, 2048)
        bs = db.background_stats()
        db.disable_background()
        assert bs["rounds"] >= 6 and bs["errors"] == 0, bs
        assert db.arena_used_bytes <= 480 << 10
        for s, v in zip(db.segments, data):
            assert np.array_equal(s.scan(0, rows), v)
        # room again once segments go away: the next round packs what was left out
        for s in list(db.segments[:12]):
            s.close()
        before = sum(1 for s in db.segments[12:] if s.compacted)
        db.policy_step(0.90)
        assert sum(1 for s in db.segments[12:] if s.compacted) >= before
    finally:
        db.close()


def test_segments_destroyed_and_created_while_the_policy_thread_runs(adac, host):
    """a policy round snapshots segment pointers; a segment destroyed while the round is in flight must not
    be touched afterwards (the reference has this race and suppresses it under TSan: race:~ColumnSegment).  A round now
    re-checks the catalog under the pool's flip lock, which the destructor takes before it leaves the catalog."""
    import threading
    rng = np.random.default_rng(4)
    db = host.Database([0, 0], adaptive=True, arena_bytes=64 << 20, decoded_cache_bytes=2 << 20)
    stop = threading.Event()
    errors = []
    keep = []
    try:
        vals = (77_000 + rng.integers(0, 1 << 12, size=20000)).astype(np.uint32)
        for i in range(16):
            s = db.create_segment(np.uint32, start=i * 20000)
            s.append(vals)
            keep.append(s)
        db.enable_background(2)

        def churn():
            try:
                k = 0
                while not stop.is_set():
                    s = host.Segment(db, np.uint32, 10_000_000 + k * 20000, 262136)
                    s.append(vals)
                    assert np.array_equal(s.scan(100, 500), vals[100:600])
                    s.close()      # ~ColumnSegment while rounds are in flight
                    k += 1
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        def read():
            try:
                while not stop.is_set():
                    s = keep[int(rng.integers(0, len(keep)))]
                    assert np.array_equal(s.scan(0, 2048), vals[:2048])
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=churn) for _ in range(2)] + [threading.Thread(target=read) for _ in range(2)]
        for t in threads:
            t.start()
        import time
        time.sleep(4.0)
        stop.set()
        for t in threads:
            t.join()
        bs = db.background_stats()
        db.disable_background()
        assert not errors, errors
        assert bs["rounds"] > 50 and bs["errors"] == 0, bs
        for s in keep:
            assert np.array_equal(s.scan(0, 20000), vals)
    finally:
        stop.set()
        db.close()
