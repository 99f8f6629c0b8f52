"""The call sequence integration/succinct_gpu.cpp issues — the DuckDB-typed adapter is a shim over
include/adacodec_host.h, so this drives, call for call, what its callbacks forward to:

  init_segment     -> adach_segment_create
  append           -> adach_segment_append (selection vector + validity mask of a UnifiedVectorFormat)
  Compact bodies   -> adach_segments_compact / adach_segment_compact
  init_scan        -> adach_segment_set_next (SegmentBase::next) + adach_scan_state_create + adach_segment_init_scan
  scan_vector/_partial -> adach_segment_scan_with   (the state lives in ColumnScanState::scan_state across calls)
  fetch_row        -> adach_segment_fetch_row
  adaptive flips   -> SuccinctAdoptUncompressed / SuccinctRestoreUncompressed: create + append the raw block + compact;
                      uncompact + scan everything back

in the pattern of ColumnData::ScanVector (/root/reference/src/storage/table/column_data.cpp:92-139): 2048-row
vectors, a vector that runs past a segment's end continues in the next segment with a partial scan.  Asserted besides
the values: the decoded-segment cache is touched once per SEGMENT, not once per vector — no per-vector device round
trip is left on this path."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VECTOR = 2048


@pytest.fixture(scope="module")
def host(adac):
    adac.build()
    return importlib.import_module(adac.__name__ + ".host")


def append_column(db, dtype, values, valid, rng, segment_rows):
    """ColumnData::Append's loop (column_data.cpp:249-271): 2048-row vectors through a selection vector into the
    current segment; what the segment does not take opens the next one."""
    dtype = np.dtype(dtype)
    segs = []
    cur = db.create_segment(dtype, start=0, segment_size=segment_rows * dtype.itemsize, chain=False)
    segs.append(cur)
    row = 0
    while row < len(values):
        n = min(VECTOR, len(values) - row)
        # a dictionary-style vector: the rows of this vector sit shuffled in a buffer and are named by the selection
        perm = rng.permutation(n).astype(np.uint32)
        buf = np.empty(n, dtype=dtype)
        buf[perm] = values[row:row + n]
        vbits = np.zeros((n + 63) // 64 * 64, dtype=bool)
        vbits[perm] = valid[row:row + n]
        validity = np.packbits(vbits, bitorder="little").view(np.uint64)
        off = 0
        while off < n:
            took = cur.append(buf, validity, sel=perm, offset=off, count=n - off)
            off += took
            if off < n:  # segment full: the caller opens a new one for the rest
                cur = db.create_segment(dtype, start=row + off, segment_size=segment_rows * dtype.itemsize, chain=False)
                segs.append(cur)
        row += n
    return segs


def scan_like_column_data(host, segs, total_rows, dtype, start_row=0):
    """ColumnData::ScanVector over the whole column with ONE scan state, as one scanning thread holds it."""
    state = host.ScanState()
    out = np.empty(total_rows - start_row, dtype=dtype)
    idx = 0
    while segs[idx].start + segs[idx].count <= start_row:
        idx += 1

    def init_scan(i):  # the shim's SuccinctInitScan
        if i + 1 < len(segs):
            host._ok(host.hlib().adach_segment_set_next(segs[i]._h, segs[i + 1]._h), "SetNext")
        segs[i].init_scan(state)

    init_scan(idx)
    row_index, calls = start_row, 0
    while row_index < total_rows:
        vec = np.empty(VECTOR, dtype=dtype)
        remaining = initial = min(VECTOR, total_rows - row_index)
        while remaining > 0:
            cur = segs[idx]
            scan_count = min(remaining, cur.start + cur.count - row_index)
            result_offset = initial - remaining
            if scan_count > 0:
                cur.scan_with(state, row_index - cur.start, scan_count, vec, result_offset,
                              entire_vector=(scan_count == initial))
                calls += 1
                row_index += scan_count
                remaining -= scan_count
            if remaining > 0:
                idx += 1
                init_scan(idx)
        out[row_index - initial - start_row:row_index - start_row] = vec[:initial]
    state.close()
    return out, calls


# (type, lo, hi): value ranges the append rule packs (succinct.cpp:286-287 orders by the sign-extended value, so a
# same-sign range has a small max - min) and one mixed-sign range, whose segments keep their full-width slots
# (8 sizeof(T) <= w, column_segment.cpp:363) and are served from the unpacked image without any decode
@pytest.mark.parametrize("dtype,lo,hi", [(np.uint32, 0, 1 << 20), (np.int64, 1 << 40, (1 << 40) + (1 << 20)),
                                         (np.uint8, 0, 100), (np.int16, -3000, -1), (np.int32, -1000, 1000)])
def test_shim_sequence_serves_vectors_from_the_pinned_block(adac, host, dtype, lo, hi):
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(5)
    segment_rows = 262136 // dtype.itemsize
    n = int(segment_rows * 6.4)
    packs = not (lo < 0 <= hi)
    values = rng.integers(lo, hi, size=n).astype(dtype)
    valid = rng.random(n) > 0.03
    db = host.Database(0, arena_bytes=64 << 20, decoded_cache_bytes=32 << 20, prefetch_segments=4)
    try:
        segs = append_column(db, dtype, values, valid, rng, segment_rows)
        assert len(segs) == 7 and sum(s.count for s in segs) == n
        assert all(s.compacted for s in segs[:-1])          # a segment that filled up compacted itself
        db.compact_segments(segs)                            # the policy's list: the last one follows
        assert all(s.compacted and s.function == host.FN_SUCCINCT for s in segs)
        expect = values  # compared on the valid rows: a NULL slot holds (NullValue<T> - min) mod 2^w, "never read"
        before = db.cache_stats()
        got, calls = scan_like_column_data(host, segs, n, dtype)
        after, pre = db.cache_stats(), db.prefetch_stats()
        assert np.array_equal(got[valid], expect[valid])
        touches = after["hits"] + after["misses"] - before["hits"] - before["misses"]
        assert calls >= n // VECTOR and touches <= len(segs), (calls, touches)   # once per segment, not per vector
        if packs:
            assert all(s.width < 8 * dtype.itemsize for s in segs)
            assert 1 <= pre["batches"] <= len(segs) and pre["prefetched"] >= len(segs) - 2, pre
        else:
            assert all(s.width == 8 * dtype.itemsize for s in segs) and pre["batches"] == 0 and touches == 0, pre
        # a scan that starts in the middle of the column (InitializeScanWithOffset) and vectors straddling segments
        mid = int(segment_rows * 2.5) + 77
        got2, _ = scan_like_column_data(host, segs, n, dtype, start_row=mid)
        assert np.array_equal(got2[valid[mid:]], expect[mid:][valid[mid:]])
        # fetch_row: the slot's row_id is segment-relative; touched segments answer from their decoded blocks
        misses = db.cache_stats()["misses"]
        for r in np.flatnonzero(valid)[rng.integers(0, int(valid.sum()), size=64)]:
            si = max(i for i, s in enumerate(segs) if s.start <= r)
            assert segs[si].fetch_row(int(r) - segs[si].start) == expect[r]
        assert db.cache_stats()["misses"] == misses
        # an append into a packed segment expands it first (column_segment.cpp:254-259), the rows come back unchanged
        last = segs[-1]
        room = segment_rows - last.count
        extra = rng.integers(lo, hi, size=min(room, 100)).astype(dtype)
        if len(extra):
            assert last.append(extra) == len(extra)
            tail = last.scan(0, last.count)
            ok = valid[last.start:]
            assert np.array_equal(tail[:-len(extra)][ok], expect[last.start:][ok])
            assert np.array_equal(tail[-len(extra):], extra)
    finally:
        db.close()


@pytest.mark.parametrize("dtype", [np.int32, np.uint64])
def test_shim_adaptive_flips_adopt_and_restore_a_raw_block(adac, oracle, host, dtype):
    """SuccinctAdoptUncompressed / SuccinctRestoreUncompressed of the adapter: an UNCOMPRESSED transient segment's
    block goes into a mirror segment, which packs it under BitCompressFromUncompressed's zero-extended rule
    (column_segment.cpp:385-456), and comes back bit for bit on Uncompact."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(9)
    rows = 262136 // dtype.itemsize
    block = rng.integers(1000, 1000 + (1 << 13), size=rows).astype(dtype)
    db = host.Database(0, adaptive=True, arena_bytes=32 << 20, decoded_cache_bytes=8 << 20)
    try:
        seg = db.create_segment(dtype, start=4096, segment_size=262136)
        assert seg.function == host.FN_UNCOMPRESSED
        assert seg.append(block) == rows          # the whole block, identity selection, all valid
        seg.compact()
        assert seg.compacted and seg.function == host.FN_SUCCINCT
        mn, mx = oracle.analyze_flat(block, 1)    # rule RECOMPACT
        assert seg.width == oracle.width_from_uncompressed(mn, mx) == 13
        state = host.ScanState()
        seg.init_scan(state)
        out = np.empty(rows, dtype=dtype)
        for r in range(0, rows, VECTOR):
            c = min(VECTOR, rows - r)
            seg.scan_with(state, r, c, out[r:r + c])
        assert np.array_equal(out, block)
        seg.uncompact()
        assert not seg.compacted and seg.function == host.FN_UNCOMPRESSED
        back = seg.scan(0, rows)                  # adach_segment_scan: what SuccinctRestoreUncompressed copies out
        assert np.array_equal(back, block)
        # the scan state of the packed form re-pins on its own: a scan through it still returns the rows
        assert np.array_equal(seg.scan_with(state, 0, VECTOR, np.empty(VECTOR, dtype=dtype)), block[:VECTOR])
        state.close()
    finally:
        db.close()
