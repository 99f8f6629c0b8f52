"""adac_scan_select_between: the filter scan with a selection-bitmap result (ColumnSegment::FilterSelection,
column_segment.cpp:575-844, evaluated on the packed bytes).  Expected bitmaps are built with numpy from the
original values; the packed words themselves are checked against the oracle by run_encode_decode."""
import numpy as np
import pytest

from test_gpu_parity import make_values, run_encode_decode, wide_sum

pytestmark = pytest.mark.gpu


def bit_pattern(v, dtype):
    dtype = np.dtype(dtype)
    return int(np.array([v]).astype(dtype).view(np.dtype("u%d" % dtype.itemsize))[0])


def pack_mask(bits, span):
    """bool per element -> u64 words of a DuckDB validity-style mask (+ one spare word)."""
    full = np.zeros(span, dtype=bool)
    full[:len(bits)] = bits
    b = np.packbits(full, bitorder="little")
    return np.concatenate([b, np.zeros((-len(b)) % 8 + 8, np.uint8)]).view(np.uint64)


def expected_bitmap(segs, offs, span, lo, hi, valid=None):
    sel = np.zeros(span, dtype=bool)
    for v, o in zip(segs, offs):
        hit = (v >= lo) & (v <= hi)
        if valid is not None:
            hit &= valid[o:o + len(v)]
        sel[o:o + len(v)] = hit
    return sel


def check_select(adac, ctx, lay, d_words, dtype, segs, offs, span, probes, valid=None):
    nw = (span + 63) // 64
    d_bm = ctx.alloc(nw * 8 + 8)
    d_cnt = ctx.alloc(len(segs) * 8)
    d_valid = None if valid is None else ctx.upload(pack_mask(valid, span))
    try:
        for templated in (1, 0):
            adac.set_tuning("templated_scan", templated)
            for group in (16, 3):
                adac.set_tuning("scan_tiles_per_wg", group)
                for lo, hi in probes:
                    d_bm.upload(np.full(nw + 1, 0xDEADBEEFDEADBEEF, dtype=np.uint64))  # the call clears it
                    lay.scan_select_between(d_words, bit_pattern(lo, dtype), bit_pattern(hi, dtype), d_bm, d_cnt, d_valid)
                    got = d_bm.download(np.uint64, nw + 1)
                    assert int(got[nw]) == 0xDEADBEEFDEADBEEF          # nothing past ceil(span / 64) words
                    exp = expected_bitmap(segs, offs, span, lo, hi, valid)
                    bits = np.unpackbits(got[:nw].view(np.uint8), bitorder="little")[:span].astype(bool)
                    bad = np.flatnonzero(bits != exp)
                    assert bad.size == 0, (np.dtype(dtype).name, templated, group, lo, hi, bad[:8].tolist())
                    tail = np.unpackbits(got[:nw].view(np.uint8), bitorder="little")[span:]
                    assert not tail.any()
                    cnt = d_cnt.download(np.uint64, len(segs)).tolist()
                    assert cnt == [int(exp[o:o + len(v)].sum()) for v, o in zip(segs, offs)]
    finally:
        adac.set_tuning("templated_scan", 1)
        adac.set_tuning("scan_tiles_per_wg", 0)


@pytest.mark.parametrize("dtype", [np.uint64, np.int64, np.uint32, np.int32, np.uint16, np.int16, np.uint8, np.int8])
def test_select_bitmap_every_path(adac, oracle, gpu_ctx, dtype):
    dtype = np.dtype(dtype)
    tb = 8 * dtype.itemsize
    info = np.iinfo(dtype)
    rng = np.random.default_rng(900 + tb + (dtype.kind == "i"))
    tile = adac.tile_values(dtype)
    widths = sorted({1, 3, 4, 5, 8, 13, 16, 21, 31, 32, 33, 47, tb - 1, tb} & set(range(1, tb + 1)))
    counts, segs = [], []
    for w in widths:
        n = int(rng.integers(50, 3 * tile)) if w % 2 else 2 * tile + 1
        counts.append(n)
        segs.append(make_values(rng, dtype, n, w))
    counts += [1, 0, 7]
    segs += [make_values(rng, dtype, 1, 3), make_values(rng, dtype, 0, 3), make_values(rng, dtype, 7, 2)]
    counts = np.array(counts, dtype=np.uint32)
    # segments at ragged element offsets: bitmap words are shared between neighbouring segments and have gaps
    offs, run = [], 0
    for i, c in enumerate(counts):
        run += (3, 0, 17, 1)[i % 4]
        offs.append(run)
        run += int(c)
    span = run
    lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs,
                                                  val_offs=np.array(offs, dtype=np.uint64))
    assert lay.value_span == span
    a, b = sorted((int(segs[3][1]), int(segs[3][5])))
    mid = segs[len(widths) // 2]
    probes = [(a, b), (int(info.min), int(info.max)), (int(mid.min()), int(mid.min())), (int(mid[0]), int(info.max)),
              (int(info.min), int(mid[1])), (5, 3)]
    check_select(adac, gpu_ctx, lay, d_words, dtype, segs, offs, span, probes)
    valid = rng.random(span) > 0.4
    check_select(adac, gpu_ctx, lay, d_words, dtype, segs, offs, span, probes[:3], valid)


def test_select_on_recompacted_and_sign_crossing_segments(adac, oracle, gpu_ctx):
    rng = np.random.default_rng(4)
    dtype = np.dtype(np.int32)
    info = np.iinfo(dtype)
    n = 30_000
    cross = ((np.uint64(info.max - 100) + rng.integers(0, 200, size=n).astype(np.uint64)) & np.uint64(0xFFFFFFFF)) \
        .astype(np.uint32).view(np.int32)
    neg = rng.integers(-5000, -4000, size=n).astype(dtype)
    pos = rng.integers(100, 90_000, size=n).astype(dtype)
    segs = [cross, neg, pos]
    counts = np.array([n, n, n], dtype=np.uint32)
    lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs, adac.RULE_RECOMPACT)
    assert all(descs["flags"] & adac.SEG_PACKED)
    offs = [0, n, 2 * n]
    probes = [(int(info.max) - 20, int(info.max)), (int(info.min), int(info.min) + 20), (-4500, 500), (-1, 1)]
    check_select(adac, gpu_ctx, lay, d_words, dtype, segs, offs, 3 * n, probes)


def test_conjunction_and_aggregate_without_materialising(adac, oracle, gpu_ctx):
    """Q6's shape: WHERE a BETWEEN .. AND b BETWEEN .. AND c < .. -> SUM(d), all on packed columns that share
    their segment layout; the selection of one scan is the validity mask of the next."""
    rng = np.random.default_rng(6)
    n = 1_000_003
    counts = adac.appender_segment_counts(n, 4)
    offs = [int(x) for x in np.concatenate([[0], np.cumsum(counts)[:-1]])]
    cols = {"shipdate": rng.integers(8000, 10600, size=n).astype(np.int32),
            "discount": rng.integers(0, 11, size=n).astype(np.int32),
            "quantity": rng.integers(1, 51, size=n).astype(np.int32),
            "price": rng.integers(90_000, 10_500_000, size=n).astype(np.int32)}
    enc = {}
    for name, v in cols.items():
        lay = adac.Layout(gpu_ctx, np.int32, counts)
        d_vals = gpu_ctx.upload(v)
        d_words = gpu_ctx.alloc(lay.max_arena_words * 8 + 16).zero()
        lay.encode(d_vals, d_words)
        enc[name] = (lay, d_words)
    nw = (n + 63) // 64
    bm = [gpu_ctx.alloc(nw * 8) for _ in range(3)]
    d_cnt = gpu_ctx.alloc(len(counts) * 8)
    d_sum = gpu_ctx.alloc(len(counts) * 8)
    lay, w = enc["shipdate"]
    lay.scan_select_between(w, 8766, 9130, bm[0], d_cnt)
    m = (cols["shipdate"] >= 8766) & (cols["shipdate"] <= 9130)
    assert int(d_cnt.download(np.uint64, len(counts)).sum()) == int(m.sum())
    lay, w = enc["discount"]
    lay.scan_select_between(w, 5, 7, bm[1], d_cnt, bm[0])
    m &= (cols["discount"] >= 5) & (cols["discount"] <= 7)
    assert int(d_cnt.download(np.uint64, len(counts)).sum()) == int(m.sum())
    lay, w = enc["quantity"]
    lay.scan_select_between(w, bit_pattern(np.iinfo(np.int32).min, np.int32), 23, bm[2], d_cnt, bm[1])
    m &= cols["quantity"] < 24
    assert int(d_cnt.download(np.uint64, len(counts)).sum()) == int(m.sum())
    lay, w = enc["price"]
    lay.scan_sum(w, d_sum, bm[2])
    got = d_sum.download(np.uint64, len(counts)).tolist()
    assert got == [wide_sum(cols["price"][o:o + int(c)][m[o:o + int(c)]]) for o, c in zip(offs, counts)]
    assert m.sum() > 1000


@pytest.mark.parametrize("dtype", [np.uint64, np.int32, np.uint16, np.int8])
def test_unpack_selected_materialises_only_the_selected_rows(adac, oracle, gpu_ctx, dtype):
    """adac_unpack_selected: values + element ids of the rows a bitmap keeps, dense and in row order, for ragged
    segments at unaligned offsets, every width class, empty and full selections."""
    dtype = np.dtype(dtype)
    tb = 8 * dtype.itemsize
    rng = np.random.default_rng(70 + tb)
    tile = adac.tile_values(dtype)
    widths = sorted({1, 3, 5, 8, 13, 16, 27, 33, 50, tb} & set(range(1, tb + 1)))
    counts = [int(rng.integers(1, 3 * tile)) if i % 2 else 2 * tile for i in range(len(widths))] + [1, 0, tile + 7]
    segs = [make_values(rng, dtype, c, w) for c, w in zip(counts, widths + [2, 2, 4])]
    counts = np.array(counts, dtype=np.uint32)
    offs, run = [], 0
    for i, c in enumerate(counts):
        run += (0, 5, 64, 1)[i % 4]
        offs.append(run)
        run += int(c)
    span = run
    lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs,
                                              val_offs=np.array(offs, dtype=np.uint64))
    total_rows = int(counts.sum())
    d_out = gpu_ctx.alloc(total_rows * dtype.itemsize + 64)
    d_ids = gpu_ctx.alloc(total_rows * 8 + 64)
    for density in (0.0, 0.003, 0.3, 1.0):
        sel = np.zeros(span, dtype=bool)
        for v, o in zip(segs, offs):
            sel[o:o + len(v)] = rng.random(len(v)) < density if 0 < density < 1 else bool(density)
        if density == 0.3:
            sel[offs[0]:offs[0] + 70] = True       # runs crossing bitmap words
        d_bm = gpu_ctx.upload(pack_mask(sel, span)[:(span + 63) // 64])   # exactly ceil(span / 64) words
        n = lay.unpack_selected(d_words, d_bm, d_out, d_ids)
        exp_vals = np.concatenate([v[sel[o:o + len(v)]] for v, o in zip(segs, offs)])
        exp_ids = np.concatenate([o + np.flatnonzero(sel[o:o + len(v)]) for v, o in zip(segs, offs)]).astype(np.uint64)
        assert n == len(exp_vals), (dtype, density)
        assert np.array_equal(d_out.download(dtype, max(n, 1))[:n], exp_vals), (dtype, density)
        assert np.array_equal(d_ids.download(np.uint64, max(n, 1))[:n], exp_ids), (dtype, density)
        n2 = lay.unpack_selected(d_words, d_bm, d_out)              # without ids
        assert n2 == n and np.array_equal(d_out.download(dtype, max(n, 1))[:n], exp_vals)


def test_filter_then_project_pipeline(adac, oracle, gpu_ctx):
    """SELECT price FROM t WHERE shipdate BETWEEN .. AND quantity < 24 on packed columns: two chained filter scans,
    then only the surviving rows of a third column are decoded."""
    rng = np.random.default_rng(9)
    n = 700_001
    counts = adac.appender_segment_counts(n, 4)
    cols = {"shipdate": rng.integers(8000, 10600, size=n).astype(np.int32),
            "quantity": rng.integers(1, 51, size=n).astype(np.int32),
            "price": rng.integers(90_000, 10_500_000, size=n).astype(np.int32)}
    enc = {}
    for name, v in cols.items():
        lay = adac.Layout(gpu_ctx, np.int32, counts)
        d_words = gpu_ctx.alloc(lay.max_arena_words * 8 + 16).zero()
        lay.encode(gpu_ctx.upload(v), d_words)
        enc[name] = (lay, d_words)
    nw = (n + 63) // 64
    bm0, bm1 = gpu_ctx.alloc(nw * 8), gpu_ctx.alloc(nw * 8)
    d_cnt = gpu_ctx.alloc(len(counts) * 8)
    enc["shipdate"][0].scan_select_between(enc["shipdate"][1], 8766, 9130, bm0, d_cnt)
    enc["quantity"][0].scan_select_between(enc["quantity"][1], bit_pattern(np.iinfo(np.int32).min, np.int32), 23, bm1,
                                           d_cnt, bm0)
    m = (cols["shipdate"] >= 8766) & (cols["shipdate"] <= 9130) & (cols["quantity"] < 24)
    hits = int(d_cnt.download(np.uint64, len(counts)).sum())
    assert hits == int(m.sum())
    d_out = gpu_ctx.alloc(hits * 4 + 64)
    d_ids = gpu_ctx.alloc(hits * 8 + 64)
    got = enc["price"][0].unpack_selected(enc["price"][1], bm1, d_out, d_ids)
    assert got == hits
    assert np.array_equal(d_out.download(np.int32, hits), cols["price"][m])
    assert np.array_equal(d_ids.download(np.uint64, hits), np.flatnonzero(m).astype(np.uint64))


@pytest.mark.parametrize("dtype", [np.uint64, np.int32, np.uint16, np.uint8])
def test_select_bitmap_words_shared_by_many_groups(adac, oracle, gpu_ctx, dtype):
    """Dense value space (segments back to back): the scan writes the words inside a group whole and leaves a record
    for every word it shares; k_sel_merge_edges ORs the records of a word.  Runs of tiny segments put up to 32 groups
    into one bitmap word, segments that end exactly on word boundaries leave unused record slots between them, and
    the last word may be the odd half of a 64-bit word."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(4242 + dtype.itemsize)
    tile = adac.tile_values(dtype)
    counts = ([1] * 70 + [32, 32, 64, 1, 31, 33, 2, 30, 5] + [int(x) for x in rng.integers(1, 40, size=150)] +
              [2 * tile, 3, tile + 17, 1, 1, 1, 64, 5 * tile + 31, 7] + [int(x) for x in rng.integers(1, 6, size=200)] + [29])
    segs = [make_values(rng, dtype, n, int(rng.integers(1, min(8 * dtype.itemsize, 20)))) for n in counts]
    counts = np.array(counts, dtype=np.uint32)
    lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs)
    offs = np.concatenate([[0], np.cumsum(counts[:-1])]).astype(np.int64).tolist()
    span = int(counts.sum())
    assert lay.value_span == span
    info = np.iinfo(dtype)
    big = segs[int(np.argmax(counts))]
    probes = [(int(info.min), int(info.max)), (int(np.median(big)), int(info.max)), (int(info.min), int(np.median(big))),
              (int(big[0]), int(big[0]))]
    check_select(adac, gpu_ctx, lay, d_words, dtype, segs, offs, span, probes)
    valid = rng.random(span) > 0.3
    check_select(adac, gpu_ctx, lay, d_words, dtype, segs, offs, span, probes[:2], valid)


@pytest.mark.parametrize("dtype", [np.uint64, np.int32, np.uint16, np.int8])
def test_scans_finish_their_results_inside_the_kernel(adac, oracle, gpu_ctx, dtype):
    """Arrival cells (round 3): SUM / COUNT / selection leave every per-segment result and every bitmap word STORED by
    the last group that contributes to it — no clearing pass before the scan, no merge kernel after it.  So: outputs
    poisoned before every call, segments without rows in between (they have no group: cleared by the call), segments
    of one row up to several scan groups, widths 2 and 3 (the narrow kernel) next to the common kernel's, calls
    repeated (the cells must be back at zero), and the clearing-pass form (scan_cells = 0) as the cross-check."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(977 + dtype.itemsize)
    tile = adac.tile_values(dtype)
    counts = [0, 1, 1, 0, 33, 2, 5 * tile + 3, 0, 0, 31, 14 * tile, 1, 64, 17 * tile + 1, 0, 3, 2 * tile, 7, 0]
    bits = [1, 1, 2, 1, 3, 2, 2, 1, 1, 3, 7, 1, 5, 3, 1, 2, 6, 2, 1]
    segs = [make_values(rng, dtype, n, min(b, 8 * dtype.itemsize)) if n else np.zeros(0, dtype) for n, b in zip(counts, bits)]
    counts = np.array(counts, dtype=np.uint32)
    lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs)
    offs = np.concatenate([[0], np.cumsum(counts[:-1])]).astype(np.int64).tolist()
    span = int(counts.sum())
    nw = (span + 63) // 64
    nseg = len(segs)
    poison = np.full(nseg, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    d_res = gpu_ctx.alloc(nseg * 8)
    d_bm = gpu_ctx.alloc(nw * 8 + 8)
    info = np.iinfo(dtype)
    big = segs[int(np.argmax(counts))]
    probes = [(int(info.min), int(info.max)), (int(np.median(big)), int(info.max)), (int(big[0]), int(big[0]))]
    try:
        for cells in (1, 0, 1):
            adac.set_tuning("scan_cells", cells)
            for group in (0, 3):
                adac.set_tuning("scan_tiles_per_wg", group)
                for rep in range(3):
                    d_res.upload(poison)
                    lay.scan_sum(d_words, d_res)
                    assert d_res.download(np.uint64, nseg).tolist() == [wide_sum(v) for v in segs], (cells, group, rep)
                    # the same with every other group sending the halves of its sum the other way round, so that two
                    # parties of a segment complete one word each and meet at the third (arrive_sum's rare branch)
                    adac.set_tuning("sel_debug", 7)
                    d_res.upload(poison)
                    lay.scan_sum(d_words, d_res)
                    adac.set_tuning("sel_debug", 0)
                    assert d_res.download(np.uint64, nseg).tolist() == [wide_sum(v) for v in segs], (cells, group, rep, "swapped")
                    for lo, hi in probes:
                        exp = expected_bitmap(segs, offs, span, lo, hi)
                        want = [int(exp[o:o + len(v)].sum()) for v, o in zip(segs, offs)]
                        d_res.upload(poison)
                        lay.scan_count_between(d_words, bit_pattern(lo, dtype), bit_pattern(hi, dtype), d_res)
                        assert d_res.download(np.uint64, nseg).tolist() == want, (cells, group, rep, lo, hi)
                        d_res.upload(poison)
                        d_bm.upload(np.full(nw + 1, 0xDEADBEEFDEADBEEF, dtype=np.uint64))
                        lay.scan_select_between(d_words, bit_pattern(lo, dtype), bit_pattern(hi, dtype), d_bm, d_res)
                        assert d_res.download(np.uint64, nseg).tolist() == want, (cells, group, rep, lo, hi)
                        got = d_bm.download(np.uint64, nw + 1)
                        assert int(got[nw]) == 0xDEADBEEFDEADBEEF
                        allbits = np.unpackbits(got[:nw].view(np.uint8), bitorder="little")
                        assert np.array_equal(allbits[:span].astype(bool), exp), (cells, group, rep, lo, hi)
                        assert not allbits[span:].any()
                    # an empty range stores zeros everywhere
                    d_res.upload(poison)
                    lay.scan_count_between(d_words, bit_pattern(5, dtype), bit_pattern(4, dtype), d_res)
                    assert not d_res.download(np.uint64, nseg).any()
    finally:
        adac.set_tuning("scan_cells", 1)
        adac.set_tuning("sel_debug", 0)
        adac.set_tuning("scan_tiles_per_wg", 0)


@pytest.mark.parametrize("dtype", [np.uint64, np.uint32])
def test_unpack_selected_follows_the_descriptors_of_a_re_encoded_layout(adac, oracle, gpu_ctx, dtype):
    """The gather reads expanded tile records (the tile with its segment's descriptor folded in, round 3): they must be
    rebuilt whenever the descriptors change — a second adac_encode of the same layout with other widths and arena
    offsets, a third through the three kernels, and adac_layout_set_descs — and both the record form and the
    tile entry -> descriptor form (knob tile_records = 0) must agree, clustered and scattered selections alike."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(31 + dtype.itemsize)
    tile = adac.tile_values(dtype)
    counts = np.array([3 * tile + 5, tile, 7, 2 * tile - 1, 5 * tile + 100], dtype=np.uint32)
    n = int(counts.sum())
    lay = adac.Layout(gpu_ctx, dtype, counts)
    d_words = gpu_ctx.alloc(lay.max_arena_words * 8 + 64).zero()
    d_out = gpu_ctx.alloc(n * dtype.itemsize + 64)
    d_ids = gpu_ctx.alloc(n * 8 + 64)
    sels = [rng.random(n) < 0.4, np.zeros(n, dtype=bool)]
    sels[1][n // 3:n // 3 + 2 * tile + 17] = True      # clustered: most tiles hold no selected row
    try:
        for step, bits in enumerate((5, 27, 13, 9)):
            vals = np.concatenate([make_values(rng, dtype, int(c), bits + (i % 3)) for i, c in enumerate(counts)])
            d_vals = gpu_ctx.upload(vals)
            adac.set_tuning("single_pass_encode", 0 if step == 2 else 1)
            lay.encode(d_vals, d_words, None, adac.RULE_APPEND, False)
            if step == 3:  # the same descriptors handed back through adac_layout_set_descs
                lay.set_descs(lay.get_descs())
            for sel in sels:
                d_bm = gpu_ctx.upload(pack_mask(sel, n)[:(n + 63) // 64])
                keep = np.flatnonzero(sel)
                for records in (1, 0, 1):
                    adac.set_tuning("tile_records", records)
                    got = lay.unpack_selected(d_words, d_bm, d_out, d_ids)
                    assert got == len(keep), (step, records)
                    assert np.array_equal(d_out.download(dtype, max(got, 1))[:got], vals[keep]), (step, records)
                    assert np.array_equal(d_ids.download(np.uint64, max(got, 1))[:got], keep.astype(np.uint64)), (step, records)
    finally:
        adac.set_tuning("tile_records", 1)
        adac.set_tuning("single_pass_encode", 1)
