"""GPU parity tests: the HIP path, called through the C ABI of libadacodec.so, against the CPU oracle on the
same seeded inputs (bit-exact: this is integer / bit work) and against the committed known answers.
Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL_DTYPES = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]
U64 = 0xFFFFFFFFFFFFFFFF


def make_values(rng, dtype, n, bits, base=None):
    """n values of `dtype` whose range (max - min) needs exactly `bits` bits (for n >= 2)."""
    dtype = np.dtype(dtype)
    tb = 8 * dtype.itemsize
    bits = min(bits, tb)
    top = 2 ** bits
    span = rng.integers(0, top, size=n, dtype=np.uint64)
    if n >= 2:  # pin both ends so the width is predictable
        i0, i1 = rng.choice(n, size=2, replace=False)
        span[i0] = 0
        span[i1] = top - 1
    if base is None:
        hi = 2 ** tb - top
        base = int(rng.integers(0, hi + 1, dtype=np.uint64)) if hi > 0 else 0
    vals = (span + np.uint64(base)) & np.uint64(2 ** tb - 1)
    return vals.astype(np.dtype("u%d" % dtype.itemsize)).view(dtype)


def oracle_encode(orc, seg_vals, rule, padded, validity=None, val_offs=None):
    """Per segment: (min, max, width, packed?, words) exactly as the reference's two encode paths."""
    out = []
    for i, v in enumerate(seg_vals):
        tb = 8 * v.dtype.itemsize
        vb0 = 0 if val_offs is None else int(val_offs[i])
        mn, mx = orc.analyze_flat(v, rule, validity, vb0)
        w = (orc.width_from_succinct if rule == 0 else orc.width_from_uncompressed)(mn, mx, padded)
        packed = tb > w
        if not packed:
            w = tb
        words = orc.pack_flat(v, mn if packed else U64, w, validity, vb0)
        out.append((mn, mx, w, packed, words))
    return out


def wide_sum(v):
    """What adac_scan_sum reports: the values widened to 64 bits by T's signedness, summed mod 2^64."""
    wide = np.int64 if v.dtype.kind == "i" else np.uint64
    return int(v.astype(wide).sum(dtype=wide)) & 0xFFFFFFFFFFFFFFFF


def run_encode_decode(adac, orc, ctx, dtype, counts, seg_vals, rule=0, padded=False, validity=None, val_offs=None):
    dtype = np.dtype(dtype)
    lay = adac.Layout(ctx, dtype, counts, val_offs)
    span = max(lay.value_span, 1)
    host_vals = np.zeros(span, dtype=dtype)
    offs = np.cumsum(np.concatenate([[0], counts[:-1]]).astype(np.uint64)) if val_offs is None else val_offs
    for v, o in zip(seg_vals, offs):
        host_vals[int(o):int(o) + len(v)] = v
    d_vals = ctx.upload(host_vals)
    d_valid = None if validity is None else ctx.upload(validity)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
    lay.encode(d_vals, d_words, d_valid, rule, padded)
    descs = lay.get_descs()
    mm = lay.get_minmax()
    exp = oracle_encode(orc, seg_vals, rule, padded, validity, offs)
    words_all = d_words.download(np.uint64, lay.max_arena_words)
    woff = 0
    for s, (mn, mx, w, packed, words) in enumerate(exp):
        d = descs[s]
        assert int(d["count"]) == len(seg_vals[s])
        if len(seg_vals[s]):
            assert (int(mm[s, 0]), int(mm[s, 1])) == (mn, mx), "min/max of segment %d" % s
        assert int(d["width"]) == w, "width of segment %d" % s
        assert bool(d["flags"] & adac.SEG_PACKED) == packed
        assert int(d["word_off"]) == woff and woff % 16 == 0
        if packed:
            assert int(d["min"]) == adac.stored_min(mn, mx, w)   # mn, except for the all-ones sentinel collision
        got = words_all[woff:woff + len(words)]
        assert np.array_equal(got, words), "packed words of segment %d (w=%d)" % (s, w)
        assert adac.size_in_bytes(len(seg_vals[s]), w) == orc.size_in_bytes(len(seg_vals[s]) * w)
        woff += adac.arena_words(len(seg_vals[s]), w)
    # decode
    d_out = ctx.alloc(span * dtype.itemsize + 16)
    ctx_fill = np.full(span, 0x5A, dtype=np.uint8).repeat(dtype.itemsize).view(dtype)[:span]
    d_out.upload(ctx_fill)
    lay.unpack(d_words, d_out)
    ctx.sync()
    out = d_out.download(dtype, span)
    for s, (mn, mx, w, packed, words) in enumerate(exp):
        o = int(offs[s])
        n = len(seg_vals[s])
        smin = adac.stored_min(mn, mx, w) if packed else mn
        add = smin if (packed and smin != U64) else 0
        ref = orc.unpack_flat(words, 0, n, w, add, dtype)
        assert np.array_equal(out[o:o + n], ref), "decode of segment %d (w=%d)" % (s, w)
        if validity is None:
            assert np.array_equal(out[o:o + n], seg_vals[s]), "round trip of segment %d" % s
    return lay, d_words, d_out, descs, out


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_encode_decode_all_types(adac, oracle, gpu_ctx, dtype):
    rng = np.random.default_rng(1234 + np.dtype(dtype).itemsize)
    tb = 8 * np.dtype(dtype).itemsize
    tile = adac.tile_values(dtype)
    counts = np.array([1, 63, 64, 65, 2047, 2048, 2049, tile - 1, tile, tile + 1, 3 * tile + 17, 5],
                      dtype=np.uint32)
    bits = [1, 2, 3, 5, 7, 8, 9, 13, 16, 17, 24, 31, 32, 33, 47, 63, 64]
    seg_vals = [make_values(rng, dtype, int(c), bits[i % len(bits)] if bits[i % len(bits)] <= tb else tb - 1)
                for i, c in enumerate(counts)]
    for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
        for padded in (False, True):
            run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, rule, padded)


@pytest.mark.parametrize("dtype", [np.uint64, np.uint32, np.uint16, np.uint8])
def test_every_width(adac, oracle, gpu_ctx, dtype):
    """Every width 1..8*sizeof(T): ragged counts straddling tile and word boundaries."""
    rng = np.random.default_rng(99)
    tb = 8 * np.dtype(dtype).itemsize
    tile = adac.tile_values(dtype)
    counts, seg_vals = [], []
    for w in range(1, tb + 1):
        n = int(rng.integers(2, 2 * tile + 100))
        counts.append(n)
        seg_vals.append(make_values(rng, dtype, n, w))
    counts = np.array(counts, dtype=np.uint32)
    lay, d_words, d_out, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
    widths = descs["width"].tolist()
    assert widths[:tb - 1] == list(range(1, tb))  # w == tb cannot shrink: stays unpacked at tb
    assert widths[tb - 1] == tb and not (descs["flags"][tb - 1] & adac.SEG_PACKED)
    # the fused scans at every width, in both kernel forms (width-templated registers / LDS image)
    udt = np.dtype("u%d" % np.dtype(dtype).itemsize)
    exp_sum = [wide_sum(v) for v in seg_vals]
    keys = [seg_vals[5][3], seg_vals[tb // 2][0]]
    d_res = gpu_ctx.alloc(len(counts) * 8)
    try:
        for templated in (1, 0):
            adac.set_tuning("templated_scan", templated)
            for group in (1, 3, 16):
                adac.set_tuning("scan_tiles_per_wg", group)
                lay.scan_sum(d_words, d_res)
                assert d_res.download(np.uint64, len(counts)).tolist() == exp_sum, (templated, group)
                for kv in keys:
                    lay.scan_count_eq(d_words, int(np.array([kv]).view(udt)[0]), d_res)
                    got = d_res.download(np.uint64, len(counts)).tolist()
                    assert got == [int((v == kv).sum()) for v in seg_vals], (templated, group)
    finally:
        adac.set_tuning("templated_scan", 1)
        adac.set_tuning("scan_tiles_per_wg", 0)


def test_signed_and_mixed_sign(adac, oracle, gpu_ctx):
    """Sign-extended min/max of the append path (succinct.cpp:286-287): all-negative segments pack, mixed-sign
    segments cannot (range spans 2^64) and must still round-trip (the reference does not: SURVEY.md §4-1)."""
    rng = np.random.default_rng(7)
    for dtype in (np.int8, np.int16, np.int32, np.int64):
        info = np.iinfo(dtype)
        neg = rng.integers(max(info.min, -1000), -1, size=5000).astype(dtype)
        mixed = rng.integers(-100, 100, size=5000).astype(dtype)
        extremes = np.array([info.min, info.max, 0, -1, 1] * 100, dtype=dtype)
        counts = np.array([len(neg), len(mixed), len(extremes)], dtype=np.uint32)
        _, _, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, [neg, mixed, extremes],
                                              adac.RULE_APPEND)
        assert descs["flags"][0] & adac.SEG_PACKED
        assert not (descs["flags"][1] & adac.SEG_PACKED)
        # the recompaction path sees zero-extended raw values
        run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, [neg, mixed, extremes], adac.RULE_RECOMPACT)


def test_nulls_validity_mask(adac, oracle, gpu_ctx):
    """NULL rows: excluded from min/max on the append path, stored as NullValue<T> - min
    (succinct.cpp:276-292); included as NullValue<T> on the recompaction path."""
    rng = np.random.default_rng(11)
    for dtype in (np.uint32, np.int32, np.uint64, np.int16):
        counts = np.array([3000, 70, 9000, 64], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), 11, base=5000) for c in counts]
        total = int(counts.sum())
        valid_bits = rng.random(total) > 0.3
        valid_bits[3000:3070] = False  # an all-NULL segment: min stays UINT64_MAX, nothing subtracted
        validity = np.packbits(valid_bits, bitorder="little")
        validity = np.concatenate([validity, np.zeros((-len(validity)) % 8 + 8, dtype=np.uint8)]).view(np.uint64)
        for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
            _, _, _, descs, out = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, rule,
                                                    validity=validity)
            flat = np.concatenate(seg_vals)
            if rule == adac.RULE_APPEND:
                assert int(descs["min"][1]) == U64
                ok = valid_bits.copy()
                ok[3000:3070] = False
                assert np.array_equal(out[:total][ok], flat[ok])


def test_constant_and_empty_segments(adac, oracle, gpu_ctx):
    dtype = np.uint32
    counts = np.array([0, 100, 0, 4096, 1, 0], dtype=np.uint32)
    seg_vals = [np.zeros(0, dtype), np.full(100, 42, dtype), np.zeros(0, dtype), np.full(4096, 7, dtype),
                np.array([123456], dtype), np.zeros(0, dtype)]
    _, _, _, d_app, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, adac.RULE_APPEND)
    assert d_app["width"].tolist()[1] == 1 and d_app["width"].tolist()[3] == 1   # hi(0)+1
    _, _, _, d_rec, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, adac.RULE_RECOMPACT)
    assert d_rec["width"].tolist()[1] == 6 and d_rec["width"].tolist()[3] == 3   # max not reduced when max == min


def test_unaligned_value_offsets(adac, oracle, gpu_ctx):
    """Segments placed at arbitrary element offsets: every 16-byte store/load stays aligned internally."""
    rng = np.random.default_rng(5)
    for dtype in (np.uint8, np.uint16, np.uint32, np.uint64):
        counts = np.array([1000, 37, 5000, 2048, 1], dtype=np.uint32)
        gaps = [3, 1, 7, 5, 2]
        offs, run = [], 0
        for c, g in zip(counts, gaps):
            run += g
            offs.append(run)
            run += int(c)
        seg_vals = [make_values(rng, dtype, int(c), 6) for c in counts]
        run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, val_offs=np.array(offs, dtype=np.uint64))


def test_known_answers_on_gpu(adac, oracle, gpu_ctx, golden):
    """K1..K6 of SURVEY.md §8c through the device encode: identical min, width and packed words."""
    for k in golden["sdsl"]:
        dtype = np.dtype(k["dtype"])
        vals = np.array(k["values"]).astype(dtype)
        counts = np.array([len(vals)], dtype=np.uint32)
        lay = adac.Layout(gpu_ctx, dtype, counts)
        d_vals = gpu_ctx.upload(vals)
        d_words = gpu_ctx.alloc(lay.max_arena_words * 8).zero()
        lay.encode(d_vals, d_words, None, adac.RULE_APPEND, k["padded"])
        d = lay.get_descs()[0]
        assert int(d["min"]) == int(k["min"], 16)
        assert int(d["width"]) == k["width"]
        nw = adac.packed_words(len(vals), k["width"])
        assert nw * 64 >= k["bit_size"] and adac.size_in_bytes(len(vals), k["width"]) == k["size_in_bytes"]
        got = d_words.download(np.uint64, nw)
        assert [int(x) for x in got] == [int(x, 16) for x in k["words"]]
        d_out = gpu_ctx.alloc(64)
        lay.unpack(d_words, d_out)
        assert d_out.download(dtype, len(vals)).tolist() == k["values"]


def test_decode_of_oracle_packed_segments(adac, oracle, gpu_ctx):
    """Decode-only entry: descriptors + words produced elsewhere (here: by the oracle's segment model, i.e.
    Append + Compact as the reference runs them) uploaded with adac_layout_set_descs."""
    rng = np.random.default_rng(21)
    dtype = np.dtype(np.uint64)
    segs, counts = [], []
    for n, bits in [(32767, 32), (2048, 11), (22531, 17), (16386, 40), (100, 64)]:
        v = make_values(rng, dtype, n, bits)
        s = oracle.Segment(dtype, segment_size=max(n, 2048) * 8)
        for off in range(0, n, 2048):
            s.append(v, offset=off, count=min(2048, n - off))
        s.compact()
        segs.append((s, v))
        counts.append(n)
    counts = np.array(counts, dtype=np.uint32)
    lay = adac.Layout(gpu_ctx, dtype, counts)
    descs = np.zeros(len(segs), dtype=adac.SEGMENT_DESC_DTYPE)
    arena, woff, voff = [], 0, 0
    for i, (s, v) in enumerate(segs):
        w = s.width
        descs[i] = (woff, voff, s.min_factor, len(v), w, adac.SEG_PACKED if w < 64 else 0, 0)
        fp = adac.arena_words(len(v), w)
        buf = np.zeros(fp, dtype=np.uint64)
        words = s.words[:adac.packed_words(len(v), w)]
        buf[:len(words)] = words
        arena.append(buf)
        woff += fp
        voff += len(v)
    lay.set_descs(descs)
    d_words = gpu_ctx.upload(np.concatenate(arena))
    d_out = gpu_ctx.alloc(int(counts.sum()) * 8)
    lay.unpack(d_words, d_out)
    out = d_out.download(dtype, int(counts.sum()))
    assert np.array_equal(out, np.concatenate([v for _, v in segs]))


def test_scan_vector_and_scan_partial_ranges(adac, oracle, gpu_ctx):
    """adac_unpack_range = the scan_vector / scan_partial slots: 2048-row vectors, ragged tails, arbitrary
    starts and result offsets (succinct.cpp:123-144,232-240)."""
    rng = np.random.default_rng(3)
    for dtype, bits in ((np.uint64, 27), (np.uint32, 13), (np.uint16, 5), (np.uint8, 3), (np.uint64, 40)):
        dtype = np.dtype(dtype)
        n = 32767 if dtype.itemsize == 8 else 65534
        counts = np.array([2048, n], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), bits) for c in counts]
        lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
        cases = [(1, 0, 2048, 0), (1, 2048, 2048, 0), (1, n - n % 2048, n % 2048, 0), (1, 1, 1, 0),
                 (1, 12345, 4097, 3), (0, 5, 100, 1), (1, 0, n, 0), (1, n - 1, 1, 7), (1, 777, 0, 0)]
        for seg, start, cnt, out_off in cases:
            d_out = gpu_ctx.alloc((cnt + out_off + 16) * dtype.itemsize)
            d_out.upload(np.full(cnt + out_off + 16, 0x77, dtype=dtype))
            lay.unpack_range(d_words, seg, start, cnt, d_out, out_off)
            got = d_out.download(dtype, cnt + out_off + 16)
            assert np.array_equal(got[out_off:out_off + cnt], seg_vals[seg][start:start + cnt])
            assert np.all(got[:out_off] == 0x77) and np.all(got[out_off + cnt:] == 0x77)  # nothing else written
        with pytest.raises(adac.AdacError):
            lay.unpack_range(d_words, 1, n - 5, 10, d_out, 0)
        with pytest.raises(adac.AdacError):
            lay.unpack_range(d_words, 2, 0, 1, d_out, 0)


def test_fetch_rows(adac, oracle, gpu_ctx):
    rng = np.random.default_rng(8)
    for dtype, bits in ((np.uint64, 33), (np.int32, 9), (np.uint16, 16), (np.uint8, 2)):
        dtype = np.dtype(dtype)
        counts = np.array([5000, 1, 2048, 30000], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), bits) for c in counts]
        lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
        k = 4000
        segs = rng.integers(0, len(counts), size=k).astype(np.uint32)
        rows = (rng.random(k) * counts[segs]).astype(np.uint32)
        d_out = gpu_ctx.alloc(k * dtype.itemsize)
        lay.fetch_rows(d_words, gpu_ctx.upload(segs), gpu_ctx.upload(rows), k, d_out)
        got = d_out.download(dtype, k)
        exp = np.array([seg_vals[s][r] for s, r in zip(segs, rows)], dtype=dtype)
        assert np.array_equal(got, exp)


def test_fused_scan_aggregates(adac, oracle, gpu_ctx):
    rng = np.random.default_rng(13)
    for dtype, bits in ((np.uint64, 32), (np.uint32, 20), (np.int16, 7), (np.uint8, 4), (np.int64, 50)):
        dtype = np.dtype(dtype)
        counts = np.array([40000, 2048, 1, 12345, 0, 65534 if dtype.itemsize < 8 else 32767], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), bits) for c in counts]
        lay, d_words, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
        d_res = gpu_ctx.alloc(len(counts) * 8)
        lay.scan_sum(d_words, d_res)
        sums = d_res.download(np.uint64, len(counts))
        udt = np.dtype("u%d" % dtype.itemsize)
        exp = [wide_sum(v) for v in seg_vals]
        assert sums.tolist() == exp
        key_val = seg_vals[0][17]
        key = int(np.array([key_val]).view(udt)[0])
        lay.scan_count_eq(d_words, key, d_res)
        cnts = d_res.download(np.uint64, len(counts))
        assert cnts.tolist() == [int((v == key_val).sum()) for v in seg_vals]


def test_argument_errors(adac, gpu_ctx):
    with pytest.raises(adac.AdacError) as e:
        adac.Layout(gpu_ctx, np.float32, [1])
    assert e.value.status == 2  # unsupported type, like InternalException in SuccinctFun::GetFunction
    lay = adac.Layout(gpu_ctx, np.uint32, np.array([10], dtype=np.uint32))
    buf = gpu_ctx.alloc(4096)
    with pytest.raises(adac.AdacError):
        lay.unpack(buf.ptr + 4, buf)  # misaligned arena
    bad = np.zeros(1, dtype=adac.SEGMENT_DESC_DTYPE)
    bad[0] = (8, 0, 0, 10, 5, 1, 0)  # word_off not a multiple of 16
    with pytest.raises(adac.AdacError):
        lay.set_descs(bad)


def test_large_single_segments(adac, oracle, gpu_ctx):
    """Segments far larger than DuckDB's 256 KiB blocks (the C ABI does not assume them): thousands of tiles per
    segment, bit offsets beyond 2^31 inside one segment, every kernel."""
    rng = np.random.default_rng(31)
    for dtype, n, bits in ((np.uint32, 9_000_001, 21), (np.uint64, 5_000_003, 47), (np.uint8, 20_000_000, 5)):
        dtype = np.dtype(dtype)
        counts = np.array([n, 3, 70_000], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), bits) for c in counts]
        lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
        assert int(descs["width"][0]) == bits
        d_res = gpu_ctx.alloc(len(counts) * 8)
        lay.scan_sum(d_words, d_res)
        udt = np.dtype("u%d" % dtype.itemsize)
        assert d_res.download(np.uint64, 3).tolist() == [wide_sum(v) for v in seg_vals]
        kv = seg_vals[0][n - 2]
        lay.scan_count_eq(d_words, int(np.array([kv]).view(udt)[0]), d_res)
        assert d_res.download(np.uint64, 3).tolist() == [int((v == kv).sum()) for v in seg_vals]
        for start, cnt in ((n - 5000, 5000), (4_194_304 - 7, 100_000), (0, 1)):
            d_out = gpu_ctx.alloc(cnt * dtype.itemsize + 64)
            lay.unpack_range(d_words, 0, start, cnt, d_out, 1)
            assert np.array_equal(d_out.download(dtype, cnt + 1)[1:], seg_vals[0][start:start + cnt])


def test_range_predicates_on_packed_columns(adac, oracle, gpu_ctx):
    """adac_scan_count_between: lo <= v <= hi in the column type's own order (signed for INT types), evaluated
    on packed fields; `==`, `<=`, `>=`, BETWEEN and empty ranges; packed, unpacked (mixed-sign) and constant
    segments; both fused-scan forms."""
    rng = np.random.default_rng(41)
    for dtype in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
        dtype = np.dtype(dtype)
        info = np.iinfo(dtype)
        udt = np.dtype("u%d" % dtype.itemsize)
        tb = 8 * dtype.itemsize
        span = min(tb - 1, 13)
        n = 30000
        base_pos = int(info.max // 2) - (1 << span)
        segs = [
            (base_pos + rng.integers(0, 1 << span, size=n)).astype(dtype),                       # packed, positive
            rng.integers(0, 1 << min(tb - 2, 5), size=4500).astype(dtype),                       # packed, narrow
            np.full(777, info.max // 3, dtype=dtype),                                            # constant
        ]
        if dtype.kind == "i":
            segs.append((int(info.min) + rng.integers(0, 1 << span, size=n)).astype(dtype))      # packed, negative
            segs.append(rng.integers(-50, 50, size=9000).astype(dtype))                          # mixed sign: unpacked
        else:
            segs.append(rng.integers(0, int(info.max), size=9000, dtype=np.uint64).astype(dtype))  # full range: unpacked
        counts = np.array([len(v) for v in segs], dtype=np.uint32)
        lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs)
        assert not (descs["flags"][-1] & adac.SEG_PACKED)
        d_res = gpu_ctx.alloc(len(counts) * 8)

        def bits(v):
            return int(np.array([v]).astype(dtype).view(udt)[0])

        probes = [(int(info.min), int(info.max)), (int(segs[0][5]), int(segs[0][5])), (base_pos + 100, base_pos + 900),
                  (int(info.min), base_pos + 4000), (base_pos + 4000, int(info.max)), (3, 17), (0, 0),
                  (int(info.max // 3), int(info.max // 3)), (10, 5)]
        if dtype.kind == "i":
            probes += [(-10, 10), (int(info.min), -1), (int(info.min) + 50, int(info.min) + 5000), (-1, 0)]
        clip = lambda x: min(max(x, int(info.min)), int(info.max))  # noqa: E731
        probes = [(clip(lo), clip(hi)) for lo, hi in probes]
        try:
            for templated in (1, 0):
                adac.set_tuning("templated_scan", templated)
                for lo, hi in probes:
                    lay.scan_count_between(d_words, bits(lo), bits(hi), d_res)
                    got = d_res.download(np.uint64, len(counts)).tolist()
                    exp = [int(((v >= lo) & (v <= hi)).sum()) for v in segs]
                    assert got == exp, (dtype, templated, lo, hi)
        finally:
            adac.set_tuning("templated_scan", 1)


def test_zonemaps(adac, gpu_ctx):
    """adac_zonemap: NumericStatistics-style typed min/max per segment over the valid rows (signed order for
    the INT types), the statistic RowGroup::CheckZonemapSegments skips segments with."""
    rng = np.random.default_rng(51)
    for dtype in (np.int8, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
        dtype = np.dtype(dtype)
        info = np.iinfo(dtype)
        counts = np.array([5000, 1, 70000, 300, 64], dtype=np.uint32)
        segs = [rng.integers(info.min, info.max, size=int(c), dtype=np.int64 if dtype.kind == "i" else np.uint64,
                             endpoint=True).astype(dtype) for c in counts]
        total = int(counts.sum())
        valid = rng.random(total) > 0.25
        valid[5001:75001][:10] = True
        valid[75301:] = False           # last segment: no valid row
        vmask = np.packbits(valid, bitorder="little")
        vmask = np.concatenate([vmask, np.zeros((-len(vmask)) % 8 + 8, np.uint8)]).view(np.uint64)
        lay = adac.Layout(gpu_ctx, dtype, counts)
        d_vals = gpu_ctx.upload(np.concatenate(segs))
        zm = lay.zonemap(d_vals)
        for s, v in enumerate(segs):
            assert (zm[s, 0], zm[s, 1]) == (v.min(), v.max()), (dtype, s)
        zm = lay.zonemap(d_vals, gpu_ctx.upload(vmask))
        off = 0
        for s, v in enumerate(segs):
            ok = valid[off:off + len(v)]
            off += len(v)
            if ok.any():
                assert (zm[s, 0], zm[s, 1]) == (v[ok].min(), v[ok].max()), (dtype, s)
            else:
                assert zm[s, 0] == info.max and zm[s, 1] == info.min  # empty interval


def test_fused_scans_skip_null_rows(adac, oracle, gpu_ctx):
    """adac_scan_sum_valid / adac_scan_count_between_valid: rows whose validity bit is clear take no part, in
    both scan forms, at many widths and ragged segment sizes (validity windows straddle mask words)."""
    rng = np.random.default_rng(61)
    for dtype, bits_list in ((np.uint64, (5, 13, 32, 47)), (np.int32, (4, 11, 20)), (np.uint16, (7, 12)), (np.uint8, (3, 6))):
        dtype = np.dtype(dtype)
        udt = np.dtype("u%d" % dtype.itemsize)
        tile = adac.tile_values(dtype)
        counts, segs = [], []
        for b in bits_list:
            for n in (int(rng.integers(100, 3 * tile)), 2 * tile, 977):
                counts.append(n)
                segs.append(make_values(rng, dtype, n, b))
        counts = np.array(counts, dtype=np.uint32)
        lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs)
        total = int(counts.sum())
        valid = rng.random(total) > 0.35
        valid[:70] = False
        vm = np.packbits(valid, bitorder="little")
        vm = np.concatenate([vm, np.zeros((-len(vm)) % 8 + 8, np.uint8)]).view(np.uint64)
        d_valid = gpu_ctx.upload(vm)
        d_res = gpu_ctx.alloc(len(counts) * 8)
        offs = [int(x) for x in np.concatenate([[0], np.cumsum(counts)[:-1]])]
        kv = segs[1][5]
        lo, hi = sorted((int(segs[0][3]), int(segs[0][9])))
        try:
            for templated in (1, 0):
                adac.set_tuning("templated_scan", templated)
                lay.scan_sum(d_words, d_res, d_valid)
                got = d_res.download(np.uint64, len(counts)).tolist()
                exp = [wide_sum(v[valid[o:o + len(v)]]) for v, o in zip(segs, offs)]
                assert got == exp, (dtype, templated)
                for a, b in ((kv, kv), (lo, hi)):
                    ab = int(np.array([a]).astype(dtype).view(udt)[0])
                    bb = int(np.array([b]).astype(dtype).view(udt)[0])
                    lay.scan_count_between(d_words, ab, bb, d_res, d_valid)
                    got = d_res.download(np.uint64, len(counts)).tolist()
                    exp = [int(((v >= a) & (v <= b) & valid[o:o + len(v)]).sum()) for v, o in zip(segs, offs)]
                    assert got == exp, (dtype, templated, a, b)
        finally:
            adac.set_tuning("templated_scan", 1)


def test_scans_on_segments_crossing_the_sign_boundary(adac, oracle, gpu_ctx):
    """BitCompressFromUncompressed orders values ZERO-extended (column_segment.cpp:405-420), so a signed segment
    such as {INT_MAX, INT_MIN} packs at w = 1 although min + field leaves T's signed range.  Fused scans must
    not treat such a segment as linear; all-negative and all-positive ones are."""
    rng = np.random.default_rng(77)
    for dtype in (np.int8, np.int16, np.int32, np.int64):
        dtype = np.dtype(dtype)
        info = np.iinfo(dtype)
        n = 50_000
        span = 200 if dtype.itemsize > 1 else 60
        udt = np.dtype("u%d" % dtype.itemsize)
        cross = ((np.uint64(info.max - span // 2) + rng.integers(0, span, size=n).astype(np.uint64))
                 & np.uint64((1 << (8 * dtype.itemsize)) - 1)).astype(udt).view(dtype)  # INT_MAX-ish .. INT_MIN-ish
        cross[0], cross[1] = info.max, info.min
        neg = rng.integers(-100, -3, size=n).astype(dtype)
        pos = rng.integers(5, 120, size=n).astype(dtype)
        edge = np.array([info.max, info.min], dtype=dtype)
        segs = [cross, neg, pos, edge]
        counts = np.array([len(v) for v in segs], dtype=np.uint32)
        lay, d_words, _, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs, adac.RULE_RECOMPACT)
        assert all(descs["flags"] & adac.SEG_PACKED) and int(descs["width"][3]) == 1
        assert (cross < 0).any() and (cross > 0).any()
        d_res = gpu_ctx.alloc(len(counts) * 8)
        try:
            for templated in (1, 0):
                adac.set_tuning("templated_scan", templated)
                lay.scan_sum(d_words, d_res)
                assert d_res.download(np.uint64, len(counts)).tolist() == [wide_sum(v) for v in segs], (dtype, templated)
                probes = [(info.min, info.min + 50), (info.max - 40, info.max), (-50, 50), (info.min, info.max),
                          (-100, -3), (int(info.max), int(info.max)), (int(info.min), int(info.min))]
                for a, b in probes:
                    ab = int(np.array([a]).astype(dtype).view(udt)[0])
                    bb = int(np.array([b]).astype(dtype).view(udt)[0])
                    lay.scan_count_between(d_words, ab, bb, d_res)
                    exp = [int(((v >= a) & (v <= b)).sum()) for v in segs]
                    assert d_res.download(np.uint64, len(counts)).tolist() == exp, (dtype, templated, a, b)
        finally:
            adac.set_tuning("templated_scan", 1)


def test_objects_outlive_their_context_safely(adac, oracle):
    """Contexts are reference-counted by the layouts / plans / graphs made on them: destroying the context first
    (and creating the next one right away) must leave everything usable and leak-free in any order."""
    ctx = adac.Context(0)
    vals = np.arange(100_000, dtype=np.uint32)
    counts = adac.appender_segment_counts(len(vals), 4)
    lay = adac.Layout(ctx, np.uint32, counts)
    d_vals = ctx.upload(vals)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
    lay.encode(d_vals, d_words)
    d_sum = ctx.alloc(len(counts) * 8)
    lay.scan_sum(d_words, d_sum)
    ctx.sync()
    assert int(d_sum.download(np.uint64, len(counts)).sum()) == int(vals.astype(np.uint64).sum())
    ctx.close()                     # the layout is still alive: it keeps the C context, buffers are freed here
    ctx2 = adac.Context(0)
    lay2 = adac.Layout(ctx2, np.uint32, counts)      # used to fail when a dangling context was touched before
    d2 = ctx2.upload(vals)
    w2 = ctx2.alloc(lay2.max_arena_words * 8 + 16).zero()
    lay2.encode(d2, w2)
    out = ctx2.alloc(len(vals) * 4 + 16)
    lay2.unpack(w2, out)
    assert np.array_equal(out.download(np.uint32, len(vals)), vals)
    lay.close()                     # last reference of the first context
    del lay2
    ctx2.close()


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.int64, np.uint64])
def test_all_ones_segment_survives_the_sentinel_collision(adac, oracle, gpu_ctx, dtype):
    """Reference defect 7: a segment whose every value is -1 (all ones) has min == max == UINT64_MAX, which the
    reference also uses for "no min": it packs without subtracting and scans without adding, returning 2^w - 1
    (1, or 255 when padded) instead of -1.  The product keeps those packed bits and decodes the original value,
    in the decode, the fused scans and the point fetch."""
    dtype = np.dtype(dtype)
    ones = np.array([-1 if dtype.kind == "i" else np.iinfo(dtype).max], dtype=dtype)[0]
    rng = np.random.default_rng(3)
    segs = [np.full(5000, ones, dtype=dtype), make_values(rng, dtype, 3000, 5), np.full(1, ones, dtype=dtype),
            np.full(70000, ones, dtype=dtype)]
    counts = np.array([len(v) for v in segs], dtype=np.uint32)
    for padded in (False, True):
        lay, d_words, d_out, descs, out = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs,
                                                            adac.RULE_APPEND, padded)
        w = 8 if padded else 1
        if w < 8 * dtype.itemsize:   # (int8 padded to 8 bits does not shrink: it stays unpacked)
            for s in (0, 2, 3):
                assert int(descs["width"][s]) == w and descs["flags"][s] & adac.SEG_PACKED
                assert int(descs["min"][s]) == 0xFFFFFFFFFFFFFFFF - ((1 << w) - 1)
            # the reference's own reading of the same bits (oracle, reference mode): 2^w - 1, not the value
            words0 = d_words.download(np.uint64, int(descs["word_off"][1]))
            assert int(oracle.unpack_flat(words0, 0, 1, w, 0, np.uint64)[0]) == (1 << w) - 1
        d_res = gpu_ctx.alloc(len(counts) * 8)
        lay.scan_sum(d_words, d_res)
        assert d_res.download(np.uint64, len(counts)).tolist() == [wide_sum(v) for v in segs]
        key = int(np.array([ones]).view(np.dtype("u%d" % dtype.itemsize))[0])
        lay.scan_count_eq(d_words, key, d_res)
        assert d_res.download(np.uint64, len(counts)).tolist() == [int((v == ones).sum()) for v in segs]
        d_f = gpu_ctx.alloc(8 * dtype.itemsize)
        lay.fetch_rows(d_words, gpu_ctx.upload(np.array([0, 3, 2], dtype=np.uint32)),
                       gpu_ctx.upload(np.array([4999, 69999, 0], dtype=np.uint32)), 3, d_f)
        assert d_f.download(dtype, 3).tolist() == [ones] * 3


def test_unpack_jobs_layout_free_batches(adac, oracle, gpu_ctx):
    """adac_unpack_jobs: many (segment, row range) jobs per launch, described inline (no layout), arbitrary output
    offsets and an output pointer that is not 16-byte aligned; more than 48 jobs exercises the launch grouping."""
    rng = np.random.default_rng(77)
    for dtype in (np.uint64, np.int32, np.uint16, np.int8):
        dtype = np.dtype(dtype)
        tb = 8 * dtype.itemsize
        tile = adac.tile_values(dtype)
        counts = np.array([int(rng.integers(1, 3 * tile)) for _ in range(40)] + [1, tile, tile + 1], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), int(rng.integers(1, tb + 1))) for c in counts]
        lay, d_words, d_out, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals)
        ranges, segs = [], []
        for _ in range(130):
            s = int(rng.integers(0, len(counts)))
            st = int(rng.integers(0, counts[s]))
            c = int(rng.integers(0, counts[s] - st + 1))
            segs.append(s)
            ranges.append((st, c))
        segs += [0, len(counts) - 1]
        ranges += [(0, int(counts[0])), (0, int(counts[-1]))]
        out_offs, run = [], 3
        for _, c in ranges:
            out_offs.append(run)
            run += c + int(rng.integers(0, 5))   # ragged, unaligned placement
        jobs = adac.jobs_from_descs([descs[s] for s in segs], ranges, out_offs)
        total = run + 8
        d_dst = gpu_ctx.alloc((total + 1) * dtype.itemsize + 64)
        d_dst.upload(np.full((total + 1) * dtype.itemsize + 64, 0x77, dtype=np.uint8))
        base = d_dst.ptr + dtype.itemsize            # element 1 of the buffer: misaligned for every type but u64... and u64
        adac.unpack_jobs(gpu_ctx, dtype, jobs, d_words, base)
        ev = adac.Event(gpu_ctx)
        ev.wait()
        assert ev.done()
        got = d_dst.download(np.uint8, (total + 1) * dtype.itemsize).view(dtype)[1:]
        expect = np.full(total, 0x77, dtype=np.uint8).repeat(dtype.itemsize).view(dtype).copy()
        for s, (st, c), o in zip(segs, ranges, out_offs):
            expect[o:o + c] = seg_vals[s][st:st + c]
        assert np.array_equal(got, expect), dtype
        with pytest.raises(adac.AdacError):   # a word offset that is not a multiple of 16
            bad = jobs[:1].copy()
            bad[0]["word_off"] += 1
            adac.unpack_jobs(gpu_ctx, dtype, bad, d_words, base)


@pytest.mark.parametrize("dtype", [np.uint64, np.int64, np.uint32, np.int32, np.uint16, np.int16, np.uint8])
def test_single_pass_encode_mixes_its_flows(adac, oracle, gpu_ctx, dtype):
    """The single-pass encode is persistent: one workgroup works through many segments and chooses per segment between
    the parked whole-dword flow (LDS parking area over the prefetch stage and the image), the staged image flow, the
    wide whole-dword flow and the unpacked copy.  Many more segments than CUs, of every size up to a full block, at
    random placements, with widths that alternate between the flows: the hand-over of the LDS pool (a parked segment
    dirties the image), of the prefetched rounds and of the early loads must never leak from one segment into the
    next.  Compared word for word with the oracle, both rules, padded on and off, and with the three-kernel form.
    (Every type is forced through the single-pass kernel here, also those adac_encode sends to the three kernels by
    default: profiles/r03_encode_forms.json, r03_encode_small_types.json.)"""
    dtype = np.dtype(dtype)
    tb = 8 * dtype.itemsize
    rng = np.random.default_rng(1234 + tb + (dtype.kind == "i"))
    full = 262136 // dtype.itemsize
    flow_widths = [tb // 2, 13, tb, tb // 4, 3 * tb // 4, 1, tb // 2 - 1, 7, tb // 2, 24 if tb == 32 else 48]
    nseg = 700
    sizes = rng.choice([0, 1, 2, 3, 63, 64, 65, 1000, 2047, 2048, 4097, 16386, 22531, full - 1, full], size=nseg,
                       p=[.03, .03, .03, .03, .04, .04, .04, .1, .1, .1, .1, .1, .1, .08, .08])
    counts = sizes.astype(np.uint32)
    gaps = rng.integers(0, 5, size=nseg)
    offs, run = [], 0
    for c, g in zip(counts, gaps):
        run += int(g)
        offs.append(run)
        run += int(c)
    seg_vals = [make_values(rng, dtype, int(c), flow_widths[int(rng.integers(0, len(flow_widths)))]) for c in counts]
    val_offs = np.array(offs, dtype=np.uint64)
    # knob 2 = the single-pass kernel whatever the type (by default adac_encode takes it for the 8-byte types only, and
    # for the 4-byte ones under first-come placement: profiles/r03_encode_forms.json)
    adac.set_tuning("single_pass_encode", 2)
    try:
        for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
            for padded in (False, True):
                run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, rule, padded, val_offs=val_offs)
        lay1, w1, _, d1, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, val_offs=val_offs)
        # (round 3) the big-image flow — a segment packed whole into the LDS pool, its stores deferred until the next
        # segment is analysed — and the deferred parked flow are on by default above; without them: the same bytes
        for knob in ("encode_big_image", "encode_publish_ahead"):
            adac.set_tuning(knob, 0)
            try:
                _, _, _, d0, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, val_offs=val_offs)
            finally:
                adac.set_tuning(knob, 1)
            assert d0.tobytes() == d1.tobytes(), knob
    finally:
        adac.set_tuning("single_pass_encode", 1)
    # the three-kernel form on the same column: identical descriptors and arena; and the default choice
    adac.set_tuning("single_pass_encode", 0)
    try:
        lay3, w3, _, d3, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, val_offs=val_offs)
    finally:
        adac.set_tuning("single_pass_encode", 1)
    assert d1.tobytes() == d3.tobytes()
    _, _, _, dd, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, seg_vals, val_offs=val_offs)
    assert dd.tobytes() == d3.tobytes()
    # first-come placement: same widths / mins / words per segment at offsets of its own choosing
    adac.set_tuning("encode_placement", 1)
    try:
        layf = adac.Layout(gpu_ctx, dtype, counts, val_offs)
        host_vals = np.zeros(max(layf.value_span, 1), dtype=dtype)
        for v, o in zip(seg_vals, offs):
            host_vals[o:o + len(v)] = v
        d_vals = gpu_ctx.upload(host_vals)
        d_words = gpu_ctx.alloc(layf.max_arena_words * 8 + 16).zero()
        layf.encode(d_vals, d_words)
        df = layf.get_descs()
    finally:
        adac.set_tuning("encode_placement", 0)
    for f in ("count", "width", "flags", "min", "val_off"):
        assert np.array_equal(df[f], d1[f]), f
    wf = d_words.download(np.uint64, layf.max_arena_words)
    w_ord = w1.download(np.uint64, lay1.max_arena_words)
    foot = ((df["count"].astype(np.uint64) * df["width"] + 64) >> np.uint64(6)) + np.uint64(15) & ~np.uint64(15)
    order = np.argsort(df["word_off"], kind="stable")
    ends = df["word_off"][order] + foot[order]
    assert np.all(df["word_off"] % 16 == 0) and np.all(ends[:-1] <= df["word_off"][order][1:])   # disjoint
    assert int(ends.max()) <= layf.max_arena_words
    for s in range(nseg):
        nw = int((int(df["count"][s]) * int(df["width"][s]) + 63) // 64)
        a, b = int(df["word_off"][s]), int(d1["word_off"][s])
        assert np.array_equal(wf[a:a + nw], w_ord[b:b + nw]), "words of segment %d" % s
    d_out = gpu_ctx.alloc(layf.value_span * dtype.itemsize + 64)
    layf.unpack(d_words, d_out)
    got = d_out.download(dtype, layf.value_span)
    for v, o in zip(seg_vals, offs):
        assert np.array_equal(got[o:o + len(v)], v)


def test_encode_form_follows_the_widths_the_host_last_saw(adac, oracle, gpu_ctx):
    """adac_encode on a 4-byte column with ordered placement: the three kernels while the host has not seen the layout's
    descriptors, the single pass once it has and the widest segment needs at most 17 bits (the big image publishes
    ahead there: profiles/r03_encode_big_image.json).  A hint, not a contract: whatever form runs, descriptors, min/max
    and every packed word are the same — checked across the switch, for a narrow column, a wide one and one whose
    widths change between the two encodes."""
    dtype = np.dtype(np.uint32)
    rng = np.random.default_rng(77)
    tile = adac.tile_values(dtype)
    counts = np.array([5 * tile + 3, 16 * tile - 2, 7, 16 * tile - 2, 3 * tile], dtype=np.uint32)
    n = int(counts.sum())
    lay = adac.Layout(gpu_ctx, dtype, counts)
    d_words = gpu_ctx.alloc(lay.max_arena_words * 8 + 64)
    for bits_first, bits_second in ((13, 13), (24, 24), (9, 22), (22, 9)):
        seen = []
        for bits in (bits_first, bits_second, bits_second):
            vals = np.concatenate([make_values(rng, dtype, int(c), bits) for c in counts])
            d_vals = gpu_ctx.upload(vals)
            d_words.zero()
            lay.encode(d_vals, d_words, None, adac.RULE_APPEND, False)   # the form follows the hint of the last get_descs
            descs = lay.get_descs()                                       # ... which this call refreshes
            words = d_words.download(np.uint64, lay.max_arena_words)
            adac.set_tuning("single_pass_encode", 0)                      # the three kernels, whatever the hint
            try:
                d_words.zero()
                lay.encode(d_vals, d_words, None, adac.RULE_APPEND, False)
                descs3 = lay.get_descs()
                words3 = d_words.download(np.uint64, lay.max_arena_words)
            finally:
                adac.set_tuning("single_pass_encode", 1)
            assert descs.tobytes() == descs3.tobytes(), (bits_first, bits_second, bits)
            assert np.array_equal(words, words3), (bits_first, bits_second, bits)
            d_out = gpu_ctx.alloc(n * 4 + 64)
            lay.unpack(d_words, d_out)
            assert np.array_equal(d_out.download(dtype, n), vals)
            seen.append(sorted(set(descs["width"].tolist())))
        assert max(seen[-1]) <= bits_second + 1
