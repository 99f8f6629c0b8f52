"""Seeded differential fuzz of the whole C ABI against the oracle and numpy: random types, segment shapes (empty,
single-row, tile-boundary +-1, multi-tile), widths, signs, value offsets, NULL masks, both reference rules and the
padding flag — encode (min / width / words vs the oracle), decode, range decode, point fetch, every fused scan in
both kernel forms, the selection bitmap and the packed -> packed re-compaction on the same random column."""
import numpy as np
import pytest

from test_gpu_parity import make_values, run_encode_decode, wide_sum
from test_gpu_repack import check_dst
from test_gpu_select import bit_pattern, expected_bitmap, pack_mask

pytestmark = pytest.mark.gpu

ALL = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]


def random_case(rng, adac):
    dtype = np.dtype(ALL[int(rng.integers(0, len(ALL)))])
    tb = 8 * dtype.itemsize
    tile = adac.tile_values(dtype)
    nseg = int(rng.integers(1, 9))
    counts, segs = [], []
    for _ in range(nseg):
        kind = int(rng.integers(0, 9))
        n = [0, 1, tile - 1, tile, tile + 1, int(rng.integers(2, 300)), int(rng.integers(300, 3 * tile)),
             int(rng.integers(tile, 6 * tile)),
             int(rng.integers(8 * tile, 40 * tile)) if rng.random() < 0.25 else 16 * tile - 1][kind]  # several scan groups
        w = int(rng.integers(1, tb + 1))
        v = make_values(rng, dtype, n, w)
        if dtype.kind == "i" and rng.random() < 0.3 and n:   # a negative frame of reference
            v = (v.astype(np.int64) % (1 << min(w, tb - 1)) - int(rng.integers(1, 1 << (tb - 1)))).astype(dtype) \
                if tb < 64 else v
        counts.append(n)
        segs.append(v)
    counts = np.array(counts, dtype=np.uint32)
    gaps = rng.integers(0, 5, size=nseg) * (16 // dtype.itemsize) if rng.random() < 0.5 else rng.integers(0, 40, size=nseg)
    offs, run = [], 0
    for c, g in zip(counts, gaps):
        run += int(g)
        offs.append(run)
        run += int(c)
    return dtype, counts, segs, np.array(offs, dtype=np.uint64), run


@pytest.mark.parametrize("seed", range(120))
def test_random_columns(adac, oracle, gpu_ctx, seed):
    rng = np.random.default_rng(10_000 + seed)
    dtype, counts, segs, offs, span = random_case(rng, adac)
    rule = adac.RULE_APPEND if rng.random() < 0.6 else adac.RULE_RECOMPACT
    padded = bool(rng.random() < 0.3)
    valid = None
    vm = None
    if rng.random() < 0.4 and span:
        valid = rng.random(span) > rng.random() * 0.8
        vm = pack_mask(valid, span)
    lay, d_words, d_out, descs, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs, rule, padded, vm, offs)
    udt = np.dtype("u%d" % dtype.itemsize)
    nz = [s for s, c in enumerate(counts) if c]
    if not nz:
        return
    d_valid = None if vm is None else gpu_ctx.upload(vm)
    # what the column decodes to (NULL slots hold NullValue<T> after an encode with a validity mask)
    dec_all = d_out.download(dtype, max(span, 1))
    dec = [dec_all[int(o):int(o) + int(c)] for o, c in zip(offs, counts)]
    for s in nz:
        ok = np.ones(len(segs[s]), bool) if valid is None else valid[int(offs[s]):int(offs[s]) + len(segs[s])]
        assert np.array_equal(dec[s][ok], segs[s][ok])
    # range decode + point fetch
    s = nz[int(rng.integers(0, len(nz)))]
    c = int(counts[s])
    start = int(rng.integers(0, c))
    cnt = int(rng.integers(1, c - start + 1))
    d_rng = gpu_ctx.alloc(cnt * dtype.itemsize + 64)
    shift = int(rng.integers(0, 3)) * (16 // dtype.itemsize)
    lay.unpack_range(d_words, s, start, cnt, d_rng, shift)
    assert np.array_equal(d_rng.download(dtype, cnt + shift)[shift:], dec[s][start:start + cnt])
    k = 64
    fs = np.array([nz[int(i)] for i in rng.integers(0, len(nz), size=k)], dtype=np.uint32)
    fr = np.array([int(rng.integers(0, counts[int(x)])) for x in fs], dtype=np.uint32)
    d_f = gpu_ctx.alloc(k * dtype.itemsize)
    lay.fetch_rows(d_words, gpu_ctx.upload(fs), gpu_ctx.upload(fr), k, d_f)
    assert np.array_equal(d_f.download(dtype, k), np.array([dec[int(a)][int(b)] for a, b in zip(fs, fr)], dtype=dtype))
    # fused scans (both kernel forms), selection bitmap
    d_res = gpu_ctx.alloc(len(counts) * 8)
    d_bm = gpu_ctx.alloc((span + 63) // 64 * 8 + 8)
    pool = np.concatenate([dec[x] for x in nz])
    a, b = sorted((int(pool[int(rng.integers(0, len(pool)))]), int(pool[int(rng.integers(0, len(pool)))])))
    info = np.iinfo(dtype)
    probes = [(a, b), (a, a), (int(info.min), b), (a, int(info.max))]
    try:
        for templated in (1, 0):
            adac.set_tuning("templated_scan", templated)
            adac.set_tuning("scan_tiles_per_wg", int(rng.integers(1, 20)))
            lay.scan_sum(d_words, d_res, d_valid)
            exp = [wide_sum(v if valid is None else v[valid[int(o):int(o) + len(v)]]) for v, o in zip(dec, offs)]
            assert d_res.download(np.uint64, len(counts)).tolist() == exp, ("sum", templated)
            for lo, hi in probes:
                expb = expected_bitmap(dec, [int(o) for o in offs], span, lo, hi, valid)
                expc = [int(expb[int(o):int(o) + len(v)].sum()) for v, o in zip(dec, offs)]
                lay.scan_count_between(d_words, bit_pattern(lo, dtype), bit_pattern(hi, dtype), d_res, d_valid)
                assert d_res.download(np.uint64, len(counts)).tolist() == expc, ("count", templated, lo, hi)
                lay.scan_select_between(d_words, bit_pattern(lo, dtype), bit_pattern(hi, dtype), d_bm, d_res, d_valid)
                assert d_res.download(np.uint64, len(counts)).tolist() == expc, ("select count", templated, lo, hi)
                got = np.unpackbits(d_bm.download(np.uint64, (span + 63) // 64).view(np.uint8), bitorder="little")
                assert np.array_equal(got[:span].astype(bool), expb) and not got[span:].any(), ("bitmap", templated, lo, hi)
    finally:
        adac.set_tuning("templated_scan", 1)
        adac.set_tuning("scan_tiles_per_wg", 0)
    # scan with selection: the last bitmap's rows, values and element ids, dense and in row order
    lo, hi = probes[-1]
    expb = expected_bitmap(dec, [int(o) for o in offs], span, lo, hi, valid)
    nsel = int(expb.sum())
    d_sel = gpu_ctx.alloc(max(nsel, 1) * dtype.itemsize + 64)
    d_ids = gpu_ctx.alloc(max(nsel, 1) * 8 + 64)
    assert lay.unpack_selected(d_words, d_bm, d_sel, d_ids) == nsel
    ids = np.flatnonzero(expb)
    assert np.array_equal(d_ids.download(np.uint64, max(nsel, 1))[:nsel], ids.astype(np.uint64))
    flat = np.zeros(span, dtype=dtype)
    for v, o in zip(dec, offs):
        flat[int(o):int(o) + len(v)] = v
    assert np.array_equal(d_sel.download(dtype, max(nsel, 1))[:nsel], flat[ids])
    # typed zonemaps of the raw values (valid rows only)
    d_raw = gpu_ctx.upload(flat if span else np.zeros(1, dtype))
    zm = lay.zonemap(d_raw, d_valid)
    for s in nz:
        v = dec[s] if valid is None else dec[s][valid[int(offs[s]):int(offs[s]) + len(dec[s])]]
        if len(v):
            assert (int(zm[s, 0]), int(zm[s, 1])) == (int(v.min()), int(v.max())), ("zonemap", s)
    # packed -> packed with the other padding choice: identical to a direct encode of what the column decodes to
    dst = adac.Layout(gpu_ctx, dtype, counts, offs)
    d_dst = gpu_ctx.alloc(dst.max_arena_words * 8 + 16).zero()
    lay.reencode(d_words, dst, d_dst, d_valid, rule, not padded)
    exp_descs = dst.get_descs()
    src_dec = [np.ascontiguousarray(v) for v in dec]
    from test_gpu_parity import oracle_encode
    exp = oracle_encode(oracle, src_dec, rule, not padded, vm, offs)
    words = d_dst.download(np.uint64, dst.max_arena_words)
    woff = 0
    for s, (mn, mx, w, packed, ew) in enumerate(exp):
        if len(src_dec[s]) and (valid is None or valid[int(offs[s]):int(offs[s]) + len(src_dec[s])].any() or rule == adac.RULE_RECOMPACT):
            assert int(exp_descs["width"][s]) == w and bool(exp_descs["flags"][s] & adac.SEG_PACKED) == packed, s
            assert np.array_equal(words[woff:woff + len(ew)], ew), s
        woff += adac.arena_words(len(src_dec[s]), int(exp_descs["width"][s]))
