"""adac_reencode / adac_analyze_packed / adac_repack: packed -> packed re-compaction (SURVEY.md §8b adac_repack,
§8d "old_w -> new_w").  The bar: the destination is bit-identical to encoding the decoded values directly, whose
words/widths/mins run_encode_decode checks against the oracle (column_segment.cpp:348-383, :385-456)."""
import numpy as np
import pytest

from test_gpu_parity import make_values, oracle_encode, run_encode_decode

pytestmark = pytest.mark.gpu


def check_dst(adac, orc, dst, d_dst, seg_vals, rule, padded, validity=None):
    descs = dst.get_descs()
    offs = np.concatenate([[0], np.cumsum([len(v) for v in seg_vals])[:-1]]).astype(np.uint64)
    exp = oracle_encode(orc, seg_vals, rule, padded, validity, offs)
    words = d_dst.download(np.uint64, dst.max_arena_words)
    woff = 0
    for s, (mn, mx, w, packed, ew) in enumerate(exp):
        d = descs[s]
        assert int(d["width"]) == w and bool(d["flags"] & adac.SEG_PACKED) == packed, s
        assert int(d["word_off"]) == woff
        if packed:
            assert int(d["min"]) == mn
        assert np.array_equal(words[woff:woff + len(ew)], ew), "segment %d (w=%d)" % (s, w)
        woff += adac.arena_words(len(seg_vals[s]), w)


@pytest.mark.parametrize("dtype", [np.uint64, np.int64, np.uint32, np.int32, np.uint16, np.int16, np.uint8, np.int8])
def test_repack_equals_direct_encode(adac, oracle, gpu_ctx, dtype):
    dtype = np.dtype(dtype)
    tb = 8 * dtype.itemsize
    rng = np.random.default_rng(300 + tb + (dtype.kind == "i"))
    tile = adac.tile_values(dtype)
    widths = sorted({1, 2, 5, 7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 40, 63, tb} & set(range(1, tb + 1)))
    counts, segs = [], []
    for w in widths:
        n = int(rng.integers(10, 3 * tile)) if w % 2 else 2 * tile
        counts.append(n)
        segs.append(make_values(rng, dtype, n, w))
    counts += [1, 0, 3]
    segs += [make_values(rng, dtype, 1, 3), make_values(rng, dtype, 0, 3), make_values(rng, dtype, 3, 2)]
    counts = np.array(counts, dtype=np.uint32)
    for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
        # source: byte-padded widths (succinct_padded_to_next_byte); destination: exact widths — and back
        src, d_src, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs, rule, padded=True)
        dst = adac.Layout(gpu_ctx, dtype, counts)
        d_dst = gpu_ctx.alloc(dst.max_arena_words * 8 + 16).zero()
        src.reencode(d_src, dst, d_dst, None, rule, False)
        check_dst(adac, oracle, dst, d_dst, segs, rule, False)
        back = adac.Layout(gpu_ctx, dtype, counts)
        d_back = gpu_ctx.alloc(back.max_arena_words * 8 + 16).zero()
        dst.reencode(d_dst, back, d_back, None, rule, True)
        check_dst(adac, oracle, back, d_back, segs, rule, True)
        d_out = gpu_ctx.alloc(int(counts.sum()) * dtype.itemsize + 64)
        back.unpack(d_back, d_out)
        assert np.array_equal(d_out.download(dtype, int(counts.sum())), np.concatenate(segs))


def test_repack_after_the_value_range_shrank(adac, oracle, gpu_ctx):
    """A segment keeps its old width after rows were rewritten with a narrower range (DuckDB updates in place):
    re-compaction finds the new min/max on the packed data and tightens the width."""
    rng = np.random.default_rng(8)
    n = 70_001
    wide = [make_values(rng, np.uint32, n, 27), make_values(rng, np.uint32, 4096, 19)]
    counts = np.array([n, 4096], dtype=np.uint32)
    src, d_src, _, sd, _ = run_encode_decode(adac, oracle, gpu_ctx, np.uint32, counts, wide)
    assert sd["width"].tolist() == [27, 19]
    # rewrite the rows at the OLD widths and mins (what an in-place update leaves behind)
    descs = src.get_descs()
    host = np.zeros(src.max_arena_words, dtype=np.uint64)
    narrow = [(v % 1000 + np.uint32(int(descs["min"][s]) + 37)).astype(np.uint32) for s, v in enumerate(wide)]
    for s, v in enumerate(narrow):
        w, mn, off = int(descs["width"][s]), int(descs["min"][s]), int(descs["word_off"][s])
        assert v.min() >= mn and int(v.max()) - mn < (1 << w)
        ws = oracle.pack_flat(v, mn, w)
        host[off:off + len(ws)] = ws
    d_src.upload(host)
    dst = adac.Layout(gpu_ctx, np.uint32, counts)
    d_dst = gpu_ctx.alloc(dst.max_arena_words * 8 + 16).zero()
    src.reencode(d_src, dst, d_dst)
    check_dst(adac, oracle, dst, d_dst, narrow, adac.RULE_APPEND, False)
    assert dst.get_descs()["width"].tolist() == [10, 10]


def test_repack_with_nulls_and_argument_errors(adac, oracle, gpu_ctx):
    rng = np.random.default_rng(12)
    dtype = np.dtype(np.int16)
    counts = np.array([9000, 5, 20000], dtype=np.uint32)
    segs = [make_values(rng, dtype, int(c), 9) for c in counts]
    total = int(counts.sum())
    valid = rng.random(total) > 0.3
    vm = np.packbits(valid, bitorder="little")
    vm = np.concatenate([vm, np.zeros((-len(vm)) % 8 + 8, np.uint8)]).view(np.uint64)
    for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
        src, d_src, _, _, _ = run_encode_decode(adac, oracle, gpu_ctx, dtype, counts, segs, rule, padded=True, validity=vm)
        # what the source decodes to (NULL slots hold NullValue<T>) is what a direct encode of those values sees
        d_out = gpu_ctx.alloc(total * 2 + 64)
        src.unpack(d_src, d_out)
        dec = d_out.download(dtype, total)
        offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
        dec_segs = [dec[int(o):int(o) + int(c)] for o, c in zip(offs, counts)]
        dst = adac.Layout(gpu_ctx, dtype, counts)
        d_dst = gpu_ctx.alloc(dst.max_arena_words * 8 + 16).zero()
        d_valid = gpu_ctx.upload(vm)
        src.reencode(d_src, dst, d_dst, d_valid, rule, False)
        check_dst(adac, oracle, dst, d_dst, dec_segs, rule, False, vm)
    other = adac.Layout(gpu_ctx, dtype, np.array([9000, 5, 20001], dtype=np.uint32))
    with pytest.raises(adac.AdacError):
        src.reencode(d_src, other, d_dst)          # different segment shapes
    with pytest.raises(adac.AdacError):
        src.repack(d_src, dst, d_src)              # in place is not supported
