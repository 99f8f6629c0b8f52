"""Grouped SUM / COUNT over two packed columns (adac_scan_group_sum: the Q1 shape of the reference's config 3,
TPCH_runtime.txt:2-6).  The reference has no grouped scan of its own (its engine aggregates decoded vectors), so the
check is numpy's GROUP BY over the decoded columns — decoded by the oracle-verified path — on the same seeded inputs;
integer results, compared exactly (mod 2^64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U64 = 0xFFFFFFFFFFFFFFFF


def encode_column(adac, ctx, vals, counts, val_offs=None):
    lay = adac.Layout(ctx, vals.dtype, counts, val_offs)
    host = np.zeros(max(lay.value_span, 1), dtype=vals.dtype)
    offs = np.cumsum(np.concatenate([[0], counts[:-1]]).astype(np.uint64)) if val_offs is None else val_offs
    pos = 0
    for c, o in zip(counts, offs):
        host[int(o):int(o) + int(c)] = vals[pos:pos + int(c)]
        pos += int(c)
    d_vals = ctx.upload(host)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 16).zero()
    lay.encode(d_vals, d_words)
    return lay, d_words


def reference_groups(vals, keys, ngroups):
    wide = np.int64 if vals.dtype.kind == "i" else np.uint64
    ukeys = keys.view(np.dtype("u%d" % keys.dtype.itemsize)).astype(np.uint64)
    bins = np.minimum(ukeys, np.uint64(ngroups)).astype(np.int64)
    sums, cnts = [], []
    v64 = vals.astype(wide).view(np.uint64)
    for g in range(ngroups + 1):
        m = bins == g
        sums.append(int(v64[m].sum(dtype=np.uint64)) & U64)
        cnts.append(int(m.sum()))
    return sums, cnts


def run_case(adac, ctx, rng, vdtype, kdtype, counts, ngroups, vbits, key_top, gaps=False):
    vdtype, kdtype = np.dtype(vdtype), np.dtype(kdtype)
    n = int(counts.sum())
    tb = 8 * vdtype.itemsize
    span = rng.integers(0, 2 ** min(vbits, tb), size=n, dtype=np.uint64)
    base = int(rng.integers(0, 2 ** tb - 2 ** min(vbits, tb) + 1, dtype=np.uint64)) if vbits < tb else 0
    vals = ((span + np.uint64(base)) & np.uint64(2 ** tb - 1)).astype(np.dtype("u%d" % vdtype.itemsize)).view(vdtype)
    keys = rng.integers(0, key_top, size=n, dtype=np.uint64).astype(np.dtype("u%d" % kdtype.itemsize)).view(kdtype)
    voffs = koffs = None
    if gaps:
        voffs = np.cumsum(np.concatenate([[3], counts[:-1] + 5]).astype(np.uint64))
        koffs = np.cumsum(np.concatenate([[1], counts[:-1] + 2]).astype(np.uint64))
    vlay, vwords = encode_column(adac, ctx, vals, counts, voffs)
    klay, kwords = encode_column(adac, ctx, keys, counts, koffs)
    d_sums = ctx.alloc((ngroups + 1) * 8)
    d_cnts = ctx.alloc((ngroups + 1) * 8)
    vlay.scan_group_sum(vwords, klay, kwords, ngroups, d_sums, d_cnts)
    got_s = d_sums.download(np.uint64, ngroups + 1).tolist()
    got_c = d_cnts.download(np.uint64, ngroups + 1).tolist()
    exp_s, exp_c = reference_groups(vals, keys, ngroups)
    assert got_c == exp_c
    assert got_s == exp_s
    assert sum(got_c) == n
    return vlay, vwords, klay, kwords


@pytest.mark.parametrize("vdtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64])
def test_group_sum_every_value_type(adac, gpu_ctx, vdtype):
    rng = np.random.default_rng(77 + np.dtype(vdtype).itemsize + (np.dtype(vdtype).kind == "i"))
    counts = np.array([2048, 32767, 1, 0, 5000, 70001, 63, 4096], dtype=np.uint32)
    tb = 8 * np.dtype(vdtype).itemsize
    for kdtype, ngroups, key_top in ((np.uint8, 6, 6), (np.uint8, 7, 9), (np.uint16, 40, 50), (np.int32, 256, 300),
                                     (np.uint8, 1, 3)):
        for vbits in (1, 6, tb // 2 + 1, tb):
            run_case(adac, gpu_ctx, rng, vdtype, kdtype, counts, ngroups, vbits, key_top)


def test_group_sum_placements_and_wide_keys(adac, gpu_ctx):
    rng = np.random.default_rng(99)
    counts = np.array([1000, 37, 5000, 2048, 1, 16385], dtype=np.uint32)
    run_case(adac, gpu_ctx, rng, np.int32, np.uint8, counts, 4, 21, 4, gaps=True)
    run_case(adac, gpu_ctx, rng, np.uint64, np.uint64, counts, 8, 47, 2 ** 40, gaps=True)   # nearly all rows overflow
    run_case(adac, gpu_ctx, rng, np.int64, np.int16, counts, 200, 33, 200, gaps=True)


def test_q1_shape_on_lineitem_like_columns(adac, gpu_ctx):
    """60 M-row lineitem is the bench harness's job; here 3 M rows of the same shape: a 6-valued group code
    (l_returnflag x l_linestatus), l_quantity in [1, 50] and a 21-bit l_partkey, in 65 534-row segments."""
    rng = np.random.default_rng(2024)
    n = 3_000_000
    counts = np.array([65534] * (n // 65534) + [n % 65534], dtype=np.uint32)
    code = rng.choice(6, size=n, p=[.25, .25, .01, .24, .24, .01]).astype(np.uint8)
    qty = rng.integers(1, 51, size=n, dtype=np.int64).astype(np.int32)
    part = rng.integers(1, 2_000_001, size=n, dtype=np.int64).astype(np.int32)
    klay, kwords = encode_column(adac, gpu_ctx, code, counts)
    d_sums = gpu_ctx.alloc(7 * 8)
    d_cnts = gpu_ctx.alloc(7 * 8)
    for col in (qty, part):
        vlay, vwords = encode_column(adac, gpu_ctx, col, counts)
        vlay.scan_group_sum(vwords, klay, kwords, 6, d_sums, d_cnts)
        exp_s, exp_c = reference_groups(col, code, 6)
        assert d_sums.download(np.uint64, 7).tolist() == exp_s
        assert d_cnts.download(np.uint64, 7).tolist() == exp_c
        assert exp_c[6] == 0


def test_group_sum_argument_errors(adac, gpu_ctx):
    a = adac.Layout(gpu_ctx, np.uint32, np.array([10, 20], dtype=np.uint32))
    b = adac.Layout(gpu_ctx, np.uint8, np.array([10, 21], dtype=np.uint32))
    c = adac.Layout(gpu_ctx, np.uint8, np.array([10, 20], dtype=np.uint32))
    d = gpu_ctx.alloc(4096).zero()
    for keys, g in ((b, 4), (c, 0), (c, 257)):
        with pytest.raises(adac.AdacError):
            a.scan_group_sum(d, keys, d, g, d, d)
