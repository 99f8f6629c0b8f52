"""The device BITPACKING codec through the C ABI (adac_bp_plan_create -> adac_bp_write -> adac_bp_unpack /
_unpack_range / _fetch_rows) against the reference's OWN fixtures for this codec: the data shapes and expected results
of /root/reference/test/sql/storage/compression/bitpacking/*.test, committed as data in
tests/golden/bitpacking_sql_cases.json (tests/test_bitpacking_sql_cases.py runs the same cases through the oracle on
the CPU).  Every case under each forced mode the file loops over; the push-down case also runs its predicate on the
device: the decoded column is packed into succinct segments and filtered / aggregated by the fused scans."""
import numpy as np
import pytest

import bp_sql_cases as sc
from oracle import bitpacking as bp

pytestmark = pytest.mark.gpu


def pack_validity(valid):
    bits = np.packbits(valid, bitorder="little")
    return np.concatenate([bits, np.zeros((-len(bits)) % 8 + 8, np.uint8)]).view(np.uint64)


@pytest.mark.parametrize("case_id", sc.case_ids())
def test_device_codec_reproduces_the_reference_sql_expectations(adac, gpu_ctx, case_id):
    ctx = gpu_ctx
    case = next(c for c in sc.load_cases() if c["id"] == case_id)
    vals, valid = sc.build_column(case)
    dtype, n = vals.dtype, len(vals)
    d_vals = ctx.upload(vals)
    d_valid = None if valid is None else ctx.upload(pack_validity(valid))
    for mode in case["forced_modes"]:
        plan = adac.BitpackingPlan(ctx, dtype, d_vals, n, d_valid, sc.MODE_CODE[mode])
        if sc.refused(case):                                 # the file expects 'Uncompressed'
            assert not plan.encodable, (case_id, mode)
            continue
        if case.get("may_be_refused"):                       # no compression assertion in the file: agree with the oracle
            try:
                bp.Compressed(vals, valid, force_mode=sc.MODE_CODE[mode])
                oracle_ok = True
            except ValueError:
                oracle_ok = False
            assert plan.encodable == oracle_ok, (case_id, mode)
            if not oracle_ok:
                continue
        assert plan.encodable, (case_id, mode)              # "compression = BitPacking" under this forced mode
        d_blocks = ctx.alloc(max(plan.nseg, 1) * plan.BLOCK_STRIDE + 64)
        plan.write(d_vals, d_blocks, d_valid)
        segs = [plan.segment(i) for i in range(plan.nseg)]   # (start, count, size)
        offs = np.arange(plan.nseg, dtype=np.uint64) * np.uint64(plan.BLOCK_STRIDE)
        counts = np.array([s[1] for s in segs], dtype=np.uint32)
        assert int(counts.sum()) == n
        lay = adac.BitpackingLayout(ctx, dtype, offs, counts)
        d_out = ctx.alloc(n * dtype.itemsize + 64)
        lay.unpack(d_blocks, d_out)
        got = d_out.download(dtype, n)
        starts = np.array([s[0] for s in segs])

        def fetch(row):  # BitpackingFetchRow (bitpacking.cpp:827-870)
            si = int(np.searchsorted(starts, row, side="right") - 1)
            d_f = ctx.alloc(16)
            lay.fetch_rows(d_blocks, ctx.upload(np.array([si], np.uint32)),
                           ctx.upload(np.array([row - starts[si]], np.uint32)), 1, d_f)
            return d_f.download(dtype, 1)[0]

        def filter_eq(key):  # the predicate on the device: fused scans over the re-packed column
            cnts = adac.appender_segment_counts(n, dtype.itemsize)
            sl = adac.Layout(ctx, dtype, cnts)
            d_words = ctx.alloc(sl.max_arena_words * 8 + 64).zero()
            sl.encode(d_out, d_words, d_valid)
            d_bm = ctx.alloc((n + 63) // 64 * 8 + 8)
            d_cnt, d_sum = ctx.alloc(len(cnts) * 8), ctx.alloc(len(cnts) * 8)
            k = int(np.array([key], dtype=dtype).view("u%d" % dtype.itemsize)[0])
            sl.scan_select_between(d_words, k, k, d_bm, d_cnt, d_valid)
            sl.scan_sum(d_words, d_sum, d_bm)
            d_sel = ctx.alloc(n * dtype.itemsize + 64)
            m = sl.unpack_selected(d_words, d_bm, d_sel)
            hit = d_sel.download(dtype, m)
            count = int(d_cnt.download(np.uint64, len(cnts)).sum())
            assert count == m
            return int(d_sum.download(np.uint64, len(cnts)).sum(dtype=np.uint64)), int(hit.min()), int(hit.max()), count

        sc.check_expectations(case, got, valid, fetch=fetch, filter_eq=filter_eq)
        # ranged scans (BitpackingScanPartial, bitpacking.cpp:736-826): the file's `limit 5 offset 12000` shape
        for e in case["expect"]:
            if e["op"] == "head":
                si = int(np.searchsorted(starts, e["offset"], side="right") - 1)
                d_r = ctx.alloc(len(e["rows"]) * dtype.itemsize + 64)
                lay.unpack_range(d_blocks, si, e["offset"] - int(starts[si]), len(e["rows"]), d_r)
                assert [int(x) for x in d_r.download(dtype, len(e["rows"]))] == e["rows"]
        # and the block images are the oracle's, byte for byte, under the forced mode too
        # (a NULL slot keeps the buffer's stale content in the reference; the device writes zero there: null_zero)
        comp = bp.Compressed(vals, valid, force_mode=sc.MODE_CODE[mode], null_zero=True)
        assert plan.groups_by_mode() == comp.groups_by_mode() and plan.nseg == comp.nseg
        img = d_blocks.download(np.uint8, plan.nseg * plan.BLOCK_STRIDE)
        for i in range(comp.nseg):
            size = segs[i][2]
            assert segs[i] == (comp.start(i), comp.count(i), comp.size(i))
            assert np.array_equal(img[i * plan.BLOCK_STRIDE:i * plan.BLOCK_STRIDE + size], comp.block(i)[:size]), (mode, i)
        if "comment_expectation" in case and valid is None:
            assert plan.groups_by_mode()[mode] >= 1, (mode, plan.groups_by_mode())
