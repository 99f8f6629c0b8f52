#!/usr/bin/env python3
"""Launch chosen kernels of the codec a few times on uniformly distributed columns of given (dtype, width), so that
`rocprofv3 --pmc ...` (or --kernel-trace --stats) can be attached to exactly those kernels.

usage: python3 tools/pmc_probe.py <ops> <cases> [rows] [reps]
  ops    comma list of: unpack, sum, count, select, gather, encode, pack, repack, analyze, groupsum (SUM / COUNT GROUP BY a
         6-valued uint8 code column, the Q1 shape)
  cases  comma list of <dtype>:<width>, e.g. u64:13,u64:16,u32:8
Prints one JSON object with the HIP-event launch times (ms) per case and op.
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
adac = importlib.import_module("duckdb-adaptive-compression_amd")


def main():
    ops = sys.argv[1].split(",")
    cases = [(np.dtype("uint" + c.split(":")[0][1:]), int(c.split(":")[1])) for c in sys.argv[2].split(",")]
    rows_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    for kv in filter(None, os.environ.get("ADAC_TUNING", "").split(",")):   # e.g. ADAC_TUNING=sel_debug=2
        k, v = kv.split("=")
        adac.set_tuning(k, int(v))
    ctx = adac.Context(0)
    rng = np.random.default_rng(3)
    out = []
    for dtype, w in cases:
        rows = rows_arg or int(400e6 * 8 / w)  # packed bytes beyond the 256 MiB Infinity Cache
        vals = rng.integers(0, 2 ** w, size=rows, dtype=np.uint64).astype(dtype)
        counts = adac.appender_segment_counts(rows, dtype.itemsize)
        lay = adac.Layout(ctx, dtype, counts)
        d_vals = ctx.upload(vals)
        d_words = ctx.alloc(lay.max_arena_words * 8 + 128).zero()
        lay.encode(d_vals, d_words)
        ctx.sync()
        descs = lay.get_descs()
        rd = int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum())
        d_res = ctx.alloc(len(counts) * 8)
        rec = {"dtype": "u%d" % (8 * dtype.itemsize), "width": w, "rows": rows, "packed_bytes": rd, "ms": {}}

        def timed(name, fn):
            fn()
            ctx.sync()
            ctx.timer_start()
            for _ in range(reps):
                fn()
            ms = ctx.timer_stop() / reps
            rec["ms"][name] = ms

        if "unpack" in ops:
            d_out = ctx.alloc(rows * dtype.itemsize + 64)
            timed("unpack", lambda: lay.unpack(d_words, d_out))
            del d_out
        if "sum" in ops:
            timed("sum", lambda: lay.scan_sum(d_words, d_res))
        if "count" in ops:
            timed("count", lambda: lay.scan_count_between(d_words, (1 << w) // 4, (1 << w) // 2, d_res))
        if "select" in ops:
            d_bm = ctx.alloc((rows + 63) // 64 * 8 + 8)
            timed("select", lambda: lay.scan_select_between(d_words, 0, 2 ** (w - 1), d_bm, d_res))
            rec["select_read_GBps"] = rd / (rec["ms"]["select"] * 1e-3) / 1e9
            del d_bm
        if "gather" in ops:  # scan-with-selection at 50 % selectivity: values + element ids of the selected rows
            d_bm = ctx.alloc((rows + 63) // 64 * 8 + 8)
            lay.scan_select_between(d_words, 0, 2 ** (w - 1) - 1, d_bm, d_res)
            nsel = int((vals < 2 ** (w - 1)).sum())
            d_go, d_gi = ctx.alloc((nsel + 16) * dtype.itemsize), ctx.alloc((nsel + 16) * 8)
            timed("gather", lambda: lay.unpack_selected(d_words, d_bm, d_go, d_gi, False))
            rec["gather_traffic_GBps"] = (rd + rows // 4 + nsel * (dtype.itemsize + 8)) / (rec["ms"]["gather"] * 1e-3) / 1e9
            del d_bm, d_go, d_gi
        if "groupsum" in ops:
            code = rng.choice(6, size=rows, p=[.2466, .2534, .0004, .2500, .2490, .0006]).astype(np.uint8)
            klay = adac.Layout(ctx, np.uint8, counts)
            d_code = ctx.upload(code)
            d_kwords = ctx.alloc(klay.max_arena_words * 8 + 128).zero()
            klay.encode(d_code, d_kwords)
            ctx.sync()
            kd = klay.get_descs()
            krd = int(((kd["count"].astype(np.uint64) * kd["width"] + 63) // 64 * 8).sum())
            d_gs, d_gc = ctx.alloc(7 * 8), ctx.alloc(7 * 8)
            timed("groupsum", lambda: lay.scan_group_sum(d_words, klay, d_kwords, 6, d_gs, d_gc))
            rec["groupsum_packed_GBps"] = (rd + krd) / (rec["ms"]["groupsum"] * 1e-3) / 1e9
            del klay, d_code, d_kwords
        if "analyze" in ops:
            timed("analyze", lambda: lay.analyze(d_vals, None, adac.RULE_APPEND))
        if "encode" in ops:
            timed("encode", lambda: lay.encode(d_vals, d_words))
            rec["encode_values_per_s"] = rows / (rec["ms"]["encode"] * 1e-3)
        if "pack" in ops:
            timed("pack", lambda: lay.pack(d_vals, d_words))
        if "repack" in ops:
            del d_vals
            pad = adac.Layout(ctx, dtype, counts)
            d_pad = ctx.alloc(pad.max_arena_words * 8 + 128).zero()
            exact = adac.Layout(ctx, dtype, counts)
            d_exact = ctx.alloc(exact.max_arena_words * 8 + 128).zero()
            lay.reencode(d_words, pad, d_pad, None, adac.RULE_APPEND, True)
            pad.reencode(d_pad, exact, d_exact, None, adac.RULE_APPEND, False)
            ctx.sync()
            pd = pad.get_descs()
            rd_pad = int(((pd["count"].astype(np.uint64) * pd["width"] + 63) // 64 * 8).sum())
            timed("repack", lambda: pad.repack(d_pad, exact, d_exact))
            rec["repack_GBps"] = (rd_pad + rd) / (rec["ms"]["repack"] * 1e-3) / 1e9
            timed("reencode", lambda: pad.reencode(d_pad, exact, d_exact, None, adac.RULE_APPEND, False))
            rec["reencode_values_per_s"] = rows / (rec["ms"]["reencode"] * 1e-3)
            del pad, d_pad, exact, d_exact
        out.append(rec)
        del lay, d_words, d_res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
