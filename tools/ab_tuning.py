#!/usr/bin/env python3
"""In-process interleaved A/B of launch-shape knobs (adac_set_tuning) on the scan kernels: per variant the
median and min launch time over several rounds, HIP events on the codec's stream (methodology: variants
interleaved in ONE process, cdna_hip_programming.md §5.4 rule 24).  Prints one JSON object.

usage: python tools/ab_tuning.py [rows] [rounds]
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    ctx = adac.Context(0)
    rng = np.random.default_rng(1)
    base = {"persistent_unpack": 0, "scan_probe": 0}
    variants = [
        ("lds x1", dict(base, templated_scan=0, scan_tiles_per_wg=1)),
        ("lds x8", dict(base, templated_scan=0, scan_tiles_per_wg=8)),
        ("templ auto", dict(base, templated_scan=1, scan_tiles_per_wg=0)),
        ("templ x1", dict(base, templated_scan=1, scan_tiles_per_wg=1)),
        ("templ x2", dict(base, templated_scan=1, scan_tiles_per_wg=2)),
        ("templ x3", dict(base, templated_scan=1, scan_tiles_per_wg=3)),
        ("templ x4", dict(base, templated_scan=1, scan_tiles_per_wg=4)),
        ("templ x6", dict(base, templated_scan=1, scan_tiles_per_wg=6)),
        ("templ x7", dict(base, templated_scan=1, scan_tiles_per_wg=7)),
        ("templ x12", dict(base, templated_scan=1, scan_tiles_per_wg=12)),
        ("templ x15", dict(base, templated_scan=1, scan_tiles_per_wg=15)),
        ("templ x8", dict(base, templated_scan=1, scan_tiles_per_wg=8)),
        ("templ x16", dict(base, templated_scan=1, scan_tiles_per_wg=16)),
        ("templ x32", dict(base, templated_scan=1, scan_tiles_per_wg=32)),
        ("probe x8", dict(base, templated_scan=1, scan_tiles_per_wg=8, scan_probe=1)),
        ("probe x16", dict(base, templated_scan=1, scan_tiles_per_wg=16, scan_probe=1)),
        ("persistent-dma-8/cu", {"persistent_unpack": 1, "blocks_per_cu": 8, "templated_scan": 0}),
    ]
    out = {"rows": rows, "rounds": rounds, "cases": []}
    cases = ((np.uint64, 8), (np.uint64, 16), (np.uint64, 32), (np.uint64, 48), (np.uint32, 8), (np.uint32, 16),
             (np.uint32, 24), (np.uint16, 7), (np.uint8, 3))
    if os.environ.get("AB_CASES"):  # e.g. AB_CASES="u64:8,u32:16"
        cases = tuple((np.dtype("uint" + c.split(":")[0][1:]), int(c.split(":")[1]))
                      for c in os.environ["AB_CASES"].split(","))
    if os.environ.get("AB_VARIANTS"):  # comma-separated substrings of variant names
        keep = os.environ["AB_VARIANTS"].split(",")
        variants = [v for v in variants if any(k in v[0] for k in keep)]
    for dtype, w in cases:
        dtype = np.dtype(dtype)
        vals = rng.integers(0, 2 ** w, size=rows, dtype=np.uint64).astype(dtype)
        counts = adac.appender_segment_counts(rows, dtype.itemsize)
        lay = adac.Layout(ctx, dtype, counts)
        d_vals = ctx.upload(vals)
        d_words = ctx.alloc(lay.max_arena_words * 8).zero()
        d_out = ctx.alloc(rows * dtype.itemsize + 64)
        d_sums = ctx.alloc(len(counts) * 8)
        lay.encode(d_vals, d_words)
        ctx.sync()
        descs = lay.get_descs()
        rd = int(((descs["count"].astype(np.uint64) * descs["width"] + 63) // 64 * 8).sum())
        wr = rows * dtype.itemsize
        ref_sum = int(vals.astype(np.uint64).sum(dtype=np.uint64))
        times = {name: {"unpack": [], "sum": [], "count": [], "select": []} for name, _ in variants}
        d_bm = ctx.alloc((rows + 63) // 64 * 8 + 8)
        lo_key, hi_key = (1 << w) // 4, (1 << w) // 2   # ~25 % selectivity on the uniform test column
        for _ in range(rounds):
            for name, knobs in variants:
                for k, v in knobs.items():
                    adac.set_tuning(k, v)
                lay.unpack(d_words, d_out)
                lay.scan_sum(d_words, d_sums)
                ctx.sync()
                ctx.timer_start()
                for _i in range(5):
                    lay.unpack(d_words, d_out)
                times[name]["unpack"].append(ctx.timer_stop() / 5)
                ctx.timer_start()
                for _i in range(5):
                    lay.scan_sum(d_words, d_sums)
                times[name]["sum"].append(ctx.timer_stop() / 5)
                ctx.timer_start()
                for _i in range(5):
                    lay.scan_count_between(d_words, lo_key, hi_key, d_sums)
                times[name]["count"].append(ctx.timer_stop() / 5)
                ctx.timer_start()
                for _i in range(5):
                    lay.scan_select_between(d_words, lo_key, hi_key, d_bm, d_sums)
                times[name]["select"].append(ctx.timer_stop() / 5)
        # every variant must still be right
        for name, knobs in variants:
            for k, v in knobs.items():
                adac.set_tuning(k, v)
            d_out.zero()
            lay.unpack(d_words, d_out)
            lay.scan_sum(d_words, d_sums)
            ctx.sync()
            assert np.array_equal(d_out.download(dtype, rows), vals), name
            if not knobs.get("scan_probe"):
                assert int(d_sums.download(np.uint64, len(counts)).sum(dtype=np.uint64)) == ref_sum, name
                ref_cnt = int(((vals >= lo_key) & (vals <= hi_key)).sum())
                lay.scan_count_between(d_words, lo_key, hi_key, d_sums)
                assert int(d_sums.download(np.uint64, len(counts)).sum()) == ref_cnt, name
                lay.scan_select_between(d_words, lo_key, hi_key, d_bm, d_sums)
                assert int(d_sums.download(np.uint64, len(counts)).sum()) == ref_cnt, name
        case = {"dtype": "u%d" % (8 * dtype.itemsize), "width": w, "variants": {}}
        for name, _ in variants:
            mu = float(np.median(times[name]["unpack"]))
            ms = float(np.median(times[name]["sum"]))
            case["variants"][name] = {
                "unpack_ms_median": mu, "unpack_ms_min": float(min(times[name]["unpack"])),
                "unpack_total_GBps": (rd + wr) / mu / 1e6, "unpack_Gvalues_s": rows / mu / 1e6,
                "sum_ms_median": ms, "sum_ms_min": float(min(times[name]["sum"])),
                "sum_read_GBps": rd / ms / 1e6, "sum_Gvalues_s": rows / ms / 1e6,
                "count_read_GBps": rd / float(np.median(times[name]["count"])) / 1e6,
                "select_read_GBps": rd / float(np.median(times[name]["select"])) / 1e6,
            }
        out["cases"].append(case)
        del lay, d_vals, d_words, d_out, d_sums, d_bm
    adac.set_tuning("scan_probe", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
