#!/usr/bin/env python3
"""Phase timeline of the single-pass encode (k_encode_1p): runs one encode with the tuning knob "encode_stamps" set and
summarises the 100 MHz wall-clock stamps thread 0 of every workgroup recorded.

usage: python3 tools/encode_stamps.py <dtype>:<width> [rows] [placement: 0 ordered look-back (default), 1 first come]
slots (per segment): 0 its iteration starts (rows loading), 1 wave 0 consumed its rows (min/max), 2 workgroup min/max
       known + next segment prefetch issued, 3 look-back done / descriptor written, 4 last store issued, 7 hardware id
"""
import ctypes
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
adac = importlib.import_module("duckdb-adaptive-compression_amd")


def main():
    c = sys.argv[1]
    dtype = np.dtype("uint" + c.split(":")[0][1:])
    w = int(c.split(":")[1])
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    parts = 1
    placement = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    adac.set_tuning("encode_placement", placement)
    ctx = adac.Context(0)
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 2 ** w, size=rows, dtype=np.uint64).astype(dtype)
    counts = adac.appender_segment_counts(rows, dtype.itemsize)
    lay = adac.Layout(ctx, dtype, counts)
    d_vals = ctx.upload(vals)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 128).zero()
    for _ in range(3):
        lay.encode(d_vals, d_words)
    ctx.sync()
    adac.set_tuning("encode_stamps", 1)
    lay.encode(d_vals, d_words)
    ctx.sync()
    adac.set_tuning("encode_stamps", 0)
    nwg = min(len(counts) * parts, 16384)
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    rc = adac.lib().adac_debug_encode_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(buf.nbytes))
    assert rc == 0
    t = buf[:, :5].astype(np.int64)
    live = t[:, 4] > 0          # second halves of short segments return before any stamp but 0/1
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0       # 100 MHz -> microseconds
    ph = np.diff(us[live], axis=1)
    names = ["loads+minmax(wave0)", "wg reduce + prefetch issue", "width+lookback", "pack+stores issued"]
    out = {"case": c, "rows": rows, "placement": placement, "parts": parts, "workgroups": int(nwg), "live": int(live.sum()),
           "kernel_span_us": float(us[live, 4].max() - us[:, 0].min()),
           "wg_lifetime_us": {"mean": float((us[live, 4] - us[live, 0]).mean()),
                              "p10": float(np.percentile(us[live, 4] - us[live, 0], 10)),
                              "p90": float(np.percentile(us[live, 4] - us[live, 0], 90))},
           "phase_us_mean": {n: float(ph[:, i].mean()) for i, n in enumerate(names)},
           "phase_us_p90": {n: float(np.percentile(ph[:, i], 90)) for i, n in enumerate(names)}}
    out["lookback_only_us_mean"] = float(((buf[live, 5].astype(np.int64) - t[live, 2]) / 100.0).mean())
    out["barrier_after_lookback_us_mean"] = float(((t[live, 3] - buf[live, 5].astype(np.int64)) / 100.0).mean())
    probe = buf[live, 6]
    out["lookback_round_trips_mean"] = float((probe >> np.uint64(16)).mean())
    out["lookback_unpublished_retries_mean"] = float((probe & np.uint64(0xffff)).mean())
    out["lookback_unpublished_retries_p90"] = float(np.percentile((probe & np.uint64(0xffff)).astype(np.int64), 90))
    hw = buf[:, 7]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    cu = (hw >> np.uint64(8)) & np.uint64(0xf)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    place = xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(16) + cu
    # per CU: how much of the kernel span is covered by a load phase [slot1, slot2] of some workgroup on it
    span = out["kernel_span_us"]
    cover = []
    for pid in np.unique(place[live]):
        m = live & (place == pid)
        iv = sorted(zip(us[m, 0], us[m, 1]))
        tot, end = 0.0, -1.0
        for a, b in iv:
            a = max(a, end)
            if b > a:
                tot += b - a
                end = b
        cover.append(tot / span)
    out["cus_seen"] = len(cover)
    out["load_phase_coverage_per_cu"] = {"mean": float(np.mean(cover)), "min": float(np.min(cover))}
    # start-time histogram: are the workgroups in lockstep?
    starts = np.sort(us[live, 0])
    out["start_us_deciles"] = [float(x) for x in np.percentile(starts, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
