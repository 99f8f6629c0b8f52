#!/usr/bin/env python3
"""Guard against silent throughput regressions between two bench lines (bench.py's "sweep" + headline figures).

  python tools/sweep_diff.py profiles/r02i_bench.json gpurun_out/r03x/bench.json [--limit 3.0]

Two runs land on two different boxes, and boxes differ by 3 - 7 % across the board (the same r02i binary: fused SUM
u64 w 8 5668 GB/s on one box, 5277 on another).  So every cell is compared twice: raw, and NORMALISED by the box
factor = the median new/old ratio over all cells — a change that slows one kernel family shows as a normalised loss
while a slower box moves every cell alike.  Exit code 1 when a cell of the north star's band (u64 / u32 columns at
widths 8 .. 32: decode, fused SUM, selection, re-compaction) loses more than --limit percent after normalisation.
tools/profile_round.sh runs it against the previous committed round and fails loudly on a loss."""
import argparse
import json
import statistics
import sys

METRICS = (("decode", lambda r: r["decode_total_GBps"]), ("sum", lambda r: r["fused_sum_read_GBps"]),
           ("select", lambda r: r["select_read_GBps"]), ("repack", lambda r: r["recompaction"]["repack_GBps"]))


def cells(bench):
    out = {}
    for r in bench.get("sweep", []):
        for name, get in METRICS:
            try:
                out["%s w%d %s" % (r["dtype"], r["width"], name)] = float(get(r))
            except (KeyError, TypeError):
                pass
    if "roofline" in bench:
        out["C2 decode"] = float(bench["roofline"]["achieved"])
    fs = bench.get("fused_scan", {})
    if "read_GBps" in fs:
        out["C2 sum"] = float(fs["read_GBps"])
    if "select_bitmap" in fs:
        out["C2 select"] = float(fs["select_bitmap"]["read_GBps"])
    if "encode" in bench:
        out["C2 encode"] = float(bench["encode"]["algorithmic_GBps"])
    return out


def in_band(key):
    t, w = key.split()[0], key.split()[1]
    if t == "C2":
        return True
    return t in ("u64", "u32") and 8 <= int(w[1:]) <= 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("new")
    ap.add_argument("--limit", type=float, default=3.0, help="allowed normalised loss in percent")
    a = ap.parse_args()
    old, new = cells(json.load(open(a.old))), cells(json.load(open(a.new)))
    common = [k for k in old if k in new]
    if not common:
        print("no common cells between %s and %s" % (a.old, a.new))
        return 1
    box = statistics.median(new[k] / old[k] for k in common)
    print("box factor (median new/old over %d cells): %.3f" % (len(common), box))
    print("%-22s %9s %9s %8s %8s" % ("cell", "old GB/s", "new GB/s", "raw %", "norm %"))
    bad = []
    for k in common:
        raw = 100.0 * (new[k] / old[k] - 1.0)
        norm = 100.0 * (new[k] / old[k] / box - 1.0)
        flag = ""
        if in_band(k) and norm < -a.limit:
            flag = "  <-- LOSS"
            bad.append(k)
        print("%-22s %9.0f %9.0f %+8.1f %+8.1f%s" % (k, old[k], new[k], raw, norm, flag))
    if bad:
        print("\nREGRESSION: %d cell(s) of the 8-32-bit band lost more than %.1f %% after normalisation: %s"
              % (len(bad), a.limit, ", ".join(bad)))
        return 1
    print("\nno cell of the 8-32-bit band lost more than %.1f %% after normalisation" % a.limit)
    return 0


if __name__ == "__main__":
    sys.exit(main())
