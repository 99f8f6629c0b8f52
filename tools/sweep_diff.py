#!/usr/bin/env python3
"""Guard against silent throughput regressions between two bench lines (bench.py's "sweep" + headline figures).

  python tools/sweep_diff.py profiles/r02i_bench.json gpurun_out/r03x/bench.json [--limit 5.0]

Two runs land on two different boxes, and boxes differ by 3 - 7 % across the board (the same r02i binary: fused SUM
u64 w 8 5668 GB/s on one box, 5277 on another).  So every cell is compared twice: raw, and NORMALISED by the box
factor = the ratio of the two lines' device-to-device copy rates (bench.py's device_copy_GBps: no kernel of this library;
the median new/old ratio over all cells where a line lacks it) — a change that slows one kernel family shows as a
normalised loss while a slower box moves every cell alike.  Exit code 1 when a cell of the north star's band (u64 / u32 columns at
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
    ap.add_argument("--limit", type=float, default=5.0,
                    help="allowed normalised loss in percent (the box factor itself is only good to +-4 %: the byte-identical "
                         "decode kernel read -1 % raw between two rounds whose copy rates differed by +4 %)")
    a = ap.parse_args()
    old_line, new_line = json.load(open(a.old)), json.load(open(a.new))
    old, new = cells(old_line), cells(new_line)
    common = [k for k in old if k in new]
    if not common:
        print("no common cells between %s and %s" % (a.old, a.new))
        return 1
    med = statistics.median(new[k] / old[k] for k in common)
    box = med
    print("median new/old over %d cells: %.3f" % (len(common), med))
    # the neutral probe of the box: the device-to-device copy rate both bench lines carry (none of this library's
    # kernels).  The median over the cells is biased as soon as a change speeds up MANY cells (round 3: arrival cells
    # raised every SUM and selection cell, the median read 1.019 on a box whose copy rate was 4.6 % LOWER, and the
    # untouched decode kernel showed as a 'loss')
    if old_line.get("device_copy_GBps") and new_line.get("device_copy_GBps"):
        box = new_line["device_copy_GBps"] / old_line["device_copy_GBps"]
        print("box factor (device copy rate new/old: %.0f / %.0f GB/s): %.3f"
              % (new_line["device_copy_GBps"], old_line["device_copy_GBps"], box))
    else:
        print("box factor: the median (no device_copy_GBps in both lines)")
    # neither probe is exact (the copy rate read +3 .. 5 % on the untouched decode kernel of the same pair of runs):
    # a cell is a LOSS when it lost more than the limit under BOTH normalisations
    print("%-22s %9s %9s %8s %9s %9s" % ("cell", "old GB/s", "new GB/s", "raw %", "/copy %", "/median %"))
    bad = []
    for k in common:
        raw = 100.0 * (new[k] / old[k] - 1.0)
        norm = 100.0 * (new[k] / old[k] / box - 1.0)
        norm_med = 100.0 * (new[k] / old[k] / med - 1.0)
        flag = ""
        if in_band(k) and norm < -a.limit and norm_med < -a.limit:
            flag = "  <-- LOSS"
            bad.append(k)
        print("%-22s %9.0f %9.0f %+8.1f %+9.1f %+9.1f%s" % (k, old[k], new[k], raw, norm, norm_med, flag))
    if bad:
        print("\nREGRESSION: %d cell(s) of the 8-32-bit band lost more than %.1f %% after normalisation: %s"
              % (len(bad), a.limit, ", ".join(bad)))
        return 1
    print("\nno cell of the 8-32-bit band lost more than %.1f %% after normalisation" % a.limit)
    return 0


if __name__ == "__main__":
    sys.exit(main())
