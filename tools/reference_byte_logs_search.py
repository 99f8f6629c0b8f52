#!/usr/bin/env python3
"""Can the only reference-held numbers for this path — the byte totals logged in /root/reference/benchmarks.csv:2-66
and experiments/data/*.csv (GetTotalDataSize before / after the runs) — be reproduced from the segment layout model
(duckdb-adaptive-compression_amd/layout.py) + sdsl::size_in_bytes?  If one row count N explained them they would be
an independent pin for the oracle.  RESULT (round 2): no.  Kept so the negative result can be re-checked.

The uncompressed totals are the cleanest probe: with succinct off every data segment reports its segment_size
(ColumnSegment::GetDataSize, src/storage/table/column_segment.cpp:204-214), i.e. total = a * 262136 + b * 2048 * sizeof(T)
with a big (Storage::BLOCK_SIZE) and b small (first-of-flush) segments.
  1. Appender model (2048-row head segment + BLOCK_SIZE segments per 204 800-row flush): no N in
     [0, 2^33) gives any of the logged totals for 4- or 8-byte types.
  2. Any mix (a, b): solutions exist (the two sizes are coprime up to the factor 8) but their b / a ratios differ from
     benchmark to benchmark (1.07, 0.15, 0.24, 1.29 for uint32) where the Appender model needs exactly 1 / 4 — the logs
     were written by a build with a different segment sizing (or other NUM_INSERTS / column sets) than the snapshot's
     sources, which define NUM_INSERTS = 100 000 000 (benchmark/micro/succinct/zipf_distribution.cpp:13): that N gives
     515 957 496 B, the log says 7 729 800 312 B.
Run here only (reads /root/reference); prints the evidence as JSON."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lay = importlib.import_module("duckdb-adaptive-compression_amd.layout")

TARGETS = {  # uncompressed GetTotalDataSize figures, benchmarks.csv line numbers
    "NonSuccinctSequentialInsert (benchmarks.csv:12-16)": 38649001560,
    "NonSuccinctNormal/ZipfDistribution (:27-31, :42-46)": 7729800312,
    "NonSuccinctScanOOM (:52-56)": 19275421456,
    "NonSuccinctZipfScanOOM (:62-66)": 1927127848,
}


def uncompressed_total(n, ts):
    return sum(cap * ts for _, cap in lay.appender_segments(n, ts))


def appender_model_matches(total, ts):
    per_flush = uncompressed_total(lay.FLUSH_COUNT, ts)
    steps, acc = [], 0
    for _, cap in lay.appender_segments(lay.FLUSH_COUNT, ts):
        acc += cap * ts
        steps.append(acc)
    out = []
    for k in (total // per_flush - 1, total // per_flush):
        rest = total - k * per_flush
        if rest == 0 or rest in steps:
            out.append(int(k))
    return out


def mixes(total, ts, limit=4):
    big, small = lay.BLOCK_SIZE, lay.STANDARD_VECTOR_SIZE * ts
    out = []
    b = 0
    while b * small <= total and len(out) < limit:
        if (total - b * small) % big == 0:
            out.append({"big": (total - b * small) // big, "small": b})
        b += 1
    return out


def main():
    res = {"snapshot_NUM_INSERTS_100M_uint32_uncompressed_bytes": uncompressed_total(100_000_000, 4), "targets": {}}
    for name, t in TARGETS.items():
        res["targets"][name] = {"bytes": t}
        for ts in (4, 8):
            res["targets"][name]["type_size_%d" % ts] = {"appender_model_flush_counts": appender_model_matches(t, ts),
                                                         "any_mix_first_solutions": mixes(t, ts)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
