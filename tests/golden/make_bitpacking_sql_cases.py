#!/usr/bin/env python3
"""Writes tests/golden/bitpacking_sql_cases.json: the data shapes and expected results of the reference's own
sqllogictest files for the BITPACKING codec (/root/reference/test/sql/storage/compression/bitpacking/
{bitpacking_simple,bitpacking_delta,bitpacking_constant_delta,bitpacking_nulls,bitpacking_index_fetch,
bitpacking_filter_pushdown}.test), transcribed BY HAND as data — each entry cites the file:line it restates; nothing
is parsed from, copied out of or run in the reference.  The expected values are the ones the .test files print."""
import json
import os

U64MAX = 18446744073709551615
REF = "test/sql/storage/compression/bitpacking/"
modes = ["delta_for", "for", "constant_delta", "constant"]   # the foreach of every file (PRAGMA force_bitpacking_mode)
cases = []
cases.append({
    "id": "simple_bigint", "source": REF + "bitpacking_simple.test:12-60", "type": "int64", "forced_modes": modes,
    "pieces": [{"range": [0, 10000], "kind": "affine", "a": -1, "b": 0},          # :24  -i FROM range(0,10000)
               {"range": [0, 10000], "kind": "const", "value": 13371337}],        # :25
    "expect": [{"op": "head", "offset": 0, "rows": [0, -1, -2, -3, -4]},         # :30-37
               {"op": "head", "offset": 12000, "rows": [13371337] * 5},          # :39-46
               {"op": "avg", "value": 6683168.75},                                # :48-51
               {"op": "compression_is_bitpacking"}],                              # :53-56
    "comment_expectation": "':23 Insert multiple ranges so that each method can be used on at least on the the "
                           "ranges' - a comment, not an assertion of the file",
})
cases.append({
    "id": "delta_full_range_u64", "source": REF + "bitpacking_delta.test:12-42", "type": "uint64",
    "forced_modes": modes,
    "pieces": [{"range": [0, 1000000], "kind": "alternate", "values": [0, U64MAX]}],   # :24
    "expect": [{"op": "group_count", "rows": [[0, 500000], [U64MAX, 500000]]},          # :26-30
               {"op": "compression_is_bitpacking"}],                                    # :32-35
})
phys = [("int8", "int8", 1), ("int16", "int16", 1), ("int32", "int32", 1), ("int64", "int64", 1),
        ("uint8", "uint8", 1), ("uint16", "uint16", 1), ("uint32", "uint32", 1), ("uint64", "uint64", 1),
        # DECIMAL(w,1) is stored as the scaled integer of its physical type (widths up to 4 / 9 / 18 digits)
        ("decimal(4,1)", "int16", 10), ("decimal(8,1)", "int32", 10), ("decimal(12,1)", "int64", 10),
        ("decimal(18,1)", "int64", 10)]
for sql, p, scale in phys:
    c = {
        "id": "constant_delta_small_" + sql.replace("(", "_").replace(",", "_").replace(")", ""),
        "source": REF + "bitpacking_constant_delta.test:17-43", "sql_type": sql, "type": p, "forced_modes": modes,
        "pieces": [{"range": [0, 5], "kind": "affine", "a": 2 * scale, "b": 2 * scale}],      # :23  2+i*2
        "expect": [{"op": "compression_is_bitpacking"},                                        # :28-30
                   {"op": "head", "offset": 0, "rows": [2 * scale * k for k in range(1, 6)]}],  # :32-39
    }
    if scale != 1:
        c["note"] = "stored integers; a DECIMAL(w,1) value v is stored as 10 v"
    cases.append(c)
cases.append({
    "id": "constant_delta_130000_int64", "source": REF + "bitpacking_constant_delta.test:46-66", "type": "int64",
    "forced_modes": modes,
    "pieces": [{"range": [0, 130000], "kind": "affine", "a": 1, "b": 0}],     # :49-50
    "expect": [{"op": "compression_is_bitpacking"}, {"op": "avg", "value": 64999.5}],   # :55-62
})
cases.append({
    "id": "nulls_bigint", "source": REF + "bitpacking_nulls.test:12-47", "type": "int64", "forced_modes": modes,
    "pieces": [{"range": [0, 10000], "kind": "const", "value": 1337, "null_every": 5},     # :24
               {"range": [0, 10000], "kind": "affine", "a": 1, "b": 0, "null_every": 5},   # :28
               {"range": [0, 10000], "kind": "floordiv", "d": 2, "null_every": 5}],        # :32
    "expect": [{"op": "compression_is_bitpacking"},                                        # :37-39
               {"op": "sum_min_max", "sum": 70694000, "min": 0, "max": 9999}],             # :41-44
})
for sql, p in (("INTEGER", "int32"), ("UINT16", "uint16")):
    cases.append({
        "id": "index_fetch_" + p, "source": REF + "bitpacking_index_fetch.test:17-54", "sql_type": sql, "type": p,
        "forced_modes": modes,
        "pieces": [{"range": [0, 10000], "kind": "affine", "a": 1, "b": 0},        # :22-23
                   {"range": [10000, 20000], "kind": "const", "value": 1337},      # :25-26
                   {"range": [20000, 30000], "kind": "affine", "a": 1, "b": 0}],   # :28-29
        "row_of_id": "id == row (ids 0..29999 inserted in order)",
        "expect": [{"op": "compression_is_bitpacking"},                            # :31-33
                   {"op": "fetch", "row": 5000, "value": 5000},                    # :38-41
                   {"op": "fetch", "row": 12000, "value": 1337},                   # :43-46
                   {"op": "fetch", "row": 22000, "value": 22000}],                 # :48-51
    })
cases.append({
    "id": "filter_pushdown_int32", "source": REF + "bitpacking_filter_pushdown.test:17-58", "type": "int32",
    "forced_modes": modes,
    "pieces": [{"range": [0, 10000], "kind": "affine", "a": 1, "b": 0},            # :21-22
               {"range": [20000, 30000], "kind": "const", "value": 1337},          # :24-25
               {"range": [30000, 40000], "kind": "affine", "a": 1, "b": 0}],       # :27-28
    "row_of_id": "ids 0..9999 -> rows 0..9999, 20000..29999 -> rows 10000..19999, 30000..39999 -> rows 20000..29999",
    "expect": [{"op": "compression_is_bitpacking"},                                 # :33-35
               {"op": "filter_eq", "key": 1337, "sum": 13371337, "min": 1337, "max": 1337, "count": 10001},  # :39-42
               {"op": "fetch", "row": 5000, "value": 5000},                         # :45-48
               {"op": "rows_agg",                                                   # :51-54  id::INT64 % 1000 = 0
                "rows": list(range(0, 10000, 1000)) + list(range(10000, 20000, 1000)) + list(range(20000, 30000, 1000)),
                "sum": 403370, "min": 0, "max": 39000, "count": 30}],
})
doc = {
    "_about": "Data shapes and expected results of the reference's own sqllogictest files for the BITPACKING codec "
              "(/root/reference/test/sql/storage/compression/bitpacking/*.test), transcribed as data: inputs as piece "
              "descriptions of the INSERT ... FROM range() statements, outputs as the rows / aggregates the files "
              "expect under every PRAGMA force_bitpacking_mode they loop over.  'compression_is_bitpacking' = the "
              "file's pragma_storage_info check: the codec must be able to encode the column under that forced mode. "
              "Written by tests/golden/make_bitpacking_sql_cases.py; no text of the .test files is kept.",
    "piece_kinds": {"affine": "value = a * i + b for i in range", "const": "value", "alternate": "values[i % 2]",
                    "floordiv": "value = i // d (integer division of BIGINT operands)",
                    "null_every": "row is NULL when i % null_every == 0"},
    "cases": cases,
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitpacking_sql_cases.json")
json.dump(doc, open(out, "w"), indent=1)
print(len(cases), "cases ->", out)
