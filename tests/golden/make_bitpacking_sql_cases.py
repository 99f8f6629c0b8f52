#!/usr/bin/env python3
"""Writes tests/golden/bitpacking_sql_cases.json: the data shapes and expected results of the reference's own
sqllogictest files for the BITPACKING codec (/root/reference/test/sql/storage/compression/bitpacking/
{bitpacking_simple,bitpacking_delta,bitpacking_constant_delta,bitpacking_nulls,bitpacking_index_fetch,
bitpacking_filter_pushdown}.test and {bitpacking_bitwidths,bitpacking_types}.test_coverage), transcribed BY HAND as data — each entry cites the file:line it restates; nothing
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

# ---- bitpacking_bitwidths.test_coverage: values that compress to every width, per type size and forced mode ----------
BW = REF + "bitpacking_bitwidths.test_coverage"
for ts in (8, 16, 32, 64):
    cases.append({
        "id": "bitwidths_unsigned_%d" % ts, "source": BW + ":20-40", "sql_type": "UINT%d" % ts, "type": "uint%d" % ts,
        "forced_modes": modes,
        "pieces": [{"range": [0, ts * 2048], "kind": "pow2_step", "step": 2048, "sign": 1}],   # :21  2**(i/2048)
        "expect": [{"op": "distinct_values", "n": ts},                # :26-29  count(*) = typesize groups
                   {"op": "every_group_count", "count": 2048}],       # :31-34  DISTINCT(c_i) = 2048
    })
    cases.append({
        "id": "bitwidths_signed_neg_%d" % ts, "source": BW + ":42-54", "sql_type": "INT%d" % ts, "type": "int%d" % ts,
        "forced_modes": modes,
        "pieces": [{"range": [0, ts * 2048], "kind": "pow2_step", "step": 2048, "sign": -1}],  # :43  -(2**(i/2048))
        "expect": [{"op": "distinct_values", "n": ts},                # :48-51
                   {"op": "every_group_count", "count": 2048}],       # :53-56
    })
    cases.append({
        "id": "bitwidths_signed_pos_%d" % ts, "source": BW + ":65-75", "sql_type": "INT%d" % ts, "type": "int%d" % ts,
        "forced_modes": modes,
        "pieces": [{"range": [0, (ts - 1) * 2048], "kind": "pow2_step", "step": 2048, "sign": 1}],  # :66
        "expect": [{"op": "distinct_values", "n": ts - 1}],           # :71-74  (the file's next query reads test_signed_neg again)
    })
for sql, p in (("TINYINT", "int8"), ("SMALLINT", "int16"), ("INTEGER", "int32"), ("BIGINT", "int64"),
               ("UTINYINT", "uint8"), ("USMALLINT", "uint16"), ("UINTEGER", "uint32"), ("UBIGINT", "uint64"),
               ("BOOL", "uint8")):
    c = {
        "id": "nullpack_" + sql.lower(), "source": BW + ":100-112", "sql_type": sql, "type": p,
        "forced_modes": ["constant"],   # the loop over the modes has ended at :98; the last PRAGMA set stays in force
        "pieces": [{"range": [0, 12000], "kind": "floordiv_mod", "d": 3000, "m": 2}],     # :103  (i/3000)%2
        "expect": [{"op": "avg", "value": 0.5}],                                          # :108-111
    }
    if sql == "BOOL":
        c["note"] = "a BOOL is one byte holding 0 / 1 (PhysicalType::BOOL); the codec is run on it as uint8"
    cases.append(c)

# ---- bitpacking_types.test_coverage: every numeric type, and the numerical limits -------------------------------------
TY = REF + "bitpacking_types.test_coverage"
for sql, p, scale in phys:
    c = {
        "id": "types_mod3_" + sql.replace("(", "_").replace(",", "_").replace(")", ""), "source": TY + ":17-31",
        "sql_type": sql, "type": p, "forced_modes": modes,
        "pieces": [{"range": [0, 10000], "kind": "mod", "m": 3, "scale": scale}],         # :19  MOD(i,3)::type
        "expect": [{"op": "min_max_avg_count", "min": 0, "max": 2 * scale, "avg": 0.9999 * scale, "count": 10000},  # :21-24
                   {"op": "filter_eq", "key": scale, "sum": 3333 * scale, "min": scale, "max": scale, "count": 3333}],  # :26-29
    }
    if scale != 1:
        c["note"] = "stored integers; a DECIMAL(w,1) value v is stored as 10 v"
    cases.append(c)
cases.append({
    "id": "types_int32_full_range_is_refused", "source": TY + ":35-46", "type": "int32", "forced_modes": modes,
    "pieces": [{"range": [0, 2], "kind": "list", "values": [-2147483648, 2147483647]}],   # :39
    "expect": [{"op": "compression_is_not_bitpacking"}],                                  # :41-45  'Uncompressed'
    "comment_expectation_refusal": "':37 Range too big to force bitpacking'",
})
for sql, p, bits in (("INT64", "int64", 64), ("INT32", "int32", 32), ("INT16", "int16", 16), ("TINYINT", "int8", 8)):
    cases.append({
        "id": "types_all_but_one_bit_" + p, "source": TY + ":51-75", "sql_type": sql, "type": p, "forced_modes": modes,
        "pieces": [{"range": [0, 2], "kind": "list", "values": [-(2 ** (bits - 2)), 2 ** (bits - 2) - 1]}],   # :57
        "expect": [{"op": "compression_is_bitpacking"},                                    # :62-71
                   {"op": "avg", "value": -0.5}, {"op": "count_rows", "count": 2}],        # :73-76
    })
limits = [("uint64", "a", 3256, 5.665461940419088e+18), ("uint32", "b", 2256, 1903811655.4255319),
          ("uint16", "c", 1256, 25716.671974522294), ("uint8", "d", 256, 127.5)]
for p, col, live, avg in limits:
    pieces = [{"range": [0, 256], "kind": "affine", "a": 1, "b": 0}]                                    # :84
    pieces.append({"range": [31768, 32768], "kind": "affine", "a": 1, "b": 0} if live >= 1256
                  else {"range": [31768, 32768], "kind": "null"})                                        # :85
    pieces.append({"range": [4294966295, 4294967295], "kind": "affine", "a": 1, "b": 0} if live >= 2256
                  else {"range": [4294966295, 4294967295], "kind": "null"})                              # :86
    pieces.append({"range": [0, 1000], "kind": "affine_u64", "a": -1, "b": U64MAX} if live >= 3256
                  else {"range": [0, 1000], "kind": "null"})                                             # :87
    cases.append({
        "id": "types_unsigned_limits_" + col, "source": TY + ":81-95", "type": p, "forced_modes": modes,
        "pieces": pieces,
        "expect": [{"op": "avg_approx", "value": avg, "rel": 1e-15},        # :92-95 (a double as the file prints it)
                   {"op": "count_rows", "count": 3256}, {"op": "count_valid", "count": live}],
        # the file does not look at pragma_storage_info for this table: where BitpackingAnalyze declines the column
        # under a forced mode (a 2048-row group of NULLs only has no FOR / DELTA form: its min / max are the initial
        # limits and their difference overflows) DuckDB stores it uncompressed and the query results hold trivially
        "may_be_refused": True,
    })
cases.append({
    "id": "types_bool_filter", "source": TY + ":97-108", "sql_type": "BOOL", "type": "uint8", "forced_modes": modes,
    "pieces": [{"range": [0, 10000], "kind": "mod", "m": 2, "scale": 1}],                 # :101  CAST(i%2 as BOOL)
    "expect": [{"op": "filter_eq", "key": 1, "sum": 5000, "min": 1, "max": 1, "count": 5000}],   # :106-109
    "note": "a BOOL is one byte holding 0 / 1 (PhysicalType::BOOL); the codec is run on it as uint8",
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
                    "null_every": "row is NULL when i % null_every == 0",
                    "pow2_step": "value = sign * 2 ** (i // step)", "floordiv_mod": "value = (i // d) % m",
                    "mod": "value = (i % m) * scale", "list": "values[i]", "null": "every row NULL",
                    "affine_u64": "value = (a * i + b) mod 2**64"},
    "cases": cases,
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitpacking_sql_cases.json")
json.dump(doc, open(out, "w"), indent=1)
print(len(cases), "cases ->", out)
