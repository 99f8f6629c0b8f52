"""Shared by tests/test_bitpacking_sql_cases.py (CPU: the oracle) and tests/test_gpu_bitpacking_sql_cases.py (GPU: the
C ABI): builds the columns tests/golden/bitpacking_sql_cases.json describes and checks a decoded column against the
results the reference's .test files expect."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODE_CODE = {"auto": 0, "constant": 1, "constant_delta": 2, "delta_for": 3, "for": 4}


def load_cases():
    with open(os.path.join(HERE, "golden", "bitpacking_sql_cases.json")) as f:
        return json.load(f)["cases"]


def case_ids():
    return [c["id"] for c in load_cases()]


def build_column(case):
    """-> (values of the case's physical type, validity as a bool array or None)"""
    dtype = np.dtype(case["type"])
    vals, valid = [], []
    for p in case["pieces"]:
        lo, hi = p["range"]
        i = np.arange(lo, hi, dtype=np.int64)
        if p["kind"] == "affine":
            v = (p["a"] * i + p["b"]).astype(dtype)
        elif p["kind"] == "const":
            v = np.full(len(i), p["value"], dtype=dtype)
        elif p["kind"] == "alternate":
            v = np.array(p["values"], dtype=dtype)[i % 2]
        elif p["kind"] == "floordiv":
            v = (i // p["d"]).astype(dtype)
        elif p["kind"] == "pow2_step":
            v = np.array([p["sign"] * 2 ** int(k) for k in i // p["step"]], dtype=object).astype(dtype)
        elif p["kind"] == "floordiv_mod":
            v = ((i // p["d"]) % p["m"]).astype(dtype)
        elif p["kind"] == "mod":
            v = ((i % p["m"]) * p["scale"]).astype(dtype)
        elif p["kind"] == "list":
            v = np.array(p["values"], dtype=object).astype(dtype)
        elif p["kind"] == "affine_u64":
            v = np.array([(p["a"] * int(k) + p["b"]) % 2 ** 64 for k in i], dtype=object).astype(dtype)
        elif p["kind"] == "null":
            v = np.zeros(len(i), dtype=dtype)
        else:
            raise ValueError(p["kind"])
        vals.append(v)
        if p["kind"] == "null":
            valid.append(np.zeros(len(i), dtype=bool))
        else:
            valid.append(i % p["null_every"] != 0 if "null_every" in p else np.ones(len(i), dtype=bool))
    valid = np.concatenate(valid)
    return np.concatenate(vals), (None if valid.all() else valid)


def refused(case):
    """the file expects 'Uncompressed': the codec must decline the column under every forced mode"""
    return any(e["op"] == "compression_is_not_bitpacking" for e in case["expect"])


def check_expectations(case, decoded, valid, fetch=None, filter_eq=None):
    """decoded: the column as the codec returns it (NULL rows hold anything).  fetch(row) -> value through the codec's
    point look-up; filter_eq(key) -> (sum, min, max, count) through the codec's own filter path (both optional: the
    decoded column is used when they are None)."""
    ok = np.ones(len(decoded), dtype=bool) if valid is None else valid
    live = decoded[ok]
    as_int = [int(x) for x in live]
    for e in case["expect"]:
        op = e["op"]
        if op in ("compression_is_bitpacking", "compression_is_not_bitpacking"):
            continue  # asserted by the caller: the column was (not) encodable under the forced mode
        if op == "head":
            assert [int(x) for x in decoded[e["offset"]:e["offset"] + len(e["rows"])]] == e["rows"], e
        elif op == "avg":
            assert sum(as_int) / len(as_int) == e["value"], e
        elif op == "sum_min_max":
            assert (sum(as_int), min(as_int), max(as_int)) == (e["sum"], e["min"], e["max"]), e
        elif op == "group_count":
            keys, counts = np.unique(live, return_counts=True)
            assert [[int(k), int(c)] for k, c in zip(keys, counts)] == e["rows"], e
        elif op == "fetch":
            got = int(fetch(e["row"])) if fetch else int(decoded[e["row"]])
            assert got == e["value"], e
        elif op == "filter_eq":
            if filter_eq:
                got = filter_eq(e["key"])
            else:
                hit = live[live == e["key"]]
                got = (int(hit.astype(np.int64).sum()), int(hit.min()), int(hit.max()), len(hit))
            assert tuple(got) == (e["sum"], e["min"], e["max"], e["count"]), (e, got)
        elif op == "rows_agg":
            r = decoded[np.array(e["rows"])]
            assert (int(r.astype(np.int64).sum()), int(r.min()), int(r.max()), len(r)) == (e["sum"], e["min"], e["max"], e["count"]), e
        elif op == "distinct_values":
            assert len(np.unique(live)) == e["n"], e
        elif op == "every_group_count":
            assert set(np.unique(live, return_counts=True)[1].tolist()) == {e["count"]}, e
        elif op == "min_max_avg_count":
            assert (min(as_int), max(as_int), len(as_int)) == (e["min"], e["max"], e["count"]), e
            assert abs(sum(as_int) / len(as_int) - e["avg"]) <= 1e-9 * max(1.0, abs(e["avg"])), e
        elif op == "avg_approx":
            assert abs(sum(as_int) / len(as_int) - e["value"]) <= e["rel"] * abs(e["value"]), (e, sum(as_int) / len(as_int))
        elif op == "count_rows":
            assert len(decoded) == e["count"], e
        elif op == "count_valid":
            assert len(live) == e["count"], e
        else:
            raise ValueError(op)
