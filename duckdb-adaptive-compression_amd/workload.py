"""Synthetic columns for the benchmark harness (host only): binding of csrc/workload.c."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libadacworkload.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("libadacworkload.so not built: run __graft_entry__.build()")
        L = C.CDLL(_PATH)
        L.adacw_zipf_fill.restype = C.c_int
        L.adacw_zipf_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_double, C.c_double, C.c_uint64,
                                      C.c_uint32, C.c_int]
        L.adacw_zipf_fill_range.restype = C.c_int
        L.adacw_zipf_fill_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint32, C.c_int]
        L.adacw_mt19937_stream.restype = None
        L.adacw_mt19937_stream.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def zipf_column(n, dtype=np.uint64, domain=2 ** 32 - 1, skew=1.0, base=0, seed=42, threads=None):
    """n values base + Zipf(domain, skew) (draws in [1, domain]) — the value distribution of config C2
    (Zipf(n = 2^32 - 1, s = 1.0), mt19937 seed 42)."""
    dtype = np.dtype(dtype)
    out = np.empty(n, dtype=dtype)
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    rc = _load().adacw_zipf_fill(out.ctypes.data, n, dtype.itemsize, float(domain), float(skew), base, seed, threads)
    if rc != 0:
        raise ValueError("adacw_zipf_fill failed: %d" % rc)
    return out


def zipf_column_range(row_lo, row_hi, dtype=np.uint64, domain=2 ** 32 - 1, skew=1.0, base=0, seed=42, threads=None):
    """Rows [row_lo, row_hi) of the global column zipf_column(n >= row_hi, ..., seed) — one shard's slice of ONE
    column (config C4: a 1 B-row column whose segments are partitioned by id across the GPUs)."""
    dtype = np.dtype(dtype)
    out = np.empty(row_hi - row_lo, dtype=dtype)
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    rc = _load().adacw_zipf_fill_range(out.ctypes.data, row_lo, row_hi, dtype.itemsize, float(domain), float(skew),
                                       base, seed, threads)
    if rc != 0:
        raise ValueError("adacw_zipf_fill_range failed: %d" % rc)
    return out


def mt19937_stream(seed, n):
    out = np.empty(n, dtype=np.uint32)
    _load().adacw_mt19937_stream(seed, out.ctypes.data, n)
    return out
