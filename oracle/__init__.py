"""ctypes loader for the CPU oracle (oracle/succinct_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from the product package.  See the header of succinct_oracle.c for the
pinning status and the reference file:line each function follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

FN_UNCOMPRESSED = 1
FN_SUCCINCT = 10
U64_MAX = 0xFFFFFFFFFFFFFFFF


def build(force=False):
    src = os.path.join(_HERE, "succinct_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64, u32, i32, vp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p
        sig = {
            "orc_hi": (u32, [u64]),
            "orc_size_in_bytes": (u64, [u64]),
            "orc_serialize": (u64, [vp, u64, C.c_uint, vp]),
            "orc_load": (u64, [vp, u64, C.POINTER(u64), C.POINTER(C.c_uint), vp, u64]),
            "orc_width_from_succinct": (u32, [u64, u64, i32]),
            "orc_width_from_uncompressed": (u32, [u64, u64, i32]),
            "orc_seg_create": (vp, [C.c_uint, i32, u64, i32, i32, i32]),
            "orc_seg_destroy": (None, [vp]),
            "orc_seg_compact": (None, [vp, i32]),
            "orc_seg_uncompact": (None, [vp, i32]),
            "orc_seg_append": (u64, [vp, vp, vp, vp, u64, u64, i32]),
            "orc_seg_scan_partial": (None, [vp, u64, u64, vp, i32, i32]),
            "orc_seg_scan": (None, [vp, u64, u64, vp, i32, i32]),
            "orc_seg_fetch_row": (None, [vp, u64, vp]),
            "orc_seg_data_size": (u64, [vp]),
            "orc_seg_count": (u64, [vp]),
            "orc_seg_min": (u64, [vp]),
            "orc_seg_max": (u64, [vp]),
            "orc_seg_width": (u32, [vp]),
            "orc_seg_bit_size": (u64, [vp]),
            "orc_seg_compacted": (i32, [vp]),
            "orc_seg_function": (i32, [vp]),
            "orc_seg_num_reads": (u64, [vp]),
            "orc_seg_reset_reads": (None, [vp]),
            "orc_seg_words": (vp, [vp]),
            "orc_seg_num_words": (u64, [vp]),
            "orc_seg_block": (vp, [vp]),
            "orc_analyze_flat": (None, [vp, u64, C.c_uint, i32, i32, vp, u64, vp, vp]),
            "orc_pack_flat": (None, [vp, u64, C.c_uint, i32, vp, u64, u64, C.c_uint, vp]),
            "orc_unpack_flat": (None, [vp, u64, u64, C.c_uint, u64, C.c_uint, vp]),
            "orc_scan_segments_mt": (None, [vp, vp, vp, vp, vp, u64, C.c_uint, i32, vp, i32]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def hi(x):
    return lib().orc_hi(x & U64_MAX)


def size_in_bytes(bit_size):
    return lib().orc_size_in_bytes(bit_size)


def serialize(words, bit_size, width):
    """sdsl::int_vector<0>::serialize of a vector with these words / m_size / m_width -> bytes."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    assert len(words) >= (bit_size + 63) // 64
    out = np.zeros(size_in_bytes(bit_size), dtype=np.uint8)
    n = lib().orc_serialize(_ptr(words), bit_size, width, _ptr(out))
    assert n == len(out)
    return out.tobytes()


def load(data):
    """sdsl::int_vector<0>::load -> (bit_size, width, words)"""
    buf = np.frombuffer(data, dtype=np.uint8)
    bs, w = C.c_uint64(), C.c_uint()
    words = np.zeros(max(1, len(buf) // 8), dtype=np.uint64)
    n = lib().orc_load(_ptr(buf), len(buf), C.byref(bs), C.byref(w), _ptr(words), len(words))
    assert n, "short image"
    return bs.value, w.value, words[:(bs.value + 63) // 64]


def width_from_succinct(mn, mx, padded=False):
    return lib().orc_width_from_succinct(mn & U64_MAX, mx & U64_MAX, int(padded))


def width_from_uncompressed(mn, mx, padded=False):
    return lib().orc_width_from_uncompressed(mn & U64_MAX, mx & U64_MAX, int(padded))


class Segment:
    """Model of the reference's ColumnSegment succinct state (column_segment.hpp:60-64,190-214)."""

    def __init__(self, dtype, segment_size=262136, succinct_enabled=True, adaptive=False, padded=False,
                 store_min=True):
        self.dtype = np.dtype(dtype)
        assert self.dtype.kind in "iu"
        self.store_min = int(store_min)
        self._h = lib().orc_seg_create(self.dtype.itemsize, int(self.dtype.kind == "i"), segment_size,
                                       int(succinct_enabled), int(adaptive), int(padded))
        if not self._h:
            raise ValueError("unsupported type")
        self.segment_size = segment_size

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_seg_destroy(self._h)
            self._h = None

    def append(self, vals, validity=None, sel=None, offset=0, count=None):
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        if count is None:
            count = len(vals) - offset if sel is None else len(sel) - offset
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        if sel is not None:
            sel = np.ascontiguousarray(sel, dtype=np.uint32)
        return lib().orc_seg_append(self._h, _ptr(vals), _ptr(validity), _ptr(sel), offset, count, self.store_min)

    def compact(self):
        lib().orc_seg_compact(self._h, self.store_min)

    def uncompact(self, correct=None):
        lib().orc_seg_uncompact(self._h, self.store_min if correct is None else int(correct))

    def scan_partial(self, start, n, mode=0, with_copy=False):
        out = np.empty(n, dtype=self.dtype)
        lib().orc_seg_scan_partial(self._h, start, n, _ptr(out), mode, int(with_copy))
        return out

    def scan(self, start, n, mode=0):
        out = np.empty(n, dtype=self.dtype)
        lib().orc_seg_scan(self._h, start, n, _ptr(out), mode, self.store_min)
        return out

    def fetch_row(self, row):
        out = np.empty(1, dtype=self.dtype)
        lib().orc_seg_fetch_row(self._h, row, _ptr(out))
        return out[0]

    count = property(lambda s: lib().orc_seg_count(s._h))
    min_factor = property(lambda s: lib().orc_seg_min(s._h))
    max_factor = property(lambda s: lib().orc_seg_max(s._h))
    width = property(lambda s: lib().orc_seg_width(s._h))
    bit_size = property(lambda s: lib().orc_seg_bit_size(s._h))
    compacted = property(lambda s: bool(lib().orc_seg_compacted(s._h)))
    function = property(lambda s: lib().orc_seg_function(s._h))
    data_size = property(lambda s: lib().orc_seg_data_size(s._h))
    num_reads = property(lambda s: lib().orc_seg_num_reads(s._h))

    def reset_reads(self):
        lib().orc_seg_reset_reads(self._h)

    @property
    def words(self):
        n = lib().orc_seg_num_words(self._h)
        p = lib().orc_seg_words(self._h)
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n,)).copy()

    @property
    def block(self):
        p = lib().orc_seg_block(self._h)
        n = self.segment_size // self.dtype.itemsize
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * self.dtype.itemsize,)).copy()
        return raw.view(self.dtype)


def analyze_flat(vals, rule=0, validity=None, vbit0=0):
    vals = np.ascontiguousarray(vals)
    mn, mx = C.c_uint64(), C.c_uint64()
    if validity is not None:
        validity = np.ascontiguousarray(validity, dtype=np.uint64)
    lib().orc_analyze_flat(_ptr(vals), len(vals), vals.dtype.itemsize, int(vals.dtype.kind == "i"), rule,
                           _ptr(validity), vbit0, C.addressof(mn), C.addressof(mx))
    return mn.value, mx.value


def pack_flat(vals, mn, w, validity=None, vbit0=0):
    vals = np.ascontiguousarray(vals)
    nwords = (len(vals) * w + 63) >> 6
    words = np.zeros(nwords + 1, dtype=np.uint64)
    if validity is not None:
        validity = np.ascontiguousarray(validity, dtype=np.uint64)
    lib().orc_pack_flat(_ptr(vals), len(vals), vals.dtype.itemsize, int(vals.dtype.kind == "i"), _ptr(validity),
                        vbit0, mn & U64_MAX, w, _ptr(words))
    return words[:nwords]


def unpack_flat(words, start, n, w, add, dtype):
    dtype = np.dtype(dtype)
    words = np.ascontiguousarray(np.concatenate([words, np.zeros(1, dtype=np.uint64)]))
    out = np.empty(n, dtype=dtype)
    lib().orc_unpack_flat(_ptr(words), start, n, w, add & U64_MAX, dtype.itemsize, _ptr(out))
    return out


def scan_segments_mt(seg_words, counts, widths, adds, out_offs, dtype, out, with_copy=False, threads=1):
    """Full scan of packed segments, 2048 values per call, on `threads` host threads (CPU baseline)."""
    dtype = np.dtype(dtype)
    n = len(seg_words)
    ptrs = (C.c_void_p * n)(*[w.ctypes.data for w in seg_words])
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    widths = np.ascontiguousarray(widths, dtype=np.uint8)
    adds = np.ascontiguousarray(adds, dtype=np.uint64)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    lib().orc_scan_segments_mt(C.cast(ptrs, C.c_void_p), _ptr(counts), _ptr(widths), _ptr(adds), _ptr(out_offs), n,
                               dtype.itemsize, int(with_copy), _ptr(out), threads)
    return out
