"""ctypes loader for oracle/bitpacking_oracle.c (DuckDB BITPACKING codec restatement) and, when it has been
built in this container, for oracle/_ref/libfastpfor_ref.so (the reference's real fastpforlib).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbporacle.so")
_REF = os.path.join(_HERE, "_ref", "libfastpfor_ref.so")

MODE_AUTO, MODE_CONSTANT, MODE_CONSTANT_DELTA, MODE_DELTA_FOR, MODE_FOR = 0, 1, 2, 3, 4
MODE_NAMES = {1: "constant", 2: "constant_delta", 3: "delta_for", 4: "for"}
GROUP = 2048
BLOCK_SIZE = 262144 - 8

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "bitpacking_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libbporacle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_LIB)
        u64, vp, i32, u32 = C.c_uint64, C.c_void_p, C.c_int, C.c_uint
        for name, res, args in (
            ("bp_pack_group", None, [vp, u32, vp]), ("bp_unpack_group", None, [vp, u32, vp]),
            ("bp_compress", vp, [vp, vp, u64, u32, i32, i32, i32]), ("bp_free", None, [vp]),
            ("bp_num_segments", u64, [vp]), ("bp_segment_block", vp, [vp, u64]), ("bp_segment_count", u64, [vp, u64]),
            ("bp_segment_start", u64, [vp, u64]), ("bp_segment_size", u64, [vp, u64]),
            ("bp_groups_by_mode", u64, [vp, i32]), ("bp_scan", None, [vp, u32, i32, u64, u64, u64, vp]),
            ("bp_group_info", i32, [vp, u32, u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        ):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def ref_available():
    """True when oracle/_ref has been built here.  Does NOT load it: the committed vectors under tests/golden/ exist so
    that the reference-built library is never mapped into a test process that does not run the live comparison."""
    return os.path.exists(_REF)


def ref():
    """The reference's fastpforlib, or None when oracle/_ref has not been built (e.g. on the GPU box)."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        _ref = C.CDLL(_REF)
    return _ref


def pack_group(vals, w):
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    assert len(vals) == 32
    out = np.zeros(max(4 * w, 1), dtype=np.uint8)
    lib().bp_pack_group(vals.ctypes.data, w, out.ctypes.data)
    return out[:4 * w]


def unpack_group(buf, w):
    buf = np.ascontiguousarray(np.concatenate([np.frombuffer(bytes(buf), dtype=np.uint8), np.zeros(8, np.uint8)]))
    out = np.zeros(32, dtype=np.uint64)
    lib().bp_unpack_group(buf.ctypes.data, w, out.ctypes.data)
    return out


class Compressed:
    """A column compressed into BITPACKING segments (256 KiB block images)."""

    def __init__(self, vals, validity=None, force_mode=MODE_AUTO, null_zero=False):
        vals = np.ascontiguousarray(vals)
        self.dtype = vals.dtype
        self.n = len(vals)
        v8 = None if validity is None else np.ascontiguousarray(validity, dtype=np.uint8)
        self._h = lib().bp_compress(vals.ctypes.data, None if v8 is None else v8.ctypes.data, self.n,
                                    vals.dtype.itemsize, int(vals.dtype.kind == "i"), force_mode, int(null_zero))
        if not self._h:
            raise ValueError("the BITPACKING codec cannot encode this column (Flush returned false)")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bp_free(self._h)
            self._h = None

    @property
    def nseg(self):
        return lib().bp_num_segments(self._h)

    def block(self, i):
        p = lib().bp_segment_block(self._h, i)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(BLOCK_SIZE,)).copy()

    def count(self, i):
        return lib().bp_segment_count(self._h, i)

    def start(self, i):
        return lib().bp_segment_start(self._h, i)

    def size(self, i):
        return lib().bp_segment_size(self._h, i)

    def groups_by_mode(self):
        return {MODE_NAMES[m]: lib().bp_groups_by_mode(self._h, m) for m in MODE_NAMES}

    def scan(self, i, start=0, n=None):
        n = self.count(i) - start if n is None else n
        out = np.empty(n, dtype=self.dtype)
        blk = self.block(i)
        lib().bp_scan(blk.ctypes.data, self.dtype.itemsize, int(self.dtype.kind == "i"), self.count(i), start, n,
                      out.ctypes.data)
        return out

    def group_info(self, i, g):
        blk = self.block(i)
        off, w = C.c_uint32(), C.c_uint32()
        mode = lib().bp_group_info(blk.ctypes.data, self.dtype.itemsize, g, C.byref(off), C.byref(w))
        return mode, off.value, w.value
