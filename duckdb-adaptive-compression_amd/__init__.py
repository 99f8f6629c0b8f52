"""adacodec — MI355X-native succinct column-segment codec (host-side Python binding of libadacodec.so).

The product is the C-ABI shared library built from csrc/ (include/adacodec.h); this module is the thin
ctypes binding used by the tests, bench.py and __graft_entry__.py.  It never computes anything itself and
has NO CPU fallback: if the HIP extension is missing or no gfx950 device is usable, calls raise AdacError.

Import with importlib (the directory name is not a Python identifier):
    adac = importlib.import_module("duckdb-adaptive-compression_amd")
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADAC_LIB: another build of the same library (same-box A/B of two builds, tools/ab_encode.py); default: in-tree
LIB_PATH = os.environ.get("ADAC_LIB") or os.path.join(_HERE, "libadacodec.so")

# adac_type == duckdb::PhysicalType codes
UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64 = 2, 3, 4, 5, 6, 7, 8, 9
RULE_APPEND, RULE_RECOMPACT = 0, 1
NO_MIN = 0xFFFFFFFFFFFFFFFF
SEG_PACKED = 1

_NP2TYPE = {
    np.dtype(np.uint8): UINT8, np.dtype(np.int8): INT8, np.dtype(np.uint16): UINT16, np.dtype(np.int16): INT16,
    np.dtype(np.uint32): UINT32, np.dtype(np.int32): INT32, np.dtype(np.uint64): UINT64, np.dtype(np.int64): INT64,
}
_TYPE2NP = {v: k for k, v in _NP2TYPE.items()}

STATUS_NAMES = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED_TYPE", 3: "DEVICE", 4: "OUT_OF_MEMORY", 5: "NO_DEVICE"}

UNPACK_JOB_DTYPE = np.dtype([
    ("word_off", np.uint64), ("min", np.uint64), ("out_off", np.uint64), ("start", np.uint32), ("count", np.uint32),
    ("width", np.uint8), ("flags", np.uint8), ("reserved", np.uint16), ("reserved2", np.uint32),
])
assert UNPACK_JOB_DTYPE.itemsize == 40

SEGMENT_DESC_DTYPE = np.dtype([
    ("word_off", np.uint64), ("val_off", np.uint64), ("min", np.uint64), ("count", np.uint32),
    ("width", np.uint8), ("flags", np.uint8), ("reserved", np.uint16),
])
assert SEGMENT_DESC_DTYPE.itemsize == 32


class AdacError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        detail = ""
        try:
            detail = lib().adac_last_error().decode() if status == 3 else ""
        except Exception:  # pragma: no cover
            pass
        super().__init__("%s failed: ADAC_ERR_%s %s" % (where, STATUS_NAMES.get(status, status), detail))


def physical_type(dtype):
    try:
        return _NP2TYPE[np.dtype(dtype)]
    except (KeyError, TypeError):
        raise AdacError(2, "physical_type(%r)" % (dtype,))


def numpy_dtype(ptype):
    return _TYPE2NP[ptype]


def kernel_source_sha256():
    """Hash of EVERY device source of libadacodec (adac_kernels.hip, adac_internal.h and each csrc/*.inl it includes):
    a PMC traffic record belongs to the kernels it was measured on, and bench.py attaches one only while this hash
    equals the record's (profiles/summarize.py writes the same hash)."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for f in ["adac_kernels.hip", "adac_internal.h"] + sorted(x for x in os.listdir(csrc) if x.endswith(".inl")):
        h.update(f.encode())
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False):
    """Compile libadacodec.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "adacodec.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "all"])
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); every symbol include/adacodec.h declares
_u64, _u32, _u8, _int, _vp, _sz = C.c_uint64, C.c_uint32, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t
_P = C.POINTER
SIGNATURES = {
    "adac_abi_version": (_int, []),
    "adac_status_string": (C.c_char_p, [_int]),
    "adac_last_error": (C.c_char_p, []),
    "adac_type_is_supported": (_int, [_int]),
    "adac_type_size": (_u32, [_int]),
    "adac_hi": (_u32, [_u64]),
    "adac_width": (_u8, [_u64, _u64, _int, _int]),
    "adac_stored_min": (_u64, [_u64, _u64, _u8]),
    "adac_packed_words": (_u64, [_u64, _u8]),
    "adac_size_in_bytes": (_u64, [_u64, _u8]),
    "adac_arena_words": (_u64, [_u64, _u8]),
    "adac_block_bytes": (_u64, [_u64, _u8]),
    "adac_block_write": (_u64, [_vp, _int, _vp, _vp, _u64]),
    "adac_block_read": (_int, [_vp, _u64, _vp, _P(_int), _vp, _u64]),
    "adac_block_peek": (_int, [_vp, _u64, _vp, _P(_int)]),
    "adac_block_stride": (_u64, [_u64, _u8]),
    "adac_blocks_write": (_int, [_vp, _int, _vp, _vp, _u64, _vp, _vp]),
    "adac_blocks_read": (_int, [_vp, _int, _vp, _vp, _u64, _vp, _vp]),
    "adac_tile_values": (_u32, [_int]),
    "adac_set_tuning": (_int, [C.c_char_p, _int]),
    "adac_debug_encode_stamps": (_int, [_vp, _u64]),
    "adac_device_count": (_int, []),
    "adac_ctx_create": (_int, [_int, _vp, _P(_vp)]),
    "adac_ctx_destroy": (None, [_vp]),
    "adac_ctx_sync": (_int, [_vp]),
    "adac_ctx_stream": (_vp, [_vp]),
    "adac_ctx_device": (_int, [_vp]),
    "adac_dev_alloc": (_int, [_vp, _sz, _P(_vp)]),
    "adac_dev_free": (_int, [_vp, _vp]),
    "adac_dev_memset": (_int, [_vp, _vp, _int, _sz]),
    "adac_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "adac_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "adac_memcpy_d2h_async": (_int, [_vp, _vp, _vp, _sz]),
    "adac_host_alloc_pinned": (_int, [_vp, _sz, _P(_vp)]),
    "adac_host_free_pinned": (_int, [_vp, _vp]),
    "adac_timer_start": (_int, [_vp]),
    "adac_timer_stop": (_int, [_vp, _P(C.c_float)]),
    "adac_layout_create": (_int, [_vp, _int, _vp, _vp, _u64, _P(_vp)]),
    "adac_layout_destroy": (None, [_vp]),
    "adac_layout_nseg": (_u64, [_vp]),
    "adac_layout_ntiles": (_u64, [_vp]),
    "adac_layout_total_values": (_u64, [_vp]),
    "adac_layout_value_span": (_u64, [_vp]),
    "adac_layout_max_arena_words": (_u64, [_vp]),
    "adac_layout_set_descs": (_int, [_vp, _vp]),
    "adac_layout_get_descs": (_int, [_vp, _vp]),
    "adac_layout_get_minmax": (_int, [_vp, _vp]),
    "adac_layout_device_descs": (_vp, [_vp]),
    "adac_analyze": (_int, [_vp, _vp, _vp, _int]),
    "adac_zonemap": (_int, [_vp, _vp, _vp, _vp]),
    "adac_plan": (_int, [_vp, _int, _int]),
    "adac_pack": (_int, [_vp, _vp, _vp, _vp]),
    "adac_encode": (_int, [_vp, _vp, _vp, _int, _int, _vp]),
    "adac_analyze_packed": (_int, [_vp, _vp, _vp, _int, _vp]),
    "adac_repack": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "adac_reencode": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp]),
    "adac_unpack": (_int, [_vp, _vp, _vp]),
    "adac_unpack_range": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _u64]),
    "adac_fetch_rows": (_int, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "adac_unpack_jobs": (_int, [_vp, _int, _vp, _u64, _vp, _vp]),
    "adac_event_record": (_int, [_vp, _P(_vp)]),
    "adac_event_wait": (_int, [_vp]),
    "adac_event_done": (_int, [_vp]),
    "adac_event_destroy": (None, [_vp]),
    "adac_scan_sum": (_int, [_vp, _vp, _vp]),
    "adac_scan_group_sum": (_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "adac_scan_count_eq": (_int, [_vp, _vp, _u64, _vp]),
    "adac_scan_count_between": (_int, [_vp, _vp, _u64, _u64, _vp]),
    "adac_scan_sum_valid": (_int, [_vp, _vp, _vp, _vp]),
    "adac_scan_select_between": (_int, [_vp, _vp, _vp, _u64, _u64, _vp, _vp]),
    "adac_unpack_selected": (_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "adac_scan_count_between_valid": (_int, [_vp, _vp, _vp, _u64, _u64, _vp]),
    "adac_bp_layout_create": (_int, [_vp, _int, _vp, _vp, _vp, _u64, _P(_vp)]),
    "adac_bp_layout_destroy": (None, [_vp]),
    "adac_bp_layout_ngroups": (_u64, [_vp]),
    "adac_bp_layout_total_values": (_u64, [_vp]),
    "adac_bp_bind": (_int, [_vp, _vp]),
    "adac_bp_unpack": (_int, [_vp, _vp, _vp]),
    "adac_bp_unpack_range": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _u64]),
    "adac_bp_fetch_rows": (_int, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "adac_bp_plan_create": (_int, [_vp, _int, _vp, _vp, _u64, _int, _P(_vp)]),
    "adac_bp_plan_destroy": (None, [_vp]),
    "adac_bp_plan_encodable": (_int, [_vp]),
    "adac_bp_plan_nseg": (_u64, [_vp]),
    "adac_bp_plan_groups_by_mode": (_u64, [_vp, _int]),
    "adac_bp_plan_segment": (_int, [_vp, _u64, _P(_u64), _P(_u64), _P(_u64)]),
    "adac_bp_write": (_int, [_vp, _vp, _vp, _vp, _u64]),
}


def lib():
    """Load libadacodec.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdacError(5, "load %s (HIP extension not built: run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(st, where):
    if st != 0:
        raise AdacError(st, where)


def _dptr(x):
    """Device pointer of a torch tensor / DeviceBuffer / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ptr"):
        return x.ptr
    raise TypeError("not a device buffer: %r" % (x,))


# host-only helpers -------------------------------------------------------------------------------

def stored_min(mn, mx, w):
    """The min a PACKED descriptor stores (differs from mn only for the all-ones sentinel collision)."""
    return lib().adac_stored_min(mn & NO_MIN, mx & NO_MIN, w)


def width(mn, mx, rule=RULE_APPEND, pad_to_byte=False):
    return lib().adac_width(mn & NO_MIN, mx & NO_MIN, rule, int(pad_to_byte))


def size_in_bytes(count, w):
    return lib().adac_size_in_bytes(count, w)


def packed_words(count, w):
    return lib().adac_packed_words(count, w)


def arena_words(count, w):
    return lib().adac_arena_words(count, w)


def block_write(desc, dtype, words):
    """Persistent block image (bytes) of one packed segment: sdsl::int_vector<0> serialisation + 16-byte trailer."""
    d = np.zeros(1, dtype=SEGMENT_DESC_DTYPE)
    d[0] = desc
    words = np.ascontiguousarray(words, dtype=np.uint64)
    n = lib().adac_block_bytes(int(d["count"][0]), int(d["width"][0]))
    out = np.zeros(n, dtype=np.uint8)
    got = lib().adac_block_write(d.ctypes.data, physical_type(dtype), words.ctypes.data, out.ctypes.data, n)
    if got != n:
        raise AdacError(1, "adac_block_write")
    return out.tobytes()


def block_read(block):
    """-> (desc record, numpy dtype, packed words)"""
    buf = np.frombuffer(block, dtype=np.uint8)
    d = np.zeros(1, dtype=SEGMENT_DESC_DTYPE)
    t = _int()
    words = np.zeros(max(1, len(buf) // 8), dtype=np.uint64)
    _check(lib().adac_block_read(buf.ctypes.data, len(buf), d.ctypes.data, C.byref(t), words.ctypes.data, len(words)),
           "adac_block_read")
    nw = lib().adac_packed_words(int(d["count"][0]), int(d["width"][0]))
    return d[0], numpy_dtype(t.value), words[:nw]


def block_peek(block):
    """-> (desc record, numpy dtype) from the header and trailer of a block image (exact or 8-byte padded length)."""
    buf = np.frombuffer(block, dtype=np.uint8)
    d = np.zeros(1, dtype=SEGMENT_DESC_DTYPE)
    t = _int()
    _check(lib().adac_block_peek(buf.ctypes.data, len(buf), d.ctypes.data, C.byref(t)), "adac_block_peek")
    return d[0], numpy_dtype(t.value)


def block_stride(count, w):
    return lib().adac_block_stride(count, w)


def blocks_write(ctx, dtype, descs, block_offs, d_words, d_blocks):
    """Block images of many segments built in HBM (adac_blocks_write); synchronous."""
    descs = np.ascontiguousarray(descs, dtype=SEGMENT_DESC_DTYPE)
    offs = np.ascontiguousarray(block_offs, dtype=np.uint64)
    _check(lib().adac_blocks_write(ctx._h, physical_type(dtype), descs.ctypes.data, offs.ctypes.data, len(descs),
                                   _dptr(d_words), _dptr(d_blocks)), "adac_blocks_write")


def blocks_read(ctx, dtype, descs, block_offs, d_blocks, d_words):
    """Packed words of many block images moved into the arena at descs[i].word_off (adac_blocks_read); synchronous."""
    descs = np.ascontiguousarray(descs, dtype=SEGMENT_DESC_DTYPE)
    offs = np.ascontiguousarray(block_offs, dtype=np.uint64)
    _check(lib().adac_blocks_read(ctx._h, physical_type(dtype), descs.ctypes.data, offs.ctypes.data, len(descs),
                                  _dptr(d_blocks), _dptr(d_words)), "adac_blocks_read")


def set_tuning(name, value):
    if lib().adac_set_tuning(name.encode(), int(value)) != 0:
        raise ValueError("unknown tuning knob %r" % name)


def tile_values(dtype):
    return lib().adac_tile_values(physical_type(dtype))


class DeviceBuffer:
    """hipMalloc'd buffer owned through the C ABI (for hosts that have no torch)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = nbytes
        p = _vp()
        _check(lib().adac_dev_alloc(ctx._h, nbytes, C.byref(p)), "adac_dev_alloc")
        self.ptr = p.value
        ctx._buffers.add(self)   # freed with the context if still alive then (raw device memory is not ref-counted)

    def free(self):
        if self.ptr:
            lib().adac_dev_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: no cover
            pass

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(lib().adac_memcpy_h2d(self.ctx._h, self.ptr, arr.ctypes.data, arr.nbytes), "adac_memcpy_h2d")
        return self

    def download(self, dtype, count, byte_offset=0):
        out = np.empty(count, dtype=dtype)
        assert byte_offset + out.nbytes <= self.nbytes
        _check(lib().adac_memcpy_d2h(self.ctx._h, out.ctypes.data, self.ptr + byte_offset, out.nbytes),
               "adac_memcpy_d2h")
        return out

    def zero(self):
        _check(lib().adac_dev_memset(self.ctx._h, self.ptr, 0, self.nbytes), "adac_dev_memset")
        return self


class Context:
    def __init__(self, device=0, stream=None):
        h = _vp()
        _check(lib().adac_ctx_create(device, stream, C.byref(h)), "adac_ctx_create")
        self._h = h.value
        self.device = device
        self._buffers = weakref.WeakSet()

    def close(self):
        """Layouts, plans and graphs made on the context keep it alive on the C side (reference counts); raw device
        buffers do not, so the ones still alive are freed here."""
        if getattr(self, "_h", None):
            for b in list(self._buffers):
                b.free()
            lib().adac_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def sync(self):
        _check(lib().adac_ctx_sync(self._h), "adac_ctx_sync")

    @property
    def stream(self):
        return lib().adac_ctx_stream(self._h)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 16)).upload(arr)

    def timer_start(self):
        _check(lib().adac_timer_start(self._h), "adac_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().adac_timer_stop(self._h, C.byref(ms)), "adac_timer_stop")
        return ms.value


def unpack_jobs(ctx, dtype, jobs, d_words, d_out):
    """Layout-free decode of a list of (segment, row range) jobs (UNPACK_JOB_DTYPE records) in one launch per 48."""
    jobs = np.ascontiguousarray(jobs, dtype=UNPACK_JOB_DTYPE)
    _check(lib().adac_unpack_jobs(ctx._h, physical_type(dtype), jobs.ctypes.data, len(jobs), _dptr(d_words),
                                  _dptr(d_out)), "adac_unpack_jobs")


def jobs_from_descs(descs, ranges, out_offs):
    """descs: SEGMENT_DESC_DTYPE records; ranges: (start, count) per job; out_offs: element offsets."""
    jobs = np.zeros(len(descs), dtype=UNPACK_JOB_DTYPE)
    for i, (d, (st, c), o) in enumerate(zip(descs, ranges, out_offs)):
        jobs[i] = (d["word_off"], d["min"], o, st, c, d["width"], d["flags"], 0, 0)
    return jobs


class Event:
    """A marker in the context's stream (adac_event): wait() blocks until everything enqueued before it is done."""

    def __init__(self, ctx):
        h = _vp()
        _check(lib().adac_event_record(ctx._h, C.byref(h)), "adac_event_record")
        self._h = h.value

    def wait(self):
        _check(lib().adac_event_wait(self._h), "adac_event_wait")

    def done(self):
        return bool(lib().adac_event_done(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().adac_event_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


class Layout:
    """A batch of column segments of one type on one device (adac_layout)."""

    def __init__(self, ctx, dtype, counts, val_offs=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.ptype = physical_type(dtype)
        self.counts = np.ascontiguousarray(counts, dtype=np.uint32)
        offs = None if val_offs is None else np.ascontiguousarray(val_offs, dtype=np.uint64)
        h = _vp()
        _check(lib().adac_layout_create(ctx._h, self.ptype, self.counts.ctypes.data,
                                        None if offs is None else offs.ctypes.data, len(self.counts), C.byref(h)),
               "adac_layout_create")
        self._h = h.value

    def close(self):
        if getattr(self, "_h", None):
            lib().adac_layout_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    nseg = property(lambda s: lib().adac_layout_nseg(s._h))
    ntiles = property(lambda s: lib().adac_layout_ntiles(s._h))
    total_values = property(lambda s: lib().adac_layout_total_values(s._h))
    value_span = property(lambda s: lib().adac_layout_value_span(s._h))
    max_arena_words = property(lambda s: lib().adac_layout_max_arena_words(s._h))

    def set_descs(self, descs):
        descs = np.ascontiguousarray(descs, dtype=SEGMENT_DESC_DTYPE)
        _check(lib().adac_layout_set_descs(self._h, descs.ctypes.data), "adac_layout_set_descs")

    def get_descs(self):
        descs = np.zeros(self.nseg, dtype=SEGMENT_DESC_DTYPE)
        _check(lib().adac_layout_get_descs(self._h, descs.ctypes.data), "adac_layout_get_descs")
        return descs

    def get_minmax(self):
        mm = np.zeros((self.nseg, 2), dtype=np.uint64)
        _check(lib().adac_layout_get_minmax(self._h, mm.ctypes.data), "adac_layout_get_minmax")
        return mm

    def analyze(self, d_vals, d_validity=None, rule=RULE_APPEND):
        _check(lib().adac_analyze(self._h, _dptr(d_vals), _dptr(d_validity), rule), "adac_analyze")

    def zonemap(self, d_vals, d_validity=None):
        """Per-segment typed (min, max) over the valid rows, as arrays of the column dtype."""
        zm = np.zeros((self.nseg, 2), dtype=np.uint64)
        _check(lib().adac_zonemap(self._h, _dptr(d_vals), _dptr(d_validity), zm.ctypes.data), "adac_zonemap")
        udt = np.dtype("u%d" % self.dtype.itemsize)
        return zm.astype(udt).view(self.dtype)

    def plan(self, rule=RULE_APPEND, pad_to_byte=False):
        _check(lib().adac_plan(self._h, rule, int(pad_to_byte)), "adac_plan")

    def pack(self, d_vals, d_words, d_validity=None):
        _check(lib().adac_pack(self._h, _dptr(d_vals), _dptr(d_validity), _dptr(d_words)), "adac_pack")

    def encode(self, d_vals, d_words, d_validity=None, rule=RULE_APPEND, pad_to_byte=False):
        _check(lib().adac_encode(self._h, _dptr(d_vals), _dptr(d_validity), rule, int(pad_to_byte), _dptr(d_words)),
               "adac_encode")

    def analyze_packed(self, d_words, dst, d_validity=None, rule=RULE_APPEND):
        """min/max of this layout's decoded values, left in `dst` (a layout over the same segments) for dst.plan()."""
        _check(lib().adac_analyze_packed(self._h, _dptr(d_words), _dptr(d_validity), rule, dst._h), "adac_analyze_packed")

    def repack(self, d_words, dst, d_dst_words, d_validity=None):
        _check(lib().adac_repack(self._h, _dptr(d_words), _dptr(d_validity), dst._h, _dptr(d_dst_words)), "adac_repack")

    def reencode(self, d_words, dst, d_dst_words, d_validity=None, rule=RULE_APPEND, pad_to_byte=False):
        """packed -> packed: dst gets the widths adac_encode would choose for the decoded values."""
        _check(lib().adac_reencode(self._h, _dptr(d_words), _dptr(d_validity), rule, int(pad_to_byte), dst._h,
                                   _dptr(d_dst_words)), "adac_reencode")

    def unpack(self, d_words, d_out):
        _check(lib().adac_unpack(self._h, _dptr(d_words), _dptr(d_out)), "adac_unpack")

    def unpack_range(self, d_words, seg, start, count, d_out, out_off=0):
        _check(lib().adac_unpack_range(self._h, _dptr(d_words), seg, start, count, _dptr(d_out), out_off),
               "adac_unpack_range")

    def fetch_rows(self, d_words, d_segs, d_rows, n, d_out):
        _check(lib().adac_fetch_rows(self._h, _dptr(d_words), _dptr(d_segs), _dptr(d_rows), n, _dptr(d_out)),
               "adac_fetch_rows")

    def scan_sum(self, d_words, d_sums, d_validity=None):
        _check(lib().adac_scan_sum_valid(self._h, _dptr(d_words), _dptr(d_validity), _dptr(d_sums)), "adac_scan_sum")

    def scan_group_sum(self, d_words, keys, d_key_words, ngroups, d_sums, d_counts):
        """SUM(self) and COUNT(*) GROUP BY `keys` (a Layout over the same rows); ngroups + 1 results each."""
        _check(lib().adac_scan_group_sum(self._h, _dptr(d_words), keys._h, _dptr(d_key_words), int(ngroups),
                                         _dptr(d_sums), _dptr(d_counts)), "adac_scan_group_sum")

    def scan_count_between(self, d_words, lo, hi, d_counts, d_validity=None):
        """lo / hi: bit patterns of the column type (use int(np.array([v], dtype).view(unsigned)[0]) for signed)."""
        _check(lib().adac_scan_count_between_valid(self._h, _dptr(d_words), _dptr(d_validity), lo & NO_MIN, hi & NO_MIN,
                                                   _dptr(d_counts)), "adac_scan_count_between")

    def scan_select_between(self, d_words, lo, hi, d_bitmap, d_counts, d_validity=None):
        """Selection bitmap over the element index space (ceil(value_span / 64) words) + per-segment hit counts."""
        _check(lib().adac_scan_select_between(self._h, _dptr(d_words), _dptr(d_validity), lo & NO_MIN, hi & NO_MIN,
                                              _dptr(d_bitmap), _dptr(d_counts)), "adac_scan_select_between")

    def unpack_selected(self, d_words, d_bitmap, d_out, d_out_ids=None, want_total=True):
        """Values (and optionally element indices) of the rows whose bitmap bit is set, dense, in row order.
        Returns the number of rows written (None when want_total is False: the call then stays asynchronous)."""
        total = _u64()
        _check(lib().adac_unpack_selected(self._h, _dptr(d_words), _dptr(d_bitmap), _dptr(d_out), _dptr(d_out_ids),
                                          C.byref(total) if want_total else None), "adac_unpack_selected")
        return total.value if want_total else None

    def scan_count_eq(self, d_words, key, d_counts):
        _check(lib().adac_scan_count_eq(self._h, _dptr(d_words), key & NO_MIN, _dptr(d_counts)), "adac_scan_count_eq")


class BitpackingLayout:
    """Segments of DuckDB's on-disk BITPACKING codec resident in one device buffer (adac_bp_layout)."""

    def __init__(self, ctx, dtype, block_offs, counts, out_offs=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.block_offs = np.ascontiguousarray(block_offs, dtype=np.uint64)
        self.counts = np.ascontiguousarray(counts, dtype=np.uint32)
        oo = None if out_offs is None else np.ascontiguousarray(out_offs, dtype=np.uint64)
        h = _vp()
        _check(lib().adac_bp_layout_create(ctx._h, physical_type(dtype), self.block_offs.ctypes.data,
                                           self.counts.ctypes.data, None if oo is None else oo.ctypes.data,
                                           len(self.counts), C.byref(h)), "adac_bp_layout_create")
        self._h = h.value

    def close(self):
        if getattr(self, "_h", None):
            lib().adac_bp_layout_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    ngroups = property(lambda s: lib().adac_bp_layout_ngroups(s._h))
    total_values = property(lambda s: lib().adac_bp_layout_total_values(s._h))

    def bind(self, d_blocks):
        _check(lib().adac_bp_bind(self._h, _dptr(d_blocks)), "adac_bp_bind")

    def unpack(self, d_blocks, d_out):
        _check(lib().adac_bp_unpack(self._h, _dptr(d_blocks), _dptr(d_out)), "adac_bp_unpack")

    def unpack_range(self, d_blocks, seg, start, count, d_out, out_off=0):
        _check(lib().adac_bp_unpack_range(self._h, _dptr(d_blocks), seg, start, count, _dptr(d_out), out_off),
               "adac_bp_unpack_range")

    def fetch_rows(self, d_blocks, d_segs, d_rows, n, d_out):
        _check(lib().adac_bp_fetch_rows(self._h, _dptr(d_blocks), _dptr(d_segs), _dptr(d_rows), n, _dptr(d_out)),
               "adac_bp_fetch_rows")


class BitpackingPlan:
    """Compress side of the BITPACKING codec: device statistics + host mode decisions and block placement."""

    BLOCK_STRIDE = 262144

    def __init__(self, ctx, dtype, d_vals, n, d_validity=None, force_mode=0):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.n = n
        h = _vp()
        _check(lib().adac_bp_plan_create(ctx._h, physical_type(dtype), _dptr(d_vals), _dptr(d_validity), n, force_mode,
                                         C.byref(h)), "adac_bp_plan_create")
        self._h = h.value

    def close(self):
        if getattr(self, "_h", None):
            lib().adac_bp_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    encodable = property(lambda s: bool(lib().adac_bp_plan_encodable(s._h)))
    nseg = property(lambda s: lib().adac_bp_plan_nseg(s._h))

    def groups_by_mode(self):
        return {name: lib().adac_bp_plan_groups_by_mode(self._h, m)
                for m, name in ((1, "constant"), (2, "constant_delta"), (3, "delta_for"), (4, "for"))}

    def segment(self, i):
        a, b, c = _u64(), _u64(), _u64()
        _check(lib().adac_bp_plan_segment(self._h, i, C.byref(a), C.byref(b), C.byref(c)), "adac_bp_plan_segment")
        return a.value, b.value, c.value

    def write(self, d_vals, d_blocks, d_validity=None, block_stride=None):
        _check(lib().adac_bp_write(self._h, _dptr(d_vals), _dptr(d_validity), _dptr(d_blocks),
                                   block_stride or self.BLOCK_STRIDE), "adac_bp_write")


from .layout import appender_segment_counts, aligned_value_offsets  # noqa: E402,F401
