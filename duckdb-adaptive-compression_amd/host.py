"""ctypes binding of include/adacodec_host.h — the C wrappers of the C++ host mirror (csrc/host/):
the reference's ColumnSegment / CompressionFunction / ColumnSegmentCatalog state machine on one GPU pool."""
import ctypes as C

import numpy as np

from . import AdacError, lib as _codec_lib, physical_type, LIB_PATH  # noqa: F401

FN_UNCOMPRESSED, FN_SUCCINCT = 1, 10
_u64, _i64, _u32, _int, _vp = C.c_uint64, C.c_int64, C.c_uint32, C.c_int, C.c_void_p

HOST_SIGNATURES = {
    "adach_last_error": (C.c_char_p, []),
    "adach_type_is_supported": (_int, [_int]),
    "adach_db_create": (_vp, [_int, _int, _int, _int, _u64]),
    "adach_db_destroy": (None, [_vp]),
    "adach_db_create_cached": (_vp, [_int, _int, _int, _int, _u64, _u64]),
    "adach_db_create_pools": (_vp, [C.POINTER(_int), _int, _int, _int, _int, _u64, _u64, _u32, _u32]),
    "adach_db_num_pools": (_u64, [_vp]),
    "adach_db_cache_stats": (None, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "adach_db_prefetch_stats": (None, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "adach_full_scan_mt": (_int, [C.POINTER(_vp), _u64, _u64, _u32, C.POINTER(_u64), C.POINTER(C.c_double),
                                  C.POINTER(_u64)]),
    "adach_checkpoint_column": (_int, [_vp, _int, _int, _u64, _vp, _vp, _u64, C.POINTER(_vp), _u64, C.POINTER(_u64),
                                       C.POINTER(_u64), C.POINTER(_u64), _vp, _u64, C.POINTER(_u64)]),
    "adach_segment_block_bytes": (_u64, [_vp]),
    "adach_segments_persist": (_int, [_vp, C.POINTER(_vp), _u64, _vp, _u64, C.POINTER(_u64)]),
    "adach_segments_load": (_int, [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), _u64, C.POINTER(_vp)]),
    "adach_segment_pool": (_int, [_vp]),
    "adach_segment_persistent": (_int, [_vp]),
    "adach_catalog_background_stats": (None, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.c_char_p, _u64]),
    "adach_db_pool_arena_used_bytes": (_u64, [_vp, _u32]),
    "adach_full_scan": (_int, [C.POINTER(_vp), _u64, _u64, C.POINTER(_u64), C.POINTER(C.c_double), C.POINTER(_u64)]),
    "adach_db_reserve_staging": (_int, [_vp, _u64]),
    "adach_db_data_size": (_i64, [_vp]),
    "adach_db_arena_used_bytes": (_u64, [_vp]),
    "adach_compress_column": (_int, [_vp, _int, _int, _u64, _vp, _vp, _u64, C.POINTER(_vp), _u64, C.POINTER(_u64),
                                     C.POINTER(_u64), C.POINTER(_u64)]),
    "adach_function_slots": (_int, [_vp, _int, _int, C.POINTER(_int)]),
    "adach_segment_create": (_vp, [_vp, _int, _u64, _u64]),
    "adach_segment_set_next": (_int, [_vp, _vp]),
    "adach_segment_destroy": (None, [_vp]),
    "adach_segment_append": (_i64, [_vp, _vp, _vp, _vp, _u64, _u64]),
    "adach_segment_scan": (_int, [_vp, _u64, _u64, _vp, _u64, _int]),
    "adach_segment_fetch_row": (_int, [_vp, _i64, _vp, _u64]),
    "adach_scan_state_create": (_vp, []),
    "adach_scan_state_destroy": (None, [_vp]),
    "adach_segment_init_scan": (_int, [_vp, _vp]),
    "adach_segment_scan_with": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _int]),
    "adach_segments_compact": (_int, [_vp, C.POINTER(_vp), _u64]),
    "adach_segment_compact": (_int, [_vp]),
    "adach_segment_uncompact": (_int, [_vp]),
    "adach_segment_count": (_u64, [_vp]),
    "adach_segment_min": (_u64, [_vp]),
    "adach_segment_max": (_u64, [_vp]),
    "adach_segment_width": (_u32, [_vp]),
    "adach_segment_compacted": (_int, [_vp]),
    "adach_segment_function": (_int, [_vp]),
    "adach_segment_data_size": (_u64, [_vp]),
    "adach_catalog_compact_all": (_int, [_vp]),
    "adach_catalog_total_data_size": (_u64, [_vp]),
    "adach_catalog_num_segments": (_u64, [_vp]),
    "adach_catalog_policy_step": (_int, [_vp, C.c_double]),
    "adach_catalog_enable_background": (_int, [_vp, C.c_uint]),
    "adach_catalog_disable_background": (_int, [_vp]),
}

_ready = False


def hlib():
    global _ready
    L = _codec_lib()
    if not _ready:
        for name, (res, args) in HOST_SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _ready = True
    return L


class HostError(RuntimeError):
    pass


def _ok(rc, where):
    if rc != 0:
        raise HostError("%s: %s" % (where, hlib().adach_last_error().decode()))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ScanState:
    """duckdb::ColumnScanState of one scanning thread (adach_scan_state)."""

    def __init__(self):
        self._h = hlib().adach_scan_state_create()

    def close(self):
        if self._h:
            hlib().adach_scan_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


class Database:
    """DBConfig flags + one segment pool per GPU + the ColumnSegmentCatalog.  device: one HIP device, or a list of
    them = one pool each (the same device may be listed several times); arena_bytes / decoded_cache_bytes per pool."""

    def __init__(self, device=0, succinct_enabled=True, adaptive=False, padded=False, arena_bytes=1 << 30,
                 decoded_cache_bytes=0, scan_lanes=0, prefetch_segments=0):
        devices = [device] if isinstance(device, int) else list(device)
        arr = (_int * len(devices))(*devices)
        self._h = hlib().adach_db_create_pools(arr, len(devices), int(succinct_enabled), int(adaptive), int(padded),
                                               arena_bytes, decoded_cache_bytes, scan_lanes, prefetch_segments)
        if not self._h:
            raise HostError("adach_db_create: %s" % hlib().adach_last_error().decode())
        self.segments = []

    def close(self):
        if self._h:
            hlib().adach_catalog_disable_background(self._h)
        for s in self.segments:
            s.close()
        self.segments = []
        if self._h:
            hlib().adach_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def create_segment(self, dtype, start=0, segment_size=262136, chain=True):
        """chain: make the new segment the `next` of the previously created one when it continues the same column
        (same type, start right after it) — the hint the sequential-scan prefetch follows."""
        s = Segment(self, dtype, start, segment_size)
        prev = self.segments[-1] if self.segments else None
        if chain and prev is not None and prev._h and prev.dtype == s.dtype and start > prev.start:
            _ok(hlib().adach_segment_set_next(prev._h, s._h), "SetNext")
        self.segments.append(s)
        return s

    SLOT_NAMES = ("init_analyze", "analyze", "final_analyze", "init_compression", "compress", "compress_finalize",
                  "init_scan", "scan_vector", "scan_partial", "fetch_row", "skip", "init_segment", "init_append",
                  "append", "finalize_append", "revert_append")

    def function_slots(self, dtype, compression_type=10):
        """Which slots of duckdb::CompressionFunction the table fills (10 = SUCCINCT, 1 = UNCOMPRESSED)."""
        present = (_int * 16)()
        _ok(hlib().adach_function_slots(self._h, compression_type, physical_type(dtype), present), "function_slots")
        return {n: bool(p) for n, p in zip(self.SLOT_NAMES, present)}

    def compress_column(self, values, validity=None, row_group_start=0, compression_type=10):
        """The checkpoint-side slots over a whole column (analyze every vector, score, compress every vector).
        Returns (segments, bytes reported per segment, analyze score)."""
        values = np.ascontiguousarray(values)
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        cap = len(values) // 2048 + 8
        arr = (_vp * cap)()
        sizes = (_u64 * cap)()
        nseg, score = _u64(), _u64()
        _ok(hlib().adach_compress_column(self._h, compression_type, physical_type(values.dtype), row_group_start,
                                         _p(values), _p(validity), len(values), arr, cap, C.byref(nseg), sizes,
                                         C.byref(score)), "compress_column")
        segs, row = [], row_group_start
        for i in range(nseg.value):
            s = Segment.__new__(Segment)
            s.db, s.dtype, s.start, s._h = self, values.dtype, row, arr[i]
            row += s.count
            segs.append(s)
            self.segments.append(s)
        return segs, [sizes[i] for i in range(nseg.value)], score.value

    def checkpoint_column(self, values, validity=None, row_group_start=0, compression_type=10):
        """compress_column + the block images ConvertToPersistent yields for the flushed segments.
        Returns (segments, sizes, score, [image bytes per segment])."""
        values = np.ascontiguousarray(values)
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        cap = len(values) // 2048 + 8
        arr = (_vp * cap)()
        sizes = (_u64 * cap)()
        offs = (_u64 * (cap + 1))()
        nseg, score = _u64(), _u64()
        blob = np.zeros(values.nbytes + cap * 64 + 4096, dtype=np.uint8)
        _ok(hlib().adach_checkpoint_column(self._h, compression_type, physical_type(values.dtype), row_group_start,
                                           _p(values), _p(validity), len(values), arr, cap, C.byref(nseg), sizes,
                                           C.byref(score), _p(blob), blob.nbytes, offs), "checkpoint_column")
        segs, row, images = [], row_group_start, []
        for i in range(nseg.value):
            s = Segment.__new__(Segment)
            s.db, s.dtype, s.start, s._h = self, values.dtype, row, arr[i]
            row += s.count
            segs.append(s)
            self.segments.append(s)
            if compression_type == 10:
                n = hlib().adach_segment_block_bytes(s._h)
                images.append(blob[offs[i]:offs[i] + n].tobytes())
        return segs, [sizes[i] for i in range(nseg.value)], score.value, images

    def persist(self, segments):
        """ConvertToPersistent for a list of segments -> [image bytes]."""
        arr = (_vp * len(segments))(*[s._h for s in segments])
        cap = sum(hlib().adach_segment_block_bytes(s._h) + 8 for s in segments) + 64
        blob = np.zeros(cap, dtype=np.uint8)
        offs = (_u64 * (len(segments) + 1))()
        _ok(hlib().adach_segments_persist(self._h, arr, len(segments), _p(blob), cap, offs), "persist")
        return [blob[offs[i]:offs[i] + hlib().adach_segment_block_bytes(s._h)].tobytes()
                for i, s in enumerate(segments)]

    def load(self, images, dtype, starts):
        """Segments from block images (CreatePersistentSegments)."""
        lens = np.array([len(b) for b in images], dtype=np.uint64)
        padded = [(int(n) + 7) & ~7 for n in lens]
        offs = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.uint64) if len(images) else np.zeros(0, np.uint64)
        blob = np.zeros(int(sum(padded)) + 8, dtype=np.uint8)
        for o, b in zip(offs, images):
            blob[int(o):int(o) + len(b)] = np.frombuffer(b, dtype=np.uint8)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        arr = (_vp * len(images))()
        _ok(hlib().adach_segments_load(self._h, _p(blob), offs.ctypes.data_as(C.POINTER(_u64)),
                                       lens.ctypes.data_as(C.POINTER(_u64)), st.ctypes.data_as(C.POINTER(_u64)),
                                       len(images), arr), "load")
        segs = []
        for i in range(len(images)):
            s = Segment.__new__(Segment)
            s.db, s.dtype, s.start, s._h = self, np.dtype(dtype), int(st[i]), arr[i]
            segs.append(s)
            self.segments.append(s)
        return segs

    def reserve_staging(self, nbytes):
        _ok(hlib().adach_db_reserve_staging(self._h, nbytes), "reserve_staging")

    def cache_stats(self):
        h, m, b = _u64(), _u64(), _u64()
        hlib().adach_db_cache_stats(self._h, C.byref(h), C.byref(m), C.byref(b))
        return {"hits": h.value, "misses": m.value, "bytes": b.value}

    def prefetch_stats(self):
        a, b, c = _u64(), _u64(), _u64()
        hlib().adach_db_prefetch_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"batches": a.value, "prefetched": b.value, "arena_exhausted": c.value}

    def background_stats(self):
        r, e = _u64(), _u64()
        buf = C.create_string_buffer(512)
        hlib().adach_catalog_background_stats(self._h, C.byref(r), C.byref(e), buf, 512)
        return {"rounds": r.value, "errors": e.value, "last_error": buf.value.decode()}

    def full_scan(self, segments=None, vector_size=2048, threads=1):
        """Scan every row of `segments` the way ColumnData::ScanVector does (InitializeScan per segment,
        vector_size-row ColumnSegment::Scan calls; C++ loop, timed on the host), with `threads` consumers each taking
        a contiguous run of the segments.  Returns (checksum, seconds, rows)."""
        segments = self.segments if segments is None else segments
        arr = (_vp * len(segments))(*[s._h for s in segments])
        cs, sec, rows = _u64(), C.c_double(), _u64()
        _ok(hlib().adach_full_scan_mt(arr, len(segments), vector_size, threads, C.byref(cs), C.byref(sec),
                                      C.byref(rows)), "full_scan")
        return cs.value, sec.value, rows.value

    def compact_segments(self, segments):
        """ColumnSegment::CompactMany over a list: one upload / analyze / pack per (pool, type, rule)."""
        arr = (_vp * len(segments))(*[s._h for s in segments])
        _ok(hlib().adach_segments_compact(self._h, arr, len(segments)), "CompactMany")

    def compact_all(self):
        _ok(hlib().adach_catalog_compact_all(self._h), "CompactAllSegments")

    def policy_step(self, rate=0.90):
        _ok(hlib().adach_catalog_policy_step(self._h, rate), "CompressLowestKSegments")

    def enable_background(self, period_ms):
        _ok(hlib().adach_catalog_enable_background(self._h, period_ms), "EnableBackgroundThreadCompaction")

    def disable_background(self):
        _ok(hlib().adach_catalog_disable_background(self._h), "DisableBackgroundThreadCompaction")

    total_data_size = property(lambda s: hlib().adach_catalog_total_data_size(s._h))
    num_segments = property(lambda s: hlib().adach_catalog_num_segments(s._h))
    data_size = property(lambda s: hlib().adach_db_data_size(s._h))
    arena_used_bytes = property(lambda s: hlib().adach_db_arena_used_bytes(s._h))
    num_pools = property(lambda s: hlib().adach_db_num_pools(s._h))

    def pool_arena_used_bytes(self, pool):
        return hlib().adach_db_pool_arena_used_bytes(self._h, pool)


class Segment:
    def __init__(self, db, dtype, start, segment_size):
        self.db = db
        self.dtype = np.dtype(dtype)
        self.start = start
        self._h = hlib().adach_segment_create(db._h, physical_type(dtype), start, segment_size)
        if not self._h:
            raise HostError("adach_segment_create: %s" % hlib().adach_last_error().decode())

    def close(self):
        if self._h:
            hlib().adach_segment_destroy(self._h)
            self._h = None

    def append(self, vals, validity=None, sel=None, offset=0, count=None):
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        if count is None:
            count = (len(vals) if sel is None else len(sel)) - offset
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        if sel is not None:
            sel = np.ascontiguousarray(sel, dtype=np.uint32)
        n = hlib().adach_segment_append(self._h, _p(vals), _p(validity), _p(sel), offset, count)
        if n < 0:
            raise HostError("Append: %s" % hlib().adach_last_error().decode())
        return n

    def scan(self, row, count, result=None, result_offset=0, entire_vector=None):
        """row: segment-relative row. Returns the rows scanned."""
        if entire_vector is None:
            entire_vector = result_offset == 0
        if result is None:
            result = np.empty(result_offset + count, dtype=self.dtype)
        _ok(hlib().adach_segment_scan(self._h, self.start + row, count, _p(result), result_offset, int(entire_vector)),
            "Scan")
        return result[result_offset:result_offset + count]

    def init_scan(self, state):
        """ColumnSegment::InitializeScan into a ScanState that lives across scan calls (and pins the decoded block)."""
        _ok(hlib().adach_segment_init_scan(self._h, state._h), "InitializeScan")

    def scan_with(self, state, row, count, result, result_offset=0, entire_vector=None):
        if entire_vector is None:
            entire_vector = result_offset == 0
        _ok(hlib().adach_segment_scan_with(self._h, state._h, self.start + row, count, _p(result), result_offset,
                                           int(entire_vector)), "Scan")
        return result[result_offset:result_offset + count]

    def fetch_row(self, row):
        out = np.empty(1, dtype=self.dtype)
        _ok(hlib().adach_segment_fetch_row(self._h, self.start + row, _p(out), 0), "FetchRow")
        return out[0]

    def compact(self):
        _ok(hlib().adach_segment_compact(self._h), "Compact")

    def uncompact(self):
        _ok(hlib().adach_segment_uncompact(self._h), "Uncompact")

    count = property(lambda s: hlib().adach_segment_count(s._h))
    min_factor = property(lambda s: hlib().adach_segment_min(s._h))
    max_factor = property(lambda s: hlib().adach_segment_max(s._h))
    width = property(lambda s: hlib().adach_segment_width(s._h))
    compacted = property(lambda s: bool(hlib().adach_segment_compacted(s._h)))
    function = property(lambda s: hlib().adach_segment_function(s._h))
    data_size = property(lambda s: hlib().adach_segment_data_size(s._h))
    pool = property(lambda s: hlib().adach_segment_pool(s._h))
    persistent = property(lambda s: bool(hlib().adach_segment_persistent(s._h)))
