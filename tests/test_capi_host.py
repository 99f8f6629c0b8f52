"""CPU-side checks of the C ABI: the library loads, exports every symbol include/adacodec.h declares, its
host-only helpers agree with the oracle, and device entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(adac):
    adac.build()
    return adac.lib()


def test_every_declared_symbol_is_exported(adac, lib):
    hdr = open(os.path.join(ROOT, "include", "adacodec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(adac_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    raw = C.CDLL(adac.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libadacodec.so does not export %s" % name
    assert declared == set(adac.SIGNATURES), declared ^ set(adac.SIGNATURES)
    assert lib.adac_abi_version() == 1
    # the C wrappers of the C++ host mirror (include/adacodec_host.h)
    import importlib
    host = importlib.import_module(adac.__name__ + ".host")
    hdr = open(os.path.join(ROOT, "include", "adacodec_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdeclared = set(re.findall(r"\b(adach_[a-z0-9_]+)\s*\(", hdr))
    for name in sorted(hdeclared):
        assert hasattr(raw, name), "libadacodec.so does not export %s" % name
    assert hdeclared == set(host.HOST_SIGNATURES), hdeclared ^ set(host.HOST_SIGNATURES)


def test_segment_desc_abi(adac):
    assert adac.SEGMENT_DESC_DTYPE.itemsize == 32
    assert [adac.SEGMENT_DESC_DTYPE.fields[f][1] for f in ("word_off", "val_off", "min", "count", "width", "flags")] \
        == [0, 8, 16, 24, 28, 29]


def test_type_table_matches_physical_type_codes(adac, lib):
    # duckdb::PhysicalType: UINT8=2 INT8=3 UINT16=4 INT16=5 UINT32=6 INT32=7 UINT64=8 INT64=9
    sizes = {2: 1, 3: 1, 4: 2, 5: 2, 6: 4, 7: 4, 8: 8, 9: 8}
    for t in range(0, 30):
        assert bool(lib.adac_type_is_supported(t)) == (t in sizes)
        assert lib.adac_type_size(t) == sizes.get(t, 0)
    for dt, t in ((np.uint8, 2), (np.int8, 3), (np.uint16, 4), (np.int16, 5), (np.uint32, 6), (np.int32, 7),
                  (np.uint64, 8), (np.int64, 9)):
        assert adac.physical_type(dt) == t
        assert lib.adac_tile_values(t) * np.dtype(dt).itemsize == 16384
    with pytest.raises(adac.AdacError):
        adac.physical_type(np.float64)


def test_width_and_size_helpers_match_oracle(adac, oracle, lib):
    rng = np.random.default_rng(0)
    cases = [(0, 0), (5, 5), (42, 42), (0, 1), (1, 0), (2 ** 64 - 1, 0), (0, 2 ** 64 - 1), (2 ** 40, 2 ** 40 + 123456789),
             (0xFFFFFFFFFFFFFFF6, 0xFFFFFFFFFFFFFFFB), (5, 0xFFFFFFFFFFFFFFFF)]
    for _ in range(2000):
        a, b = (int(x) for x in rng.integers(0, 2 ** 64, size=2, dtype=np.uint64))
        sh = int(rng.integers(0, 64))
        cases.append((a >> sh, b >> int(rng.integers(0, 64))))
    for mn, mx in cases:
        for pad in (0, 1):
            assert adac.width(mn, mx, adac.RULE_APPEND, pad) == oracle.width_from_succinct(mn, mx, pad) & 0xFF
            assert adac.width(mn, mx, adac.RULE_RECOMPACT, pad) == oracle.width_from_uncompressed(mn, mx, pad) & 0xFF
    for x in [0, 1, 2, 3, 255, 256, 2 ** 63, 2 ** 64 - 1]:
        assert lib.adac_hi(x) == oracle.hi(x)
    for n in [0, 1, 5, 63, 64, 65, 1000, 32767, 65534]:
        for w in [1, 3, 12, 16, 27, 32, 33, 64]:
            assert adac.size_in_bytes(n, w) == oracle.size_in_bytes(n * w)
            assert adac.packed_words(n, w) == (n * w + 63) // 64
            aw = adac.arena_words(n, w)
            assert aw % 16 == 0 and aw >= (n * w + 64) // 64 and aw - (n * w + 64) // 64 < 16
    assert adac.size_in_bytes(1000, 12) == 1513


def test_device_entry_points_fail_loudly_without_a_gpu(adac, lib):
    """No CPU fallback behind the device ABI: on a box without a HIP device context creation reports
    ADAC_ERR_NO_DEVICE (on the GPU box it succeeds and this test only checks the status path)."""
    h = C.c_void_p()
    st = lib.adac_ctx_create(0, None, C.byref(h))
    if st == 0:
        lib.adac_ctx_destroy(h)
        pytest.skip("a GPU is present")
    assert st == 5
    assert b"device" in lib.adac_status_string(st)
    with pytest.raises(adac.AdacError) as e:
        adac.Context(0)
    assert e.value.status == 5
    # NULL handles are rejected, not dereferenced
    assert lib.adac_ctx_sync(None) == 1
    assert lib.adac_unpack(None, None, None) == 1
    assert lib.adac_analyze(None, None, None, 0) == 1


def test_workload_generator(adac):
    import importlib
    wl = importlib.import_module(adac.__name__ + ".workload")
    assert int(wl.mt19937_stream(5489, 10000)[-1]) == 4123659995  # the C++ standard's mt19937 check value
    a = wl.zipf_column(300000, np.uint64, seed=42, threads=1)
    b = wl.zipf_column(300000, np.uint64, seed=42, threads=5)
    assert np.array_equal(a, b)  # independent of the thread count
    assert a.min() >= 1 and a.max() <= 2 ** 32 - 1
    # Zipf(1.0): P(1) = 1/H_n ~ 1/22.8 for n = 2^32-1
    frac1 = float((a == 1).mean())
    assert 0.03 < frac1 < 0.06
    s2 = wl.zipf_column(200000, np.uint32, domain=2 ** 32 - 1, skew=2.0, seed=1)
    assert float((s2 == 1).mean()) > 0.55  # 1/zeta(2) = 0.608
    small = wl.zipf_column(100000, np.uint8, domain=200, skew=0.5, seed=3)
    assert small.max() <= 200 and small.min() >= 1


def test_persistent_block_image(adac, oracle, golden):
    """adac_block_write/read: the sdsl::int_vector<0> serialisation (uint64 bit size, uint8 width, words —
    int_vector.hpp:602-609,1546-1578) of the packed vector + a 16-byte trailer.  Checked on K1..K6: the sdsl part
    is exactly size_in_bytes long and carries the known words."""
    import struct
    for k in golden["sdsl"]:
        dtype = np.dtype(k["dtype"])
        words = np.array([int(w, 16) for w in k["words"]], dtype=np.uint64)
        n = len(k["values"])
        desc = (0, 0, int(k["min"], 16), n, k["width"], adac.SEG_PACKED, 0)
        blob = adac.block_write(desc, dtype, words)
        assert len(blob) == k["size_in_bytes"] + 16
        bit_size, width = struct.unpack_from("<QB", blob, 0)
        assert (bit_size, width) == (k["bit_size"], k["width"])
        assert np.array_equal(np.frombuffer(blob, dtype="<u8", count=len(words), offset=9), words)
        d, dt, w2 = adac.block_read(blob)
        assert dt == dtype and int(d["count"]) == n and int(d["width"]) == k["width"]
        assert int(d["min"]) == int(k["min"], 16) and int(d["flags"]) == adac.SEG_PACKED
        assert np.array_equal(w2, words)
        # the decoded block equals the original values (through the oracle's reader)
        assert oracle.unpack_flat(w2, 0, n, int(d["width"]), int(d["min"]), dtype).tolist() == k["values"]
    # malformed blocks are rejected
    good = adac.block_write((0, 0, 5, 100, 7, 1, 0), np.uint32, np.arange(11, dtype=np.uint64))
    for bad in (good[:-1], good + b"\0", b"\0" * 8 + bytes([0]) + good[9:], good[:8] + bytes([40]) + good[9:]):
        with pytest.raises(adac.AdacError):
            adac.block_read(bad)


def test_null_arguments_are_refused_not_dereferenced(adac):
    """Every entry point called with NULL handles / pointers and zero sizes: status-returning ones answer
    ADAC_ERR_INVALID_ARGUMENT (or NO_DEVICE for adac_ctx_create-like calls), getters answer 0 / NULL, destroy
    functions accept NULL — nothing dereferences a NULL handle (errors are statuses, never crashes: the C++ adapter
    turns them into InternalException, SURVEY.md §8b 'Errors')."""
    import ctypes as C
    L = adac.lib()
    host_only = {"adac_status_string", "adac_last_error", "adac_abi_version", "adac_type_is_supported", "adac_type_size",
                 "adac_hi", "adac_width", "adac_packed_words", "adac_size_in_bytes", "adac_arena_words",
                 "adac_tile_values", "adac_set_tuning", "adac_block_bytes", "adac_stored_min", "adac_block_write",
                 "adac_bp_plan_encodable", "adac_block_stride"}
    assert L.adac_event_done(None) == 0               # a query, not a status: "not finished" for a NULL event
    host_only.add("adac_event_done")
    assert L.adac_device_count() >= 0                 # 0 without a device, never an error code
    host_only.add("adac_device_count")
    for name, (res, args) in adac.SIGNATURES.items():
        if name in host_only:
            continue
        zero = [None if (a is C.c_void_p or a is C.c_char_p or (isinstance(a, type) and issubclass(a, C._Pointer))) else 0
                for a in args]
        r = getattr(L, name)(*zero)
        if res is C.c_int:
            assert r != 0, name                      # an adac_status other than ADAC_OK
        elif res is None:
            assert r is None
        else:
            assert not r or r == -1, (name, r)       # counts are 0, pointers NULL, adac_ctx_device(NULL) is -1
