"""integration/succinct_gpu.cpp and integration/bitpacking_gpu.cpp — the DuckDB-side adapters as real code — are
COMPILED TO OBJECTS (g++ -c: templates instantiated, code generated) against the REFERENCE'S OWN HEADERS when the
reference checkout is present (this container; the GPU box has no /root/reference, where the test is skipped).  Every
slot signature is thereby checked against duckdb::CompressionFunction's typedefs and ColumnSegment's members; nothing
of the reference is copied, linked or run.  The succinct adapter is a shim over include/adacodec_host.h: the second
test holds it to that (every call it makes is an adach_* export of libadacodec.so, none a device call of its own),
and tests/test_gpu_adapter_sequence.py drives the same call sequence on the GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "include", "duckdb")),
                    reason="the reference checkout (its headers) exists only in the build container")
@pytest.mark.parametrize("source", ["succinct_gpu.cpp", "bitpacking_gpu.cpp"])
def test_adapter_compiles_against_the_reference_headers(source, tmp_path):
    inc = [os.path.join(REF, "src", "include")] + [os.path.join(REF, "third_party", d) for d in (
        "sdsl/include", "fmt/include", "re2", "utf8proc/include", "concurrentqueue", "fsst", "fastpforlib")]
    obj = str(tmp_path / (source + ".o"))
    cmd = ["g++", "-std=c++11", "-c", "-O1", "-fPIC", "-o", obj, "-Wall", "-Wno-unused-function", "-Wno-unused-parameter"]
    for d in inc:
        cmd += ["-I", d]
    cmd += ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", source)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-4000:]
    # warnings inside the reference's own headers are not ours; the adapter itself must be clean
    ours = [l for l in out.stdout.decode().splitlines() if "integration/" + source in l and "warning" in l]
    assert not ours, "\n".join(ours)
    syms = subprocess.run(["nm", "-C", obj], stdout=subprocess.PIPE).stdout.decode()
    if source.startswith("succinct"):
        assert re.search(r" T duckdb::SuccinctFun::GetFunction\(", syms)
        assert re.search(r" T duckdb::SuccinctFun::TypeIsSupported\(", syms)
        # the object defines the two entry points the engine links against (compression.hpp:56-59) and reaches the codec only through the host mirror's C interface: no adac_* device entry point but the
        # device count, and every adach_* it needs is a declared export
        und = set(re.findall(r" U (adac[h]?_\w+)", syms))
        assert und, "the shim calls nothing?"
        assert {u for u in und if u.startswith("adac_")} <= {"adac_device_count"}, und
        header = open(os.path.join(ROOT, "include", "adacodec_host.h")).read()
        missing = [u for u in und if u.startswith("adach_") and not re.search(r"\b%s\(" % u, header)]
        assert not missing, missing
        for needed in ("adach_segment_init_scan", "adach_segment_scan_with", "adach_segments_compact",
                       "adach_segment_append", "adach_segment_fetch_row"):
            assert needed in und, needed
