"""integration/succinct_gpu.cpp and integration/bitpacking_gpu.cpp — the DuckDB-side adapters as real code — are
syntax-checked against the REFERENCE'S
OWN HEADERS when the reference checkout is present (this container; the GPU box has no /root/reference, where the
test is skipped).  Every slot signature is thereby checked against duckdb::CompressionFunction's typedefs and
ColumnSegment's members; nothing of the reference is copied, linked or run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "include", "duckdb")),
                    reason="the reference checkout (its headers) exists only in the build container")
@pytest.mark.parametrize("source", ["succinct_gpu.cpp", "bitpacking_gpu.cpp"])
def test_adapter_compiles_against_the_reference_headers(source):
    inc = [os.path.join(REF, "src", "include")] + [os.path.join(REF, "third_party", d) for d in (
        "sdsl/include", "fmt/include", "re2", "utf8proc/include", "concurrentqueue", "fsst", "fastpforlib")]
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-unused-function", "-Wno-unused-parameter"]
    for d in inc:
        cmd += ["-I", d]
    cmd += ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", source)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-4000:]
    # warnings inside the reference's own headers are not ours; the adapter itself must be clean
    ours = [l for l in out.stdout.decode().splitlines() if "integration/" + source in l and "warning" in l]
    assert not ours, "\n".join(ours)
