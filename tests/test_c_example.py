"""examples/c_abi_demo.c: the C ABI used from strict C99 (no Python, no HIP headers).  On CPU the test proves the
headers are plain C and that the binary fails loudly without a device (exit 3: no CPU fallback); on the GPU box the
same binary must run the whole encode / scan / filter / project sequence and exit 0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "duckdb-adaptive-compression_amd")


def build(tmp_path, adac):
    adac.build()
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", PKG, "-ladacodec",
                           "-Wl,-rpath," + PKG, "-o", exe])
    return exe


def test_headers_are_plain_c_and_the_demo_links(adac, tmp_path):
    for header in ("adacodec.h", "adacodec_host.h"):
        src = tmp_path / ("inc_%s.c" % header.replace(".", "_"))
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % header)
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I",
                               os.path.join(ROOT, "include"), str(src)])
    exe = build(tmp_path, adac)
    rc = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode
    assert rc in (0, 3)   # 3 = "no HIP device: the codec has no CPU fallback"


@pytest.mark.gpu
def test_c_demo_runs_on_the_gpu(adac, tmp_path):
    exe = build(tmp_path, adac)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode() + out.stdout.decode()
    assert out.stdout.decode().strip().endswith("ok")
