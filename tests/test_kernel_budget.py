"""Kernel resource budget (no GPU needed): the compiler's own -Rpass-analysis=kernel-resource-usage remarks of the
build that ships (duckdb-adaptive-compression_amd/build/adac_kernels.resources.txt, written by the package Makefile)
are held against profiles/kernel_budget.json.

Why: in round 2 a width-2/3 register walk was inlined into the common fused-scan kernel; k_scan_agg<u64, sum> went from
42 to 74 VGPRs, occupancy 8 -> 6 waves per SIMD, and the fused SUM at w = 8 .. 32 lost 10 % without any test noticing.
A change that costs a budgeted kernel occupancy, or makes it spill, now fails here; a deliberate change refreshes the
budget with `python tools/kernel_resources.py --write-budget` next to the sweep that justifies it."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402


@pytest.fixture(scope="module")
def table():
    importlib.import_module("duckdb-adaptive-compression_amd").build()  # make: a no-op when the objects are current
    src = os.path.join(ROOT, "duckdb-adaptive-compression_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".inl", ".h")))
    assert os.path.getmtime(kr.REMARKS) >= newest, "resource remarks are older than the kernel sources"
    return kr.parse()


@pytest.fixture(scope="module")
def budget():
    with open(kr.BUDGET) as f:
        return json.load(f)["kernels"]


def test_every_budgeted_kernel_is_in_the_budget_file(table, budget):
    built = set(kr.budgeted(table))
    assert built == set(budget), "kernels without a budget / budgets without a kernel: %s" % sorted(built ^ set(budget))


def test_no_budgeted_kernel_lost_occupancy_or_started_to_spill(table, budget):
    bad = []
    for name, b in budget.items():
        r = table[name]
        for field in ("vgpr_spills", "sgpr_spills", "scratch", "lds", "agprs"):
            if r[field] > b[field]:
                bad.append("%s: %s %d > budget %d" % (name, field, r[field], b[field]))
        if r["occupancy"] < b["occupancy"]:
            bad.append("%s: occupancy %d < budget %d (vgprs %d, budget %d)" % (name, r["occupancy"], b["occupancy"],
                                                                                r["vgprs"], b["vgprs"]))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("utype", ["u64", "u32"])
@pytest.mark.parametrize("op", [0, 1])
def test_common_fused_scan_stays_thin(table, utype, op):
    """The kernel of the north star's 8-32-bit band: SUM / COUNT without a validity mask on 4- and 8-byte columns."""
    r = table["k_scan_agg<%s,%d,false,false>" % (utype, op)]
    assert r["vgprs"] <= 48 and r["vgpr_spills"] == 0 and r["scratch"] == 0 and r["occupancy"] == 8, r


@pytest.mark.parametrize("utype", ["u64", "u32", "u16", "u8"])
def test_decode_kernel_stays_thin(table, utype):
    for k in ("k_unpack<%s,false,false>" % utype, "k_unpack_jobs<%s>" % utype):
        r = table[k]
        assert r["occupancy"] == 8 and r["vgpr_spills"] == 0 and r["scratch"] == 0, (k, r)
