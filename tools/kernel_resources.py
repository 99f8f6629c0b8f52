#!/usr/bin/env python3
"""Per-kernel resource usage of libadacodec's device code, from the compiler's own remarks.

The package Makefile compiles csrc/adac_kernels.hip with -Rpass-analysis=kernel-resource-usage and keeps the remarks
in build/adac_kernels.resources.txt.  This module parses them into {demangled kernel name: {vgprs, agprs, sgprs,
vgpr_spills, sgpr_spills, scratch, occupancy, lds}} — no GPU needed.

  python tools/kernel_resources.py                  # table of every kernel
  python tools/kernel_resources.py --write-budget   # refresh profiles/kernel_budget.json for the budgeted kernels
                                                    # (review the diff: the budget is a committed decision)

tests/test_kernel_budget.py holds the build against profiles/kernel_budget.json.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REMARKS = os.path.join(ROOT, "duckdb-adaptive-compression_amd", "build", "adac_kernels.resources.txt")
BUDGET = os.path.join(ROOT, "profiles", "kernel_budget.json")
CXXFILT = "c++filt"  # binutils (the ROCm image carries no llvm-cxxfilt)

FIELDS = {
    "TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
    "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
    "LDS Size [bytes/block]": "lds",
}

# the kernels whose registers / occupancy decide a measured throughput; matched as prefixes of the
# short name below
BUDGETED = ("k_unpack<", "k_unpack_jobs<", "k_scan_agg<", "k_encode_1p<", "k_repack_g<", "k_analyze_packed_g<",
            "k_group_sum", "k_gather<", "k_gather_c<", "k_pack<", "k_analyze<", "k_bp_unpack<")


def short_name(demangled):
    """adac::(anonymous namespace)::k_scan_agg<unsigned long, 0, false, false>(args...) -> k_scan_agg<u64,0,false,false>"""
    s = demangled
    s = re.sub(r"^void ", "", s)
    s = s.replace("adac::(anonymous namespace)::", "").replace("adac::", "")
    # cut the argument list: the last top-level '(' of the function itself
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            s = s[:i]
            break
    for a, b in (("unsigned long", "u64"), ("unsigned int", "u32"), ("unsigned short", "u16"),
                 ("unsigned char", "u8"), ("(bool)", ""), ("(int)", "")):
        s = s.replace(a, b)
    return s.replace(", ", ",")


def parse(path=REMARKS):
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run `make -C duckdb-adaptive-compression_amd` first)")
    mangled, rows, cur = [], {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            mangled.append(cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if m and cur and m.group(1).strip() in FIELDS:
            v = m.group(2)
            rows[cur][FIELDS[m.group(1).strip()]] = int(v) if v.isdigit() else v
    dem = subprocess.run([CXXFILT], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    return {short_name(d): rows[m] for m, d in zip(mangled, dem)}


def budgeted(table):
    return {k: v for k, v in sorted(table.items()) if k.startswith(BUDGETED)}


def main():
    # --remarks <file>: the remarks of another build (tools/snapbuild.sh keeps them next to its library)
    table = parse(sys.argv[sys.argv.index("--remarks") + 1]) if "--remarks" in sys.argv else parse()
    if "--write-budget" in sys.argv:
        out = {k: {f: v[f] for f in ("vgprs", "agprs", "vgpr_spills", "sgpr_spills", "scratch", "occupancy", "lds")}
               for k, v in budgeted(table).items()}
        json.dump({"_comment": "upper bounds on vgprs/agprs/spills/scratch/lds, lower bound on occupancy; "
                               "written by tools/kernel_resources.py --write-budget from a build whose sweep is committed",
                   "kernels": out}, open(BUDGET, "w"), indent=1, sort_keys=True)
        print("wrote", BUDGET, len(out), "kernels")
        return
    w = max(len(k) for k in table)
    print("%-*s %5s %5s %6s %7s %4s %7s" % (w, "kernel", "vgpr", "agpr", "spills", "scratch", "occ", "lds"))
    for k, v in sorted(table.items()):
        print("%-*s %5s %5s %6s %7s %4s %7s" % (w, k, v.get("vgprs"), v.get("agprs"), v.get("vgpr_spills"),
                                               v.get("scratch"), v.get("occupancy"), v.get("lds")))


if __name__ == "__main__":
    main()
