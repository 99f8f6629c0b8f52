#!/usr/bin/env python3
"""Per kernel and counter: dispatches and the mean counter value per dispatch, from the counter_collection CSVs that
`rocprofv3 --pmc ... --output-format csv -d <dir>` leaves under <dir>.  usage: python3 tools/pmc_sum.py <dir> [name filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|adac::|void ", "", name)
    name = name.replace("unsigned long", "u64").replace("unsigned int", "u32").replace("unsigned short", "u16").replace("unsigned char", "u8")
    return name.split("(")[0]


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-28s n=%-3d mean %.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
