#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/<name>/**) into the small summaries committed under
profiles/: per-kernel stats of the --kernel-trace --stats run and HBM traffic per launch from the two
separate --pmc passes (FETCH_SIZE, WRITE_SIZE).

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly HALF of the bytes of a wide coalesced (16 B/lane) streaming read, so it is doubled; WRITE_SIZE is
exact for 16 B/lane streaming stores.

usage: summarize.py <round-tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir> <rows>
"""
import collections
import csv
import glob
import json
import os
import sys


def one(pattern):
    fs = glob.glob(pattern, recursive=True)
    if not fs:
        raise SystemExit("no file matches %s" % pattern)
    return fs[0]


def short(name):
    for k in ("k_unpack_jobs", "k_unpack", "k_scan_agg", "k_encode_1p", "k_sel_clear_edges", "k_gather", "k_pack",
              "k_analyze", "k_plan", "k_minmax_init", "k_fetch"):
        if k in name:
            if "unsigned long" in name:
                t = "u64"
            elif "unsigned int" in name:
                t = "u32"
            elif "unsigned short" in name:
                t = "u16"
            elif "unsigned char" in name:
                t = "u8"
            else:
                t = ""
            extra = ""
            if k == "k_scan_agg":  # k_scan_agg<U, OP, V>: OP 0 sum, 1 count(range), 2 probe, 3 select bitmap
                import re
                m = re.search(r"k_scan_agg<[^,]+, (\d), (true|false)(?:, (true|false))?>", name)
                op = {"0": "sum", "1": "count", "2": "probe", "3": "select"}.get(m.group(1), "?") if m else "?"
                extra = "," + op + (",valid" if m and m.group(2) == "true" else "") + \
                    (",narrow" if m and m.group(3) == "true" else "")
            if k == "k_unpack" and ", true>" in name:
                extra = ",range"
            return "%s<%s%s>" % (k, t, extra) if t else k
    return None


def main():
    tag, trace_dir, fetch_dir, write_dir, rows = sys.argv[1:6]
    here = os.path.dirname(os.path.abspath(__file__))
    stats = {}
    rows_out = []
    for r in csv.DictReader(open(one(os.path.join(trace_dir, "**", "*_kernel_stats.csv")))):
        s = short(r["Name"])
        if s:
            stats[s] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                        "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
            rows_out.append(r)
    with open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows_out[0].keys()))
        w.writeheader()
        w.writerows(rows_out)
    pmc = collections.defaultdict(dict)
    for d, counter in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        acc = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
            s = short(r["Kernel_Name"])
            if s and r["Counter_Name"] == counter:
                acc[s].append(float(r["Counter_Value"]))
                meta[s] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                           "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}
        for s, v in acc.items():
            pmc[s][counter + "_KiB_avg"] = sum(v) / len(v)
            pmc[s]["launches_" + counter] = len(v)
            pmc[s].update(meta[s])
    summary = {"round": tag, "rows": int(rows), "units": "bytes per launch", "kernels": {}}
    for s, p in pmc.items():
        fetch = 2.0 * p.get("FETCH_SIZE_KiB_avg", 0.0) * 1024.0
        write = p.get("WRITE_SIZE_KiB_avg", 0.0) * 1024.0
        e = {"hbm_read_bytes": fetch, "hbm_write_bytes": write, "hbm_bytes": fetch + write}
        e.update(p)
        if s in stats:
            e.update(stats[s])
            e["hbm_GBps_at_avg_ns"] = (fetch + write) / stats[s]["avg_ns"]
        summary["kernels"][s] = e
    json.dump(summary, open(os.path.join(here, "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)
    if "k_unpack<u64>" in summary["kernels"]:
        k = summary["kernels"]["k_unpack<u64>"]
        import importlib
        sys.path.insert(0, os.path.join(here, ".."))
        sha = importlib.import_module("duckdb-adaptive-compression_amd").kernel_source_sha256()  # bench.py's hash
        per_kernel = {n: {f: summary["kernels"][n][f] for f in ("hbm_bytes", "hbm_read_bytes", "hbm_write_bytes")}
                      for n in ("k_unpack<u64>", "k_scan_agg<u64,sum>", "k_scan_agg<u64,select>", "k_encode_1p<u64>")
                      if n in summary["kernels"]}
        json.dump({"round": tag, "rows": int(rows), "kernel": "k_unpack<u64>", "kernel_source_sha256": sha,
                   "hbm_bytes_per_launch": k["hbm_bytes"], "hbm_read_bytes": k["hbm_read_bytes"],
                   "hbm_write_bytes": k["hbm_write_bytes"], "kernels": per_kernel,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB*1024; "
                             "FETCH_SIZE doubled (gfx950 16 B/lane streaming-read correction)"},
                  open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(summary["kernels"], indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
