#!/usr/bin/env python3
"""Print the JSON of tools/ab_select.py / tools/ab_encode.py as a table.  usage: python3 tools/ab_show.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, d.get("lib", "")[-44:])
    for c in d["cases"]:
        print(" ", c["case"], c["rows"], c.get("selected", ""))
        for s in c["settings"]:
            k = ";".join("%s=%s" % kv for kv in s["knobs"].items())
            if "select_ms" in s:
                print("    %-40s sel %.4f (%4.0f) sum %.4f (%4.0f) cnt %.4f (%4.0f) gather %.4f (%4.0f GB/s)" % (
                    k, s["select_ms"], s["select_read_GBps"], s.get("sum_ms", 0), s.get("sum_read_GBps", 0),
                    s.get("count_ms", 0), s.get("count_read_GBps", 0), s["gather_ms"], s["gather_traffic_GBps"]))
                if "unpack_ms" in s:
                    print("    %-40s unpack %.4f ms (%4.0f GB/s of read + written bytes)" % ("", s["unpack_ms"], s["unpack_total_GBps"]))
                if "clustered_gather_ms" in s:
                    print("    %-40s clustered %.0f %%: gather %.4f ms" % ("", 100 * s["clustered_fraction"], s["clustered_gather_ms"]))
                if "masked_sum_ms" in s:
                    print("    %-40s masked: sum %.4f (%4.0f) cnt %.4f (%4.0f) sel %.4f (%4.0f GB/s)" % (
                        "", s["masked_sum_ms"], s["masked_sum_read_GBps"], s["masked_count_ms"], s["masked_count_read_GBps"],
                        s["masked_select_ms"], s["masked_select_read_GBps"]))
            else:
                print("    %-40s median %.4f min %.4f max %.4f ms" % (k, s["median_ms"], s["min_ms"], s["max_ms"]))
