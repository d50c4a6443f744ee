#!/usr/bin/env python3
"""Timeline of the LAST SET forward in a rocprofv3 kernel trace of tools/quick_bench_set.py (usage: set_timeline.py trace.csv)."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pack" in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
def short(n):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::|sgrl_gemm::)?([a-z_0-9A-Z]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:64]
busy = {}
for r in rows[s:e]:
    st, d = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("%8.1f %7.1f  %-64s grid=%s" % (st / 1e3, d / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size"))))
    k = short(r["Kernel_Name"]).split("<")[0]
    busy[k] = busy.get(k, 0) + d / 1e3
print("forward span %.1f us" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e3))
print("sum of kernel durations by name:", {k: round(v, 1) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])})
