#!/usr/bin/env python3
"""Print a per-kernel timeline (start offset, duration, gap) from a rocprofv3 kernel_trace.csv."""
import csv, glob, os, sys
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip: skip + n]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void crsdr::", "").replace("crsdr::", "")[:28]
    print(f"{name:28s} q={r.get('Queue_Id','?'):>3s} start={(s - t0) / 1e3:9.1f}us dur={(e - s) / 1e3:7.1f}us gap_after_prev_end={(s - prev_end) / 1e3:7.1f}us")
    prev_end = max(prev_end, e)
