#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + per-kernel PMC averages) into a short text summary."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def short(n): return n.split("(")[0].replace("void crsdr::", "")[:60]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for r in csv.DictReader(open(f)):
        print(f"{short(r['Name']):60s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):12.1f} pct={r['Percentage']}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== pmc:", os.path.basename(d))
    for k, cs in acc.items():
        if "crsdr" not in k and "k_" not in k: continue
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
