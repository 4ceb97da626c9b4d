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

# ---- HBM traffic per launch (bench.py's roofline.traffic): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is
# doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a streamed read), WRITE_SIZE as is.
import json
def pmc_avg(counter):
    res = {}
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = sum(v) / len(v)
    return res
fetch, write = pmc_avg("FETCH_SIZE"), pmc_avg("WRITE_SIZE")
traffic = {}
for key, pat in (("k_xcorr_lag", "k_xcorr_lag"), ("k_phase_dot", "k_phase_dot"), ("k_align_quant", "k_align_quant"), ("k_align_fused", "k_align_fused"), ("k_ref_spectrum", "k_ref_spectrum")):
    ks = [k for k in fetch if pat in k]
    if not ks or ks[0] not in write:
        continue
    k = ks[0]
    traffic[key] = {"bytes_per_launch": int(2 * fetch[k] * 1024 + write[k] * 1024), "fetch_size_raw_kib": round(fetch[k], 1),
                    "write_size_raw_kib": round(write[k], 1), "kernel": k}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("== traffic.json:", json.dumps(traffic))
