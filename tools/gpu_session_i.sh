#!/bin/bash
set -o pipefail
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 120 tools/mfma_corun 20000 2>&1 | tee $O/mfma_corun.log
timeout -k 10 300 python bench.py > $O/r03_bench_default.json 2> $O/default.err; echo "default rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/r03_bench_driver.json 2> $O/driver.err; echo "driver rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3i/*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(d["value"], 1), "fenced", round(d["value_fenced_median"], 1), "first5", round(d["value_first5"], 1), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["roofline"]["traffic"], d["roofline"]["traffic_source"][:40])
PY
