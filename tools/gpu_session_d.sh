#!/bin/bash
# round-3 GPU session D: after the clean-up -- full parity suite; per-rank shape (nsig 128, T = 20) with the packed / two-row K1
set -o pipefail
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for v in auto q packed; do
  CRSDR_K1_VARIANT=$v timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/nsig128_$v.json 2> $O/nsig128_$v.err; echo "nsig128 $v rc=$?"
done
for v in auto q; do
  CRSDR_K1_VARIANT=$v timeout -k 10 120 python bench.py --steps 64 --warmup 64 --nsig 128 --no-extras --no-cpu-baseline > $O/nsig128_T64_$v.json 2> $O/nsig128_T64_$v.err; echo "nsig128 T64 $v rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3d/nsig128*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), "fenced", round(d["value_fenced_median"]), "ev", round(d["timing"]["value_gpu_events"]),
              "host_ms/batch", round(d["host_issue_ms_per_batch"], 4), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["lags_exact"], d["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
