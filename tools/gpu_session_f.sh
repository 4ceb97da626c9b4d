#!/bin/bash
# round-3 GPU session F: reference spectra folded into K1 -- parity, then fold on / off at the per-rank shape and the one-GPU shape
set -o pipefail
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for f in 1 0; do
  CRSDR_K1_FOLD=$f timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/nsig128_fold$f.json 2> $O/nsig128_fold$f.err; echo "nsig128 fold$f rc=$?"
  CRSDR_K1_FOLD=$f timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/driver_fold$f.json 2> $O/driver_fold$f.err; echo "driver fold$f rc=$?"
  CRSDR_K1_FOLD=$f timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 256 --no-extras --no-cpu-baseline > $O/nsig256_fold$f.json 2> $O/nsig256_fold$f.err; echo "nsig256 fold$f rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), "fenced", round(d["value_fenced_median"]), "first5", round(d["value_first5"]),
              "host_ms/batch", round(d["host_issue_ms_per_batch"], 4), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["lags_exact"], d["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
