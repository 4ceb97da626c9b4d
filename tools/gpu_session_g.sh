#!/bin/bash
# round-3 GPU session G: fold v2 (early peek, reference items first) -- K1 variant parity, then fold on / off
set -o pipefail
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_cfg4.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for f in 1 0; do
  CRSDR_K1_FOLD=$f timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/nsig128_fold${f}_$rep.json 2> $O/nsig128_fold$f.err; echo "nsig128 fold$f rc=$?"
  CRSDR_K1_FOLD=$f timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/driver_fold${f}_$rep.json 2> $O/driver_fold$f.err; echo "driver fold$f rc=$?"
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), "fenced", round(d["value_fenced_median"]), "first5", round(d["value_first5"]),
              "host_ms/batch", round(d["host_issue_ms_per_batch"], 4), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["lags_exact"], d["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
