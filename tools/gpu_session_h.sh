#!/bin/bash
# round-3 GPU session H: cfg5 -- stage B with whole rounds (CRSDR_LONG_BALANCE) on / off, with and without the fractional-delay pass
set -o pipefail
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_fracdelay.py -m gpu -x -q -k "long or cfg5 or frac" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for bal in 1 0; do
  CRSDR_LONG_BALANCE=$bal timeout -k 10 200 python bench.py --cfg5 > $O/cfg5_bal${bal}_$rep.json 2> $O/cfg5_bal$bal.err; echo "cfg5 bal$bal rc=$?"
  CRSDR_LONG_BALANCE=$bal timeout -k 10 200 python bench.py --cfg5 --frac-apply > $O/cfg5f_bal${bal}_$rep.json 2> $O/cfg5f_bal$bal.err; echo "cfg5 frac bal$bal rc=$?"
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"], 1), "fenced", round(d["value_fenced_median"], 1), "ms/step", round(d["ms_per_step"], 4), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["lags_exact"], d["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
