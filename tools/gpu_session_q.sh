#!/bin/bash
# K1 variant thresholds at the per-rank shapes of the driver's runs (--steps 20: T = 20; nsig = 1024 / world)
set -o pipefail
O=gpurun_out/r3q; mkdir -p $O
for nsig in 128 256; do for v in auto packed; do
  CRSDR_K1_VARIANT=$v timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig $nsig --no-extras --no-cpu-baseline > $O/n${nsig}_$v.json 2> $O/n${nsig}_$v.err
  python - $O/n${nsig}_$v.json $nsig $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('nsig',sys.argv[2],sys.argv[3], round(d['value']), 'fenced', round(d['value_fenced_median']), 'k1_ms', round(d['roofline']['avg_launch_ms'],4), d['lags_exact'])
PY
done; done
