#!/bin/bash
# the fractional-delay correction at the reference's block size (k_frac_apply behind the phase kernels): its price on the default workload
set -o pipefail
O=gpurun_out/r3s; mkdir -p $O
for fa in "" "--frac-apply"; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline $fa > $O/default$fa.json 2> $O/default$fa.err; echo "rc=$?"
  python - "$O/default$fa.json" "$fa" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2] or 'plain', round(d['value']), 'fenced', round(d['value_fenced_median']), d['kernel_ms'], d['lags_exact'])
PY
done
