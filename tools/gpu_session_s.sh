#!/bin/bash
# the fractional-delay correction at the reference's block size: parity, then its price on the default workload (K1-network kernel / generic)
set -o pipefail
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_fracdelay.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "frac tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
for v in 0 1; do
  CRSDR_FRAC_GENERIC=$v timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --frac-apply > $O/frac_generic$v.json 2> $O/frac_generic$v.err; echo "rc=$?"
  python - "$O/frac_generic$v.json" "$v" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('generic='+sys.argv[2], round(d['value']), 'fenced', round(d['value_fenced_median']), d['kernel_ms'], d['lags_exact'])
PY
done
