#!/bin/bash
set -o pipefail
O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_plan.py -m gpu -x -q -k "long_row_phase_kernel_variants" > $O/tests.log 2>&1; rc=$?; echo "variants test rc=$rc"; tail -15 $O/tests.log
[ $rc -eq 0 ] || exit 1
for v in 1 0 4 1 0 4; do CRSDR_LONG_K2_U=$v CRSDR_LONG_K2=$v timeout -k 10 200 python bench.py --cfg5 > $O/cfg5_k2_$v.json 2> $O/cfg5_$v.err; python - $O/cfg5_k2_$v.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('LONG_K2='+sys.argv[2], round(d['value'],1), 'fenced', round(d['value_fenced_median'],1), d.get('kernel_ms'))
PY
done
