#!/bin/bash
# Per-kernel averages of the long-block stages (cfg5) under diagnostic switches: tools/lb_ab.sh "<CRSDR_LB_DBG values>" [lib]
# (a -DCRSDR_LB_EXPERIMENT build of csrc/crsdr.hip as tools/libcrsdr_lbexp.so; one rocprofv3 kernel trace per value)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${2:-$ROOT/tools/libcrsdr_lbexp.so}
cd /tmp && export TMPDIR=/tmp
for D in $1; do
  OUT=$ROOT/gpurun_out/lbab_$D
  rm -rf $OUT
  CRSDR_LB_DBG=$D CRSDR_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --cfg5 --steps 120 --warmup 60 --repeats 1 > $OUT.log 2>&1
  F=$(ls $OUT/*/*kernel_stats.csv | head -1)
  echo "== CRSDR_LB_DBG=$D"
  python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("k_long_fwd_cols<7, false>", "k_rows14_cf32q", "k_long_inv_cols<7, false>"):
        if k in n:
            print(f"   {k:28s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $OUT
done
