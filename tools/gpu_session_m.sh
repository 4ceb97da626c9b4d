#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3m; mkdir -p $O
R=$PWD
cat > /tmp/covt.py <<'PY'
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, sys.argv[1])
b = importlib.import_module("coherent-rtlsdr_amd.binding")
dev = torch.device("cuda", 0)
nsig, B = 1024, 16384
m = torch.randint(-128, 128, ((nsig + 1), B), dtype=torch.int8, device=dev)
rxx = torch.empty((nsig, nsig, 2), dtype=torch.float32, device=dev)
for _ in range(30): b.covariance_device(rxx.data_ptr(), m.data_ptr(), nsig + 1, B)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/covtrace -- python3 /tmp/covt.py $R > $O/covtrace.log 2>&1; echo "trace rc=$?"
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r3m/covtrace/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
