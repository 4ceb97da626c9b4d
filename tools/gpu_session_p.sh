#!/bin/bash
# the N > 1 code path of bench.py on one GPU: forced exchange (one-rank RCCL group) and the world-2 rehearsal
set -o pipefail
O=gpurun_out/r3p; mkdir -p $O
CRSDR_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/force.json 2> $O/force.err; echo "force rc=$?"
CRSDR_BENCH_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/reh2.json 2> $O/reh2.err; echo "reh2 rc=$?"
python - <<'PY'
import json
for f in ("force", "reh2"):
    d = json.loads([l for l in open("gpurun_out/r3p/%s.json" % f) if l.startswith("{")][-1])
    print(f, round(d["value"], 1), "fenced", round(d["value_fenced_median"], 1), "host_issue_ms_per_batch", d.get("host_issue_ms_per_batch"), d["lags_exact"], d.get("matrix_assembled"), d.get("scalars_assembled"))
PY
