#!/bin/bash
# round-3 GPU session A: tests, driver-style bench, one rank's shape of the 8-GPU run, rehearsal of world 2
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/driver.json 2> $O/driver.err; echo "driver rc=$?"
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/nsig128.json 2> $O/nsig128.err; echo "nsig128 rc=$?"
CRSDR_BENCH_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/reh2.json 2> $O/reh2.err; echo "reh2 rc=$?"
python - <<'PY'
import json
for f in ("driver", "nsig128", "reh2"):
    try:
        d = json.loads([l for l in open("gpurun_out/r3a/%s.json" % f) if l.startswith("{")][-1])
        print(f, round(d["value"]), "fenced", round(d["value_fenced_median"]), "first5", round(d["value_first5"]), "ev", round(d["timing"]["value_gpu_events"]),
              "host_ms/batch", round(d["host_issue_ms_per_batch"], 4), "k1_ms", d["roofline"]["avg_launch_ms"], d["lags_exact"], d["batches_timed"], d.get("matrix_assembled"), d.get("scalars_assembled"))
    except Exception as e:
        print(f, "ERR", e)
PY
