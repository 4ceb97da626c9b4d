#!/bin/bash
# round-3 GPU session E: timeline of the per-rank shape (nsig 128, T = 20) and the host cost of the N > 1 code path with a one-rank group
set -o pipefail
O=$PWD/gpurun_out/r3e; mkdir -p $O
R=$PWD
CRSDR_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/force128.json 2> $O/force128.err; echo "force128 rc=$?"
CRSDR_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/force1024.json 2> $O/force1024.err; echo "force1024 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace128 -- python3 $R/bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline --repeats 40 > $O/trace128.log 2>&1; echo "trace rc=$?"
cd $R
python3 profiles/timeline.py $O/trace128 60 700 > $O/timeline128.txt 2>&1; head -70 $O/timeline128.txt
python - <<'PY'
import json
for f in ("force128", "force1024"):
    try:
        d = json.loads([l for l in open("gpurun_out/r3e/%s.json" % f) if l.startswith("{")][-1])
        print(f, round(d["value"]), "fenced", round(d["value_fenced_median"]), "ev", round(d["timing"]["value_gpu_events"]), "host_ms/batch", round(d["host_issue_ms_per_batch"], 4),
              "k1_ms", round(d["roofline"]["avg_launch_ms"], 4), d["lags_exact"], d.get("matrix_assembled"), d.get("scalars_assembled"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/force128.err
