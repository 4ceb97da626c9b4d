#!/bin/bash
# round-3 records: full parity suite, rocprofv3 trace + PMC passes of the default workload, the bench lines of every configuration
set -o pipefail
O=gpurun_out/r3rec; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 500 bash profiles/run_profile.sh r03 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
timeout -k 10 300 python bench.py > $O/r03_bench_default.json 2> $O/default.err; echo "default rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/r03_bench_driver.json 2> $O/driver.err; echo "driver rc=$?"
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --frac-apply > $O/r03_bench_default_frac.json 2> $O/default_frac.err; echo "default frac rc=$?"
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/r03_bench_nsig128.json 2> $O/nsig128.err; echo "nsig128 rc=$?"
timeout -k 10 120 python bench.py --nsig 128 --no-extras --no-cpu-baseline > $O/r03_bench_nsig128_T64.json 2> $O/nsig128b.err; echo "nsig128 T64 rc=$?"
timeout -k 10 200 python bench.py --nsig 21 --steps 4096 --warmup 256 --nbuf 128 --no-extras --no-cpu-baseline > $O/r03_bench_cfg2.json 2> $O/cfg2.err; echo "cfg2 rc=$?"
timeout -k 10 200 python bench.py --nsig 256 --steps 2048 --warmup 128 --no-extras --no-cpu-baseline > $O/r03_bench_cfg3.json 2> $O/cfg3.err; echo "cfg3 rc=$?"
timeout -k 10 200 python bench.py --cfg5 > $O/r03_bench_cfg5.json 2> $O/cfg5.err; echo "cfg5 rc=$?"
timeout -k 10 200 python bench.py --cfg5 --frac-apply > $O/r03_bench_cfg5_frac.json 2> $O/cfg5f.err; echo "cfg5 frac rc=$?"
CRSDR_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --nsig 128 --no-extras --no-cpu-baseline > $O/r03_bench_nsig128_forced_exchange.json 2> $O/force.err; echo "force rc=$?"
for n in 2 4; do CRSDR_BENCH_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/reh$n.json 2> $O/reh$n.err; echo "reh$n rc=$?"; done
timeout -k 10 300 coherent-rtlsdr_amd/host/coherent_demo --bench --nsig 1024 --batch 16 --blocks 512 > $O/cpp_bench.log 2>&1; echo "cpp bench rc=$?"; tail -2 $O/cpp_bench.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3rec/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        loc = d.get("locked", {})
        print(f.split("/")[-1], round(d["value"], 1), "fenced", round(d["value_fenced_median"], 1), "first5", round(d["value_first5"], 1), "k1_ms", round(d["roofline"]["avg_launch_ms"], 4),
              "frac", round(d["roofline"]["frac"], 4), d["lags_exact"], d.get("matrix_assembled"), "locked", round(loc.get("blocks_per_s", 0)), round(loc.get("roofline", {}).get("read_plus_write_GBs", 0)),
              "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
