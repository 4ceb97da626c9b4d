#!/bin/bash
set -o pipefail
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_music.py -m gpu -x -q > $O/tests.log 2>&1; echo "music tests rc=$?"; tail -12 $O/tests.log
python - <<'PY' 2>&1 | tee gpurun_out/r3l/cov_time.log
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, ".")
b = importlib.import_module("coherent-rtlsdr_amd.binding")
dev = torch.device("cuda", 0)
for nsig, B in ((1024, 16384), (256, 16384), (2048, 16384)):
    m = torch.randint(-128, 128, ((nsig + 1), B), dtype=torch.int8, device=dev)
    rxx = torch.empty((nsig, nsig, 2), dtype=torch.float32, device=dev)
    for _ in range(3): b.covariance_device(rxx.data_ptr(), m.data_ptr(), nsig + 1, B)
    t0 = time.perf_counter(); n = 20
    for _ in range(n): b.covariance_device(rxx.data_ptr(), m.data_ptr(), nsig + 1, B)
    dt = (time.perf_counter() - t0) / n
    print(nsig, B, "ms", round(1e3 * dt, 4), "int8 TOPS", round(3 * 2 * nsig * nsig * B / dt / 1e12, 1))
PY
