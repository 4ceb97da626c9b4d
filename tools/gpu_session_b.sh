#!/bin/bash
# round-3 GPU session B: K1 epilogue change -- parity subset, then A/B against the previous build in the track cadence
set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_cfg4.py tests/test_gpu_fuzz.py tests/test_host_cpp.py tests/test_gpu_exchange.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 400 python tools/k2_ab.py --track --rounds 3 "CRSDR_LIB=$PWD/tools/libcrsdr_old.so" "CRSDR_K1_NEW=1" 2>&1 | tee $O/k1_ab.log
