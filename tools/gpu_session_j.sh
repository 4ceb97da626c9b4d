#!/bin/bash
set -o pipefail
O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fracdelay.py -m gpu -x -q > $O/tests.log 2>&1; echo "frac tests rc=$?"; tail -15 $O/tests.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fracdelay.py > $O/tests_all.log 2>&1; echo "all tests rc=$?"; tail -3 $O/tests_all.log
