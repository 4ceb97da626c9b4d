#!/bin/bash
set -o pipefail
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -m gpu -x -q -k "phase_path_variants" > $O/tests.log 2>&1; rc=$?; echo "variants test rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/k2_ab.py "CRSDR_K2_PERSIST=0" "CRSDR_K2_PERSIST=4" "CRSDR_K2_PERSIST=3" "CRSDR_K2_PERSIST=5 CRSDR_LIB=$PWD/tools/libcrsdr_b5.so" "CRSDR_K2_PERSIST=4 CRSDR_LIB=$PWD/tools/libcrsdr_b5.so" --rounds 2 2>&1 | tee $O/k2_persist_ab.log
