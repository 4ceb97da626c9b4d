#!/bin/bash
# round-3 GPU session C: two-row phase kernel -- parity, then A/B in the locked cadence (one row per workgroup / pairs / the previous build)
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 500 python tools/k2_ab.py --rounds 3 "CRSDR_LIB=$PWD/tools/libcrsdr_old.so" "CRSDR_K2_FUSED=1" "CRSDR_K2_FUSED=2" 2>&1 | tee $O/k2_ab.log
