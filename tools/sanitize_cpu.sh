#!/bin/bash
# ASan + UBSan pass over the CPU-side C code (oracle, synthetic source).  GPU sanitizers are not available
# on the MI355X pool; the HIP kernels are covered by the parity tests instead.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/crsdr_san; mkdir -p $OUT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -std=c11 -D_GNU_SOURCE -o $OUT/liboracle.so $ROOT/oracle/coherent_oracle.c $ROOT/oracle/beamformer_oracle.c -lm -lpthread
gcc -O1 -g -fsanitize=address,undefined -ffp-contract=off -fPIC -shared -std=c11 -o $OUT/libcsynth.so $ROOT/coherent-rtlsdr_amd/host/csynth.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) CRSDR_SAN_DIR=$OUT python3 $ROOT/tools/sanitize_cpu.py
