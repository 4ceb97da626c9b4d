#!/bin/bash
set -o pipefail
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 600 python tools/k2_ab.py --rounds 2 "CRSDR_K2_PF=0" "CRSDR_K2_PF=1024" "CRSDR_K2_PF=2048" "CRSDR_K2_PF=4096" "CRSDR_K2_PF=8192" 2>&1 | tee $O/k2_pf_ab.log
