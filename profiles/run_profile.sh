#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <tag> [bench args...]
# Writes rocprofv3 outputs under gpurun_out/prof_<tag>/ ; copy the summaries you keep into profiles/.
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 128 --warmup 64 --no-cpu-baseline --skip-pcie $@"   # every launch carries the default 64 blocks (the PCIe extra submits 16 at a time: skipped)
# pass 1: kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
# separate PMC passes (never combined with sys/hip/hsa traces)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || echo "pmc pass $N failed"
done
python3 $ROOT/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
