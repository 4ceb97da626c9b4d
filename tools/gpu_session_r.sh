#!/bin/bash
# rocprofv3 trace + PMC passes of the long-block configuration (cfg5), plain and with the fractional-delay correction; only the
# summaries travel back (the raw traces of 1 349 launches x 8 kernels exceed what gpurun merges)
set -o pipefail
for tag in r03cfg5 r03cfg5f; do
  extra="--cfg5"; [ $tag = r03cfg5f ] && extra="--cfg5 --frac-apply"
  timeout -k 10 500 bash profiles/run_profile.sh $tag $extra > gpurun_out/prof_$tag.log 2>&1; echo "$tag profile rc=$?"
  mkdir -p gpurun_out/keep_$tag
  cp gpurun_out/prof_$tag/summary.txt gpurun_out/prof_$tag/traffic.json gpurun_out/keep_$tag/ 2>/dev/null
  cp gpurun_out/prof_$tag/trace/*/*kernel_stats.csv gpurun_out/keep_$tag/kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/prof_$tag
done
ls -la gpurun_out/keep_r03cfg5 gpurun_out/keep_r03cfg5f
