#!/bin/bash
# tools/copy_artifacts.sh: what tools/final_artifacts.sh left in gpurun_out/ -> profiles/r06/ (the names profiles/r06/README.md lists)
set -e
cd "$(dirname "$0")/.."
P=profiles/r06
cp gpurun_out/bench_final.json $P/bench_default_s26_final.json
cp gpurun_out/gpu_tests.log $P/gpu_tests.log
cp gpurun_out/kc24_r6.json gpurun_out/kc26_r6.json gpurun_out/kc26_bins_r6.txt gpurun_out/bk_r6final.txt gpurun_out/upload_phases_s26.txt $P/
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
mkdir -p $P/s26_r6 $P/kc22_r6
cp gpurun_out/prof_s26_r6final/summary.txt gpurun_out/prof_s26_r6final/bench_trace.json $P/s26_r6/
cp $(find gpurun_out/prof_s26_r6final/trace -name "*kernel_stats.csv" | head -1) $P/s26_r6/trace_kernel_stats.csv
cp gpurun_out/prof_kc22_r6final/summary.txt gpurun_out/prof_kc22_r6final/stdout.txt $P/kc22_r6/
cp $(find gpurun_out/prof_kc22_r6final/trace -name "*kernel_stats.csv" | head -1) $P/kc22_r6/trace_kernel_stats.csv
for f in mfma_bits.txt kc4_mfma_probe6.txt; do [ -f gpurun_out/r6b/$f ] && cp gpurun_out/r6b/$f $P/$f; done
echo copied
