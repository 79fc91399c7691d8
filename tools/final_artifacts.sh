#!/bin/bash
# the artifacts a round ends with (run on the GPU box: gpurun -- 'bash tools/final_artifacts.sh'); everything lands in gpurun_out/ and is copied to profiles/ by hand
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/gpu_tests.log; tail -2 gpurun_out/gpu_tests.log
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
python bench.py --dump-traffic gpurun_out/hbm_traffic.json 2>gpurun_out/bench_final.err | tail -1 > gpurun_out/bench_final.json; wc -c gpurun_out/bench_final.json
if [ "${1:-}" = "bk" ]; then
  bash tools/profile_pmc.sh bk_r6final tools/bk_probe.py --default-only > gpurun_out/prof_bk_r6final.log 2>&1; tail -2 gpurun_out/prof_bk_r6final.log
  (python tools/bk_probe.py --default-only | tail -1; bash tools/probes/bk_trace.sh; python tools/bk_probe.py 18 64 --default-only | tail -1) > gpurun_out/bk_r6final.txt 2>&1
  exit 0
fi
bash tools/profile_bench.sh s26_r6final > gpurun_out/prof_s26_r6final.log 2>&1; tail -3 gpurun_out/prof_s26_r6final.log
bash tools/profile_pmc.sh kc22_r6final tools/kc_probe.py 22 > gpurun_out/prof_kc22_r6final.log 2>&1; tail -2 gpurun_out/prof_kc22_r6final.log
python tools/kc_probe.py 24 --ab 2>&1 | tail -4 > gpurun_out/kc24_r6.json
python tools/kc_probe.py 26 --ab 2>&1 | tail -4 > gpurun_out/kc26_r6.json
(bash tools/kc_trace.sh 26; bash tools/probes/kc_bin_traffic.sh 26; GMSX_OPT_TIMING=1 python tools/kc_probe.py 26 2>&1 | grep "kclique\]" | sort -u) > gpurun_out/kc26_bins_r6.txt 2>&1
(python tools/bk_probe.py --default-only | tail -1; bash tools/probes/bk_trace.sh) > gpurun_out/bk_r6final.txt 2>&1
GMSX_OPT_TIMING=1 python tools/probes/upload_phases.py 26 2>&1 | grep "gmsx\|rep" | grep -v "rmat\|host\]" > gpurun_out/upload_phases_s26.txt
