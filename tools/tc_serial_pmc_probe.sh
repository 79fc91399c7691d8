#!/bin/bash
# GPU: the triangle-count kernels SERIALISED (GMSX_TC_OVERLAP=0 inside tools/tc_phase_probe.py) under rocprofv3: a kernel trace, then
# separate --pmc passes for VALU issue / lane utilisation and for FETCH_SIZE (never combined with tracing).  Writes gpurun_out/ts/.
export TMPDIR=/tmp
mkdir -p gpurun_out/ts
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts/trace -o trace -- python3 tools/tc_phase_probe.py 26 > gpurun_out/ts/out.json 2> gpurun_out/ts/err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/ts/p -o pmc -- python3 tools/tc_phase_probe.py 26 >> gpurun_out/ts/out.json 2>> gpurun_out/ts/err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ts/f -o pmc -- python3 tools/tc_phase_probe.py 26 >> gpurun_out/ts/out.json 2>> gpurun_out/ts/err.txt
cat gpurun_out/ts/out.json
grep "k_tc_" gpurun_out/ts/trace/trace_kernel_stats.csv | cut -c1-50,180-400
find gpurun_out/ts -name "*kernel_trace.csv" -size +2M -delete
