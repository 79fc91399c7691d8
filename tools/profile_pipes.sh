#!/bin/bash
# usage: tools/profile_pipes.sh <tag> <python script + args...>
# Pipe-busy counters of the triangle-count kernels, one after the other (GMSX_OPT_TC_OVERLAP=0 unless the caller sets it): kernel trace, then
# separate --pmc passes (never combined with tracing; 8 SQ slots per pass).  What is busy: SQ_ACTIVE_INST_{VALU,LDS,VMEM,SCA} (quad-cycles a
# wave had an instruction of that kind executing), SQ_WAIT_* (parked), SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT (LDS-array cycles).
set -u
TAG=$1; shift
OUT=gpurun_out/pipes_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export GMSX_OPT_TC_OVERLAP=${GMSX_OPT_TC_OVERLAP:-0}
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/stdout.txt 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_active -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_active.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_lds -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_misc -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_misc.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_grbm.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A10 "per-kernel totals" $OUT/summary.txt | cut -c1-120
tail -2 $OUT/*.err | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
