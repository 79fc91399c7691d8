#!/bin/bash
# usage: tools/profile_pmc.sh <tag> <python script + args...> : kernel trace + separate PMC passes (never combined) for an arbitrary probe script
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/stdout.txt 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_l2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -o pmc -- python3 "$@" > /dev/null 2> $OUT/pmc_sq.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A10 "per-kernel totals" $OUT/summary.txt | cut -c1-120
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
