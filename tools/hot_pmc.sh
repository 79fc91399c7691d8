#!/bin/bash
# usage: tools/hot_pmc.sh <windows> <KB> <min> : kernel trace + FETCH_SIZE + L2 hit/miss passes of the scale-26 triangle count with the given hot-window setting
export TMPDIR=/tmp
export GMSX_OPT_TC_HOT_WINDOWS=$1 GMSX_OPT_TC_HOT_KB=$2 GMSX_OPT_TC_HOT_MIN=$3
OUT=gpurun_out/hot_$1_$2_$3
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/tc_probe.py 26 --passes 3 > $OUT/stdout.txt 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 tools/tc_probe.py 26 --passes 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- python3 tools/tc_probe.py 26 --passes 2 > /dev/null 2> $OUT/pmc_l2.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep "k_tc_items\|k_tc_light" $OUT/summary.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
