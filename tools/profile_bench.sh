#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then separate PMC passes (never combined with tracing).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --check-scale 0 --ref-scale 0 --pmc 0 --side 0 --big 0 $*"   # --pmc 0: no nested rocprofv3 children under the profiler
# the first pass leaves the generated graph as a .sg cache (bench.py default), the PMC passes load it
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_l2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_lds.err
find $OUT -name "*.csv" | head -50
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back payload small: drop the per-dispatch traces, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
