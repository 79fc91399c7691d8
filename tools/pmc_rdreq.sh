#!/bin/bash
# usage: tools/pmc_rdreq.sh <tag> <python script + args…>: memory-side read requests of the L2 by size class (calibration of FETCH_SIZE, which tallies
# every request at 64 B): TCC_EA0_RDREQ_sum and its 32-B / 64-B / 128-B classes in one pass, FETCH_SIZE and TCC_MISS_sum in passes of their own.
set -u
TAG=$1; shift
OUT=gpurun_out/rdreq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_rdreq -o pmc -- python3 "$@" > $OUT/stdout.txt 2> $OUT/rdreq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- python3 "$@" > /dev/null 2> $OUT/l2.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "RDREQ|FETCH_SIZE|TCC_MISS" $OUT/summary.txt | cut -c1-170
find $OUT -name "*counter_collection.csv" -size +2M -delete
