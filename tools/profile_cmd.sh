#!/bin/bash
# usage: tools/profile_cmd.sh <tag> <python script + args...> : kernel trace + stats for an arbitrary probe script
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/stdout.txt 2> $OUT/trace.err
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A14 "per-kernel totals" $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +2M -delete
