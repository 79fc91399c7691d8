#!/bin/bash
# kernel-trace summaries of the non-headline paths (k-clique, per-vertex counts, Bron-Kerbosch): bash tools/profile_misc.sh
set -u
export TMPDIR=/tmp
for job in "kc22:tools/kc_one.py 22 4" "kc24:tools/kc_one.py 24 4" "vc22:tools/vc_one.py 22" "bk21:tools/bk_one.py 21 56"; do
  tag=${job%%:*}; cmd=${job#*:}
  bash tools/profile_cmd.sh $tag $cmd > /dev/null 2>&1
  echo "== $tag: python3 $cmd"; tail -1 gpurun_out/prof_$tag/stdout.txt; grep -A8 "per-kernel totals" gpurun_out/prof_$tag/summary.txt | cut -c1-150
done
