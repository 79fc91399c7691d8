#!/bin/bash
export TMPDIR=/tmp
D=$(mktemp -d /tmp/bktrace_XXXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 tools/bk_probe.py --default-only > $D/stdout.txt 2>&1
tail -1 $D/stdout.txt | cut -c1-200
python3 - "$D" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bk" in r["Name"]: print(r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["TotalDurationNs"])/3e6, 2), "ms/call-set")
PY
