#!/bin/bash
# usage: tools/trace_kernels.sh <filter> <python script + args…> : rocprofv3 kernel trace of a probe script, per-kernel totals for kernels matching <filter>
export TMPDIR=/tmp
F=$1; shift
D=$(mktemp -d /tmp/trace_XXXXXX)
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 "$@" > $D/stdout.txt 2>&1
tail -3 $D/stdout.txt | cut -c1-300
python3 - "$D" "$F" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0, []])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sys.argv[2] not in k:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        agg[k][0] += 1
        agg[k][1] += d
        agg[k][2].append(round(d, 1))
for k, (c, ms, l) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{ms:9.1f} ms {c:4d} calls {k[:70]}  {l[:16]}")
PY
rm -rf $D
