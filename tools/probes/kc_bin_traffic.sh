#!/bin/bash
# usage: tools/probes/kc_bin_traffic.sh [scale]: beyond-L2 read traffic per k = 4 bin (FETCH_SIZE x 1024 x 2: the requests are 128-byte ones), bins on one stream
export TMPDIR=/tmp
D=$(mktemp -d /tmp/kcpmc_XXXXXX)
GMSX_OPT_KC_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -o p -- python3 tools/kc_probe.py ${1:-26} --k 4 > $D/stdout.txt 2>&1
python3 - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_kc" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-34:], float(r["Counter_Value"])))
rows.sort()
n = len(rows) // 3
for d, k, v in rows[-n:]:
    print(d, k, round(v * 1024 * 2 / 1e9, 1), "GB")
PY
