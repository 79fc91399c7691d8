#!/bin/bash
# usage: tools/kc_trace.sh [scale] : kernel trace of the k = 4 call at scale 22 with the bins on ONE stream (per-bin times)
export TMPDIR=/tmp
D=$(mktemp -d /tmp/trace_XXXXXX)
GMSX_OPT_KC_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 tools/kc_probe.py ${1:-22} --k 4 > $D/stdout.txt 2>&1
tail -1 $D/stdout.txt | cut -c1-300
python3 - "$D" <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_kc" in r["Kernel_Name"]:
            rows.append(r)
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//3
for r in rows[-n:]:
    print(round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2), r["Kernel_Name"].split("(")[0][-40:], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "lds", r.get("LDS_Block_Size"), "vgpr", r.get("VGPR_Count"))
PY
