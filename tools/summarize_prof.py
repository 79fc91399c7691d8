#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: per-kernel time (kernel trace) and per-kernel PMC sums."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True):
    print(open(f).read())
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows("trace/**/*kernel_trace.csv"):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("== per-kernel totals from the trace (ms) ==")
for k, (c, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:20]:
    print(f"{ms:12.3f} ms  {c:6d} calls  avg {ms/c:10.3f} ms  {k[:100]}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in rows(os.path.basename(d) + "/**/*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    print(f"== {os.path.basename(d)}: counter sums over all dispatches (and per dispatch) ==")
    for k in tot:
        if not k.startswith("gmsx") and "k_tc" not in k and "k_kc" not in k and "k_bk" not in k: continue
        for cname, v in tot[k].items():
            print(f"  {k[:60]:60s} {cname:22s} sum {v:18.1f}  n={cnt[(k,cname)]}  per-dispatch {v/cnt[(k,cname)]:16.1f}")
