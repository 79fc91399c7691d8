export TMPDIR=/tmp
python3 bench.py --scale 26 --ref-scale 0 2>gpurun_out/bench26.err | tail -1 > gpurun_out/bench_s26_v8.json
cut -c1-400 gpurun_out/bench_s26_v8.json
mkdir -p gpurun_out/pmc26
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc26/f -o p -- python3 bench.py --scale 26 --steps 2 --warmup 1 --cpu-seconds 0 --ref-scale 0 > /dev/null 2> gpurun_out/pmc26.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc26/t -o t -- python3 bench.py --scale 26 --steps 2 --warmup 1 --cpu-seconds 0 --ref-scale 0 > /dev/null 2>> gpurun_out/pmc26.err
python3 - <<'PY'
import csv, glob, collections
f=glob.glob("gpurun_out/pmc26/f/*counter_collection.csv")[0]
agg=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_tc_" in k and "stats" not in k and r["Counter_Name"]=="FETCH_SIZE":
        name=k.split("(")[0].split("::")[-1]
        agg[name]+=float(r["Counter_Value"]); n[name]+=1
tot=0
for k in agg: print(k, "FETCH_SIZE per dispatch KB", agg[k]/n[k], "n", n[k]); tot+=agg[k]/n[k]
print("total KB per pass", tot, "-> bytes x2:", tot*1024*2)
PY
grep "k_tc" gpurun_out/pmc26/t/*kernel_stats.csv | cut -c1-40,150-260
