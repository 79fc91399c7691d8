# per-launch durations of the k-clique kernels (rocprofv3 kernel trace): bash tools/kc_dbg.sh <scale> <k>
S=${1:-24}; K=${2:-4}
bash tools/profile_cmd.sh kcd tools/kc_one.py $S $K > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_kcd/trace/trace_kernel_trace.csv")))
out=[]
for r in rows:
    if "k_kc" in r["Kernel_Name"]:
        out.append("%s:%.1f" % ("S" if "small" in r["Kernel_Name"] else ("L" if "true" in r["Kernel_Name"] else "M"), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
print(" ".join(out[:len(out)//2]))
PY
