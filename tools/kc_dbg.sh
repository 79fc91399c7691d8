for d in 0 1 2 3; do GMSX_DBG=$d bash tools/profile_cmd.sh kcd$d tools/kc_one.py 24 4 > /dev/null 2>&1; done
python3 - <<'PY'
import csv
for d in range(4):
    rows=list(csv.DictReader(open(f"gpurun_out/prof_kcd{d}/trace/trace_kernel_trace.csv")))
    out=[]
    for r in rows:
        if "k_kc" in r["Kernel_Name"]:
            out.append("%.1f" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
    print(d, " ".join(out[:11]))
PY
