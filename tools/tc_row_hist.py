"""GPU: prints gmsx_tc_row_histogram for an RMAT graph — which row lengths the heavy pivots stream and what the light pivots' gathers cost.
usage: python tools/tc_row_hist.py <scale>"""
import os, sys, json
os.environ.setdefault("GMSX_OPT_TC_KEEP_ROWS", "1")  # the what-if estimates at the end read the per-vertex row descriptors
sys.path.insert(0, ".")
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", scale), flags=capi.UPLOAD_TRUSTED)
hist, light = g.tc_row_histogram()
names = ["hub list", "hub bitset", "hub delta", "tail list", "tail delta"]
bins = [str(i) for i in range(1, 17)] + ["17-32", "33-64", "65-128", "129-256", "257-512", "513-1024", "1025-2048", "2049+"]
tot_rows, tot_units, tot_slots = 0, 0, 0
for c, name in enumerate(names):
    rows, units = int(hist[c, :, 0].sum()), int(hist[c, :, 1].sum())
    # lane slots a 16-lane group spends on a row of n units: single step for the remainder, double steps of 32 before it
    slots = 0
    print(f"== {name}: rows {rows/1e6:.1f} M, units {units/1e9:.3f} G ({units*16/1e9:.1f} GB)")
    for b, lab in enumerate(bins):
        r, u = int(hist[c, b, 0]), int(hist[c, b, 1])
        if r == 0:
            continue
        avg = u / r
        steps = -(-avg // 16)                       # 16-unit steps per row (approximation inside a bin)
        slots += r * steps * 16
        print(f"   {lab:>9s} units: rows {r/1e6:9.2f} M  units {u/1e9:8.3f} G  ({100*u/max(units,1):5.1f} % of the class)")
    print(f"   lane-slot utilisation of the class ≈ {units/max(slots,1):.3f}")
    tot_rows += rows; tot_units += units; tot_slots += slots
print(f"all classes: rows {tot_rows/1e6:.1f} M, units {tot_units/1e9:.2f} G, lane-slot utilisation ≈ {tot_units/max(tot_slots,1):.3f}")
print(f"entries {int(light[0])/1e6:.1f} M (inline {int(light[1])/1e6:.2f} M) in {int(light[2])/1e6:.2f} M work items; pivots' own containers {int(light[3])/1e9:.2f} GB")
for k, lab in ((8, "heavy pivots"), (10, "light pivots")):
    cur, best = int(light[k]), int(light[k + 1])
    print(f"{lab}: streaming the member's rows for every oriented edge = {cur*16/1e9:.1f} GB; streaming the smaller endpoint's rows = {best*16/1e9:.1f} GB")
print(f"heavy pivots, handed-over rows cut at the receiving pivot's id (estimate): {int(light[12])*16/1e9:.1f} GB")
print(json.dumps({"scale": scale, "hist": hist.tolist(), "light": light.tolist()}))
