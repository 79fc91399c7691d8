"""Ad-hoc: time the BK path. usage: python tools/gpu_probe_bk.py [max_scale]"""
import sys, time, json
sys.path.insert(0, ".")
from gms_amd import capi
GOLD = {10: 25467, 12: 692903, 14: 120747027}
max_scale = int(sys.argv[1]) if len(sys.argv) > 1 else 12
capi.init(0)
for kind, s in [("kronecker", 10), ("kronecker", 12), ("uniform", 16), ("uniform", 20), ("kronecker", 13), ("kronecker", 14)]:
    if s > max_scale and kind == "kronecker": break
    csr = capi.HostCSR.generate(kind, s)
    g = capi.DeviceGraph.from_csr(csr)
    t0 = time.time(); c, st = g.bk_count(stats=True); t1 = time.time()
    print(json.dumps({"kind": kind, "scale": s, "bk": c, "ok": GOLD.get(s) == c if kind == "kronecker" and s in GOLD else None,
                      "kernel_ms": round(st["kernel_ms"], 2), "setup_ms": round(st["setup_ms"], 2), "wall_s": round(t1 - t0, 3), "launches": st["launches"]}), flush=True)
    g.free(); del csr
