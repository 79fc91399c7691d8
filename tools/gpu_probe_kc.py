"""Ad-hoc: time the k-clique path. usage: python tools/gpu_probe_kc.py [max_scale] [k]"""
import sys, time, json
sys.path.insert(0, ".")
from gms_amd import capi
GOLD4 = {10: 9831960, 12: 96513528, 14: 877984296, 16: 6993215424}
max_scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
capi.init(0)
for s in [10, 12, 14, 16, 18, 20, 22]:
    if s > max_scale: break
    csr = capi.HostCSR.generate("kronecker", s)
    g = capi.DeviceGraph.from_csr(csr)
    ms = []
    for it in range(2):
        o, c, st = g.kclique_count(k, stats=True); ms.append(round(st["kernel_ms"], 3))
    print(json.dumps({"scale": s, "k": k, "ordered": o, "cliques": c, "ok": (GOLD4.get(s) == o) if k == 4 and s in GOLD4 else None,
                      "kernel_ms": ms, "launches": st["launches"]}), flush=True)
    g.free(); del csr
