import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
for s in (16, 18, 20):
    g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", s), flags=capi.UPLOAD_TRUSTED)
    row = {"scale": s}
    for k in (3, 4, 5, 6):
        o, c, st = g.kclique_count(k, stats=True)
        o, c, st = g.kclique_count(k, stats=True)
        row[f"k{k}"] = {"cliques": c, "ms": round(st["kernel_ms"], 2)}
    print(json.dumps(row), flush=True)
    g.free()
