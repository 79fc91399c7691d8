import sys, time, json
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
for s in (16, 18, 20, 22):
    if len(sys.argv) > 1 and s > int(sys.argv[1]): break
    csr = capi.HostCSR.generate("kronecker", s)
    g = capi.DeviceGraph.from_csr(csr)
    t, st = g.tc_total(capi.TC_FULL, stats=True)
    c, st2 = g.tc_vertex_count2(stats=True)
    print(json.dumps({"scale": s, "tri": t, "full_ms": round(st["kernel_ms"], 2), "vertex_ms": round(st2["kernel_ms"], 2), "sum_ok": int(c.sum()) == 6 * t}), flush=True)
    g.free()
