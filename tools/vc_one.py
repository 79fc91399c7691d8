import sys, json, time
sys.path.insert(0, ".")
from gms_amd import capi
s = int(sys.argv[1])
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", s), flags=capi.UPLOAD_TRUSTED)
tri = g.tc_total()
for _ in range(2):
    t0 = time.perf_counter(); c = g.tc_vertex_count2(); wall = time.perf_counter() - t0
print(json.dumps({"scale": s, "sum_over_6": int(c.sum()) // 6, "triangles": tri, "ok": int(c.sum()) == 6 * tri, "wall_ms": round(1e3 * wall, 1), "max": int(c.max())}))
