import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
s, k = int(sys.argv[1]), int(sys.argv[2])
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", s))
for _ in range(2):
    o, c, st = g.kclique_count(k, stats=True)
print(json.dumps({"scale": s, "k": k, "ordered": o, "kernel_ms": st["kernel_ms"]}))
