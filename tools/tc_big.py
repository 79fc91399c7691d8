import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
s = int(sys.argv[1])
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", s), flags=capi.UPLOAD_TRUSTED)
for _ in range(2):
    t, st = g.tc_total(stats=True)
print(json.dumps({"scale": s, "tri": t, "kernel_ms": st["kernel_ms"]}))
