import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", int(sys.argv[1])), flags=capi.UPLOAD_TRUSTED)
c, st = g.bk_count(stats=True)
print(json.dumps({"bk": c, "kernel_ms": round(st["kernel_ms"], 1), "rounds": st["probes"]}))
