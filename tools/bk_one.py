import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
scale, ef = int(sys.argv[1]), int(sys.argv[2])
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate_rmat(scale, ef, 0.45, 0.22, 0.22), flags=capi.UPLOAD_TRUSTED)
c, st = g.bk_count(stats=True)
print(json.dumps({"bk": c, "kernel_ms": st["kernel_ms"], "rounds": st["probes"], "launches": st["launches"]}))
