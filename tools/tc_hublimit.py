"""Sweep of the hub boundary (upload test hook GMSX_UPLOAD_HUB_LIMIT) on one graph: kernel time of the triangle count."""
import sys, json
sys.path.insert(0, ".")
from gms_amd import capi
s = int(sys.argv[1])
capi.init(0)
csr = capi.HostCSR.generate("kronecker", s)
for hl in [int(x) for x in sys.argv[2:]]:
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED | (hl << 8))
    for _ in range(2):
        t, st = g.tc_total(stats=True)
    print(json.dumps({"scale": s, "hub_limit": hl or 65535, "tri": t, "kernel_ms": round(st["kernel_ms"], 2), "device_GB": round(g.device_bytes / 1e9, 2)}), flush=True)
    g.free()
