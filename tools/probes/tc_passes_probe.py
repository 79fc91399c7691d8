import os, sys, json
sys.path.insert(0, os.getcwd())
from gms_amd import capi
capi.init(0)
csr = capi.HostCSR.generate("kronecker", 18)
g = capi.DeviceGraph.from_csr(csr)
base = g.device_bytes
g.free()
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_FOR_TC)
full = g.device_bytes
g.free()
out = {"base_MB": base / 2**20, "tc_MB": (full - base) / 2**20}
for lim in (96, 80, 72, 64, 56, 48, 40, 32, 24, 16):
    capi.set_option("TC_MEM_LIMIT_MB", lim)
    g = capi.DeviceGraph.from_csr(csr)
    try:
        g.tc_total()
        out[lim] = g.tc_passes
    except capi.GmsxError as e:
        out[lim] = "err %d" % e.status
    g.free()
print(json.dumps(out))
