import json, os, sys
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
csr = capi.HostCSR.generate("kronecker", int(sys.argv[1]), 16)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
for ov, w in (("0", "2"), ("1", "1"), ("1", "2"), ("1", "3"), ("1", "4"), ("1", "6"), ("1", "8")):
    os.environ["GMSX_TC_OVERLAP"] = ov; os.environ["GMSX_TC_WAVE_WGS"] = w
    ms = []
    for _ in range(4):
        t, st = g.tc_total(stats=True); ms.append(st["kernel_ms"])
    print(ov, w, round(min(ms[1:]), 2), [round(x, 1) for x in ms], flush=True)
