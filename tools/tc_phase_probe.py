"""GPU probe for phase builds of k_tc_block (-DGMSX_TC_PHASES=1/2/4: only the hub members / the hub parts of tail members / the tail
parts are scanned — the COUNT IS WRONG by construction, only time and counters mean anything).  usage: tc_phase_probe.py <scale>"""
import os, sys, json
sys.path.insert(0, ".")
from gms_amd import capi
scale = int(sys.argv[1])
capi.init(0)
path = f"/tmp/phase_probe_{scale}.sg"
if os.path.exists(path):
    csr = capi.HostCSR.load(path)
else:
    csr = capi.HostCSR.generate("kronecker", scale)
    csr.save_sg(path)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
os.environ["GMSX_TC_OVERLAP"] = "0"
ms = []
for _ in range(3):
    t, st = g.tc_total(stats=True)
    ms.append(round(st["kernel_ms"], 1))
print(json.dumps({"count": t, "ms": ms}))
