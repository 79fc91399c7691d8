"""GPU probe: three serialised triangle-count passes (GMSX_TC_OVERLAP=0) on a cached RMAT graph, meant to run under rocprofv3
(tools/tc_serial_pmc_probe.sh); also used with one-off phase builds of k_tc_block.  usage: tc_phase_probe.py <scale>"""
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
