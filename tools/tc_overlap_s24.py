"""GPU probe: serial / default / forced co-scheduling of the two triangle-count kernels.  usage: tc_overlap_s24.py [scale]"""
import os, sys
sys.path.insert(0, ".")
from gms_amd import capi
capi.init(0)
g = capi.DeviceGraph.from_csr(capi.HostCSR.generate("kronecker", int(sys.argv[1]) if len(sys.argv) > 1 else 24), flags=capi.UPLOAD_TRUSTED)
for ov, w in (("0", "2"), ("1", "2"), ("2", "1"), ("2", "2"), ("2", "4")):
    os.environ["GMSX_TC_OVERLAP"] = ov; os.environ["GMSX_TC_WAVE_WGS"] = w
    ms = []
    for _ in range(5):
        t, st = g.tc_total(stats=True); ms.append(st["kernel_ms"])
    print("overlap", ov, "wave_wgs", w, round(min(ms[1:]), 2), [round(x, 2) for x in ms], flush=True)
