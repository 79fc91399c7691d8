"""GPU probe: upload phases of a validated (GMSX_UPLOAD_DEFAULT) upload — run with GMSX_OPT_TIMING=1 for the per-phase lines.
usage: GMSX_OPT_TIMING=1 python tools/upload_probe.py 24 [26]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi  # noqa: E402

capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
for s in (int(a) for a in sys.argv[1:]):
    csr = capi.HostCSR.generate("kronecker", s)
    for flags, name in ((capi.UPLOAD_DEFAULT, "validated"), (capi.UPLOAD_TRUSTED, "trusted")):
        t0 = time.perf_counter()
        g = capi.DeviceGraph.from_csr(csr, flags=flags)
        t = time.perf_counter() - t0
        print({"scale": s, "upload": name, "seconds": round(t, 3)}, flush=True)
        g.free()
