"""GPU probe: the phases of bench.py's upload (validated base upload + triangle-count containers) — run with GMSX_OPT_TIMING=1.
usage: GMSX_OPT_TIMING=1 python tools/probes/upload_phases.py [scale]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gms_amd import capi  # noqa: E402

capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", int(sys.argv[1]) if len(sys.argv) > 1 else 26)
for rep in range(2):
    t0 = time.perf_counter()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_DEFAULT)
    t1 = time.perf_counter()
    g.prepare(capi.PREPARE_TC)
    t2 = time.perf_counter()
    print({"rep": rep, "upload_base": round(t1 - t0, 3), "build_tc": round(t2 - t1, 3)}, file=sys.stderr, flush=True)
    g.free()
