"""GPU probe: triangle-count pass time vs the stream-row forms (GMSX_TC_DELTA = 0 lists+bitsets, 1 delta when smaller, 2 delta wherever
possible).  usage: tc_delta_sweep.py [scale] [modes...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
modes = sys.argv[2:] or ["0", "1", "2"]   # "1:95" = mode 1 with GMSX_TC_DELTA_PCT=95; "1::0" = hub delta on, tail delta off
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max": capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", scale, 16)
for mode in modes:
    os.environ["GMSX_TC_DELTA"] = mode.split(":")[0]
    os.environ["GMSX_TC_DELTA_PCT"] = mode.split(":")[1] if ":" in mode and mode.split(":")[1] else "85"
    os.environ["GMSX_TC_TAIL_DELTA"] = mode.split(":")[2] if mode.count(":") >= 2 else "1"
    t0 = time.time()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
    up = time.time() - t0
    ms = []
    for _ in range(4):
        tri, st = g.tc_total(stats=True)
        ms.append(st["kernel_ms"])
    print(json.dumps({"delta": mode, "triangles": tri, "kernel_ms": round(min(ms[1:]), 2), "all": [round(x, 1) for x in ms], "upload_s": round(up, 2),
                      "device_GB": round(g.device_bytes / 1e9, 2), "alg_GB": round(st["stream_bytes"] / 1e9, 1)}), flush=True)
    g.free()
