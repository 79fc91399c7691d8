"""GPU probe: triangle-count time at RMAT scale S as a function of the bitset limit R (rows of rank id < R get a bitset container;
light pivots resolve those members by inverted gathers instead of streaming their rows).  usage: tc_bitset_sweep.py [scale] [R ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
limits = [int(x) for x in sys.argv[2:]] or [65535, 131072, 262144, 524288, 1048576]
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max": capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
t0 = time.time()
csr = capi.HostCSR.generate("kronecker", scale, 16)
print(json.dumps({"scale": scale, "m": csr.num_edges, "gen_s": round(time.time() - t0, 1)}), flush=True)
for R in limits:
    os.environ["GMSX_INLINE_LIMIT"] = str(R)  # GMSX_BITSET_LIMIT until the inline rows replaced the near-tail bitsets
    t0 = time.time()
    g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
    up = time.time() - t0
    ms = []
    for _ in range(4):
        tri, st = g.tc_total(stats=True)
        ms.append(st["kernel_ms"])
    print(json.dumps({"R": R, "triangles": tri, "kernel_ms": round(min(ms[1:]), 2), "upload_s": round(up, 2), "device_GB": round(g.device_bytes / 1e9, 2),
                      "alg_GB": round(st["stream_bytes"] / 1e9, 1)}), flush=True)
    g.free()
