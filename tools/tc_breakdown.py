"""GPU probe: where the algorithmic stream bytes of one triangle-count pass go (gmsx_tc_stream_breakdown).  usage: tc_breakdown.py [scale]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max": capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", scale, 16)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
tri, st = g.tc_total(stats=True)
b = g.tc_stream_breakdown()
total = sum(b[k] for k in capi.DeviceGraph.BREAKDOWN_BYTES)
print(json.dumps({"scale": scale, "triangles": tri, "kernel_ms": st["kernel_ms"], "stream_bytes": st["stream_bytes"], "sum_of_categories": total}))
for k, v in b.items():
    if k.startswith("reserved"):
        continue
    print("%-40s %12.3f GB" % (k, v / 1e9) if not k.startswith("count_") else "%-40s %12.3f M" % (k, v / 1e6))
