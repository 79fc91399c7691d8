"""GPU: gmsx_tc_comembership for an RMAT graph — what a pass of the hub items would stream if B consecutive heavy pivots (d+ order) were staged together
and every distinct stream row of the batch were loaded once ("stream each row ONCE against several staged pivots").
usage: tc_comembership.py [scale] [ef]   -> one JSON line per batch size"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", scale, ef, True)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED | capi.UPLOAD_FOR_TC)
total, st = g.tc_total(stats=True)
print(json.dumps({"scale": scale, "m": csr.num_edges, "triangles": total, "kernel_ms": round(st["kernel_ms"], 2), "stream_bytes": st.get("stream_bytes")}), flush=True)
for b in (1, 2, 4, 8, 16, 64, 256, 4096, 1 << 30):
    r = g.tc_comembership(b)
    r.update(batch=b, GB_today=round(r["units"] * 16 / 1e9, 2), GB_batched=round(r["batched_units"] * 16 / 1e9, 2),
             saved_pct=round(100.0 * (r["units"] - r["batched_units"]) / max(r["units"], 1), 2))
    print(json.dumps(r), flush=True)
