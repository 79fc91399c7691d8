"""GPU probe: gmsx_set_op_batch on every edge (u < v) of an RMAT graph — materialised N(u) ∩ N(v) and N(u) \\ N(v): pairs/s, ids/s, kernel time.
usage: python tools/setop_probe.py [scale]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
capi.init(0)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        capi.set_host_threads(max(1, int(int(q) / int(p) + 0.999)))
except (OSError, ValueError):
    pass
csr = capi.HostCSR.generate("kronecker", scale, 16, True)
off, ng = csr.offsets(), csr.neighbors()
u = np.repeat(np.arange(csr.num_nodes, dtype=np.int32), np.diff(off))
keep = u < ng
u, v = u[keep], ng[keep].astype(np.int32)
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED)
cnt = g.intersect_count_batch(u, v)
for op in ("intersect", "difference"):
    t0 = time.perf_counter()
    o, ids, st = g.set_op_batch(op, u, v, stats=True)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(np.diff(o), cnt if op == "intersect" else np.diff(off)[u] - cnt))
    print(json.dumps({"scale": scale, "op": op, "pairs": int(u.size), "ids_out": int(ids.size), "sizes_ok": ok, "triangles_x3": int(cnt.sum()) if op == "intersect" else None,
                      "kernel_ms_fill_call": round(st["kernel_ms"], 2), "wall_s_sizing_plus_fill_incl_copies": round(dt, 3),
                      "pairs_per_s_kernel": round(u.size / (st["kernel_ms"] * 1e-3)), "ids_per_s_kernel": round(ids.size / (st["kernel_ms"] * 1e-3))}), flush=True)
