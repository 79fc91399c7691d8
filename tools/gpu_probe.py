"""Ad-hoc GPU probe used during development: parity of the HIP triangle count against the oracle at small
scales, timings at larger ones.  Usage: python tools/gpu_probe.py [max_scale]"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from gms_amd import capi
from oracle.bindings import Oracle

GOLD = {10: 74720, 12: 483489, 14: 2862425, 16: 15656287, 18: 82728031, 20: 423625371, 22: 2111140967, 24: 10283205554}
max_scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
capi.init(0)
print(capi.device_info())
O = Oracle()
for s in [4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24]:
    if s > max_scale: break
    t0 = time.time(); csr = capi.HostCSR.generate("kronecker", s); t1 = time.time()
    g = capi.DeviceGraph.from_csr(csr); t2 = time.time()
    res = []
    for it in range(3):
        t, st = g.tc_total(stats=True); res.append(st["kernel_ms"])
    exp = GOLD.get(s)
    if exp is None or s <= 12:
        exp2 = O.tc_total(csr.offsets(), csr.neighbors())
        assert exp is None or exp == exp2
        exp = exp2
    print(json.dumps({"scale": s, "n": csr.num_nodes, "m": csr.num_edges, "tri": t, "ok": t == exp, "gen_s": round(t1-t0,2),
        "upload_s": round(t2-t1,2), "kernel_ms": res, "max_dplus": g.max_out_degree, "probes": st["probes"], "units": st["units"],
        "alg_elems": st["alg_elements"], "Medges_s": round(csr.num_edges/ (min(res)/1e3)/1e6,1)}), flush=True)
    g.free(); del csr
