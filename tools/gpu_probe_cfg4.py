"""Config-4 calibration: BK on lower-skew RMAT (A=.45,B=C=.22) of growing size + CPU oracle cross-check at small scale."""
import sys, time, json
sys.path.insert(0, ".")
from gms_amd import capi
from oracle.bindings import Oracle
capi.init(0)
O = Oracle()
for scale, deg in [(12, 38), (14, 38), (16, 38), (18, 38), (20, 28), (21, 56), (22, 28)]:
    if len(sys.argv) > 1 and scale > int(sys.argv[1]): break
    t0 = time.time(); csr = capi.HostCSR.generate_rmat(scale, deg, 0.45, 0.22, 0.22); t1 = time.time()
    g = capi.DeviceGraph.from_csr(csr)
    c, st = g.bk_count(stats=True)
    rec = {"scale": scale, "ef": deg, "n": csr.num_nodes, "m": csr.num_edges, "bk": c, "kernel_ms": round(st["kernel_ms"], 1), "rounds": st["probes"], "launches": st["launches"], "gen_s": round(t1 - t0, 1), "max_dplus": g.max_out_degree}
    if scale <= 14:
        t0 = time.time(); want = O.bk_count(csr.offsets(), csr.neighbors()); rec["oracle_ok"] = (want == c); rec["oracle_s"] = round(time.time() - t0, 2); rec["oracle_threads"] = O.max_threads()
    tri, st2 = g.tc_total(stats=True); rec["tri"] = tri; rec["tc_ms"] = round(st2["kernel_ms"], 2)
    print(json.dumps(rec), flush=True)
    g.free(); del csr
