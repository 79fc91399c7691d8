"""RMAT scale 26 (the BASELINE.json target size): generate on the host, count on one MI355X, cross-check the triangle
count three independent ways (no reference golden exists at this size): bitmap kernels, 8-way sharded partials, and the
k-clique kernels at k=3 (hash-map build + popcount)."""
import sys, time, json
sys.path.insert(0, ".")
from gms_amd import capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
capi.init(0)
t0 = time.time(); csr = capi.HostCSR.generate("kronecker", scale); t1 = time.time()
print(json.dumps({"scale": scale, "n": csr.num_nodes, "m": csr.num_edges, "gen_s": round(t1 - t0, 1)}), flush=True)
elems = csr.merge_elements()
g = capi.DeviceGraph.from_csr(csr, flags=capi.UPLOAD_TRUSTED); t2 = time.time()
ms = []
for _ in range(3):
    tri, st = g.tc_total(stats=True); ms.append(round(st["kernel_ms"], 2))
parts = sum(g.tc_partial(p, 8) for p in range(8))
o3, c3, st3 = g.kclique_count(3, stats=True)
b_alg = 4 * elems + 8 * (csr.num_nodes + 1) + 4 * csr.nnz
print(json.dumps({"scale": scale, "triangles": tri, "partials_ok": parts == tri, "kclique3_ok": c3 == tri, "kernel_ms": ms, "upload_s": round(t2 - t1, 2),
                  "merge_elements": elems, "B_alg_TB": round(b_alg / 1e12, 2), "Gedges_s": round(csr.num_edges / (min(ms) / 1e3) / 1e9, 2),
                  "max_dplus": g.max_out_degree, "device_GB": round(g.device_bytes / 1e9, 2), "probes": st["probes"], "kc3_ms": round(st3["kernel_ms"], 1)}), flush=True)
# k = 4 at the target size (BASELINE.json north_star: "bit-exact triangle and k-clique counts on RMAT scale-26"): no oracle runs
# at this size, so the count is checked through its shards (8 partial counts, disjoint pivots) and its divisibility by 4!
t = time.time(); o4, c4, st4 = g.kclique_count(4, stats=True); t4 = time.time() - t
p4 = sum(g.kclique_partial(4, p, 8) for p in range(8))
print(json.dumps({"scale": scale, "k": 4, "ordered": o4, "cliques": c4, "partials_ok": p4 == c4, "divisible": o4 % 24 == 0, "kernel_ms": round(st4["kernel_ms"], 1), "wall_s": round(t4, 2)}), flush=True)
