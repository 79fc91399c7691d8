"""Cross-check the independent device paths against each other on large graphs of other shapes (no CPU oracle at
these sizes): oriented TC vs sharded partials vs k-clique(k=3) vs the full-row formulation vs Σ vertex counts / 6."""
import sys, json
sys.path.insert(0, ".")
import numpy as np
from gms_amd import capi
capi.init(0)


def check(name, csr, full=True):
    g = capi.DeviceGraph.from_csr(csr)
    t = g.tc_total()
    rec = {"graph": name, "n": csr.num_nodes, "m": csr.num_edges, "triangles": t, "max_dplus": g.max_out_degree,
           "partials": sum(g.tc_partial(p, 5) for p in range(5)) == t, "kclique3": g.kclique_count(3)[1] == t}
    if full:
        rec["full"] = g.tc_total(capi.TC_FULL) == t
        rec["vertex_sum"] = int(g.tc_vertex_count2().sum()) == 6 * t
    print(json.dumps(rec), flush=True)
    assert all(v for k, v in rec.items() if isinstance(v, bool)), rec
    g.free()


check("uniform-20-32", capi.HostCSR.generate("uniform", 20, 32))
check("rmat-20-16 A=.45", capi.HostCSR.generate_rmat(20, 16, 0.45, 0.22, 0.22))
check("rmat-19-64 A=.65 (very skewed)", capi.HostCSR.generate_rmat(19, 64, 0.65, 0.15, 0.15))
# 2-D grid 1200 x 1200 with diagonals in every other cell: known triangle count
k = 1200
idx = np.arange(k * k, dtype=np.int64).reshape(k, k)
e = [np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1), np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()], 1)]
diag = np.stack([idx[:-1:2, :-1].ravel(), idx[1::2, 1:].ravel()], 1)
e.append(diag)
e = np.concatenate(e).astype(np.int32)
csr = capi.HostCSR.from_edges(e[:, 0], e[:, 1])
g = capi.DeviceGraph.from_csr(csr)
assert g.tc_total() == 2 * diag.shape[0] == g.kclique_count(3)[1], (g.tc_total(), 2 * diag.shape[0])
print(json.dumps({"graph": "grid+diagonals", "n": csr.num_nodes, "triangles": g.tc_total(), "expected": 2 * int(diag.shape[0]), "bk": g.bk_count()}), flush=True)
g.free()
# many stars sharing leaves (n > 65535, tail containers carry real rows), plus one clique among 300 leaves
rng = np.random.default_rng(1)
n = 200000
hubs = rng.integers(0, n, (2_000_000, 1))
leaves = rng.integers(0, n, (2_000_000, 1))
cl = np.arange(150000, 150300)
ce = np.array([(a, b) for i, a in enumerate(cl) for b in cl[i + 1:]])
ee = np.concatenate([np.concatenate([hubs % 500, leaves], 1), ce]).astype(np.int32)
check("stars+clique n=200k", capi.HostCSR.from_edges(ee[:, 0], ee[:, 1], num_nodes=n))
print("crosscheck ok")
