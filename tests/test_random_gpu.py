"""GPU: randomised sweep — every device entry point against the oracle on many small random graphs of varied shape
(Erdős–Rényi of several densities, bipartite, stars+cliques, isolated vertices), with the hub-limit test hook varied
so that both container kinds and the bitset rows are exercised.  Bit-exact."""
import numpy as np
import pytest

from conftest import edges_to_csr

pytestmark = pytest.mark.gpu


def _graphs(rng):
    for n, p in [(1, 0.0), (2, 1.0), (7, 0.5), (33, 0.3), (64, 0.9), (65, 0.1), (130, 0.5), (257, 0.08), (300, 0.3), (90, 1.0)]:
        iu = np.triu_indices(n, 1)
        keep = rng.random(iu[0].size) < p
        yield f"gnp-{n}-{p}", n, np.stack([iu[0][keep], iu[1][keep]], 1)
    a, b = 40, 55  # complete bipartite: no triangles, a*b maximal cliques
    yield "bipartite", a + b, np.array([(i, a + j) for i in range(a) for j in range(b)])
    # a hub joined to three cliques plus isolated vertices and a pendant path
    edges = []
    base = 1
    for size in (5, 12, 34):
        ids = list(range(base, base + size))
        edges += [(x, y) for i, x in enumerate(ids) for y in ids[i + 1:]] + [(0, x) for x in ids]
        base += size
    edges += [(base, base + 1), (base + 1, base + 2)]
    yield "hub-cliques", base + 10, np.array(edges)
    # power-law-ish: preferential attachment
    n = 400
    e = [(0, 1)]
    deg = np.zeros(n)
    deg[:2] = 1
    for v in range(2, n):
        targets = rng.choice(v, size=min(v, 6), replace=False, p=deg[:v] / deg[:v].sum())
        for t in targets:
            e.append((int(t), v))
            deg[t] += 1
            deg[v] += 1
    yield "pref-attach", n, np.array(e)


def test_all_entry_points_on_random_graphs(gpu, oracle):
    rng = np.random.default_rng(2026)
    checked = 0
    for name, n, edges in _graphs(rng):
        csr = edges_to_csr(gpu, edges.reshape(-1, 2), n=n)
        off, ng = csr.offsets(), csr.neighbors()
        want_t, raw = oracle.tc_total(off, ng, raw=True)
        want_v = oracle.tc_vertex_count2(off, ng)
        want_k = {k: oracle.kclique(off, ng, k) for k in (3, 4, 5)}
        want_bk = oracle.bk_count(off, ng)
        for hub_limit in (0, 1, 9, 70):  # 0 = production value
            g = gpu.DeviceGraph.from_csr(csr, flags=(hub_limit << 8))
            assert g.tc_total() == want_t, (name, hub_limit)
            assert g.tc_total(gpu.TC_FULL) == want_t
            assert sum(g.tc_partial(p, 3) for p in range(3)) == want_t
            assert np.array_equal(g.tc_vertex_count2(), want_v), (name, hub_limit)
            for k in (3, 4, 5):
                assert g.kclique_count(k)[0] == want_k[k], (name, hub_limit, k)
            assert g.bk_count() == want_bk, (name, hub_limit)
            if n > 1:
                u = rng.integers(0, n, 64).astype(np.int32)
                v = rng.integers(0, n, 64).astype(np.int32)
                want = np.array([oracle.intersect_count(ng[off[a]:off[a + 1]], ng[off[b]:off[b + 1]]) for a, b in zip(u, v)], dtype=np.uint32)
                assert np.array_equal(g.intersect_count_batch(u, v), want)
            g.free()
            checked += 1
    assert checked >= 50
