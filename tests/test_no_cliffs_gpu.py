"""GPU: no request fails for its SIZE.  The bit-matrix kernels hold pivots up to d+ = 8192 (k <= 4) /
4096 (k >= 5) and k <= 10, the Bron–Kerbosch register-resident search kernels grow to eight words per lane (16384 candidates); wider
pivots and larger k fall back to the generic list recursion (kclique.hip, k_kc_generic), wider start vertices to the memory-resident
search (bk.hip, bk_search_mem), instead of GMSX_ERR_UNSUPPORTED.

Shapes: K_{a,b} plus a sparse random graph H inside side A, b = a + 1.  Every B vertex has all of A as its oriented row
(d+ = a), B is independent, so every clique has at most one B vertex:
    C_k(G) = b * C_{k-1}(H) + C_k(H),      maximal cliques(G) = b * MC(H)    (isolated vertices of H count)
with the H terms from the oracle on H alone."""
import math
import os

import numpy as np
import pytest

from conftest import host_graph

pytestmark = pytest.mark.gpu


def kab_plus_h(gpu, oracle, a, b, eh, seed):
    rng = np.random.default_rng(seed)
    hu, hv = rng.integers(0, a, eh), rng.integers(0, a, eh)
    keep = hu != hv
    hu, hv = hu[keep].astype(np.int32), hv[keep].astype(np.int32)
    hcsr = gpu.HostCSR.from_edges(hu, hv, num_nodes=a)
    bu, bv = np.meshgrid(np.arange(a, dtype=np.int32), np.arange(a, a + b, dtype=np.int32), indexing="ij")
    csr = gpu.HostCSR.from_edges(np.concatenate([bu.ravel(), hu]), np.concatenate([bv.ravel(), hv]))
    return csr, hcsr


def cliques_of(oracle, csr, k):
    if k == 1:
        return csr.num_nodes
    if k == 2:
        return csr.num_edges
    ordered = oracle.kclique(csr.offsets(), csr.neighbors(), k)
    assert ordered % math.factorial(k) == 0
    return ordered // math.factorial(k)


@pytest.mark.parametrize("a,ks", [(4200, (5, 6)), (8300, (3, 4))])
def test_kclique_beyond_the_bit_matrix_width(gpu, oracle, a, ks):
    b = a + 1
    csr, hcsr = kab_plus_h(gpu, oracle, a, b, 3 * a, seed=a)
    g = gpu.DeviceGraph.from_csr(csr)
    assert g.max_out_degree >= a
    for k in ks:
        want = b * cliques_of(oracle, hcsr, k - 1) + cliques_of(oracle, hcsr, k)
        ordered, cliques = g.kclique_count(k)
        assert cliques == want and ordered == (want * math.factorial(k)) & ((1 << 64) - 1), (a, k)
        assert sum(g.kclique_partial(k, p, 3) for p in range(3)) == want
    g.free()


_WANT = {}


def _want(oracle, csr, key, k):
    if (key, k) not in _WANT:
        _WANT[(key, k)] = oracle.kclique(csr.offsets(), csr.neighbors(), k)
    return _WANT[(key, k)]


@pytest.mark.parametrize("maxd", [1, 8, 40, 300])
def test_generic_path_equals_bit_matrix_path(gpu, oracle, maxd):
    """GMSX_KC_MAXD (test hook) lowers the width limit, so ordinary graphs go through the generic recursion: same counts."""
    gpu.set_option("KC_MAXD", str(maxd))
    try:
        for kind, scale, deg, ks in (("kronecker", 10, 16, (3, 4, 5, 6)), ("uniform", 10, 30, (3, 4)), ("kronecker", 12, 8, (3, 4, 5))):
            csr = host_graph(gpu, kind, scale, deg, True)
            g = gpu.DeviceGraph.from_csr(csr)
            for k in ks:
                assert g.kclique_count(k)[0] == _want(oracle, csr, (kind, scale, deg), k), (kind, scale, k, maxd)
            g.free()
    finally:
        gpu.reset_options()


def test_k_beyond_ten(gpu):
    """k = 11 … 14 (the unrolled bit-matrix recursions stop at 10): K_n has C(n, k) k-cliques; a dense random graph against
    a plain Python recursion on the same DAG orientation."""
    for n in (12, 16, 20):
        iu = np.triu_indices(n, 1)
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32)))
        for k in (11, 12, 14):
            assert g.kclique_count(k)[1] == math.comb(n, k), (n, k)
        g.free()
    rng = np.random.default_rng(5)
    n = 36
    adj = np.triu(rng.random((n, n)) < 0.85, 1)
    adj = adj | adj.T
    nbrs = [set(np.flatnonzero(adj[i])) for i in range(n)]

    def count(k, cand):
        if k == 0:
            return 1
        total = 0
        for v in sorted(cand):
            total += count(k - 1, {w for w in cand & nbrs[v] if w > v})
        return total
    iu = np.nonzero(np.triu(adj, 1))
    g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32), num_nodes=n))
    for k in (11, 13):
        assert g.kclique_count(k)[1] == count(k, set(range(n))), k
    with pytest.raises(gpu.GmsxError) as ei:
        g.kclique_count(65)
    assert ei.value.status == gpu.ERR_UNSUPPORTED  # the one remaining bound: per-level cursors of the generic recursion
    g.free()


def planted_cliques_graph(gpu, scale=14, sizes=(16, 17, 18), seed=5):
    """uniform G(n, m) background (avg degree 32) + cliques planted on random vertex sets; returns (HostCSR, adjacency as python int bitmasks
    over a DAG order)."""
    base = gpu.HostCSR.generate("uniform", scale)
    n = base.num_nodes
    off, ng = base.offsets(), base.neighbors()
    src = np.repeat(np.arange(n, dtype=np.int32), np.diff(off))
    rng = np.random.default_rng(seed)
    es, ed = [src[src < ng]], [ng[src < ng]]
    for sz in sizes:
        mem = rng.choice(n, size=sz, replace=False).astype(np.int32)
        iu = np.triu_indices(sz, 1)
        es.append(mem[iu[0]])
        ed.append(mem[iu[1]])
    return gpu.HostCSR.from_edges(np.concatenate(es), np.concatenate(ed), num_nodes=n)


def dag_clique_count(off, ng, k):
    """independent CPU count of k-cliques: degree-oriented DAG, python-int bitmasks, pruned recursion (each clique once)"""
    n = off.size - 1
    deg = np.diff(off)
    rank = np.empty(n, dtype=np.int64)
    rank[np.lexsort((np.arange(n), deg))] = np.arange(n)     # low degree first; edges point to higher rank
    out = [[int(v) for v in ng[off[u]:off[u + 1]] if rank[v] > rank[u]] for u in range(n)]
    total = 0
    for u in range(n):
        mem = out[u]
        if len(mem) < k - 1:
            continue
        idx = {v: i for i, v in enumerate(mem)}
        adj = [0] * len(mem)
        for v in mem:
            for w in out[v]:
                if w in idx:
                    adj[idx[v]] |= 1 << idx[w]
                    adj[idx[w]] |= 1 << idx[v]

        def rec(need, cand):
            if need == 0:
                return 1
            if bin(cand).count("1") < need:
                return 0
            t, c = 0, cand
            while c:
                low = c & -c
                i = low.bit_length() - 1
                c ^= low
                t += rec(need - 1, c & adj[i])   # members after i only: every clique once
            return t
        total += rec(k - 1, (1 << len(mem)) - 1)
    return total


@pytest.mark.parametrize("slab_mb", [None, 1])
def test_generic_path_k15_on_a_midsize_graph(gpu, slab_mb):
    """k >= 11 sends EVERY pivot with d+ >= k-1 through the generic list recursion, whose level lists live in one budgeted slab: pivots per
    chunk = budget / (levels * stride of the chunk's widest pivot) (fixed 60000-pivot chunks sized by the GLOBAL max d+ asked
    for tens of GB at k >= 12 on RMAT-26).  16 384 vertices, 262 k background edges, planted 16/17/18-cliques: k = 15 with the default
    budget and with a 1 MB one (GMSX_KC_SLAB_MB test hook: many chunks through one reused slab) against an independent CPU recursion.
    (The reference's own recursion enumerates k!·C ordered cliques — 15! per clique — and cannot produce this value in any time.)"""
    csr = planted_cliques_graph(gpu)
    want = dag_clique_count(csr.offsets(), csr.neighbors(), 15)
    assert want >= math.comb(16, 15) + math.comb(17, 15) + math.comb(18, 15)
    if slab_mb is not None:
        gpu.set_option("KC_SLAB_MB", str(slab_mb))
    try:
        g = gpu.DeviceGraph.from_csr(csr)
        ordered, cliques, st = g.kclique_count(15, stats=True)
        assert cliques == want and ordered == (want * math.factorial(15)) & ((1 << 64) - 1)
        if slab_mb is not None:
            assert st["launches"] >= 8                               # the tiny budget really cut the pivots into many chunks
        assert sum(g.kclique_partial(15, p, 3) for p in range(3)) == cliques
        g.free()
    finally:
        gpu.reset_options()


@pytest.mark.parametrize("a", [4200, 8200])
def test_bk_beyond_4096_candidates(gpu, oracle, a):
    """Start vertices with 4097..8192 / 8193..16384 candidates: four / eight words per lane in the search kernels."""
    b = a + 1
    csr, hcsr = kab_plus_h(gpu, oracle, a, b, 4 * a, seed=a + 1)
    mc_h = oracle.bk_count(hcsr.offsets(), hcsr.neighbors())
    g = gpu.DeviceGraph.from_csr(csr)
    assert g.max_out_degree >= a
    want = b * mc_h
    got, st = g.bk_count(stats=True)
    assert got == want
    assert sum(g.bk_partial(p, 2) for p in range(2)) == want
    g.free()


@pytest.mark.parametrize("maxc", [1, 20, 200])
def test_bk_memory_resident_search_equals_register_search(gpu, oracle, maxc):
    """GMSX_BK_MAXC (test hook) lowers the width above which a start vertex runs on the memory-resident search (k_bk_wave<false, 0>:
    P / Xc / ext of every level in the global slab) — production value 16384.  Same counts as the oracle and as the shard sums."""
    gpu.set_option("BK_MAXC", str(maxc))
    try:
        for kind, scale, deg in (("kronecker", 10, 16), ("uniform", 10, 30), ("kronecker", 11, 8)):
            csr = host_graph(gpu, kind, scale, deg, True)
            want = _WANT.setdefault(("bk", kind, scale, deg), oracle.bk_count(csr.offsets(), csr.neighbors()))
            g = gpu.DeviceGraph.from_csr(csr)
            assert g.bk_count() == want, (kind, scale, maxc)
            assert sum(g.bk_partial(p, 3) for p in range(3)) == want
            g.free()
        # K_n: one maximal clique, search depth n; a complete multipartite graph: 3^4 maximal cliques
        iu = np.triu_indices(300, 1)
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32)))
        assert g.bk_count() == 1
        g.free()
        parts = [range(3 * i, 3 * i + 3) for i in range(4)]
        edges = [(a, b) for i, pa in enumerate(parts) for pb in parts[i + 1:] for a in pa for b in pb]
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(np.array([e[0] for e in edges], np.int32), np.array([e[1] for e in edges], np.int32)))
        assert g.bk_count() == 81
        g.free()
    finally:
        gpu.reset_options()


@pytest.mark.parametrize("chunk", [1, 700, 50000])
def test_rows_sorted_in_ranges(gpu, oracle, chunk):
    """rocPRIM's segmented sort counts items in 32 bits, so the upload sorts the rows in vertex ranges of < 2^31 container entries
    (graphs beyond 2^32 entries used to keep unsorted rows and lose the delta forms and the k-clique kernels).  GMSX_SORT_CHUNK (test
    hook) forces many ranges on a small graph: every kernel that relies on ascending rows still agrees with the oracle."""
    gpu.set_option("SORT_CHUNK", str(chunk))
    try:
        for kind, scale, deg, hub_limit in (("kronecker", 12, 16, 0), ("uniform", 11, 40, 64)):
            csr = host_graph(gpu, kind, scale, deg, True)
            g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
            want = _WANT.setdefault(("tc", kind, scale, deg), oracle.tc_total(csr.offsets(), csr.neighbors()))
            t, st = g.tc_total(stats=True)
            assert t == want and st["units"] == csr.num_edges
            assert g.kclique_count(3)[1] == want
            assert g.kclique_count(4)[0] == _WANT.setdefault(("kc4", kind, scale, deg), oracle.kclique(csr.offsets(), csr.neighbors(), 4))
            assert g.bk_count() == _WANT.setdefault(("bk", kind, scale, deg), oracle.bk_count(csr.offsets(), csr.neighbors()))
            g.free()
    finally:
        gpu.reset_options()
