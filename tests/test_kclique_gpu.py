"""GPU parity tests of the HIP k-clique path (gmsx_kclique_count = the reference's CliqueCount, k!*C_k) against the
reference goldens, the oracle, and the known-answer cases of the reference's own clique-counting tests."""
import math

import numpy as np
import pytest

from conftest import edges_to_csr, host_graph, load_golden

pytestmark = pytest.mark.gpu
GRAPHS = load_golden("graphs.json")
U64 = (1 << 64) - 1


@pytest.mark.parametrize("key", [k for k, v in GRAPHS.items() if any(f"kc{i}" in v for i in (3, 4, 5))])
def test_kclique_equals_reference_golden(gpu, key):
    rec = GRAPHS[key]
    g = gpu.DeviceGraph.from_csr(host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"]))
    for k in (3, 4, 5):
        if f"kc{k}" in rec:
            ordered, cliques, st = g.kclique_count(k, stats=True)
            assert ordered == rec[f"kc{k}"] and cliques * math.factorial(k) == ordered
            assert st["kernel_ms"] > 0
    if "triangles" in rec:
        assert g.kclique_count(3)[1] == rec["triangles"] == g.tc_total()
    assert g.kclique_count(2) == (2 * rec["m"], rec["m"])
    g.free()


@pytest.mark.parametrize("spec", [("kronecker", 9, 3), ("kronecker", 10, 30), ("uniform", 10, 40), ("kronecker", 12, 8)])
def test_kclique_vs_oracle_k3_to_k7(gpu, oracle, spec):
    kind, scale, deg = spec
    csr = host_graph(gpu, kind, scale, deg, True)
    g = gpu.DeviceGraph.from_csr(csr)
    for k in (3, 4, 5, 6, 7):
        if (k >= 6 and (scale, deg) == (10, 30)) or (k >= 7 and (scale, deg) == (12, 8)):
            continue  # the oracle's k! * C recursion takes minutes there
        want = oracle.kclique(csr.offsets(), csr.neighbors(), k)
        assert g.kclique_count(k)[0] == want, (spec, k)
    g.free()


def test_known_answers_and_reference_test_graphs(gpu):
    ka = load_golden("known_answers.json")
    for c in ka["kclique"]:  # testing/clique_counting/CliqueCounter2_tests.h:45-271
        g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, c["edges"]))
        assert g.kclique_count(c["k"]) == (c["ordered"], c["cliques"]), c["name"]
        g.free()
    for c in ka["bk_random"]:
        g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, c["edges"], n=c["n"]))
        assert g.kclique_count(4)[0] == c["kc4"]
        g.free()
    import os
    from conftest import GOLDEN
    for name, rec in load_golden("testgraphs.json").items():
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name)))
        assert g.kclique_count(3)[0] == rec["kc3"] and g.kclique_count(4)[0] == rec["kc4"]
        g.free()


def test_partials_sum_to_total(gpu):
    g = gpu.DeviceGraph.from_csr(host_graph(gpu, "kronecker", 13, 16, True))
    for k in (3, 4, 5):
        total = g.kclique_count(k)[1]
        for nparts in (2, 3, 8):
            assert sum(g.kclique_partial(k, p, nparts) for p in range(nparts)) == total
    g.free()


def test_complete_graphs_exercise_every_bin(gpu):
    # K_n: C_k = C(n,k); d+ runs 0..n-1, so n = 40 / 300 / 1100 / 2300 cover the wave kernel, the LDS bit-matrix bins,
    # the global-slab kernel and its two-words-per-lane variant.  k! * C_k wraps mod 2^64 like the reference's size_t.
    # (round 6: for k = 4 the LDS matrices of 512 < d+ <= 1472 are triangular — n = 1100, 1600, 2300 put a pivot on every width of both triangular bins
    #  and of the slab bins behind them; KC_TRI = 0 keeps the rectangular / slab split of rounds 1-5 covered)
    for n, ks in [(5, (3, 4, 5, 6)), (40, (3, 4, 5, 8, 10)), (300, (3, 4, 5)), (1100, (3, 4)), (1600, (4,)), (2300, (3, 4))]:
        iu = np.triu_indices(n, 1)
        csr = gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32))
        for tri in ((None, 0) if n >= 1100 else (None,)):
            gpu.set_option("KC_TRI", tri)
            try:
                g = gpu.DeviceGraph.from_csr(csr)
                for k in ks:
                    ordered, cliques = g.kclique_count(k)
                    assert cliques == math.comb(n, k), (n, k, tri)
                    assert ordered == (math.comb(n, k) * math.factorial(k)) & U64
                if n >= 1100:
                    assert sum(g.kclique_partial(4, q, 3) for q in range(3)) == math.comb(n, 4)
                g.free()
            finally:
                gpu.set_option("KC_TRI", None)


def test_edge_cases_and_errors(gpu):
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, [(0, 1), (1, 2)]))
    assert g.kclique_count(3) == (0, 0) and g.kclique_count(2) == (4, 2) and g.kclique_count(10) == (0, 0)
    with pytest.raises(gpu.GmsxError) as ei:
        g.kclique_count(1)
    assert ei.value.status == gpu.ERR_INVALID
    assert g.kclique_count(11) == (0, 0)  # k > 10 runs on the generic list recursion (tests/test_no_cliffs_gpu.py)
    with pytest.raises(gpu.GmsxError) as ei:
        g.kclique_count(65)
    assert ei.value.status == gpu.ERR_UNSUPPORTED
    g.free()
    e = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, []))
    assert e.kclique_count(4) == (0, 0)
    e.free()


def test_dense_random_graph_large_out_degrees(gpu):
    """G(n, p) with p = 0.85: out-degrees up to ~n, so pivots fall into the widest LDS bins and the global-slab kernel with
    rows that are NOT all-ones (complete graphs cannot catch a wrong bit position).  k = 3 must equal the triangle
    kernels (an independent code path); k = 4 is checked against a dense-matrix count: a 4-clique is its smallest vertex a
    plus a triangle among the later neighbours of a (trace-style count on the 0/1 sub-matrix)."""
    n, p = 1500, 0.85
    rng = np.random.default_rng(5)
    A = np.triu(rng.random((n, n)) < p, 1)
    A = (A | A.T)
    iu = np.nonzero(np.triu(A, 1))
    csr = gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32))
    Af = A.astype(np.float32)
    A2A = (Af @ Af) * Af
    tri = int(round(float(A2A.sum(dtype=np.float64)))) // 6
    per_vertex = np.rint(A2A.sum(axis=1, dtype=np.float64)).astype(np.int64)  # vertex ids survive from_edges (no relabelling there)
    k4 = 0
    for a in range(n):
        nb = np.nonzero(A[a])[0]
        nb = nb[nb > a]
        if len(nb) < 3:
            continue
        S = Af[np.ix_(nb, nb)]
        k4 += int(round(float(np.einsum("ij,ij->", S @ S, S, dtype=np.float64)))) // 6  # triangles among the later neighbours of a
    for hub_limit, tri_opt in ((0, None), (900, None), (900, 0)):
        gpu.set_option("KC_TRI", tri_opt)
        g = gpu.DeviceGraph.from_csr(csr, flags=gpu.UPLOAD_DEFAULT | (hub_limit << 8))
        assert g.max_out_degree > 1024
        assert g.tc_total() == tri
        assert g.kclique_count(3)[1] == tri
        assert g.kclique_count(4)[1] == k4
        assert sum(g.kclique_partial(4, q, 3) for q in range(3)) == k4
        # per-vertex counts run on the same bit-matrices (row popcounts + column sums): counts[u] = Σ_v A²[u,v]·A[u,v]
        assert np.array_equal(g.tc_vertex_count2(), per_vertex)
        g.free()
    gpu.set_option("KC_TRI", None)


def test_matrix_core_count_equals_popcount_count(gpu):
    """k = 4 on the matrix cores (round 6, kc4_mfma.hpp: the pivots of d+ > 512 leave their bit matrices in a pool, k_kc4_mfma counts them as a masked
    bit-GEMM) against the AND + popcount count inside the BUILD kernels (KC_MFMA = 0), against a closed form, with a pool so small that every bin is cut
    into many chunks, on one stream, and shard by shard.  Complete multipartite graph: a 4-clique takes one vertex from each of four parts, so
    C_4 = e_4(part sizes); the members of a pivot are NOT a clique (no edges inside a part), d+ runs up to n - |largest part| = 2 500: both triangular
    LDS bins and the slab bins of 2 048 and 4 096.  G(n, 0.6) adds irregular rows."""
    import itertools
    parts = [800, 700, 600, 500, 400, 300]
    starts = np.concatenate([[0], np.cumsum(parts)])
    n = int(starts[-1])
    part_of = np.repeat(np.arange(len(parts)), parts)
    iu = np.triu_indices(n, 1)
    keep = part_of[iu[0]] != part_of[iu[1]]
    csr_mp = gpu.HostCSR.from_edges(iu[0][keep].astype(np.int32), iu[1][keep].astype(np.int32))
    e4 = sum(a * b * c * d for a, b, c, d in itertools.combinations(parts, 4))
    rng = np.random.default_rng(11)
    n2 = 3800
    A = np.triu(rng.random((n2, n2)) < 0.6, 1)
    ju = np.nonzero(A)
    csr_gnp = gpu.HostCSR.from_edges(ju[0].astype(np.int32), ju[1].astype(np.int32))
    for csr, closed in ((csr_mp, e4), (csr_gnp, None)):
        got = {}
        for name, opts in (("default", {}), ("popcount", {"KC_MFMA": 0}), ("small_pool", {"KC_POOL_MB": 96}), ("one_stream", {"KC_STREAMS": 1})):
            for kk, vv in opts.items():
                gpu.set_option(kk, vv)
            try:
                g = gpu.DeviceGraph.from_csr(csr)
                assert g.max_out_degree > 2048
                ordered, cliques, st = g.kclique_count(4, stats=True)
                assert ordered == (cliques * 24) & U64
                got[name] = (cliques, st["launches"])
                if name in ("default", "small_pool"):
                    assert sum(g.kclique_partial(4, q, 3) for q in range(3)) == cliques
                g.free()
            finally:
                for kk in opts:
                    gpu.set_option(kk, None)
        assert len({v[0] for v in got.values()}) == 1, got
        if closed is not None:
            assert got["default"][0] == closed
        assert got["default"][1] > got["popcount"][1]       # a BUILD and a count launch per chunk
        assert got["small_pool"][1] > got["default"][1]     # more, shorter chunks
    gpu.set_option("KC_POOL_MB", None)


def test_out_degrees_above_4096(gpu, oracle):
    """Pivots with 4096 < d+ <= 8192 (four-words-per-lane slab kernel, k <= 4 and the per-vertex counts).  K_{a,b} plus a sparse
    random graph H inside side A: every B vertex sees all of A as higher-ranked neighbours (d+ = a > 4096), triangles are
    T(H) + b·|E(H)|, 4-cliques K4(H) + b·T(H); k = 5 is beyond the two-word rows of the deeper recursion: the generic list recursion takes it."""
    a, b, eh = 4200, 4300, 20000
    rng = np.random.default_rng(11)
    hu, hv = rng.integers(0, a, eh), rng.integers(0, a, eh)
    keep = hu != hv
    H = np.zeros((a, a), dtype=bool)
    H[hu[keep], hv[keep]] = True
    H = H | H.T
    hi = np.nonzero(np.triu(H, 1))
    Hf = H.astype(np.float32)
    H2 = (Hf @ Hf) * Hf
    th = int(round(float(H2.sum(dtype=np.float64)))) // 6
    k4h = 0
    for x in np.nonzero(H.any(axis=1))[0]:
        nb = np.nonzero(H[x])[0]
        nb = nb[nb > x]
        if len(nb) >= 3:
            S = Hf[np.ix_(nb, nb)]
            k4h += int(round(float(((S @ S) * S).sum(dtype=np.float64)))) // 6
    bu, bv = np.meshgrid(np.arange(a, dtype=np.int32), np.arange(a, a + b, dtype=np.int32), indexing="ij")
    src = np.concatenate([bu.ravel(), hi[0].astype(np.int32)])
    dst = np.concatenate([bv.ravel(), hi[1].astype(np.int32)])
    g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(src, dst))
    assert 4096 < g.max_out_degree <= 8192
    eh_real = len(hi[0])
    tri = th + b * eh_real
    assert g.tc_total() == tri
    assert g.kclique_count(3)[1] == tri
    assert g.kclique_count(4)[1] == k4h + b * th
    per_vertex = np.zeros(a + b, dtype=np.int64)
    per_vertex[:a] = np.rint(H2.sum(axis=1, dtype=np.float64)).astype(np.int64) + 2 * b * H.sum(axis=1)  # 2 x (triangles in H + b per H-edge at the vertex)
    per_vertex[a:] = 2 * eh_real
    assert np.array_equal(g.tc_vertex_count2(), per_vertex)
    # k = 5 is past the 4096-wide bit rows of the k >= 5 kernels: the B pivots go through the generic list recursion
    # (every 5-clique has at most one B vertex: b * C_4(H) + C_5(H))
    hcsr = gpu.HostCSR.from_edges(hi[0].astype(np.int32), hi[1].astype(np.int32), num_nodes=a)
    k5h = oracle.kclique(hcsr.offsets(), hcsr.neighbors(), 5) // 120
    assert g.kclique_count(5)[1] == b * k4h + k5h
    g.free()


def test_kclique_star_count(gpu, oracle):
    """gmsx_kclique_star_count = KCliqueStar::Par::CliqueStar in count mode (k_clique_star_list/parallel/recursive.h:19-35): the number of
    k-clique-stars and the total size of the stars, bit-equal to the oracle's restatement of the reference recursion (itself pinned against
    the compiled reference's list in tests/test_oracle.py) and to the reference goldens."""
    for kind, scale, deg in (("kronecker", 10, 16), ("uniform", 11, 16), ("kronecker", 8, 40)):
        csr = host_graph(gpu, kind, scale, deg, True)
        g = gpu.DeviceGraph.from_csr(csr)
        for k in (1, 2, 3, 4, 5):
            want = oracle.kclique_star_count(csr.offsets(), csr.neighbors(), k)
            assert g.kclique_star_count(k) == want, (kind, scale, k)
            assert g.kclique_star_count(k, members=False) == (want[0], None)
        g.free()
    csr = host_graph(gpu, "kronecker", 14)
    g = gpu.DeviceGraph.from_csr(csr)
    rec = GRAPHS["kronecker-14-16-relabel"]
    assert g.kclique_star_count(3) == (rec["triangles"], 4 * (rec["kc4"] // 24))
    assert g.kclique_star_count(2)[0] == csr.num_edges and g.kclique_star_count(2)[1] == 3 * rec["triangles"]
    with pytest.raises(gpu.GmsxError):
        g.kclique_star_count(0)
    g.free()


@pytest.mark.parametrize("rev_min", [1, 8, 64])
def test_reverse_rows_equal_forward_rows(gpu, oracle, rev_min):
    """Round 6: the BUILD takes rows[i] = N+(v_i) ∩ N+(u) from the cheaper end of the edge — streamed forward (N+(v_i) against u's bitmap)
    or, for hub members whose receiver takes enough edges, copied from the arena k_kc_reverse filled (u's prefix below v_i against v_i's bitset container).
    Same matrix either way: counts for k = 3 … 6, the per-vertex counts and the shards' partial sums with the reverse rows at several receiver thresholds
    (KC_REV_MIN; 1 = every cheaper edge is handed over) equal those with KC_REVERSE = 0, the oracle's and the reference goldens; small hub ranges
    (GMSX_UPLOAD_HUB_LIMIT) move the hub / tail boundary through the pivots' rows."""
    specs = (("kronecker", 14, 16, 0), ("kronecker", 16, 16, 0), ("kronecker", 13, 40, 0), ("kronecker", 14, 16, 3000), ("uniform", 12, 120, 0), ("kronecker", 12, 64, 500))
    for kind, scale, deg, hub_limit in specs:
        csr = host_graph(gpu, kind, scale, deg, True)
        flags = gpu.UPLOAD_DEFAULT | (hub_limit << 8)
        with gpu.options(KC_REVERSE=0):
            g0 = gpu.DeviceGraph.from_csr(csr, flags=flags)
            want = {k: g0.kclique_count(k, stats=True) for k in (3, 4, 5, 6)}
            vc0 = g0.tc_vertex_count2()
            g0.free()
        with gpu.options(KC_REV_MIN=rev_min):
            g = gpu.DeviceGraph.from_csr(csr, flags=flags)
            base = g.device_bytes
            for k in (4, 3, 5, 6):
                ordered, cliques, st = g.kclique_count(k, stats=True)
                assert (ordered, cliques) == want[k][:2], (kind, scale, deg, hub_limit, k, rev_min)
                assert st["stream_bytes"] > 0
            assert np.array_equal(g.tc_vertex_count2(), vc0)
            # shards on a FRESH upload each (an arena that a whole-graph call filled before would hide a receiver pass that skips the wrong pivots:
            # the ranks of a multi-GPU run are separate processes that only ever run their own shard)
            for nparts in (3, 5):
                total = 0
                for p_ in range(nparts):
                    gs = gpu.DeviceGraph.from_csr(csr, flags=flags)
                    total += gs.kclique_partial(4, p_, nparts)
                    gs.free()
                assert total == want[4][1], (kind, scale, deg, hub_limit, nparts, rev_min)
            # under the BYTE rule of rounds 6a (hub receivers only, handed over when at least twice cheaper) handing an edge over never costs more bytes than
            # streaming it; the default since 6b prices the resolution of a forward row's hits too and hands over up to ten times the bytes
            with gpu.options(KC_REV_MIN=rev_min, KC_REV_FACTOR=20, KC_REV_TAIL=0):
                gb = gpu.DeviceGraph.from_csr(csr, flags=flags)
                ordered, cliques, st = gb.kclique_count(4, stats=True)
                assert (ordered, cliques) == want[4][:2] and st["stream_bytes"] <= want[4][2]["stream_bytes"], (kind, scale, rev_min)
                gb.free()
            with gpu.options(KC_REV_MIN=rev_min, KC_REV_TAIL_MIN=1):   # every tail member that can be handed over is
                gt = gpu.DeviceGraph.from_csr(csr, flags=flags)
                for k in (4, 5):
                    assert gt.kclique_count(k)[:2] == want[k][:2], (kind, scale, deg, hub_limit, k, rev_min, "tail receivers")
                assert np.array_equal(gt.tc_vertex_count2(), vc0)
                gt.free()
            if (kind, scale, deg, hub_limit) == ("kronecker", 16, 16, 0) and rev_min <= 8:
                assert g.device_bytes > base      # this graph does hand edges over: the lists and the arena stay with the graph
            g.free()
        key = f"{kind}-{scale}-{deg}-relabel"
        if key in GRAPHS and "kc4" in GRAPHS[key]:
            assert want[4][0] == GRAPHS[key]["kc4"]
    csr = host_graph(gpu, "kronecker", 12, 16, True)
    assert gpu.DeviceGraph.from_csr(csr).kclique_count(4)[0] == oracle.kclique(csr.offsets(), csr.neighbors(), 4)
