"""GPU parity tests of the HIP Bron–Kerbosch maximal-clique count (gmsx_bk_count = BK_CLIQUE_COUNTER after
BkEppsteinPar::mceBench) against the reference goldens, the oracle and the reference's own small test graphs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, edges_to_csr, host_graph, load_golden

pytestmark = pytest.mark.gpu
GRAPHS = load_golden("graphs.json")


@pytest.mark.parametrize("key", [k for k, v in GRAPHS.items() if "bk" in v and v["scale"] <= 14])
def test_bk_equals_reference_golden(gpu, key):
    rec = GRAPHS[key]
    g = gpu.DeviceGraph.from_csr(host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"]))
    got, st = g.bk_count(stats=True)
    assert got == rec["bk"]
    assert st["kernel_ms"] > 0
    g.free()


def test_bk_uniform_goldens(gpu):
    for key in ("uniform-14-16-relabel", "uniform-16-16-relabel"):
        rec = GRAPHS[key]
        g = gpu.DeviceGraph.from_csr(host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"]))
        assert g.bk_count() == rec["bk"]
        g.free()


def test_bk_reference_test_graphs_and_random_instances(gpu, oracle):
    for name, rec in load_golden("testgraphs.json").items():  # eppsteinExample.el, tomitaExample.el, ...
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name)))
        assert g.bk_count() == rec["bk"], name
        g.free()
    ka = load_golden("known_answers.json")
    for c in ka["bk_random"]:  # concrete instances of testing/bron_kerbosch.cpp:256-268
        g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, c["edges"], n=c["n"]))
        assert g.bk_count() == c["bk"]
        # a rank vector is validated (it must be a permutation, like every rank-format ordering) and does not change the count
        assert g.bk_count(rank=np.arange(c["n"], dtype=np.int32)[::-1].copy()) == c["bk"]
        for bad in (np.zeros(c["n"], dtype=np.int32), np.arange(1, c["n"] + 1, dtype=np.int32)):
            with pytest.raises(gpu.GmsxError) as ei:
                g.bk_count(rank=bad)
            assert ei.value.status == gpu.ERR_INVALID
        g.free()
    for c in ka["kclique"]:
        g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, c["edges"]))
        assert g.bk_count() == c["bk"], c["name"]
        g.free()


@pytest.mark.parametrize("spec", [("kronecker", 9, 3), ("kronecker", 10, 30), ("uniform", 9, 60), ("kronecker", 11, 8)])
def test_bk_vs_oracle(gpu, oracle, spec):
    kind, scale, deg = spec
    for relabel in (True, False):
        csr = host_graph(gpu, kind, scale, deg, relabel)
        g = gpu.DeviceGraph.from_csr(csr)
        assert g.bk_count() == oracle.bk_count(csr.offsets(), csr.neighbors()), (spec, relabel)
        g.free()


def test_bk_partials_and_tail_containers(gpu, oracle):
    csr = host_graph(gpu, "kronecker", 11, 16, True)
    want = oracle.bk_count(csr.offsets(), csr.neighbors())
    g = gpu.DeviceGraph.from_csr(csr)
    for nparts in (2, 3, 8):
        assert sum(g.bk_partial(p, nparts) for p in range(nparts)) == want
    g.free()
    g = gpu.DeviceGraph.from_csr(csr, flags=(16 << 8))  # hub-limit test hook: ids >= 16 live in tail containers
    assert g.bk_count() == want
    g.free()


def test_bk_edge_cases(gpu):
    # isolated vertices are maximal cliques (eppsteinPAR.h:32-47 + tomita.h:73-78); K_n has exactly one
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, []))
    assert g.bk_count() == 1
    g.free()
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, [(0, 1)], n=5))
    assert g.bk_count() == 4  # the edge + three isolated vertices
    g.free()
    for k in (3, 40, 700):
        iu = np.triu_indices(k, 1)
        g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(iu[0].astype(np.int32), iu[1].astype(np.int32)))
        assert g.bk_count() == 1
        g.free()
    # a star: every edge is a maximal clique; a path likewise; a triangle with a pendant edge: 2
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, [(0, i) for i in range(1, 200)]))
    assert g.bk_count() == 199
    g.free()
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, [(0, 1), (1, 2), (2, 0), (2, 3)]))
    assert g.bk_count() == 2
    g.free()
    # complete multipartite K_{3,3,3}: 27 maximal cliques (one vertex per part); Moon-Moser extremal shape
    parts = [range(0, 3), range(3, 6), range(6, 9)]
    edges = [(a, b) for i, pa in enumerate(parts) for pb in parts[i + 1:] for a in pa for b in pb]
    g = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, edges))
    assert g.bk_count() == 27
    g.free()


def test_bk_more_than_2048_candidates(gpu, oracle):
    """Start vertices with 2049..4096 candidates run on the two-words-per-lane kernels.  K_{a,b} plus a sparse random graph H
    inside side A: B is independent and sees all of A, so every maximal clique is (a maximal clique of H, isolated vertices
    included) + one B vertex: b · MC(H), with MC(H) from the oracle on H alone.  Also the partial counts and the wide +
    narrow mix of resumable records (the B searches exceed the node budget)."""
    a, b, eh = 2200, 2300, 9000
    rng = np.random.default_rng(21)
    hu, hv = rng.integers(0, a, eh), rng.integers(0, a, eh)
    keep = hu != hv
    hcsr = gpu.HostCSR.from_edges(hu[keep].astype(np.int32), hv[keep].astype(np.int32), num_nodes=a)
    mc_h = oracle.bk_count(hcsr.offsets(), hcsr.neighbors())
    bu, bv = np.meshgrid(np.arange(a, dtype=np.int32), np.arange(a, a + b, dtype=np.int32), indexing="ij")
    src = np.concatenate([bu.ravel(), hu[keep].astype(np.int32)])
    dst = np.concatenate([bv.ravel(), hv[keep].astype(np.int32)])
    g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(src, dst))
    assert 2048 < g.max_out_degree <= 4096
    want = b * mc_h
    assert g.bk_count() == want
    assert sum(g.bk_partial(p, 3) for p in range(3)) == want
    g.free()


def test_bk_late_hubs_are_built_in_pieces(gpu, oracle):
    """Three mutually adjacent hubs that see every vertex of a sparse random graph H: whatever the order, a start vertex ends up with
    20 000 CSR positions — a dozen pieces of the workgroup build (k_bk_block takes 2048 row jobs of a start vertex per work item, the
    pieces of one vertex on different workgroups) — and, ranked last, with two candidates against 20 000 finished neighbours.  Every
    maximal clique is (a maximal clique of H, isolated vertices included) + the three hubs: MC(H) from the oracle on H alone."""
    n, eh, hubs = 20000, 50000, 3
    rng = np.random.default_rng(33)
    hu, hv = rng.integers(0, n, eh), rng.integers(0, n, eh)
    keep = hu != hv
    hcsr = gpu.HostCSR.from_edges(hu[keep].astype(np.int32), hv[keep].astype(np.int32), num_nodes=n)
    want = oracle.bk_count(hcsr.offsets(), hcsr.neighbors())
    hub_ids = np.arange(n, n + hubs, dtype=np.int32)
    su, sv = np.meshgrid(hub_ids, np.arange(n, dtype=np.int32), indexing="ij")
    hh = np.array([(a, b) for i, a in enumerate(hub_ids) for b in hub_ids[i + 1:]], dtype=np.int32)
    src = np.concatenate([su.ravel(), hh[:, 0], hu[keep].astype(np.int32)])
    dst = np.concatenate([sv.ravel(), hh[:, 1], hv[keep].astype(np.int32)])
    g = gpu.DeviceGraph.from_csr(gpu.HostCSR.from_edges(src, dst))
    assert g.bk_count() == want
    ident = np.arange(n + hubs, dtype=np.int32)
    assert g.bk_count(rank=ident) == want              # hubs last: few candidates, 20 000 finished neighbours each
    assert g.bk_count(rank=ident[::-1].copy()) == want  # hubs first: 20 000 candidates each (the wide search kernels)
    assert sum(g.bk_partial(p, 3) for p in range(3)) == want
    g.free()


def test_bk_roots_that_do_not_fit_the_arena_are_built_in_chunks(gpu):
    """GMSX_BK_ARENA_MB (test hook) caps the arena: the start vertices too big for an LDS slab are then built and searched chunk by chunk
    (as many heavy-first start vertices as fit 3/4 of the arena), every chunk with its own span of build pieces, arena fill and record
    pool.  Same count as the oracle and as the one-chunk run; more launches."""
    rec = GRAPHS["kronecker-14-16-relabel"]
    want = rec["bk"]  # the compiled reference's count
    csr = host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
    g = gpu.DeviceGraph.from_csr(csr)
    got, st = g.bk_count(stats=True)
    assert got == want
    gpu.set_option("BK_ARENA_MB", "1")
    try:
        got2, st2 = g.bk_count(stats=True)
        assert got2 == want
        assert st2["launches"] > st["launches"], (st, st2)
        assert sum(g.bk_partial(p, 2) for p in range(2)) == want
    finally:
        gpu.reset_options()
    g.free()


@pytest.mark.parametrize("budget", [16, 48, 300])
def test_bk_tiny_budget_splits_everything(gpu, oracle, budget):
    """A node budget of a few dozen nodes makes every non-trivial search split, again and again: every level with pending branches is cut
    into up to eight records whose runs of siblings start from the state their predecessors leave behind (P minus / X plus the earlier
    branch vertices).  Same counts as the oracle on graphs with deep searches (dense blocks), wide nodes (hubs) and both."""
    gpu.set_option("BK_BUDGET", str(budget))
    gpu.set_option("BK_BUDGET0", str(budget))
    try:
        rng = np.random.default_rng(5)
        graphs = [host_graph(gpu, "kronecker", 11, 16, True), host_graph(gpu, "uniform", 9, 60, True)]
        k = 220  # a dense random block: long chains of recursion
        iu = np.triu_indices(k, 1)
        sel = rng.random(iu[0].size) < 0.5
        graphs.append(gpu.HostCSR.from_edges(iu[0][sel].astype(np.int32), iu[1][sel].astype(np.int32)))
        parts = [range(4 * i, 4 * i + 4) for i in range(6)]  # complete multipartite K_{4,4,4,4,4,4}: 4^6 maximal cliques, wide nodes
        edges = [(a, b) for i, pa in enumerate(parts) for pb in parts[i + 1:] for a in pa for b in pb]
        graphs.append(gpu.HostCSR.from_edges(np.array([e[0] for e in edges], np.int32), np.array([e[1] for e in edges], np.int32)))
        for i, csr in enumerate(graphs):
            want = oracle.bk_count(csr.offsets(), csr.neighbors())
            g = gpu.DeviceGraph.from_csr(csr)
            got, st = g.bk_count(stats=True)
            assert got == want, (i, budget)
            assert sum(g.bk_partial(p, 3) for p in range(3)) == want
            g.free()
        assert oracle.bk_count(graphs[3].offsets(), graphs[3].neighbors()) == 4 ** 6
    finally:
        gpu.reset_options()


@pytest.mark.parametrize("knobs", [{"BK_GROUPS": "0"}, {"BK_TINY_BESIDE": "2"}, {"BK_TINY_ROOTS": "1"}, {"BK_SMALL_P_GROUPS": "0"},
                                   {"BK_SMALL_P_GROUPS": "600"}, {"BK_BUDGET": "40", "BK_BUDGET0": "40", "BK_SMALL_P_GROUPS": "0"},
                                   {"BK_BUDGET": "24", "BK_BUDGET0": "24", "BK_GROUPS": "0"}])
def test_bk_search_kernel_variants(gpu, oracle, knobs):
    """Round 5: records with at most 512 candidates are searched four to a wave (k_bk_resume4: 16-lane groups, Xf below level 0 as a list of
    non-zero words — in registers up to 16 pairs, in memory beyond —, the pivot's row doubling as the first branch row, levels kept in
    registers / not kept at all when no branch is left).  The knobs switch it off (every record on k_bk_resume), move the LDS-slab tasks
    beside the first resume round, force / forbid pivot scoring, and split every search after a few dozen nodes (records written from
    every level in list and in dense form).  Same counts as the oracle on: an R-MAT graph, a dense block (deep searches), a late hub with
    thousands of in-neighbours (Xf far longer than 16 words at level 0, lists in memory below), and K_{4,4,4,4,4,4}."""
    for k_opt, v_opt in knobs.items():
        gpu.set_option(k_opt, v_opt)
    try:
        rng = np.random.default_rng(11)
        graphs = [host_graph(gpu, "kronecker", 11, 16, True)]
        k = 200
        iu = np.triu_indices(k, 1)
        sel = rng.random(iu[0].size) < 0.5
        graphs.append(gpu.HostCSR.from_edges(iu[0][sel].astype(np.int32), iu[1][sel].astype(np.int32)))
        # a clique-rich core of 120 vertices + a hub adjacent to the whole core and to 6000 leaves, each leaf adjacent to a few core vertices:
        # the hub comes late in the degree order of its core neighbours' searches, its in-neighbours fill hundreds of Xf words
        core = 120
        cu, cv = np.triu_indices(core, 1)
        keep = rng.random(cu.size) < 0.6
        hub = core
        leaves = np.arange(core + 1, core + 1 + 6000, dtype=np.int32)
        lu = np.repeat(leaves, 3)
        lv = rng.integers(0, core, lu.size).astype(np.int32)
        src = np.concatenate([cu[keep].astype(np.int32), np.full(core, hub, np.int32), np.full(leaves.size, hub, np.int32), lu])
        dst = np.concatenate([cv[keep].astype(np.int32), np.arange(core, dtype=np.int32), leaves, lv])
        graphs.append(gpu.HostCSR.from_edges(src, dst))
        parts = [range(4 * i, 4 * i + 4) for i in range(6)]
        edges = [(a, b) for i, pa in enumerate(parts) for pb in parts[i + 1:] for a in pa for b in pb]
        graphs.append(gpu.HostCSR.from_edges(np.array([e[0] for e in edges], np.int32), np.array([e[1] for e in edges], np.int32)))
        for i, csr in enumerate(graphs):
            want = oracle.bk_count(csr.offsets(), csr.neighbors())
            g = gpu.DeviceGraph.from_csr(csr)
            assert g.bk_count() == want, (i, knobs)
            assert sum(g.bk_partial(p, 2) for p in range(2)) == want, (i, knobs)
            g.free()
    finally:
        gpu.reset_options()
