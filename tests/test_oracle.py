"""CPU: pins the oracle (oracle/gms_oracle.c) against the golden vectors generated from the compiled reference
(tools/make_golden.py) and, when oracle/_ref is present, against the compiled reference directly."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, edges_to_csr, host_graph, load_golden


def test_set_ops_literal_cases(oracle):
    for case in load_golden("set_ops.json")["cases"]:
        a, b, exp = oracle.make_set(case["a"]), oracle.make_set(case["b"]), case["expect"]
        assert oracle.intersect_count(a, b) == exp["intersect_count"]
        assert oracle.intersect_count(b, a) == exp["intersect_count"]  # symmetric (testing/sets.cpp:108-115)
        assert oracle.intersect(a, b).tolist() == exp["intersect"] == exp["intersect_inplace"]
        assert oracle.difference(a, b).tolist() == exp["difference"] == exp["difference_inplace"]
        assert oracle.union(a, b).tolist() == exp["union"] == exp["union_inplace"]
        assert oracle.union_count(a, b) == exp["union_count"]
        assert a.size == exp["cardinality"]


def test_set_ops_random_cases(oracle):
    z = np.load(os.path.join(GOLDEN, "set_ops_random.npz"))
    i = 0
    while f"a{i}" in z:
        a, b = oracle.make_set(z[f"a{i}"]), oracle.make_set(z[f"b{i}"])
        ic, uc, card = (int(x) for x in z[f"counts{i}"])
        assert oracle.intersect_count(a, b) == ic and oracle.union_count(a, b) == uc and a.size == card
        assert np.array_equal(oracle.intersect(a, b), z[f"intersect{i}"])
        assert np.array_equal(oracle.difference(a, b), z[f"difference{i}"])
        assert np.array_equal(oracle.union(a, b), z[f"union{i}"])
        i += 1
    assert i >= 10


GRAPHS = load_golden("graphs.json")
SMALL = [k for k, v in GRAPHS.items() if v.get("scale", 99) <= 14 and "offsets_fnv64" in v]


@pytest.mark.parametrize("key", SMALL)
def test_graph_counts_match_reference_goldens(oracle, capi, key):
    rec = GRAPHS[key]
    csr = host_graph(capi, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
    off, ng = csr.offsets(), csr.neighbors()
    assert (csr.num_nodes, csr.num_edges) == (rec["n"], rec["m"])
    if "triangles" in rec:
        t, raw = oracle.tc_total(off, ng, raw=True)
        assert raw % 3 == 0 and t == rec["triangles"]
    for k in (3, 4, 5):
        if f"kc{k}" in rec and (rec["scale"] <= 12 or k == 3):
            assert oracle.kclique(off, ng, k) == rec[f"kc{k}"]
    if "bk" in rec and (rec["scale"] <= 12 or rec["generator"] == "rmat"):  # the low-skew family enumerates in seconds at scale 14
        assert oracle.bk_count(off, ng) == rec["bk"]


def test_kclique_s14_k4(oracle, capi):
    rec = GRAPHS["kronecker-14-16-relabel"]
    csr = host_graph(capi, "kronecker", 14)
    assert oracle.kclique(csr.offsets(), csr.neighbors(), 4) == rec["kc4"]


def test_vertex_count2(oracle, capi):
    z = np.load(os.path.join(GOLDEN, "vertex_count2.npz"))
    for key in z.files:
        rec = GRAPHS[key]
        csr = host_graph(capi, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
        off, ng = csr.offsets(), csr.neighbors()
        assert np.array_equal(oracle.tc_vertex_count2(off, ng), z[key])
        assert np.array_equal(oracle.tc_vertex_count2(off, ng, once=True), z[key])
        assert int(z[key].sum()) == 6 * rec["triangles"]  # verifier.h:44-85: 3*T == Σ counts / 2


def test_reference_test_graphs(oracle, capi):
    tg = load_golden("testgraphs.json")
    assert tg["triangles_3.el"]["triangles"] == 3 and tg["triangles_3.el"]["bk"] == 7
    for name, rec in tg.items():
        csr = capi.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name))
        off, ng = csr.offsets(), csr.neighbors()
        assert off.tolist() == rec["offsets"] and ng.tolist() == rec["neigh"]
        assert oracle.tc_total(off, ng) == rec["triangles"]
        assert oracle.kclique(off, ng, 3) == rec["kc3"] and oracle.kclique(off, ng, 4) == rec["kc4"]
        assert oracle.bk_count(off, ng) == rec["bk"]


def test_known_answers(oracle, capi):
    ka = load_golden("known_answers.json")
    fact = {2: 2, 3: 6, 4: 24}
    for c in ka["kclique"]:  # testing/clique_counting/CliqueCounter2_tests.h:45-271
        csr = edges_to_csr(capi, c["edges"])
        off, ng = csr.offsets(), csr.neighbors()
        assert oracle.kclique(off, ng, c["k"]) == c["ordered"] == fact[c["k"]] * c["cliques"], c["name"]
        assert oracle.tc_total(off, ng) == c["triangles"] and oracle.bk_count(off, ng) == c["bk"]
    for c in ka["bk_random"]:  # concrete instances of testing/bron_kerbosch.cpp:256-268
        csr = edges_to_csr(capi, c["edges"], n=c["n"])
        off, ng = csr.offsets(), csr.neighbors()
        assert oracle.bk_count(off, ng) == c["bk"]
        assert oracle.bk_count(off, ng, rank=np.arange(c["n"], dtype=np.int32)) == c["bk"]  # rank-independent
        assert oracle.tc_total(off, ng) == c["triangles"] and oracle.kclique(off, ng, 4) == c["kc4"]


def test_sample_covers_the_whole_loop(oracle, capi):
    csr = host_graph(capi, "kronecker", 10)
    off, ng = csr.offsets(), csr.neighbors()
    _, raw = oracle.tc_total(off, ng, raw=True)
    parts = [oracle.tc_total_sample(off, ng, 5, p) for p in range(5)]
    assert sum(p[0] for p in parts) == raw
    assert sum(p[1] for p in parts) == csr.num_edges
    assert sum(p[2] for p in parts) == oracle.tc_elements(off, ng) == csr.merge_elements()


def test_edge_cases(oracle):
    off = np.zeros(1, dtype=np.int64)
    ng = np.zeros(0, dtype=np.int32)
    assert oracle.tc_total(off, ng) == 0 and oracle.kclique(off, ng, 4) == 0 and oracle.bk_count(off, ng) == 0
    off = np.zeros(4, dtype=np.int64)  # three isolated vertices: each is a maximal clique (eppsteinPAR.h:32-47)
    assert oracle.bk_count(off, ng) == 3 and oracle.tc_total(off, ng) == 0


# ---- against the compiled reference itself (when oracle/_ref is present) --------------------------------

@pytest.mark.parametrize("spec", [("kronecker", 9, 8), ("kronecker", 11, 16), ("uniform", 11, 12), ("kronecker", 7, 30)])
def test_oracle_equals_compiled_reference(oracle, reference, spec):
    kind, scale, deg = spec
    g = reference.generate(kind, scale, deg, relabel=True)
    try:
        off, ng = reference.csr(g)
        assert oracle.tc_total(off, ng) == reference.tc_total(g, 0) == reference.tc_total(g, 1)
        assert oracle.kclique(off, ng, 4) == reference.kclique(g, 4, 0)
        assert oracle.kclique(off, ng, 5) == reference.kclique(g, 5, 1)
        assert oracle.bk_count(off, ng) == reference.bk_count(g, 1, 0) == reference.bk_count(g, 0, 2)
        assert np.array_equal(oracle.tc_vertex_count2(off, ng), reference.tc_vertex_count2(g, 0, 0))
        assert np.array_equal(oracle.degree_rank(off), reference.rank(g, 0))
    finally:
        reference.free(g)


@pytest.mark.parametrize("spec", [("kronecker", 8, 16), ("kronecker", 12, 16), ("kronecker", 16, 16), ("uniform", 11, 12), ("kronecker", 7, 40)])
def test_reference_sliced_triangle_count_equals_the_unsliced_call(oracle, reference, spec):
    """The scale-27 golden (tests/golden/graphs.json: 106 873 365 648 triangles, BASELINE configs[4]) comes from ref_tc_total_sliced — the reference's
    RoaringSet::intersect_count per edge, accumulated over id-range slices of the neighbourhoods, because the whole RoaringGraph of that graph (~90 GB)
    does not fit the build container.  |N(u) ∩ N(v)| is additive over any partition of the id space, so every slice count gives the reference's own
    Par::count_total — asserted here for 1, 2, 3, 5 and 8 slices on every graph both calls can hold, SortedSet flavour and oracle included."""
    kind, scale, deg = spec
    g = reference.generate(kind, scale, deg, relabel=True)
    try:
        want = reference.tc_total(g, 1)
        assert want == reference.tc_total(g, 0)
        for k in (1, 2, 3, 5, 8):
            assert reference.tc_total_sliced(g, k) == want, k
        for kind_ in (0, 1):  # bench.py's cpu_baseline leg: the same call with SetGraph::FromCGraph and kernel(sgraph) clocked apart (common/benchmark.h:105-116)
            tri, build_s, count_s = reference.tc_total_timed(g, kind_)
            assert tri == want and build_s >= 0 and count_s > 0
        if scale <= 12:
            off, ng = reference.csr(g)
            assert oracle.tc_total(off, ng) == want
    finally:
        reference.free(g)


@pytest.mark.parametrize("spec", [("kronecker", 10, 16), ("kronecker", 12, 16), ("uniform", 11, 12), ("kronecker", 8, 40)])
def test_reference_kclist_agrees_with_its_set_based_count(oracle, reference, spec):
    """The reference has two k-clique paths: the set-based CliqueCount (k! x each clique; what the hot path replaces) and kClist on the
    degeneracy DAG (each clique once; its source of true counts).  They agree — which is what lets kClist pin the device at
    sizes the set-based recursion cannot reach (tests/golden/graphs.json kc4_true at scales 24 / 26) — and so does the oracle; the
    64-bit-safe driver loop of oracle/ref_shim.cc (ref_kclist_count_wide) equals the reference's own Par::NP_kclisting."""
    kind, scale, deg = spec
    g = reference.generate(kind, scale, deg, relabel=True)
    try:
        off, ng = reference.csr(g)
        for k, f in ((3, 6), (4, 24), (5, 120)):
            ordered = reference.kclique(g, k, 1)
            assert ordered % f == 0
            assert reference.kclist_count(g, k) == reference.kclist_count_wide(g, k) == ordered // f == oracle.kclique(off, ng, k) // f
        assert reference.kclist_count(g, 4, order=1) == reference.kclist_count(g, 4, order=0)  # any acyclic orientation
    finally:
        reference.free(g)


@pytest.mark.parametrize("spec", [("kronecker", 8, 16), ("kronecker", 10, 16), ("uniform", 10, 16), ("kronecker", 7, 40)])
def test_kclique_star_count_against_compiled_reference(oracle, reference, spec):
    """KCliqueStar::Par::CliqueStarList (k_clique_star_list/parallel/recursive.h:19-43): the oracle's count-mode restatement equals the
    reference's list — its length and the total size of its stars — and both equal (C_k, (k+1) C_{k+1}), the identity the device entry
    point gmsx_kclique_star_count stands on (every (k+1)-clique puts each member into the star of the k-clique of the others)."""
    kind, scale, deg = spec
    g = reference.generate(kind, scale, deg, relabel=True)
    try:
        off, ng = reference.csr(g)
        fact = [1, 1, 2, 6, 24, 120]
        for k in (1, 2, 3, 4):
            want = reference.kclique_star(g, k)
            assert oracle.kclique_star_count(off, ng, k) == want
            ck = off.size - 1 if k == 1 else oracle.kclique(off, ng, k) // fact[k]
            assert want == (ck, (k + 1) * (oracle.kclique(off, ng, k + 1) // fact[k + 1]))
    finally:
        reference.free(g)


def test_kclique_star_count_goldens(oracle, capi):
    """… and against the committed goldens (the reference's own counts): scale 10: C_3 = 74 720, C_4 = 409 665, C_5 = 1 745 925."""
    from conftest import host_graph
    csr = host_graph(capi, "kronecker", 10)
    off, ng = csr.offsets(), csr.neighbors()
    assert oracle.kclique_star_count(off, ng, 3) == (74720, 4 * 409665)
    assert oracle.kclique_star_count(off, ng, 4) == (409665, 5 * 1745925)
    assert oracle.kclique_star_count(off, ng, 1) == (1024, 2 * 10496)


def test_golden_kc4_true_is_kc4_over_24():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphs.json")) as f:
        graphs = json.load(f)
    both = [k for k, r in graphs.items() if "kc4" in r and "kc4_true" in r]
    assert {"kronecker-16-16-relabel", "kronecker-18-16-relabel", "kronecker-20-16-relabel", "kronecker-22-16-relabel"} <= set(both)
    for k in both:
        assert graphs[k]["kc4"] == 24 * graphs[k]["kc4_true"], k
    assert graphs["kronecker-24-16-relabel"]["kc4_true"] == 879950888260


def test_set_ops_against_compiled_reference(oracle, reference):
    rng = np.random.default_rng(7)
    for _ in range(200):
        hi = int(rng.choice([8, 64, 1000, 1 << 17, 1 << 24]))
        a = rng.choice(hi, size=int(rng.integers(0, min(hi, 400))), replace=False).astype(np.int32)
        b = rng.choice(hi, size=int(rng.integers(0, min(hi, 400))), replace=False).astype(np.int32)
        sa, sb = oracle.make_set(a), oracle.make_set(b)
        for kind in (0, 1):
            assert oracle.intersect_count(sa, sb) == reference.set_op(kind, "intersect_count", a, b)
            assert np.array_equal(oracle.intersect(sa, sb), np.sort(reference.set_op(kind, "intersect", a, b)))
            assert np.array_equal(oracle.difference(sa, sb), np.sort(reference.set_op(kind, "difference", a, b)))
            assert np.array_equal(oracle.union(sa, sb), np.sort(reference.set_op(kind, "union", a, b)))
            assert oracle.union_count(sa, sb) == reference.set_op(kind, "union_count", a, b)


SIM = ["jaccard", "overlap", "adamic_adar", "resource", "common_neighbors", "total_neighbors", "pref_attachment"]


def test_vertex_similarity_goldens(oracle, capi):
    """GMS::VertexSim::vertex_similarity (vertex_similarity.h:205-222): bit-identical doubles, NaN/inf positions included."""
    z = np.load(os.path.join(GOLDEN, "vertex_similarity.npz"))
    csr = host_graph(capi, "kronecker", 10)
    off, ng = csr.offsets(), csr.neighbors()
    for m, name in enumerate(SIM):
        got = oracle.vertex_similarity(m, off, ng, z["u"], z["v"])
        assert np.array_equal(got, z[name], equal_nan=True), name
    assert np.isnan(z["overlap"]).any() and (z["jaccard"] == 1.0).any()  # the empty-set edge cases are in the vectors


# ---- orderings -------------------------------------------------------------------------------

def _adg_consistent(ref_rank, rnd, deg):
    """The reference leaves ties inside a round to its partition/sort: its output must walk the oracle's (round, degree) staircase."""
    by_ref = np.argsort(ref_rank)
    keys = rnd[by_ref].astype(np.int64) * (1 << 32) + deg[by_ref]
    return bool(np.all(np.diff(keys) >= 0))


def test_orderings_match_reference_goldens(oracle, capi):
    path = os.path.join(GOLDEN, "orderings.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/orderings.npz not generated (tools/make_golden_orderings.py)")
    G = np.load(path)
    seen = 0
    for key in G.files:
        if not key.startswith("adg_") or key.startswith("adg_file_"):
            continue
        kind, scale, deg = key[4:].rsplit("_", 2)
        csr = capi.HostCSR.generate(kind, int(scale), int(deg))
        off, ng = csr.offsets(), csr.neighbors()
        rank, rnd, dat, rounds = oracle.adg_rank(off, ng, 0.001)
        assert np.array_equal(np.sort(rank), np.arange(csr.num_nodes)) and rounds == rnd.max() + 1
        assert _adg_consistent(G[key], rnd, dat), key
        # inside a round the oracle orders by (degree, id): ranks strictly follow that key
        by_rank = np.argsort(rank)
        assert np.all(np.diff((rnd[by_rank].astype(np.int64) << 52) + (dat[by_rank].astype(np.int64) << 28) + by_rank) > 0)
        counts = oracle.tc_vertex_count2(off, ng)
        mine = oracle.tc_ordering(off, ng)
        assert np.array_equal(counts[G["tco_" + key[4:]]], counts[mine]), key     # same count sequence as the reference's order
        assert np.all(np.diff(counts[mine]) >= 0)
        seen += 1
    for name in sorted(os.listdir(os.path.join(GOLDEN, "testGraphs"))):
        key = "adg_file_" + name[:-3]
        if key in G.files:
            csr = capi.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name))
            rank, rnd, dat, _ = oracle.adg_rank(csr.offsets(), csr.neighbors(), 0.001)
            assert _adg_consistent(G[key], rnd, dat), key
            seen += 1
    assert seen >= 6


@pytest.mark.parametrize("spec", [("kronecker", 9, 8), ("kronecker", 11, 16), ("uniform", 11, 12)])
def test_orderings_against_compiled_reference(oracle, reference, spec):
    kind, scale, deg = spec
    g = reference.generate(kind, scale, deg, True, threads=1)
    off, ng = reference.csr(g)
    rank, rnd, dat, _ = oracle.adg_rank(off, ng, 0.001)
    assert _adg_consistent(reference.rank(g, 1), rnd, dat)
    if hasattr(reference.L, "ref_tc_ordering"):
        counts = oracle.tc_vertex_count2(off, ng)
        assert np.array_equal(counts[reference.tc_ordering(g, 0)], counts[oracle.tc_ordering(off, ng)])
    reference.free(g)
