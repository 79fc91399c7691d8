"""GPU parity tests of the full-row intersect kernels: gmsx_intersect_count_batch (Set::intersect_count), the
reference-verbatim GMSX_TC_FULL triangle count, and gmsx_tc_vertex_count2 (Par::vertex_count2)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, edges_to_csr, host_graph, load_golden

pytestmark = pytest.mark.gpu
GRAPHS = load_golden("graphs.json")


def test_batch_intersect_count_vs_oracle(gpu, oracle):
    csr = host_graph(gpu, "kronecker", 13, 16, True)
    off, ng = csr.offsets(), csr.neighbors()
    g = gpu.DeviceGraph.from_csr(csr)
    rng = np.random.default_rng(11)
    n = csr.num_nodes
    u = np.concatenate([rng.integers(0, n, 3000), rng.integers(0, 64, 500), np.arange(50)]).astype(np.int32)
    v = np.concatenate([rng.integers(0, n, 3000), rng.integers(0, 64, 500), np.arange(50)]).astype(np.int32)
    got, st = g.intersect_count_batch(u, v, stats=True)
    want = np.array([oracle.intersect_count(ng[off[a]:off[a + 1]], ng[off[b]:off[b + 1]]) for a, b in zip(u, v)], dtype=np.uint32)
    assert np.array_equal(got, want) and st["units"] == u.size
    # symmetric, and |N(u) ∩ N(u)| = d_u
    assert np.array_equal(g.intersect_count_batch(v, u), want)
    assert np.array_equal(g.intersect_count_batch(np.arange(50), np.arange(50)), np.diff(off)[:50].astype(np.uint32))
    assert g.intersect_count_batch([], []).size == 0
    with pytest.raises(gpu.GmsxError) as ei:
        g.intersect_count_batch([0, n], [1, 1])
    assert ei.value.status == gpu.ERR_INVALID
    g.free()


def test_set_op_goldens_through_the_batch_kernel(gpu):
    """The literal SortedSet/RoaringSet cases of testing/sets.cpp: build a graph whose rows are the two sets."""
    for case in load_golden("set_ops.json")["cases"]:
        a, b = sorted(set(case["a"])), sorted(set(case["b"]))
        if not a or not b or max(a + b) > 100000:
            continue
        # vertices 0 and 1 are the two sets; make the graph symmetric by adding the reverse entries
        n = max(a + b) + 3
        x, y = n - 2, n - 1  # two fresh vertices whose rows are exactly a and b
        edges = [(x, t) for t in a] + [(y, t) for t in b]
        csr = edges_to_csr(gpu, edges, n=n)
        g = gpu.DeviceGraph.from_csr(csr)
        assert int(g.intersect_count_batch([x], [y])[0]) == case["expect"]["intersect_count"]
        g.free()


@pytest.mark.parametrize("key", ["kronecker-10-16-relabel", "kronecker-12-16-relabel", "uniform-10-16-relabel", "kronecker-14-16-raw"])
def test_tc_full_is_the_reference_formulation(gpu, oracle, key):
    rec = GRAPHS[key]
    csr = host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
    g = gpu.DeviceGraph.from_csr(csr)
    t, raw = oracle.tc_total(csr.offsets(), csr.neighbors(), raw=True)
    got, st = g.tc_total(gpu.TC_FULL, stats=True)
    assert got == t == g.tc_total(gpu.TC_ORIENTED)
    assert st["units"] == rec["m"]  # exactly m full-row intersect_count calls were executed
    parts = [g.tc_partial(p, 3, gpu.TC_FULL) for p in range(3)]
    assert sum(parts) == raw and gpu.lib().gmsx_tc_divisor(gpu.TC_FULL) == 3
    g.free()


def test_vertex_count2_vs_goldens_and_oracle(gpu, oracle):
    z = np.load(os.path.join(GOLDEN, "vertex_count2.npz"))
    for key in z.files:
        rec = GRAPHS[key]
        g = gpu.DeviceGraph.from_csr(host_graph(gpu, rec["generator"], rec["scale"], rec["degree"], rec["relabel"]))
        c = g.tc_vertex_count2()
        assert np.array_equal(c, z[key]) and int(c.sum()) == 6 * rec["triangles"]
        g.free()
    csr = host_graph(gpu, "uniform", 12, 24, True)
    g = gpu.DeviceGraph.from_csr(csr)
    assert np.array_equal(g.tc_vertex_count2(), oracle.tc_vertex_count2(csr.offsets(), csr.neighbors()))
    g.free()
    e = gpu.DeviceGraph.from_csr(edges_to_csr(gpu, [(0, 1), (1, 2), (2, 0), (2, 3)]))
    assert e.tc_vertex_count2().tolist() == [2, 2, 2, 0]
    e.free()


def test_vertex_similarity_batch(gpu, oracle):
    """gmsx_vertex_similarity_batch vs the reference goldens: count-based metrics bit-identical (NaN positions included);
    Adamic-Adar / resource allocation within 1e-12 relative (different summation order, device log())."""
    z = np.load(os.path.join(GOLDEN, "vertex_similarity.npz"))
    g = gpu.DeviceGraph.from_csr(host_graph(gpu, "kronecker", 10))
    for name in ("jaccard", "overlap", "common_neighbors", "total_neighbors", "pref_attachment"):
        got, st = g.vertex_similarity_batch(name, z["u"], z["v"], stats=True)
        assert np.array_equal(got, z[name], equal_nan=True), name
        assert st["units"] == z["u"].size
    for name in ("adamic_adar", "resource"):
        got, want = g.vertex_similarity_batch(name, z["u"], z["v"]), z[name]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin) and np.array_equal(got[~fin], want[~fin], equal_nan=True)
        assert np.allclose(got[fin], want[fin], rtol=1e-12, atol=0.0), name
    with pytest.raises(gpu.GmsxError) as ei:
        g.vertex_similarity_batch(7, [0], [1])
    assert ei.value.status == gpu.ERR_INVALID
    g.free()
    csr = host_graph(gpu, "uniform", 12, 24, True)
    g = gpu.DeviceGraph.from_csr(csr)
    rng = np.random.default_rng(3)
    u, v = rng.integers(0, 4096, 500), rng.integers(0, 4096, 500)
    for m, name in enumerate(["jaccard", "overlap", "adamic_adar", "resource", "common_neighbors", "total_neighbors", "pref_attachment"]):
        want = oracle.vertex_similarity(m, csr.offsets(), csr.neighbors(), u, v)
        got = g.vertex_similarity_batch(name, u, v)
        assert np.allclose(got, want, rtol=1e-12, atol=0.0, equal_nan=True), name
    g.free()
