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
        # … and the materialised results (round 5): the reference's intersect / difference lists of the same cases
        for op in ("intersect", "difference"):
            o, ids = g.set_op_batch(op, [x], [y])
            assert ids.tolist() == case["expect"][op] and o.tolist() == [0, len(case["expect"][op])], (case["a"], case["b"], op)
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


@pytest.mark.parametrize("maxd", [1, 8, 40, 300])
def test_vertex_count2_on_pivots_wider_than_the_bit_matrix(gpu, oracle, maxd):
    """per-vertex counts of pivots wider than the bit-matrix kernels hold run on the generic list recursion (k_kc_generic, one atomic per
    triangle) — those pivots only; the rest of the graph stays on the bit-matrix kernels.  KC_MAXD lowers the width limit so that ordinary graphs split
    between the two paths at different points: the counts are those of the oracle's Par::vertex_count2 (parallel/vertex.h:14-27) every time."""
    with gpu.options(KC_MAXD=maxd):
        for kind, scale, deg in (("kronecker", 10, 16), ("uniform", 10, 30), ("kronecker", 12, 8)):
            csr = host_graph(gpu, kind, scale, deg, True)
            g = gpu.DeviceGraph.from_csr(csr)
            got, st = g.tc_vertex_count2(stats=True)
            want = oracle.tc_vertex_count2(csr.offsets(), csr.neighbors())
            assert np.array_equal(got, want), (kind, scale, maxd)
            g.free()


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


def test_set_op_batch_materialises_intersect_and_difference(gpu, oracle):
    """gmsx_set_op_batch (round 5): Set::intersect / Set::difference of full rows, MATERIALISED for a batch of pairs (count pass, scan, fill pass).
    Bit-equal to the oracle's merges (oracle/gms_oracle.c: sorted_set_operations.h:16-42, 73-99) and — where the compiled reference is on the box
    — to the reference's own SortedSet and RoaringSet operators; ascending ids, CSR-shaped offsets, hubs / random / equal / isolated vertices."""
    csr = host_graph(gpu, "kronecker", 13, 16, True)
    off, ng = csr.offsets(), csr.neighbors()
    g = gpu.DeviceGraph.from_csr(csr)
    rng = np.random.default_rng(23)
    n = csr.num_nodes
    iso = np.flatnonzero(np.diff(off) == 0)[:20]
    u = np.concatenate([rng.integers(0, n, 2000), rng.integers(0, 64, 300), np.arange(40), iso, rng.integers(0, 64, iso.size)]).astype(np.int32)
    v = np.concatenate([rng.integers(0, n, 2000), rng.integers(0, 64, 300), np.arange(40), rng.integers(0, 64, iso.size), iso]).astype(np.int32)
    ref = None
    try:
        from oracle import bindings
        if bindings.have_ref():
            ref = bindings.Reference()
    except Exception:  # noqa: BLE001
        ref = None
    for op in ("intersect", "difference"):
        o, ids, st = g.set_op_batch(op, u, v, stats=True)
        assert o.shape == (u.size + 1,) and o[0] == 0 and np.all(np.diff(o) >= 0) and ids.size == o[-1] and st["units"] == u.size
        for i, (a, b) in enumerate(zip(u, v)):
            ra, rb = ng[off[a]:off[a + 1]], ng[off[b]:off[b + 1]]
            want = oracle.intersect(ra, rb) if op == "intersect" else oracle.difference(ra, rb)
            got = ids[o[i]:o[i + 1]]
            assert np.array_equal(got, want), (op, i, a, b)
            if ref is not None and i % 37 == 0 and ra.size and rb.size:
                for kind in (ref.SORTED, ref.ROARING):
                    assert np.array_equal(got, ref.set_op(kind, op, ra, rb)), (op, i, kind)
        cnt = g.intersect_count_batch(u, v).astype(np.int64)
        assert np.array_equal(np.diff(o), cnt if op == "intersect" else np.diff(off)[u] - cnt)
    # N(u) ∩ N(u) = N(u), N(u) \ N(u) = {}
    o, ids = g.set_op_batch("intersect", np.arange(30), np.arange(30))
    assert np.array_equal(ids, ng[:off[30]]) and np.array_equal(o, off[:31])
    o, ids = g.set_op_batch("difference", np.arange(30), np.arange(30))
    assert ids.size == 0 and not o.any()
    o, ids = g.set_op_batch("intersect", [], [])
    assert o.tolist() == [0] and ids.size == 0
    # errors: a vertex id outside the graph; a result that does not fit the caller's array (the offsets still say what it needs)
    with pytest.raises(gpu.GmsxError) as ei:
        g.set_op_batch("intersect", [0, n], [1, 1])
    assert ei.value.status == gpu.ERR_INVALID
    import ctypes as C
    uu, vv = np.array([0, 1], np.int32), np.array([1, 0], np.int32)
    o = np.zeros(3, np.int64)
    small = np.zeros(1, np.int32)
    rc = gpu.lib().gmsx_set_op_batch(g._h, gpu.SETOP_INTERSECT, 2, uu, vv, o, small.ctypes.data_as(C.c_void_p), 1, None)
    assert rc == gpu.ERR_INVALID and o[2] == 2 * int(g.intersect_count_batch([0], [1])[0]) > 1
    g.free()


def test_cpp_adaptor_set_op_batch_on_the_device(gpu, tmp_path):
    """The C++ adaptor's gmsx::set_op_batch (include/gmsx_set_graph.hpp) over HipSetGraph and HipRoaringGraph: tests/cpp/test_set_concept.cpp with its
    device block switched on."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "t")
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_set_concept.cpp"),
           "-L", os.path.join(ROOT, "gms_amd", "lib"), "-lgmsx", "-Wl,-rpath," + os.path.join(ROOT, "gms_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, env=dict(os.environ, GMSX_TEST_DEVICE_SET_OPS="1")).stdout
    assert "set concept ok" in out
