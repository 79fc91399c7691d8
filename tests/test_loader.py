"""CPU: the host graph substrate (gms_amd/csrc/host/loader.cpp) against the reference loader's fingerprints
(SURVEY.md Appendix B = tests/golden/graphs.json) and, when present, the compiled reference's arrays."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, edges_to_csr, host_graph, load_golden

GRAPHS = load_golden("graphs.json")
FP = [k for k, v in GRAPHS.items() if "offsets_fnv64" in v and v["scale"] <= 16 and v["generator"] != "rmat"]
RMAT = [k for k, v in GRAPHS.items() if v.get("generator") == "rmat" and v["scale"] <= 16]


@pytest.mark.parametrize("key", FP)
def test_fingerprints(capi, key):
    rec = GRAPHS[key]
    csr = host_graph(capi, rec["generator"], rec["scale"], rec["degree"], rec["relabel"])
    fo, fn = csr.fingerprint()
    assert (csr.num_nodes, csr.num_edges) == (rec["n"], rec["m"])
    assert "%016x" % fo == rec["offsets_fnv64"] and "%016x" % fn == rec["neigh_fnv64"]


@pytest.mark.parametrize("key", RMAT)
def test_own_rmat_generator_fingerprints(capi, key):
    """gmsx_csr_generate_rmat (custom skew, outside the reference generator): the reference's Bron–Kerbosch goldens of this family
    (tools/make_golden_big.py bk-rmat) were counted on exactly these arrays, so the generator must keep producing them."""
    rec = GRAPHS[key]
    csr = capi.HostCSR.generate_rmat(rec["scale"], rec["degree"], rec["a"], rec["b"], rec["c"])
    assert (csr.num_nodes, csr.num_edges) == (rec["n"], rec["m"])
    assert ["%016x" % x for x in csr.fingerprint()] == [rec["offsets_fnv64"], rec["neigh_fnv64"]]


def test_fingerprint_scale18_and_thread_independence(capi):
    rec = GRAPHS["kronecker-18-16-relabel"]
    a = capi.HostCSR.generate("kronecker", 18, 16, capi.RELABEL_AUTO, threads=0)
    assert ["%016x" % x for x in a.fingerprint()] == [rec["offsets_fnv64"], rec["neigh_fnv64"]]
    raw = GRAPHS["kronecker-14-16-raw"]
    for t in (1, 3):
        b = capi.HostCSR.generate("kronecker", 14, 16, capi.RELABEL_NEVER, threads=t)
        assert ["%016x" % x for x in b.fingerprint()] == [raw["offsets_fnv64"], raw["neigh_fnv64"]]


def test_rows_are_canonical(capi):
    csr = host_graph(capi, "kronecker", 12)
    off, ng = csr.offsets(), csr.neighbors()
    n = csr.num_nodes
    assert off[0] == 0 and off[-1] == ng.size == 2 * csr.num_edges
    src = np.repeat(np.arange(n), np.diff(off))
    assert (ng != src).all()                                  # no self loops
    same_row = src[1:] == src[:-1]
    assert (ng[1:][same_row] > ng[:-1][same_row]).all()       # strictly ascending rows
    fwd = set(zip(src.tolist(), ng.tolist()))
    assert all((v, u) in fwd for u, v in list(fwd)[:5000])    # symmetric
    deg = np.diff(off)
    assert (deg[:-1] >= deg[1:]).all()                        # relabelled by decreasing degree (builder.h:1699-1733)


def test_relabel_heuristic(capi):
    assert host_graph(capi, "kronecker", 12, relabel=False).worth_relabelling()
    assert not host_graph(capi, "uniform", 10, relabel=False).worth_relabelling()
    raw = host_graph(capi, "kronecker", 10, relabel=False)
    rel = raw.relabel_by_degree()
    assert ["%016x" % x for x in rel.fingerprint()] == [GRAPHS["kronecker-10-16-relabel"][k] for k in ("offsets_fnv64", "neigh_fnv64")]


def test_from_edges_and_files(capi, tmp_path):
    # duplicates, both directions, a self loop, an isolated middle vertex: n = max id + 1 (builder.h:284-285)
    csr = edges_to_csr(capi, [(0, 1), (1, 0), (0, 1), (2, 2), (5, 0)])
    assert csr.num_nodes == 6 and csr.num_edges == 2
    assert csr.offsets().tolist() == [0, 2, 3, 3, 3, 3, 4] and csr.neighbors().tolist() == [1, 5, 0, 0]
    # .el text and .sg binary round trips
    el = tmp_path / "g.el"
    el.write_text("0 1\n1 2\n2 0\n3 4\n")
    a = capi.HostCSR.load(str(el), relabel=capi.RELABEL_NEVER)
    assert a.num_nodes == 5 and a.num_edges == 4
    sg = tmp_path / "g.sg"
    a.save_sg(str(sg))
    b = capi.HostCSR.load(str(sg), relabel=capi.RELABEL_NEVER)
    assert a.fingerprint() == b.fingerprint()
    raw = sg.read_bytes()  # writer.h:39-69: bool directed; int64 nnz; int64 n; offsets; neigh
    assert raw[0] == 0 and np.frombuffer(raw[1:17], dtype=np.int64).tolist() == [8, 5]
    assert len(raw) == 1 + 16 + 6 * 8 + 8 * 4
    big = host_graph(capi, "kronecker", 10)
    big.save_sg(str(tmp_path / "k10.sg"))
    assert capi.HostCSR.load(str(tmp_path / "k10.sg"), relabel=capi.RELABEL_NEVER).fingerprint() == big.fingerprint()


def test_empty_and_errors(capi, tmp_path):
    e = capi.HostCSR.from_edges(np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert e.num_nodes == 1 and e.num_edges == 0          # FindMaxNodeId starts at 0 (builder.h:108-117)
    with pytest.raises(capi.GmsxError) as ei:
        capi.HostCSR.load(str(tmp_path / "missing.el"))
    assert ei.value.status == capi.ERR_IO
    (tmp_path / "x.foo").write_text("0 1\n")
    with pytest.raises(capi.GmsxError) as ei:
        capi.HostCSR.load(str(tmp_path / "x.foo"))
    assert ei.value.status == capi.ERR_FORMAT
    with pytest.raises(capi.GmsxError) as ei:
        capi.HostCSR.generate("kronecker", 31)
    assert ei.value.status == capi.ERR_OVERFLOW
    with pytest.raises(capi.GmsxError) as ei:
        capi.HostCSR.from_arrays(np.array([0, 2, 1], dtype=np.int64), np.array([1, 0], dtype=np.int32))
    assert ei.value.status == capi.ERR_INVALID


def test_reference_test_graph_files_load(capi):
    tg = load_golden("testgraphs.json")
    for name, rec in tg.items():
        csr = capi.HostCSR.load(os.path.join(GOLDEN, "testGraphs", name))
        assert (csr.num_nodes, csr.num_edges) == (rec["n"], rec["m"])


@pytest.mark.parametrize("spec", [("kronecker", 13, 16, True), ("kronecker", 9, 3, True), ("uniform", 12, 16, True), ("kronecker", 16, 16, False)])
def test_arrays_equal_compiled_reference(capi, reference, spec):
    kind, scale, deg, relabel = spec
    g = reference.generate(kind, scale, deg, relabel=relabel)
    try:
        off, ng = reference.csr(g)
        csr = capi.HostCSR.generate(kind, scale, deg, capi.RELABEL_AUTO if relabel else capi.RELABEL_NEVER)
        assert np.array_equal(csr.offsets(), off) and np.array_equal(csr.neighbors(), ng)
    finally:
        reference.free(g)


def test_custom_rmat_skew(capi, tmp_path):
    """gmsx_csr_generate_rmat with the reference's A/B/C reproduces the reference graph; other skews are valid graphs
    that the compiled reference loads back unchanged from the .sg file (when present)."""
    a = capi.HostCSR.generate_rmat(10, 16, 0.57, 0.19, 0.19, capi.RELABEL_AUTO)
    assert ["%016x" % x for x in a.fingerprint()] == [GRAPHS["kronecker-10-16-relabel"][k] for k in ("offsets_fnv64", "neigh_fnv64")]
    b = capi.HostCSR.generate_rmat(11, 20, 0.45, 0.22, 0.22, capi.RELABEL_NEVER)
    assert b.num_nodes <= 2048 and b.num_edges > 15000 and b.fingerprint() != a.fingerprint()
    with pytest.raises(capi.GmsxError):
        capi.HostCSR.generate_rmat(10, 16, 0.6, 0.3, 0.3)
    from oracle import bindings
    if bindings.have_ref():
        R = bindings.Reference()
        p = str(tmp_path / "r.sg")
        b.save_sg(p)
        g = R.load_file(p, relabel=False)
        off, ng = R.csr(g)
        assert np.array_equal(off, b.offsets()) and np.array_equal(ng, b.neighbors())
        O = bindings.Oracle()
        assert R.bk_count(g, 1, 0) == O.bk_count(off, ng) and R.tc_total(g, 0) == O.tc_total(off, ng)
        R.free(g)


def test_other_text_formats(capi, tmp_path):
    """The same small graph through every text format of the reference's Reader (gapbs/reader.h:49-218): identical CSR,
    and identical to what the compiled reference loads from the same files (when present)."""
    edges = [(0, 1), (1, 2), (2, 0), (2, 3), (5, 3), (4, 4)]
    (tmp_path / "g.el").write_text("".join(f"{a} {b}\n" for a, b in edges))
    (tmp_path / "g.wel").write_text("".join(f"{a} {b} {7 + i}\n" for i, (a, b) in enumerate(edges)))
    (tmp_path / "g.gr").write_text("c comment\np sp 6 6\n" + "".join(f"a {a + 1} {b + 1} 3\n" for a, b in edges))
    adj = {i: [] for i in range(6)}
    for a, b in edges:
        adj[a].append(b + 1)
    # METIS rows end in a blank here; "q.graph" has none, which the reference's reader answers by dropping the last
    # neighbour of every row (reader.h:127-136) -- kept, see loader.cpp
    (tmp_path / "g.graph").write_text("% metis\n6 6\n" + "".join("".join(f"{v} " for v in adj[i]) + "\n" for i in range(6)))
    (tmp_path / "q.graph").write_text("% metis\n6 6\n" + "".join(" ".join(map(str, adj[i])) + "\n" for i in range(6)))
    (tmp_path / "g.mtx").write_text("%%MatrixMarket matrix coordinate pattern general\n% c\n6 6 6\n" + "".join(f"{a + 1} {b + 1}\n" for a, b in edges))
    (tmp_path / "w.mtx").write_text("%%MatrixMarket matrix coordinate real symmetric\n6 6 6\n" + "".join(f"{a + 1} {b + 1} 0.5\n" for a, b in edges))
    base = capi.HostCSR.load(str(tmp_path / "g.el"), relabel=capi.RELABEL_NEVER)
    assert base.num_nodes == 6 and base.num_edges == 5  # the self loop 4-4 is dropped, vertex 4 stays isolated
    for name in ("g.wel", "g.gr", "g.graph", "g.mtx", "w.mtx"):
        got = capi.HostCSR.load(str(tmp_path / name), relabel=capi.RELABEL_NEVER)
        assert got.fingerprint() == base.fingerprint(), name
    quirk = capi.HostCSR.load(str(tmp_path / "q.graph"), relabel=capi.RELABEL_NEVER)
    assert quirk.offsets().tolist() == [0, 1, 1, 2] and quirk.neighbors().tolist() == [2, 0]
    (tmp_path / "bad.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n")
    with pytest.raises(capi.GmsxError) as ei:
        capi.HostCSR.load(str(tmp_path / "bad.mtx"))
    assert ei.value.status == capi.ERR_FORMAT
    from oracle import bindings
    if bindings.have_ref():
        R = bindings.Reference()
        for name in ("g.el", "g.gr", "g.graph", "g.mtx"):
            g = R.load_file(str(tmp_path / name), relabel=False)
            off, ng = R.csr(g)
            assert np.array_equal(off, base.offsets()) and np.array_equal(ng, base.neighbors()), name
            R.free(g)
        g = R.load_file(str(tmp_path / "q.graph"), relabel=False)
        off, ng = R.csr(g)
        assert np.array_equal(off, quirk.offsets()) and np.array_equal(ng, quirk.neighbors())
        R.free(g)


def test_host_threads_do_not_change_the_graph(capi):
    """gmsx_set_host_threads only changes how many OpenMP threads the generator / builder use: same CSR bit for bit."""
    base = capi.HostCSR.generate("kronecker", 12, 16).fingerprint()
    prev = capi.set_host_threads(1)
    assert prev >= 1
    try:
        assert capi.HostCSR.generate("kronecker", 12, 16).fingerprint() == base
        assert capi.set_host_threads(3) == 1
        assert capi.HostCSR.generate("kronecker", 12, 16).fingerprint() == base
    finally:
        capi.set_host_threads(prev)


def test_smaller_omp_team_than_requested_builds_the_same_graph(capi):
    """the builder slices the edge list by omp_get_max_threads(); under OMP_THREAD_LIMIT the team is smaller than
    that.  Every slice must still be counted and written: same fingerprint as the golden."""
    import subprocess
    import sys
    from conftest import ROOT, load_golden
    rec = load_golden("graphs.json")["kronecker-12-16-relabel"]
    code = ("import sys; sys.path.insert(0, %r); from gms_amd import capi; capi.set_host_threads(8); "
            "c = capi.HostCSR.generate('kronecker', 12, 16, capi.RELABEL_AUTO, 8); f = c.fingerprint(); "
            "print('%%016x %%016x %%d %%d' %% (f[0], f[1], c.num_nodes, c.num_edges))" % ROOT)
    for limit in ("1", "2", "3"):
        env = dict(os.environ, OMP_THREAD_LIMIT=limit, OMP_DYNAMIC="true")
        out = subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout.split()
        assert out == [rec["offsets_fnv64"], rec["neigh_fnv64"], str(rec["n"]), str(rec["m"])], (limit, out)


def test_sg_reader_rejects_corrupt_files(capi, tmp_path):
    """a stale / truncated / planted .sg cache must be refused by the reader, not walked out of bounds later."""
    import struct
    csr = capi.HostCSR.generate("kronecker", 8, 16)
    good = str(tmp_path / "g.sg")
    csr.save_sg(good)
    raw = bytearray(open(good, "rb").read())
    n, nnz = csr.num_nodes, csr.nnz
    assert struct.unpack_from("<qq", raw, 1) == (nnz, n)
    off0 = 1 + 16
    adj0 = off0 + 8 * (n + 1)

    def variant(name, mutate):
        b = bytearray(raw)
        mutate(b)
        p = str(tmp_path / name)
        open(p, "wb").write(b)
        return p

    cases = {
        "nonmono.sg": lambda b: struct.pack_into("<q", b, off0 + 8 * 5, nnz + 7),          # offsets[5] beyond offsets[6]
        "negid.sg": lambda b: struct.pack_into("<i", b, adj0 + 4 * 3, -1),
        "bigid.sg": lambda b: struct.pack_into("<i", b, adj0 + 4 * 3, n),
        "trunc.sg": lambda b: b.__delitem__(slice(len(b) - 10, len(b))),
        "badend.sg": lambda b: struct.pack_into("<q", b, off0 + 8 * n, nnz - 1),
    }
    for name, mut in cases.items():
        with pytest.raises(capi.GmsxError) as ei:
            capi.HostCSR.load(variant(name, mut), relabel=capi.RELABEL_NEVER)
        assert ei.value.status == capi.ERR_FORMAT, name
    again = capi.HostCSR.load(good, relabel=capi.RELABEL_NEVER)
    assert again.fingerprint() == csr.fingerprint()


def test_sgx_cache_is_mapped_shared_and_validated(capi, tmp_path):
    """the cache the ranks of a multi-GPU run load is MAPPED (gmsx_csr_load of ".sgx"), so N processes share one page-cache
    copy instead of fread-ing N private ones; it holds the same arrays as the reference-format .sg, is refused when corrupt exactly like .sg, and
    a mapped handle behaves like any other (relabel, save, numpy views)."""
    import struct
    import subprocess
    import sys
    csr = capi.HostCSR.generate("kronecker", 10, 16)
    n, nnz = csr.num_nodes, csr.nnz
    p = str(tmp_path / "k10.sgx")
    csr.save_sgx(p)
    raw = bytearray(open(p, "rb").read())
    assert raw[:8] == b"GMSXCSR1" and struct.unpack_from("<qqq", raw, 8) == (0, n, nnz)
    adj0 = (64 + 8 * (n + 1) + 63) // 64 * 64
    assert len(raw) == adj0 + 4 * nnz
    m = capi.HostCSR.load(p, relabel=capi.RELABEL_NEVER)
    assert m.is_mapped and not csr.is_mapped
    assert m.fingerprint() == csr.fingerprint() and m.merge_elements() == csr.merge_elements()
    assert np.array_equal(m.offsets(), csr.offsets()) and np.array_equal(m.neighbors(), csr.neighbors())
    assert m.offsets().ctypes.data % 64 == 0 and m.neighbors().ctypes.data % 64 == 0
    # a mapped graph goes through everything a read one does: .sg out, relabel (a new, owned CSR)
    m.save_sg(str(tmp_path / "back.sg"))
    assert capi.HostCSR.load(str(tmp_path / "back.sg"), relabel=capi.RELABEL_NEVER).fingerprint() == csr.fingerprint()
    r = m.relabel_by_degree()
    assert not r.is_mapped and r.num_edges == csr.num_edges
    # private anonymous memory of a process that maps the cache stays far below the arrays' size (they are file pages, shared)
    big = capi.HostCSR.generate("kronecker", 18, 16)
    pb = str(tmp_path / "k18.sgx")
    big.save_sgx(pb)
    arrays = 8 * (big.num_nodes + 1) + 4 * big.nnz
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gms_amd import capi\n"
            "def anon():\n"
            "    return next(int(l.split()[1]) for l in open('/proc/self/status') if l.startswith('RssAnon')) * 1024\n"
            "a0 = anon(); g = capi.HostCSR.load(%r, relabel=capi.RELABEL_NEVER); s = int(g.neighbors().sum()); print(anon() - a0, g.is_mapped, s)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pb)
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.split()
    assert out[1] == "True" and int(out[0]) < arrays // 4 and int(out[2]) == int(big.neighbors().sum(dtype=np.int64)), (out, arrays)
    # corrupt caches are refused, not walked
    def variant(name, mutate):
        b = bytearray(raw)
        mutate(b)
        q = str(tmp_path / name)
        open(q, "wb").write(b)
        return q
    cases = {
        "magic.sgx": lambda b: b.__setitem__(0, ord("X")),
        "nonmono.sgx": lambda b: struct.pack_into("<q", b, 64 + 8 * 5, nnz + 7),
        "negid.sgx": lambda b: struct.pack_into("<i", b, adj0 + 4 * 3, -1),
        "bigid.sgx": lambda b: struct.pack_into("<i", b, adj0 + 4 * 3, n),
        "trunc.sgx": lambda b: b.__delitem__(slice(len(b) - 10, len(b))),
        "badend.sgx": lambda b: struct.pack_into("<q", b, 64 + 8 * n, nnz - 1),
        "hugennz.sgx": lambda b: struct.pack_into("<q", b, 24, 1 << 61),
    }
    for name, mut in cases.items():
        with pytest.raises(capi.GmsxError) as ei:
            capi.HostCSR.load(variant(name, mut), relabel=capi.RELABEL_NEVER)
        assert ei.value.status == capi.ERR_FORMAT, name
