#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref/libgms_ref.so,
built by oracle/Makefile from /root/reference where it lies).  Run in the build container only:

    make -C oracle ref && python tools/make_golden.py

Everything written is data: inputs + the reference's outputs (+ the tiny .el data files the reference's
own tests hold).  No reference source text is stored.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bindings import Reference, fnv1a64  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF_TESTGRAPHS = "/root/reference/testing/testGraphs"
R = Reference()
S, RO = Reference.SORTED, Reference.ROARING


def graph_record(g, tc=True, kc=(3, 4), bk=True, fingerprint=True):
    off, ng = R.csr(g)
    rec = {"n": int(off.size - 1), "m": int(ng.size // 2)}
    if fingerprint:
        rec["offsets_fnv64"] = "%016x" % fnv1a64(off)
        rec["neigh_fnv64"] = "%016x" % fnv1a64(ng)
    if tc:
        t = R.tc_total(g, S)
        assert t == R.tc_total(g, RO) == R.tc_total(g, S, seq=True)
        rec["triangles"] = t
    for k in kc:
        v = R.kclique(g, k, S)
        assert v == R.kclique(g, k, RO)
        rec["kc%d" % k] = v
    if bk:
        c = R.bk_count(g, RO, 0)
        assert c == R.bk_count(g, S, 0) == R.bk_count(g, RO, 1)
        rec["bk"] = c
    return rec, off, ng


def similarity_goldens():
    """GMS::VertexSim::vertex_similarity<Metric> (vertex_similarity/vertex_similarity.h:205-222) for 7 metrics on random and
    edge-case pairs of the scale-10 Kronecker graph; SortedSetGraph and RoaringGraph agree bit for bit."""
    g = R.generate("kronecker", 10, 16, True)
    off, ng = R.csr(g)
    deg = np.diff(off)
    rng = np.random.default_rng(77)
    iso = np.flatnonzero(deg == 0)[:4]
    u = np.concatenate([rng.integers(0, 1024, 400), np.arange(8), iso, iso[::-1], [0, 1, 2]]).astype(np.int32)
    v = np.concatenate([rng.integers(0, 1024, 400), np.arange(8), iso, np.arange(iso.size), [1023, 1022, 1021]]).astype(np.int32)
    arrays = {"u": u, "v": v}
    for m, name in enumerate(["jaccard", "overlap", "adamic_adar", "resource", "common_neighbors", "total_neighbors", "pref_attachment"]):
        a, b = R.vertex_similarity(g, m, u, v, S), R.vertex_similarity(g, m, u, v, RO)
        assert np.array_equal(a, b, equal_nan=True), name
        arrays[name] = a
    R.free(g)
    np.savez_compressed(os.path.join(OUT, "vertex_similarity.npz"), **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)

    # ---- 1. set algebra vectors ------------------------------------------------------------------
    lit = [
        ([], []), ([], [1, 2, 3]), ([1, 2, 3], []), ([1, 2, 3], [4, 5, 6]), ([1, 2, 3, 4, 5], [3, 4, 5, 6, 7]),
        ([1, 2, 3, 4, 5, 6, 7], [2, 4, 6, 8]), ([1, 2, 3], [1, 2, 3]), ([1, 2, 3, 4, 5], [3, 4, 5, 6, 8]),
        ([3, 4, 5, 6, 8], [1, 2, 3, 4, 5]), ([2, 4, 8], [4, 2, 8]), ([1, 5, 2, 7, 9, 0, 3], [9, 0]), ([7], [7]), ([7], [8]),
        ([0, 65535, 65536, 131071, 131072], [65535, 65536, 200000]),
    ]
    rng = np.random.default_rng(20261002)
    rnd = []
    for na, nb, hi in [(1, 1, 4), (5, 50, 64), (50, 5, 64), (64, 64, 128), (100, 10000, 20000), (3000, 3000, 8000),
                       (4096, 4097, 70000), (5000, 100, 1 << 20), (12000, 12000, 1 << 16), (7, 30000, 1 << 16),
                       (30000, 1000, 1 << 22), (1000, 1000, 1 << 30)]:
        a = rng.choice(hi, size=min(na, hi), replace=False).astype(np.int32)
        b = rng.choice(hi, size=min(nb, hi), replace=False).astype(np.int32)
        rnd.append((a, b))

    def run_case(a, b):
        res = {}
        for kind in (S, RO):
            r = {op: R.set_op(kind, op, a, b) for op in ("intersect_count", "union_count", "cardinality")}
            for op in ("intersect", "difference", "union", "intersect_inplace", "difference_inplace", "union_inplace"):
                r[op] = np.sort(R.set_op(kind, op, a, b)).astype(np.int32)
            res[kind] = r
        for k in res[S]:  # the two set types agree as sets on duplicate-free input
            assert np.array_equal(res[S][k], res[RO][k]), (k, a, b)
        r = res[S]
        assert np.array_equal(r["intersect"], r["intersect_inplace"]) and np.array_equal(r["difference"], r["difference_inplace"])
        assert np.array_equal(r["union"], r["union_inplace"])
        return r

    cases = []
    for a, b in lit:
        r = run_case(a, b)
        cases.append({"a": a, "b": b, "expect": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in r.items()}})
    with open(os.path.join(OUT, "set_ops.json"), "w") as f:
        json.dump({"note": "results of the reference's SortedSet and RoaringSet (identical) for each (a, b); "
                           "*_inplace results equal the out-of-place ones", "cases": cases}, f)
    arrays = {}
    for i, (a, b) in enumerate(rnd):
        r = run_case(a, b)
        arrays["a%d" % i], arrays["b%d" % i] = a, b
        for op in ("intersect", "difference", "union"):
            arrays["%s%d" % (op, i)] = r[op]
        arrays["counts%d" % i] = np.array([r["intersect_count"], r["union_count"], r["cardinality"]], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "set_ops_random.npz"), **arrays)
    if len(sys.argv) > 1 and sys.argv[1] == "--only-sets":
        return
    similarity_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "--only-similarity":
        return

    # ---- 2. generated graphs: fingerprints + counts ---------------------------------------------------
    graphs = {}
    vcounts = {}
    for kind, scale, deg, relabel, opts in [
        ("kronecker", 4, 16, True, dict(kc=(3, 4, 5))),
        ("kronecker", 6, 16, True, dict(kc=(3, 4, 5))),
        ("kronecker", 8, 16, True, dict(kc=(3, 4, 5))),
        ("kronecker", 10, 16, True, dict(kc=(3, 4, 5))),
        ("kronecker", 10, 16, False, dict(kc=(3,), bk=False)),
        ("kronecker", 12, 16, True, dict(kc=(3, 4))),
        ("kronecker", 14, 16, True, dict(kc=(3, 4))),
        ("kronecker", 14, 16, False, dict(kc=(), bk=False, tc=False)),
        ("kronecker", 16, 16, True, dict(kc=(3,), bk=False)),
        ("kronecker", 18, 16, True, dict(kc=(), bk=False)),
        ("kronecker", 18, 16, False, dict(kc=(), bk=False, tc=False)),
        ("kronecker", 12, 4, True, dict(kc=(3, 4))),
        ("kronecker", 11, 40, True, dict(kc=(3, 4), bk=False)),
        ("uniform", 10, 16, True, dict(kc=(3, 4))),
        ("uniform", 14, 16, True, dict(kc=(3, 4))),
        ("uniform", 16, 16, True, dict(kc=(4,))),
    ]:
        g = R.generate(kind, scale, deg, relabel)
        rec, off, ng = graph_record(g, **opts)
        key = "%s-%d-%d-%s" % (kind, scale, deg, "relabel" if relabel else "raw")
        rec.update(generator=kind, scale=scale, degree=deg, relabel=relabel)
        if relabel and scale <= 12 and kind == "kronecker" and deg == 16:
            vcounts[key] = R.tc_vertex_count2(g, S, 0)
            assert (vcounts[key] == R.tc_vertex_count2(g, RO, 1)).all()
            assert (vcounts[key] == R.tc_vertex_count2(g, S, 2)).all()
        graphs[key] = rec
        print(key, rec, flush=True)
        R.free(g)
    # big triangle goldens measured in the survey with the reference; re-measured here up to s20
    g = R.generate("kronecker", 20, 16, True)
    rec, _, _ = graph_record(g, kc=(), bk=False, fingerprint=False)
    R.free(g)
    rec.update(generator="kronecker", scale=20, degree=16, relabel=True)
    graphs["kronecker-20-16-relabel"] = rec
    print("kronecker-20", rec, flush=True)
    graphs["kronecker-22-16-relabel"] = dict(generator="kronecker", scale=22, degree=16, relabel=True, n=4194302, m=64155718,
                                             triangles=2111140967, source="SURVEY.md Appendix A (reference RoaringGraph run)")
    graphs["kronecker-24-16-relabel"] = dict(generator="kronecker", scale=24, degree=16, relabel=True, n=16777212, m=260376709,
                                             triangles=10283205554, merge_elements=3566604866880,
                                             source="SURVEY.md Appendix A (reference RoaringGraph run)")
    with open(os.path.join(OUT, "graphs.json"), "w") as f:
        json.dump(graphs, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "vertex_count2.npz"), **vcounts)

    # ---- 3. the reference's own tiny test graphs (data files) + counts --------------------------------
    tg_dir = os.path.join(OUT, "testGraphs")
    os.makedirs(tg_dir, exist_ok=True)
    tg = {}
    for name in sorted(os.listdir(REF_TESTGRAPHS)):
        if not name.endswith(".el"):
            continue
        shutil.copyfile(os.path.join(REF_TESTGRAPHS, name), os.path.join(tg_dir, name))
        g = R.load_file(os.path.join(REF_TESTGRAPHS, name), relabel=True)
        rec, off, ng = graph_record(g, kc=(3, 4), fingerprint=False)
        rec["offsets"] = off.tolist()
        rec["neigh"] = ng.tolist()
        tg[name] = rec
        R.free(g)
    with open(os.path.join(OUT, "testgraphs.json"), "w") as f:
        json.dump(tg, f, indent=1)

    # ---- 4. known-answer k-clique cases of testing/clique_counting/CliqueCounter2_tests.h:45-271 ---------
    # (edge lists and true counts are the literals of those tests; kc = the set-based reference value k!*count)
    ka = [
        ("Counts2CliquesCorrect", 2, 2, [(0, 1), (1, 2)]),
        ("CountsMany2CliquesCorrect", 2, 5, [(0, 1), (1, 2), (0, 3), (0, 4), (0, 5)]),
        ("CountsNo3Clique", 3, 0, [(0, 1), (1, 2)]),
        ("CountsNo3Clique2", 3, 0, [(0, 1), (1, 2), (0, 3), (0, 4), (0, 5)]),
        ("Counts3Clique", 3, 1, [(0, 1), (1, 2), (2, 0)]),
        ("CountsMany3CliquesCorrect", 3, 6, [(1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 1), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6)]),
        ("CountsNo4Clique", 4, 0, [(1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 1), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6)]),
        ("Counts4CliquesCorrect", 4, 6, [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (2, 3), (2, 4), (2, 5), (2, 6), (3, 4), (3, 7), (4, 8), (5, 6), (6, 7), (7, 8)]),
        ("Counts4CliquesCorrect2", 4, 4, [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (2, 8), (3, 12), (4, 5), (4, 6), (4, 7), (4, 9), (5, 6), (5, 7), (6, 7), (6, 12), (7, 13), (8, 9), (8, 10), (8, 11), (9, 10), (9, 11), (10, 11), (11, 14), (12, 13), (12, 14), (12, 15), (13, 14), (13, 15), (14, 15)]),
    ]
    fact = {2: 2, 3: 6, 4: 24}
    out = []
    with tempfile.TemporaryDirectory() as td:
        for name, k, true_count, edges in ka:
            p = os.path.join(td, name + ".el")
            with open(p, "w") as f:
                f.writelines("%d %d\n" % e for e in edges)
            g = R.load_file(p, relabel=False)
            ordered = R.kclique(g, k, S)
            assert ordered == fact[k] * true_count == R.kclique(g, k, RO), (name, ordered)
            out.append(dict(name=name, k=k, cliques=true_count, ordered=ordered, edges=edges,
                            triangles=R.tc_total(g, S), bk=R.bk_count(g, RO, 0)))
            R.free(g)
    # ---- 5. concrete instances of the randomised BK regression tests (testing/bron_kerbosch.cpp:256-268: G(n, 0.5)) ----
    bk_inst = []
    with tempfile.TemporaryDirectory() as td:
        for n, seed in [(10, 1), (10, 2), (50, 3), (100, 4), (64, 5)]:
            r = np.random.default_rng(seed)
            edges = [(i, j) for i in range(n) for j in range(i + 1, n) if r.random() < 0.5]
            p = os.path.join(td, "g.el")
            with open(p, "w") as f:
                f.writelines("%d %d\n" % e for e in edges)
            g = R.load_file(p, relabel=False)
            c = R.bk_count(g, RO, 0)
            assert c == R.bk_count(g, S, 2) == R.bk_count(g, S, 1)
            bk_inst.append(dict(n=n, seed=seed, edges=edges, bk=c, triangles=R.tc_total(g, S), kc4=R.kclique(g, 4, S)))
            R.free(g)
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump({"kclique": out, "bk_random": bk_inst}, f)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
