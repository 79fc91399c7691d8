#!/usr/bin/env python3
"""Reference-produced goldens at the sizes BASELINE.json / north_star name (run in the BUILD CONTAINER only).

    make -C oracle ref
    python tools/make_golden_big.py tc 26          # Par::count_total<RoaringGraph>, ~35 min on 8 threads, ~45 GB
    python tools/make_golden_big.py tc-sliced 27 16 5   # the same count where the whole RoaringGraph does not fit the host (scale 27: ~90 GB):
                                                   # the reference's RoaringSet::intersect_count per edge, accumulated over 5 id-range
                                                   # slices of the neighbourhoods (oracle/ref_shim.cc ref_tc_total_sliced)
    python tools/make_golden_big.py kc4 16|18|20   # CliqueCount<RoaringSet,RoaringGraph,RoaringSet>(g, 4)
    python tools/make_golden_big.py bk 14          # BkEppsteinPar::mceBench<RoaringGraph>, degree rank
    python tools/make_golden_big.py kclist 24|26   # TRUE 4-clique count (each clique once) by the reference's kClist pipeline
                                                   # (CliqueCountPipeline<true>: getDegeneracyOrderingDanischHeap + InduceDirectedGraph +
                                                   # Par::NP_kclisting, bench_helper.h:33-38,71-77) -> field kc4_true; where the set-based
                                                   # kc4 (= 24 * kc4_true) exists it is asserted equal
    python tools/make_golden_big.py kclist-wide 26 # the same count where the reference's own node-parallel loop overflows (`new NodeId[count*count]`,
                                                   # uint count = a hub's out-degree: segfault at scale 26): reference Preprocess + reference
                                                   # KcListing::count under the loop of oracle/ref_shim.cc ref_kclist_count_wide (64-bit sizes)
    python tools/make_golden_big.py bk-rmat 21 56  # BASELINE configs[3]: the com-Orkut-shaped R-MAT (A=.45 B=C=.22) of the gmsx loader,
                                                   # written as .sg (gmsx_csr_save_sg), read back by the reference (cli.h:96 `-f`,
                                                   # reader.h:252-305) and enumerated by BkEppsteinPar::mceBench<RoaringGraph>

Every value is computed by the COMPILED REFERENCE (oracle/_ref/libgms_ref.so = spcl/gms headers + vendored
CRoaring, see oracle/ref_shim.cc) on the graph its own loader generates (`-g kronecker <scale> --deg 16`,
parse_and_load incl. relabel-by-degree), and merged into tests/golden/graphs.json under the key
`kronecker-<scale>-16-relabel` together with n, m and the wall time of the reference call.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bindings import Reference  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "graphs.json")


def bk_rmat(scale, deg, a=0.45, b=0.22, c=0.22):
    """configs[3]: the graph is NOT one the reference generator can make (fixed Graph500 skew), so the gmsx loader builds it,
    the reference reads the .sg file and counts maximal cliques on exactly that adjacency (relabel off: same ids)."""
    import tempfile
    import numpy as np
    from gms_amd import capi
    R = Reference()
    t0 = time.time()
    csr = capi.HostCSR.generate_rmat(scale, deg, a, b, c)
    n, m, fp = csr.num_nodes, csr.num_edges, csr.fingerprint()
    path = os.path.join(tempfile.gettempdir(), "gmsx_rmat_%d_%d.sg" % (scale, deg))
    csr.save_sg(path)
    del csr
    g = R.load_file(path, relabel=False)
    os.unlink(path)
    assert (R.num_nodes(g), R.nnz(g) // 2) == (n, m)
    print("rmat %d ef %d (%.2f/%.2f/%.2f): n=%d m=%d through %s in %.1f s" % (scale, deg, a, b, c, n, m, path, time.time() - t0), flush=True)
    t0 = time.time()
    val = R.bk_count(g, Reference.ROARING, 0)
    dt = time.time() - t0
    R.free(g)
    key = "rmat-%d-%d-a%02d-b%02d-c%02d" % (scale, deg, round(a * 100), round(b * 100), round(c * 100))
    with open(PATH) as f:
        graphs = json.load(f)
    rec = graphs.setdefault(key, dict(generator="rmat", scale=scale, degree=deg, a=a, b=b, c=c, relabel="auto"))
    if "bk" in rec:
        assert rec["bk"] == val, (rec["bk"], val)
    rec.update(n=n, m=m, offsets_fnv64="%016x" % fp[0], neigh_fnv64="%016x" % fp[1], bk=val)
    rec.setdefault("sources", {})["bk"] = ("oracle/_ref BkEppsteinPar::mceBench<RoaringGraph>, degree rank, on the .sg file written by "
                                           "gmsx_csr_save_sg, %d threads, %.0f s (tools/make_golden_big.py bk-rmat)" % (R.omp_threads(), dt))
    with open(PATH, "w") as f:
        json.dump(graphs, f, indent=1)
    print(json.dumps({key: {"bk": val, "n": n, "m": m, "run_s": round(dt, 1)}}), flush=True)


def main():
    what, scale = sys.argv[1], int(sys.argv[2])
    deg = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    if what == "bk-rmat":
        return bk_rmat(scale, deg)
    R = Reference()
    t0 = time.time()
    g = R.generate("kronecker", scale, deg, True)
    n, m = R.num_nodes(g), R.nnz(g) // 2
    t_load = time.time() - t0
    print("loaded scale %d: n=%d m=%d in %.1f s" % (scale, n, m, t_load), flush=True)
    t0 = time.time()
    if what == "tc":
        field, val, how = "triangles", R.tc_total(g, Reference.ROARING), "Par::count_total<RoaringGraph>"
    elif what == "tc-sliced":
        slices = int(sys.argv[4]) if len(sys.argv) > 4 else 5
        val, build_s, count_s = R.tc_total_sliced(g, slices, times=True)
        field = "triangles"
        how = ("Par::count_total on the reference's RoaringSet, accumulated over %d id-range slices of the neighbourhoods (the whole "
               "RoaringGraph does not fit this host): RoaringSet build %.0f s + RoaringSet::intersect_count per edge %.0f s under the "
               "loop of oracle/ref_shim.cc ref_tc_total_sliced; equal to Par::count_total<RoaringGraph> wherever that fits "
               "(tests/test_oracle.py)" % (slices, build_s, count_s))
    elif what == "kc4":
        field, val, how = "kc4", R.kclique(g, 4, Reference.ROARING), "CliqueCount<RoaringSet,RoaringGraph,RoaringSet>(g,4)"
    elif what == "bk":
        field, val, how = "bk", R.bk_count(g, Reference.ROARING, 0), "BkEppsteinPar::mceBench<RoaringGraph>, degree rank"
    elif what == "kclist":
        val, prep_s, count_s = R.kclist_count(g, 4, 0, times=True)
        field = "kc4_true"
        how = ("kClist CliqueCountPipeline<true,CSRGraph>: Preprocess (DanischHeap degeneracy order + InduceDirectedGraph, %.0f s) + "
               "Par::NP_kclisting k=4 (%.0f s)" % (prep_s, count_s))
    elif what == "kclist-wide":
        val, prep_s, count_s = R.kclist_count_wide(g, 4, times=True)
        field = "kc4_true"
        how = ("kClist: the reference's CliqueCountPipeline::Preprocess (DanischHeap degeneracy order + InduceDirectedGraph, %.0f s) + the "
               "reference's KcListing::count per node subgraph (%.0f s) under the node-parallel loop of oracle/ref_shim.cc "
               "ref_kclist_count_wide — Par::NP_kclisting itself segfaults on this graph (uint `count*count` in SubGraphBuilder.h:49 wraps "
               "for a hub); equal to it wherever it runs (scales 10-20 checked)" % (prep_s, count_s))
    else:
        raise SystemExit("what = tc | tc-sliced | kc4 | bk | kclist | kclist-wide")
    dt = time.time() - t0
    R.free(g)
    key = "kronecker-%d-%d-relabel" % (scale, deg)
    with open(PATH) as f:
        graphs = json.load(f)
    rec = graphs.setdefault(key, dict(generator="kronecker", scale=scale, degree=deg, relabel=True))
    assert rec.get("n", n) == n and rec.get("m", m) == m, (rec, n, m)
    if field in rec:
        assert rec[field] == val, "existing golden %s=%d disagrees with the reference run: %d" % (field, rec[field], val)
    rec.update(n=n, m=m)
    rec[field] = val
    if field == "kc4_true" and "kc4" in rec:
        assert rec["kc4"] == 24 * val, "the reference's two k-clique paths disagree: set-based %d vs 24 * kClist %d" % (rec["kc4"], val)
    if field == "kc4" and "kc4_true" in rec:
        assert val == 24 * rec["kc4_true"], (val, rec["kc4_true"])
    rec.setdefault("sources", {})[field] = "oracle/_ref %s, %d threads, %.0f s (tools/make_golden_big.py)" % (how, R.omp_threads(), dt)
    with open(PATH, "w") as f:
        json.dump(graphs, f, indent=1)
    print(json.dumps({key: {field: val, "n": n, "m": m, "load_s": round(t_load, 1), "run_s": round(dt, 1)}}), flush=True)


if __name__ == "__main__":
    main()
