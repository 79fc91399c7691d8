"""ctypes bindings for the CPU oracle (oracle/libgms_oracle.so) and, when built, the compiled
reference (oracle/_ref/libgms_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (gms_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(ref=True):
    """(Re)build the oracle and, if /root/reference is present, oracle/_ref (see oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", _HERE, "libgms_oracle.so"] + (["ref"] if ref else []), check=True)


class Oracle:
    """oracle/gms_oracle.h"""

    def __init__(self):
        path = os.path.join(_HERE, "libgms_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        sz = C.c_size_t
        L.gmso_intersect_count.restype = C.c_uint64
        L.gmso_intersect_count.argtypes = [_i32p, sz, _i32p, sz]
        for f in (L.gmso_intersect, L.gmso_difference, L.gmso_union):
            f.restype = sz
            f.argtypes = [_i32p, sz, _i32p, sz, _i32p]
        L.gmso_union_count.restype = C.c_uint64
        L.gmso_union_count.argtypes = [_i32p, sz, _i32p, sz]
        L.gmso_make_set.argtypes = [_i32p, sz, _i32p]
        L.gmso_tc_total.restype = C.c_uint64
        L.gmso_tc_total.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, C.POINTER(C.c_uint64)]
        L.gmso_tc_total_sample.restype = C.c_uint64
        L.gmso_tc_total_sample.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, C.c_int64, C.c_int64,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gmso_tc_vertex_count2.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, _i64p]
        L.gmso_tc_vertex_count2_once.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, _i64p]
        L.gmso_kclique.restype = C.c_uint64
        L.gmso_kclique.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, C.c_int]
        L.gmso_kclique_star_count.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gmso_bk_count.restype = C.c_uint64
        L.gmso_bk_count.argtypes = [C.c_int64, _i64p, _i32p, _i32p, C.c_int]
        L.gmso_degree_rank.argtypes = [C.c_int64, _i64p, _i32p]
        L.gmso_adg_rank.restype = C.c_int32
        L.gmso_adg_rank.argtypes = [C.c_int64, _i64p, _i32p, C.c_double, _i32p, _i32p, _i32p]
        L.gmso_tc_ordering.argtypes = [C.c_int64, _i64p, _i32p, C.c_int, _i32p]
        L.gmso_tc_elements.restype = C.c_uint64
        L.gmso_tc_elements.argtypes = [C.c_int64, _i64p, _i32p]
        L.gmso_max_threads.restype = C.c_int
        L.gmso_vertex_similarity.restype = C.c_double
        L.gmso_vertex_similarity.argtypes = [C.c_int, C.c_int64, _i64p, _i32p, C.c_int32, C.c_int32]

    # -- sets ---------------------------------------------------------------
    @staticmethod
    def _s(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    def make_set(self, a):
        a = self._s(a)
        out = np.empty_like(a)
        self.L.gmso_make_set(a, a.size, out)
        return out

    def intersect_count(self, a, b):
        a, b = self._s(a), self._s(b)
        return int(self.L.gmso_intersect_count(a, a.size, b, b.size))

    def _binop(self, fn, a, b):
        a, b = self._s(a), self._s(b)
        out = np.empty(a.size + b.size, dtype=np.int32)
        k = fn(a, a.size, b, b.size, out)
        return out[:k].copy()

    def intersect(self, a, b):
        return self._binop(self.L.gmso_intersect, a, b)

    def difference(self, a, b):
        return self._binop(self.L.gmso_difference, a, b)

    def union(self, a, b):
        return self._binop(self.L.gmso_union, a, b)

    def union_count(self, a, b):
        a, b = self._s(a), self._s(b)
        return int(self.L.gmso_union_count(a, a.size, b, b.size))

    # -- graph kernels --------------------------------------------------------
    def tc_total(self, off, neigh, threads=0, raw=False):
        r = C.c_uint64(0)
        t = int(self.L.gmso_tc_total(off.size - 1, off, neigh, threads, C.byref(r)))
        return (t, int(r.value)) if raw else t

    def tc_total_sample(self, off, neigh, stride, phase=0, threads=0):
        e, el = C.c_uint64(0), C.c_uint64(0)
        s = int(self.L.gmso_tc_total_sample(off.size - 1, off, neigh, threads, stride, phase, C.byref(e), C.byref(el)))
        return s, int(e.value), int(el.value)

    def tc_vertex_count2(self, off, neigh, threads=0, once=False):
        c = np.zeros(off.size - 1, dtype=np.int64)
        (self.L.gmso_tc_vertex_count2_once if once else self.L.gmso_tc_vertex_count2)(off.size - 1, off, neigh, threads, c)
        return c

    def kclique(self, off, neigh, k, threads=0):
        return int(self.L.gmso_kclique(off.size - 1, off, neigh, k, threads))

    def kclique_star_count(self, off, neigh, k, threads=0):
        """(number of k-clique-stars = k-cliques, total cardinality of their stars)"""
        c, m = C.c_uint64(0), C.c_uint64(0)
        self.L.gmso_kclique_star_count(off.size - 1, off, neigh, k, threads, C.byref(c), C.byref(m))
        return int(c.value), int(m.value)

    def degree_rank(self, off):
        r = np.empty(off.size - 1, dtype=np.int32)
        self.L.gmso_degree_rank(off.size - 1, off, r)
        return r

    def bk_count(self, off, neigh, rank=None, threads=0):
        if rank is None:
            rank = self.degree_rank(off)
        return int(self.L.gmso_bk_count(off.size - 1, off, neigh, np.ascontiguousarray(rank, dtype=np.int32), threads))

    def adg_rank(self, off, neigh, epsilon=0.001):
        """(rank, round_of, degree_at_removal, rounds)"""
        n = off.size - 1
        rank, rnd, deg = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(3))
        rounds = int(self.L.gmso_adg_rank(n, off, neigh, float(epsilon), rank, rnd, deg))
        return rank[:n], rnd[:n], deg[:n], rounds

    def tc_ordering(self, off, neigh, threads=0):
        o = np.zeros(max(off.size - 1, 1), dtype=np.int32)
        self.L.gmso_tc_ordering(off.size - 1, off, neigh, threads, o)
        return o[:off.size - 1]

    def vertex_similarity(self, metric, off, neigh, u, v):
        return np.array([self.L.gmso_vertex_similarity(metric, off.size - 1, off, neigh, int(a), int(b)) for a, b in zip(u, v)], dtype=np.float64)

    def tc_elements(self, off, neigh):
        return int(self.L.gmso_tc_elements(off.size - 1, off, neigh))

    def max_threads(self):
        return int(self.L.gmso_max_threads())


REF_PATH = os.path.join(_HERE, "_ref", "libgms_ref.so")


def have_ref():
    return os.path.exists(REF_PATH)


class Reference:
    """oracle/ref_shim.cc — the compiled reference (spcl/gms headers + vendored CRoaring)."""
    SORTED, ROARING = 0, 1
    OPS = {"intersect_count": 0, "intersect": 1, "difference": 2, "union": 3, "union_count": 4,
           "intersect_inplace": 5, "difference_inplace": 6, "union_inplace": 7, "cardinality": 8, "contains": 9}

    def __init__(self):
        L = self.L = C.CDLL(REF_PATH)
        vp = C.c_void_p
        L.ref_graph_generate.restype = vp
        L.ref_graph_generate.argtypes = [C.c_int] * 5
        L.ref_graph_file.restype = vp
        L.ref_graph_file.argtypes = [C.c_char_p, C.c_int]
        L.ref_graph_free.argtypes = [vp]
        L.ref_num_nodes.restype = C.c_int64
        L.ref_num_nodes.argtypes = [vp]
        L.ref_nnz.restype = C.c_int64
        L.ref_nnz.argtypes = [vp]
        L.ref_csr_copy.argtypes = [vp, _i64p, _i32p]
        L.ref_tc_total.restype = C.c_uint64
        L.ref_tc_total.argtypes = [vp, C.c_int, C.c_int]
        L.ref_tc_vertex_count2.argtypes = [vp, C.c_int, C.c_int, _i64p]
        L.ref_kclique.restype = C.c_uint64
        L.ref_kclique.argtypes = [vp, C.c_int, C.c_int]
        L.ref_bk_count.restype = C.c_uint64
        L.ref_bk_count.argtypes = [vp, C.c_int, C.c_int]
        if hasattr(L, "ref_kclique_star"):
            L.ref_kclique_star.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        if hasattr(L, "ref_kclist_count"):
            L.ref_kclist_count.restype = C.c_uint64
            L.ref_kclist_count.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if hasattr(L, "ref_kclist_count_wide"):
            L.ref_kclist_count_wide.restype = C.c_uint64
            L.ref_kclist_count_wide.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if hasattr(L, "ref_tc_total_timed"):
            L.ref_tc_total_timed.restype = C.c_uint64
            L.ref_tc_total_timed.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if hasattr(L, "ref_tc_total_sliced"):
            L.ref_tc_total_sliced.restype = C.c_uint64
            L.ref_tc_total_sliced.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_rank.argtypes = [vp, C.c_int, _i32p]
        if hasattr(L, "ref_tc_ordering"):
            L.ref_tc_ordering.argtypes = [vp, C.c_int, _i32p]
        L.ref_set_op.restype = C.c_int64
        L.ref_set_op.argtypes = [C.c_int, C.c_int, _i32p, C.c_int64, _i32p, C.c_int64, _i32p]
        L.ref_omp_threads.restype = C.c_int
        L.ref_vertex_similarity.argtypes = [vp, C.c_int, C.c_int, C.c_int64, _i32p, _i32p, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]

    def generate(self, kind, scale, deg=16, relabel=True, threads=0):
        return self.L.ref_graph_generate(1 if kind == "uniform" else 0, scale, deg, int(relabel), threads)

    def load_file(self, path, relabel=True):
        return self.L.ref_graph_file(os.fsencode(path), int(relabel))

    def free(self, g):
        self.L.ref_graph_free(g)

    def num_nodes(self, g):
        return int(self.L.ref_num_nodes(g))

    def nnz(self, g):
        return int(self.L.ref_nnz(g))

    def omp_threads(self):
        return int(self.L.ref_omp_threads())

    def csr(self, g):
        n, nnz = self.L.ref_num_nodes(g), self.L.ref_nnz(g)
        off = np.empty(n + 1, dtype=np.int64)
        neigh = np.empty(nnz, dtype=np.int32)
        self.L.ref_csr_copy(g, off, neigh)
        return off, neigh

    def tc_total(self, g, set_kind=0, seq=False):
        return int(self.L.ref_tc_total(g, set_kind, int(seq)))

    def tc_total_timed(self, g, set_kind=0):
        """(triangles, SetGraph build seconds, count seconds): Par::count_total with the phases the reference harness separates
        (common/benchmark.h:105-116: FromCGraph untimed, the trial clock around kernel(sgraph) alone)."""
        b, c = C.c_double(0), C.c_double(0)
        v = int(self.L.ref_tc_total_timed(g, set_kind, C.byref(b), C.byref(c)))
        return v, b.value, c.value

    def tc_total_sliced(self, g, slices, times=False):
        """Par::count_total on RoaringSets, accumulated over `slices` id ranges of the neighbourhoods (ref_shim.cc: ref_tc_total_sliced) —
        the reference's RoaringSet / intersect_count under a restated loop, for graphs whose whole RoaringGraph does not fit the host."""
        b, c = C.c_double(0), C.c_double(0)
        v = int(self.L.ref_tc_total_sliced(g, slices, C.byref(b), C.byref(c)))
        return (v, b.value, c.value) if times else v

    def tc_vertex_count2(self, g, set_kind=0, variant=0):
        c = np.zeros(self.L.ref_num_nodes(g), dtype=np.int64)
        self.L.ref_tc_vertex_count2(g, set_kind, variant, c)
        return c

    def kclique(self, g, k, set_kind=0):
        return int(self.L.ref_kclique(g, k, set_kind))

    def bk_count(self, g, set_kind=1, order=0):
        return int(self.L.ref_bk_count(g, set_kind, order))

    def kclique_star(self, g, k, set_kind=1):
        """Par::CliqueStarList<SGraph>(g, k): (number of (clique, star) pairs listed, total cardinality of the stars)"""
        c, m = C.c_uint64(0), C.c_uint64(0)
        self.L.ref_kclique_star(g, k, set_kind, C.byref(c), C.byref(m))
        return int(c.value), int(m.value)

    def kclist_count(self, g, k, order=0, times=False):
        """TRUE k-clique count (each clique once) by the reference's kClist pipeline (ref_shim.cc: ref_kclist_count).
        order 0 = degeneracy (Danisch heap), 1 = degree, 2 = id."""
        p, c = C.c_double(0), C.c_double(0)
        v = int(self.L.ref_kclist_count(g, k, order, C.byref(p), C.byref(c)))
        return (v, p.value, c.value) if times else v

    def kclist_count_wide(self, g, k, times=False):
        """The same count with the reference's Preprocess and KcListing::count under a node-parallel loop whose subgraph arrays are sized in
        64 bits (ref_shim.cc: ref_kclist_count_wide) — for graphs whose hubs overflow the reference's own `new NodeId[count * count]`."""
        p, c = C.c_double(0), C.c_double(0)
        v = int(self.L.ref_kclist_count_wide(g, k, C.byref(p), C.byref(c)))
        return (v, p.value, c.value) if times else v

    def rank(self, g, order=0):
        r = np.empty(self.L.ref_num_nodes(g), dtype=np.int32)
        self.L.ref_rank(g, order, r)
        return r

    def tc_ordering(self, g, set_kind=0):
        o = np.empty(self.L.ref_num_nodes(g), dtype=np.int32)
        self.L.ref_tc_ordering(g, set_kind, o)
        return o

    def vertex_similarity(self, g, metric, u, v, set_kind=0):
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        out = np.zeros(u.size, dtype=np.float64)
        self.L.ref_vertex_similarity(g, metric, set_kind, u.size, u, v, out)
        return out

    def set_op(self, set_kind, op, a, b):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        out = np.empty(a.size + b.size + 1, dtype=np.int32)
        code = self.OPS[op]
        r = int(self.L.ref_set_op(set_kind, code, a, a.size, b, b.size, out))
        if code in (0, 4, 8, 9):
            return r
        return out[:r].copy()


def fnv1a64(arr):
    """FNV-1a-64 over the raw bytes of `arr` (SURVEY.md Appendix B fingerprint)."""
    data = np.ascontiguousarray(arr).view(np.uint8)
    h = np.uint64(1469598103934665603)
    p = np.uint64(1099511628211)
    # vectorising FNV is not possible (serial dependency); chunk through python ints for small arrays
    hv = int(h)
    pv = int(p)
    mask = (1 << 64) - 1
    for b in data.tobytes():
        hv = ((hv ^ b) * pv) & mask
    return hv
