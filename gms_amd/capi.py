"""ctypes binding of the gmsx C-ABI (include/gmsx.h -> gms_amd/lib/libgmsx.so).

This is plumbing for tests and bench.py; the product is the shared library.  There is no CPU fallback:
if libgmsx.so is missing this module raises at import of the library handle, and every device entry
point returns GMSX_ERR_NO_DEVICE without a GPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMSX_LIB") or os.path.join(_HERE, "lib", "libgmsx.so")  # GMSX_LIB: an A/B build of the same library (tools/)

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")

GEN_KRONECKER, GEN_UNIFORM = 0, 1
RELABEL_NEVER, RELABEL_AUTO, RELABEL_ALWAYS = 0, 1, 2
TC_AUTO, TC_ORIENTED, TC_FULL = 0, 1, 2
UPLOAD_DEFAULT, UPLOAD_TRUSTED, UPLOAD_FOR_TC = 0, 1, 2
PREPARE_TC = 1
SETOP_INTERSECT, SETOP_DIFFERENCE = 0, 1
OK, ERR_INVALID, ERR_NOMEM, ERR_IO, ERR_FORMAT, ERR_DIRECTED, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
ERR_DEVICE_MEM, ERR_NOT_CANONICAL, ERR_OVERFLOW, ERR_UNSUPPORTED, ERR_KERNEL, ERR_COMM, ERR_TIMEOUT = -7, -8, -9, -10, -11, -12, -13
COMM_ID_BYTES = 128

# every symbol include/gmsx.h declares (tests/test_capi_symbols.py checks the header against this list)
SYMBOLS = [
    "gmsx_strerror", "gmsx_version",
    "gmsx_csr_generate", "gmsx_csr_generate_rmat", "gmsx_csr_from_edges", "gmsx_csr_load", "gmsx_csr_save_sg", "gmsx_csr_save_sgx", "gmsx_csr_is_mapped", "gmsx_csr_from_arrays",
    "gmsx_csr_worth_relabelling", "gmsx_csr_relabel_by_degree", "gmsx_csr_num_nodes", "gmsx_csr_num_edges",
    "gmsx_csr_num_edges_directed", "gmsx_csr_offsets", "gmsx_csr_neighbors", "gmsx_csr_merge_elements",
    "gmsx_csr_fingerprint", "gmsx_csr_free", "gmsx_set_host_threads", "gmsx_set_option", "gmsx_reset_options", "gmsx_option_name",
    "gmsx_init", "gmsx_set_stream", "gmsx_device_info", "gmsx_hbm_read_probe",
    "gmsx_graph_upload", "gmsx_graph_upload_csr", "gmsx_graph_upload_shard", "gmsx_graph_upload_csr_shard", "gmsx_graph_prepare", "gmsx_graph_tc_passes", "gmsx_graph_free", "gmsx_graph_num_nodes", "gmsx_graph_num_edges",
    "gmsx_graph_device_bytes", "gmsx_graph_max_out_degree",
    "gmsx_tc_total", "gmsx_tc_partial", "gmsx_tc_divisor", "gmsx_tc_stream_breakdown", "gmsx_tc_row_histogram", "gmsx_tc_comembership", "gmsx_tc_vertex_count2",
    "gmsx_intersect_count_batch", "gmsx_set_op_batch", "gmsx_vertex_similarity_batch", "gmsx_kclique_count", "gmsx_kclique_partial", "gmsx_kclique_star_count", "gmsx_bk_count", "gmsx_bk_partial",
    "gmsx_adg_rank", "gmsx_tc_ordering",
    "gmsx_comm_unique_id", "gmsx_comm_init", "gmsx_comm_allreduce_u64", "gmsx_comm_rank", "gmsx_comm_size", "gmsx_comm_finalize",
]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("setup_ms", C.c_double), ("units", C.c_uint64),
                ("alg_elements", C.c_uint64), ("probes", C.c_uint64), ("launches", C.c_int32), ("reserved", C.c_int32),
                ("stream_bytes", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class GmsxError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: gmsx status {status} ({lib().gmsx_strerror(status).decode()})")


_LIB = None


def lib():
    """The loaded libgmsx.so; raises if it has not been built (python __graft_entry__.py / make -C gms_amd/csrc)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C gms_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    L.gmsx_strerror.restype = C.c_char_p
    L.gmsx_strerror.argtypes = [C.c_int]
    L.gmsx_csr_generate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vpp]
    L.gmsx_csr_generate_rmat.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, vpp]
    L.gmsx_csr_from_edges.argtypes = [C.c_int64, C.c_int64, _i32p, _i32p, C.c_int, C.c_int, vpp]
    L.gmsx_csr_load.argtypes = [C.c_char_p, C.c_int, C.c_int, vpp]
    L.gmsx_csr_save_sg.argtypes = [vp, C.c_char_p]
    L.gmsx_csr_save_sgx.argtypes = [vp, C.c_char_p]
    L.gmsx_csr_is_mapped.argtypes = [vp]
    L.gmsx_csr_from_arrays.argtypes = [C.c_int64, _i64p, _i32p, vpp]
    L.gmsx_csr_worth_relabelling.argtypes = [vp]
    L.gmsx_csr_relabel_by_degree.argtypes = [vp, vpp]
    for f in (L.gmsx_csr_num_nodes, L.gmsx_csr_num_edges, L.gmsx_csr_num_edges_directed):
        f.restype = C.c_int64
        f.argtypes = [vp]
    L.gmsx_csr_offsets.restype = C.POINTER(C.c_int64)
    L.gmsx_csr_offsets.argtypes = [vp]
    L.gmsx_csr_neighbors.restype = C.POINTER(C.c_int32)
    L.gmsx_csr_neighbors.argtypes = [vp]
    L.gmsx_csr_merge_elements.restype = C.c_uint64
    L.gmsx_csr_merge_elements.argtypes = [vp]
    L.gmsx_csr_fingerprint.restype = C.c_uint64
    L.gmsx_csr_fingerprint.argtypes = [vp, C.c_int]
    L.gmsx_csr_free.argtypes = [vp]
    L.gmsx_init.argtypes = [C.c_int]
    L.gmsx_set_stream.argtypes = [vp]
    L.gmsx_device_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.gmsx_graph_upload.argtypes = [C.c_int64, _i64p, _i32p, C.c_uint32, vpp]
    L.gmsx_graph_upload_csr.argtypes = [vp, C.c_uint32, vpp]
    L.gmsx_graph_upload_shard.argtypes = [C.c_int64, _i64p, _i32p, C.c_uint32, C.c_int, C.c_int, vpp]
    L.gmsx_graph_upload_csr_shard.argtypes = [vp, C.c_uint32, C.c_int, C.c_int, vpp]
    L.gmsx_graph_free.argtypes = [vp]
    L.gmsx_graph_prepare.argtypes = [vp, C.c_uint32]
    L.gmsx_graph_tc_passes.argtypes = [vp]
    for f in (L.gmsx_graph_num_nodes, L.gmsx_graph_num_edges, L.gmsx_graph_device_bytes):
        f.restype = C.c_int64
        f.argtypes = [vp]
    L.gmsx_graph_max_out_degree.restype = C.c_int32
    L.gmsx_graph_max_out_degree.argtypes = [vp]
    sp = C.POINTER(Stats)
    u64p = C.POINTER(C.c_uint64)
    L.gmsx_tc_total.argtypes = [vp, C.c_int, u64p, sp]
    L.gmsx_tc_partial.argtypes = [vp, C.c_int, C.c_int, C.c_int, u64p, sp]
    L.gmsx_tc_divisor.argtypes = [C.c_int]
    L.gmsx_tc_stream_breakdown.argtypes = [vp, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.gmsx_tc_row_histogram.argtypes = [vp, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.gmsx_tc_comembership.argtypes = [vp, C.c_int, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.gmsx_tc_vertex_count2.argtypes = [vp, _i64p, sp]
    L.gmsx_intersect_count_batch.argtypes = [vp, C.c_int64, _i32p, _i32p, _u32p, sp]
    L.gmsx_set_op_batch.argtypes = [vp, C.c_int, C.c_int64, _i32p, _i32p, _i64p, C.c_void_p, C.c_int64, sp]
    L.gmsx_vertex_similarity_batch.argtypes = [vp, C.c_int, C.c_int64, _i32p, _i32p, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), sp]
    L.gmsx_kclique_count.argtypes = [vp, C.c_int, u64p, u64p, sp]
    L.gmsx_kclique_partial.argtypes = [vp, C.c_int, C.c_int, C.c_int, u64p, sp]
    L.gmsx_kclique_star_count.argtypes = [vp, C.c_int, u64p, u64p, sp]
    L.gmsx_bk_count.argtypes = [vp, C.c_void_p, u64p, sp]
    L.gmsx_bk_partial.argtypes = [vp, C.c_void_p, C.c_int, C.c_int, u64p, sp]
    L.gmsx_adg_rank.argtypes = [vp, C.c_double, C.c_int, _i32p, C.POINTER(C.c_int32), sp]
    L.gmsx_tc_ordering.argtypes = [vp, _i32p, sp]
    L.gmsx_comm_unique_id.argtypes = [C.c_char_p]
    L.gmsx_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p, vpp]
    L.gmsx_comm_allreduce_u64.argtypes = [vp, u64p]
    L.gmsx_comm_rank.argtypes = [vp]
    L.gmsx_comm_size.argtypes = [vp]
    L.gmsx_comm_finalize.argtypes = [vp]
    L.gmsx_hbm_read_probe.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.gmsx_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.gmsx_reset_options.restype = None
    L.gmsx_option_name.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    _LIB = L
    # convenience of the tools (tools/*.sh, probes): GMSX_OPT_<NAME>=<value> in the environment of a PYTHON process becomes
    # gmsx_set_option(NAME, value) here, in the binding — the library itself reads no tuning from the environment
    for k, v in os.environ.items():
        if k.startswith("GMSX_OPT_"):
            _check(L.gmsx_set_option(k[len("GMSX_OPT_"):].encode(), v.encode()), f"gmsx_set_option({k})")
    return L


def _check(status, what):
    if status != 0:
        raise GmsxError(status, what)


class HostCSR:
    """gmsx_csr*: the host CSR the loader builds (replaces CSRGraph + Builder + Generator + Reader)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def generate(cls, generator="kronecker", scale=10, degree=16, relabel=RELABEL_AUTO, threads=0):
        h = C.c_void_p()
        gen = GEN_UNIFORM if generator in ("uniform", "u", GEN_UNIFORM) else GEN_KRONECKER
        _check(lib().gmsx_csr_generate(gen, scale, degree, relabel, threads, C.byref(h)), "gmsx_csr_generate")
        return cls(h)

    @classmethod
    def generate_rmat(cls, scale, degree, a, b, c, relabel=RELABEL_AUTO, threads=0):
        h = C.c_void_p()
        _check(lib().gmsx_csr_generate_rmat(scale, degree, a, b, c, relabel, threads, C.byref(h)), "gmsx_csr_generate_rmat")
        return cls(h)

    @classmethod
    def from_edges(cls, src, dst, num_nodes=-1, symmetrize=True, relabel=RELABEL_NEVER):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().gmsx_csr_from_edges(num_nodes, src.size, src, dst, int(symmetrize), relabel, C.byref(h)),
               "gmsx_csr_from_edges")
        return cls(h)

    @classmethod
    def load(cls, path, symmetrize=True, relabel=RELABEL_AUTO):
        h = C.c_void_p()
        _check(lib().gmsx_csr_load(os.fsencode(path), int(symmetrize), relabel, C.byref(h)), "gmsx_csr_load")
        return cls(h)

    @classmethod
    def from_arrays(cls, off, neigh):
        off = np.ascontiguousarray(off, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().gmsx_csr_from_arrays(off.size - 1, off, neigh if neigh.size else np.zeros(1, np.int32), C.byref(h)),
               "gmsx_csr_from_arrays")
        return cls(h)

    def save_sg(self, path):
        _check(lib().gmsx_csr_save_sg(self._h, os.fsencode(path)), "gmsx_csr_save_sg")

    def save_sgx(self, path):
        """The mappable cache form (gmsx_csr_save_sgx); HostCSR.load of a ".sgx" maps it instead of reading it."""
        _check(lib().gmsx_csr_save_sgx(self._h, os.fsencode(path)), "gmsx_csr_save_sgx")

    @property
    def is_mapped(self):
        return bool(lib().gmsx_csr_is_mapped(self._h))

    def relabel_by_degree(self):
        h = C.c_void_p()
        _check(lib().gmsx_csr_relabel_by_degree(self._h, C.byref(h)), "gmsx_csr_relabel_by_degree")
        return HostCSR(h)

    def worth_relabelling(self):
        return bool(lib().gmsx_csr_worth_relabelling(self._h))

    @property
    def num_nodes(self):
        return lib().gmsx_csr_num_nodes(self._h)

    @property
    def num_edges(self):
        return lib().gmsx_csr_num_edges(self._h)

    @property
    def nnz(self):
        return lib().gmsx_csr_num_edges_directed(self._h)

    def offsets(self):
        """Zero-copy numpy view (valid while this object lives)."""
        return np.ctypeslib.as_array(lib().gmsx_csr_offsets(self._h), shape=(self.num_nodes + 1,))

    def neighbors(self):
        nnz = self.nnz
        if nnz == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(lib().gmsx_csr_neighbors(self._h), shape=(nnz,))

    def merge_elements(self):
        return int(lib().gmsx_csr_merge_elements(self._h))

    def fingerprint(self):
        return int(lib().gmsx_csr_fingerprint(self._h, 0)), int(lib().gmsx_csr_fingerprint(self._h, 1))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gmsx_csr_free(self._h)
            self._h = None


def set_host_threads(n=0):
    """Threads of the host substrate (OpenMP); n <= 0 = all processors.  Returns the previous maximum."""
    return int(lib().gmsx_set_host_threads(int(n)))


def set_option(name, value):
    """gmsx_set_option: value None = back to the default.  Unknown names raise (GMSX_ERR_INVALID)."""
    _check(lib().gmsx_set_option(name.encode(), None if value is None else str(value).encode()), f"gmsx_set_option({name})")


def reset_options():
    lib().gmsx_reset_options()


def option_names():
    out, i, p = [], 0, C.c_char_p()
    while lib().gmsx_option_name(i, C.byref(p)) == 0:
        out.append(p.value.decode())
        i += 1
    return out


class options:
    """with capi.options(KC_MAXD=8, BK_BUDGET=64): … — sets the options for the block and restores the defaults after it."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, None)
        return False


def init(device=-1):
    _check(lib().gmsx_init(device), "gmsx_init")


def set_stream(stream_ptr):
    _check(lib().gmsx_set_stream(C.c_void_p(stream_ptr)), "gmsx_set_stream")


def hbm_read_probe(nbytes=4 << 30, iterations=20):
    """GB/s of a read-only stream over a buffer of `nbytes` (gmsx_hbm_read_probe)."""
    v = C.c_double(0)
    _check(lib().gmsx_hbm_read_probe(int(nbytes), int(iterations), C.byref(v)), "gmsx_hbm_read_probe")
    return v.value


def device_info():
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(0), C.c_int64(0)
    _check(lib().gmsx_device_info(name, 256, C.byref(cu), C.byref(mem)), "gmsx_device_info")
    return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}


class DeviceGraph:
    """gmsx_graph*: the HBM-resident graph (replaces SetGraph<Set>::FromCGraph)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def upload(cls, off, neigh, flags=UPLOAD_DEFAULT):
        off = np.ascontiguousarray(off, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().gmsx_graph_upload(off.size - 1, off, neigh if neigh.size else np.zeros(1, np.int32), flags, C.byref(h)),
               "gmsx_graph_upload")
        return cls(h)

    @classmethod
    def from_csr(cls, csr, flags=UPLOAD_DEFAULT, shard=None):
        """shard = (part, nparts): gmsx_graph_upload_csr_shard — the triangle-count containers of that shard's pivots only."""
        h = C.c_void_p()
        if shard is None:
            _check(lib().gmsx_graph_upload_csr(csr._h, flags, C.byref(h)), "gmsx_graph_upload_csr")
        else:
            _check(lib().gmsx_graph_upload_csr_shard(csr._h, flags, int(shard[0]), int(shard[1]), C.byref(h)), "gmsx_graph_upload_csr_shard")
        return cls(h)

    num_nodes = property(lambda self: lib().gmsx_graph_num_nodes(self._h))
    num_edges = property(lambda self: lib().gmsx_graph_num_edges(self._h))
    device_bytes = property(lambda self: lib().gmsx_graph_device_bytes(self._h))
    max_out_degree = property(lambda self: lib().gmsx_graph_max_out_degree(self._h))
    tc_passes = property(lambda self: lib().gmsx_graph_tc_passes(self._h))

    def prepare(self, what=PREPARE_TC):
        """gmsx_graph_prepare: build the optional containers (triangle-count task lists) now instead of on first use."""
        _check(lib().gmsx_graph_prepare(self._h, what), "gmsx_graph_prepare")

    def tc_total(self, algo=TC_AUTO, stats=False):
        out, st = C.c_uint64(0), Stats()
        _check(lib().gmsx_tc_total(self._h, algo, C.byref(out), C.byref(st)), "gmsx_tc_total")
        return (int(out.value), st.as_dict()) if stats else int(out.value)

    def tc_partial(self, part, nparts, algo=TC_AUTO, stats=False):
        out, st = C.c_uint64(0), Stats()
        _check(lib().gmsx_tc_partial(self._h, algo, part, nparts, C.byref(out), C.byref(st)), "gmsx_tc_partial")
        return (int(out.value), st.as_dict()) if stats else int(out.value)

    BREAKDOWN = ["hub_rows_list", "hub_rows_bitset", "hub_rows_delta", "tail_rows_list", "tail_rows_delta", "entries", "pivot_containers",
                 "of_which_inline_rows", "light_streamed_hub_rows", "light_streamed_tail_rows", "light_pivot_lists_and_descriptors",
                 "count_entries", "count_inline_entries", "count_work_items", "count_light_streamed_members",
                 "reserved15", "reserved16", "reserved17", "reserved18", "reserved19", "reserved20"]
    BREAKDOWN_BYTES = ("hub_rows_list", "hub_rows_bitset", "hub_rows_delta", "tail_rows_list", "tail_rows_delta", "entries", "pivot_containers",
                       "light_streamed_hub_rows", "light_streamed_tail_rows", "light_pivot_lists_and_descriptors")  # these add up to stats.stream_bytes

    def tc_stream_breakdown(self):
        out = np.zeros(21, dtype=np.uint64)
        _check(lib().gmsx_tc_stream_breakdown(self._h, out), "gmsx_tc_stream_breakdown")
        return dict(zip(self.BREAKDOWN, (int(x) for x in out)))

    def tc_row_histogram(self):
        """(hist[5 classes][24 bins][rows, units], light[8]) — see gmsx_tc_row_histogram."""
        out = np.zeros(256, dtype=np.uint64)
        _check(lib().gmsx_tc_row_histogram(self._h, out), "gmsx_tc_row_histogram")
        return out[:240].reshape(5, 24, 2).astype(np.int64), out[240:].astype(np.int64)

    def tc_comembership(self, batch):
        """gmsx_tc_comembership: dict(entries, units, distinct_rows, batched_units, items) for batches of `batch` consecutive heavy pivots."""
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().gmsx_tc_comembership(self._h, int(batch), out), "gmsx_tc_comembership")
        return dict(entries=int(out[0]), units=int(out[1]), distinct_rows=int(out[2]), batched_units=int(out[3]), items=int(out[4]))

    def tc_vertex_count2(self, stats=False):
        c, st = np.zeros(self.num_nodes, dtype=np.int64), Stats()
        _check(lib().gmsx_tc_vertex_count2(self._h, c, C.byref(st)), "gmsx_tc_vertex_count2")
        return (c, st.as_dict()) if stats else c

    def intersect_count_batch(self, u, v, stats=False):
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        out, st = np.zeros(max(u.size, 1), dtype=np.uint32), Stats()
        _check(lib().gmsx_intersect_count_batch(self._h, u.size, u if u.size else np.zeros(1, np.int32),
                                                v if v.size else np.zeros(1, np.int32), out, C.byref(st)),
               "gmsx_intersect_count_batch")
        out = out[:u.size]
        return (out, st.as_dict()) if stats else out

    def set_op_batch(self, op, u, v, stats=False):
        """gmsx_set_op_batch: (offsets[n_pairs + 1], ids) of N(u[i]) ∩ N(v[i]) (op = "intersect") or N(u[i]) \\ N(v[i]) ("difference"), ascending — the sizing
        call first, then the fill into an array of exactly that size."""
        code = {"intersect": SETOP_INTERSECT, "difference": SETOP_DIFFERENCE}[op]
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        uu, vv = (u if u.size else np.zeros(1, np.int32)), (v if v.size else np.zeros(1, np.int32))
        off, st = np.zeros(u.size + 1, dtype=np.int64), Stats()
        _check(lib().gmsx_set_op_batch(self._h, code, u.size, uu, vv, off, None, 0, C.byref(st)), "gmsx_set_op_batch (sizing)")
        ids = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
        _check(lib().gmsx_set_op_batch(self._h, code, u.size, uu, vv, off, ids.ctypes.data_as(C.c_void_p), int(off[-1]), C.byref(st)), "gmsx_set_op_batch")
        ids = ids[:int(off[-1])]
        return (off, ids, st.as_dict()) if stats else (off, ids)

    SIM = {"jaccard": 0, "overlap": 1, "adamic_adar": 2, "resource": 3, "common_neighbors": 4, "total_neighbors": 5, "pref_attachment": 6}

    def vertex_similarity_batch(self, metric, u, v, stats=False):
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        out, st = np.zeros(max(u.size, 1), dtype=np.float64), Stats()
        m = self.SIM[metric] if isinstance(metric, str) else int(metric)
        _check(lib().gmsx_vertex_similarity_batch(self._h, m, u.size, u if u.size else np.zeros(1, np.int32),
                                                  v if v.size else np.zeros(1, np.int32), out, C.byref(st)), "gmsx_vertex_similarity_batch")
        out = out[:u.size]
        return (out, st.as_dict()) if stats else out

    def kclique_count(self, k, stats=False):
        ordered, cliques, st = C.c_uint64(0), C.c_uint64(0), Stats()
        _check(lib().gmsx_kclique_count(self._h, k, C.byref(ordered), C.byref(cliques), C.byref(st)), "gmsx_kclique_count")
        r = (int(ordered.value), int(cliques.value))
        return (r + (st.as_dict(),)) if stats else r

    def kclique_star_count(self, k, members=True, stats=False):
        """KCliqueStar::Par::CliqueStar in count mode: (number of k-clique-stars = C_k, total cardinality of the stars = (k+1) C_{k+1})."""
        stars, mem, st = C.c_uint64(0), C.c_uint64(0), Stats()
        _check(lib().gmsx_kclique_star_count(self._h, k, C.byref(stars), C.byref(mem) if members else None, C.byref(st)), "gmsx_kclique_star_count")
        r = (int(stars.value), int(mem.value) if members else None)
        return (r + (st.as_dict(),)) if stats else r

    def kclique_partial(self, k, part, nparts, stats=False):
        out, st = C.c_uint64(0), Stats()
        _check(lib().gmsx_kclique_partial(self._h, k, part, nparts, C.byref(out), C.byref(st)), "gmsx_kclique_partial")
        return (int(out.value), st.as_dict()) if stats else int(out.value)

    def bk_count(self, rank=None, stats=False):
        out, st = C.c_uint64(0), Stats()
        rp = None
        if rank is not None:
            rank = np.ascontiguousarray(rank, dtype=np.int32)
            rp = rank.ctypes.data_as(C.c_void_p)
        _check(lib().gmsx_bk_count(self._h, rp, C.byref(out), C.byref(st)), "gmsx_bk_count")
        return (int(out.value), st.as_dict()) if stats else int(out.value)

    def bk_partial(self, part, nparts, rank=None, stats=False):
        out, st = C.c_uint64(0), Stats()
        rp = None
        if rank is not None:
            rank = np.ascontiguousarray(rank, dtype=np.int32)
            rp = rank.ctypes.data_as(C.c_void_p)
        _check(lib().gmsx_bk_partial(self._h, rp, part, nparts, C.byref(out), C.byref(st)), "gmsx_bk_partial")
        return (int(out.value), st.as_dict()) if stats else int(out.value)

    def adg_rank(self, epsilon=0.001, rank_format=True, stats=False):
        """gmsx_adg_rank: (rank or order vector, number of peeling rounds)"""
        out, rounds, st = np.zeros(max(self.num_nodes, 1), dtype=np.int32), C.c_int32(0), Stats()
        _check(lib().gmsx_adg_rank(self._h, float(epsilon), int(bool(rank_format)), out, C.byref(rounds), C.byref(st)), "gmsx_adg_rank")
        r = (out[:self.num_nodes], int(rounds.value))
        return (r + (st.as_dict(),)) if stats else r

    def tc_ordering(self, stats=False):
        out, st = np.zeros(max(self.num_nodes, 1), dtype=np.int32), Stats()
        _check(lib().gmsx_tc_ordering(self._h, out, C.byref(st)), "gmsx_tc_ordering")
        out = out[:self.num_nodes]
        return (out, st.as_dict()) if stats else out

    def free(self):
        if getattr(self, "_h", None):
            lib().gmsx_graph_free(self._h)
            self._h = None

    def __del__(self):
        self.free()


class Comm:
    """gmsx_comm*: the native RCCL communicator of the path's single collective (one u64 all-reduce)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib().gmsx_comm_unique_id(buf), "gmsx_comm_unique_id")
        return buf.raw

    @classmethod
    def init(cls, rank, nranks, uid):
        assert len(uid) == COMM_ID_BYTES
        h = C.c_void_p()
        _check(lib().gmsx_comm_init(rank, nranks, uid, C.byref(h)), "gmsx_comm_init")
        return cls(h)

    rank = property(lambda self: lib().gmsx_comm_rank(self._h))
    size = property(lambda self: lib().gmsx_comm_size(self._h))

    def allreduce_u64(self, value):
        v = C.c_uint64(value & 0xFFFFFFFFFFFFFFFF)
        _check(lib().gmsx_comm_allreduce_u64(self._h, C.byref(v)), "gmsx_comm_allreduce_u64")
        return int(v.value)

    def finalize(self):
        if getattr(self, "_h", None):
            _check(lib().gmsx_comm_finalize(self._h), "gmsx_comm_finalize")
            self._h = None
