// libgmsx device layer: process/device binding, graph upload and the device-side construction of the
// set representations (the analogue of SetGraph::FromCGraph, representations/graphs/set_graph.h:86-89,152-181).
// gfx950 only.
#include "device_graph.hpp"

#include <cstdio>
#include <cstring>  // before rocprim: its texture iterator calls the host memset
#include <new>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../host/gmsx_internal.hpp"

namespace gmsx {

Ctx &ctx() {
    static Ctx c;
    return c;
}

int ensure_init() { return ctx().device >= 0 ? GMSX_OK : gmsx_init(-1); }

// ---- preprocessing kernels ------------------------------------------------------------------

__device__ __forceinline__ bool oriented_before(int64_t du, int32_t u, int64_t dv, int32_t v) {
    // u -> v  iff  (deg u, u) < (deg v, v): edges point from the lower- to the higher-degree endpoint,
    // which bounds d+ by O(sqrt(m)) on any graph.
    return du < dv || (du == dv && u < v);
}

// One wave per row: checks the canonical-row invariant and accumulates Σ_{u<v}(d_u+d_v) and max degree.
// flags[0] |= 1 unsorted/duplicate, 2 id out of range, 4 self loop, 8 asymmetric
__global__ __launch_bounds__(256) void k_validate(int64_t n, const int64_t *__restrict__ off,
                                                  const int32_t *__restrict__ adj, int check_symmetry,
                                                  unsigned long long *__restrict__ acc /* [0]=flags [1]=elements [2]=maxdeg */) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned flags = 0;
    unsigned long long elems = 0;
    unsigned long long maxdeg = 0;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const unsigned long long du = (unsigned long long)(e - b);
        if (du > maxdeg) maxdeg = du;
        for (int64_t j = b + lane; j < e; j += 64) {
            const int32_t v = adj[j];
            if (v < 0 || v >= n) { flags |= 2; continue; }
            if (v == u) flags |= 4;
            if (j + 1 < e && adj[j + 1] <= v) flags |= 1;
            const int64_t vb = off[v], ve = off[v + 1];
            if (u < v) elems += du + (unsigned long long)(ve - vb);
            if (check_symmetry) {
                int64_t lo = vb, hi = ve;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (adj[mid] < u) lo = mid + 1; else hi = mid;
                }
                if (lo >= ve || adj[lo] != u) flags |= 8;
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        elems += __shfl_down(elems, s);
        flags |= __shfl_down(flags, s);
        const unsigned long long o = __shfl_down(maxdeg, s);
        maxdeg = o > maxdeg ? o : maxdeg;
    }
    if (lane == 0) {
        if (flags) atomicOr(&acc[0], (unsigned long long)flags);
        if (elems) atomicAdd(&acc[1], elems);
        atomicMax(&acc[2], maxdeg);
    }
}

// d+ per vertex (as int64 so the exclusive scan runs in 64 bits)
__global__ __launch_bounds__(256) void k_out_degree(int64_t n, const int64_t *__restrict__ off,
                                                    const int32_t *__restrict__ adj, int64_t *__restrict__ dplus) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const int64_t du = e - b;
        int c = 0;
        for (int64_t j = b + lane; j < e; j += 64) {
            const int32_t v = adj[j];
            c += oriented_before(du, int32_t(u), off[v + 1] - off[v], v) ? 1 : 0;
        }
        for (int s = 32; s > 0; s >>= 1) c += __shfl_down(c, s);
        if (lane == 0) dplus[u] = c;
    }
}

// order-preserving compaction of the oriented neighbours into dadj
__global__ __launch_bounds__(256) void k_fill_dag(int64_t n, const int64_t *__restrict__ off,
                                                  const int32_t *__restrict__ adj, const int64_t *__restrict__ doff,
                                                  int32_t *__restrict__ dadj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const int64_t du = e - b;
        int64_t out = doff[u];
        for (int64_t base = b; base < e; base += 64) {
            const int64_t j = base + lane;
            bool keep = false;
            int32_t v = 0;
            if (j < e) {
                v = adj[j];
                keep = oriented_before(du, int32_t(u), off[v + 1] - off[v], v);
            }
            const unsigned long long mask = __ballot(keep);
            if (keep) dadj[out + __popcll(mask & ((1ull << lane) - 1ull))] = v;
            out += __popcll(mask);
        }
    }
}

__global__ void k_sort_keys(int64_t n, const int64_t *__restrict__ doff, int32_t *__restrict__ keys, int32_t *__restrict__ vals) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = int32_t(doff[i + 1] - doff[i]);
        vals[i] = int32_t(i);
    }
}

// keys are descending; out[t] = number of keys >= thr[t]
__global__ void k_bin_bounds(int64_t n, const int32_t *__restrict__ keys, int nthr, const int32_t *__restrict__ thr,
                             int64_t *__restrict__ out) {
    const int t = threadIdx.x;
    if (t >= nthr) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] >= thr[t]) lo = mid + 1; else hi = mid;
    }
    out[t] = lo;
}

static int grid_for_waves(int64_t rows) {
    // wave-per-row grid-stride kernels: enough 256-thread blocks to fill the chip a few times over
    const int64_t want = (rows + 3) / 4;
    const int64_t cap = int64_t(ctx().compute_units > 0 ? ctx().compute_units : 256) * 32;
    return int(want < 1 ? 1 : (want > cap ? cap : want));
}

template <class T>
static int dmalloc(T **p, int64_t count, gmsx_graph *g) {
    const size_t bytes = size_t(count > 0 ? count : 1) * sizeof(T);
    if (hipMalloc(reinterpret_cast<void **>(p), bytes) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return GMSX_ERR_DEVICE_MEM;
    }
    if (g) g->device_bytes += int64_t(bytes);
    return GMSX_OK;
}

static void free_graph(gmsx_graph *g) {
    if (!g) return;
    (void)hipFree(g->off);
    (void)hipFree(g->adj);
    (void)hipFree(g->doff);
    (void)hipFree(g->dadj);
    (void)hipFree(g->order);
    (void)hipFree(g->scratch);
    delete g;
}

static int build_device_sets(gmsx_graph *g, uint32_t flags) {
    hipStream_t s = ctx().stream;
    const int64_t n = g->n;
    if (int rc = dmalloc(&g->scratch, 16, g)) return rc;
    GMSX_HIP(hipMemsetAsync(g->scratch, 0, 16 * sizeof(unsigned long long), s));

    // 1. invariant check + Σ(d_u+d_v) + max degree
    const int check_sym = (flags & GMSX_UPLOAD_TRUSTED) ? 0 : 1;
    if (n > 0) hipLaunchKernelGGL(k_validate, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, check_sym, g->scratch);
    unsigned long long acc[3] = {0, 0, 0};
    GMSX_HIP(hipMemcpyAsync(acc, g->scratch, sizeof(acc), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    if (!(flags & GMSX_UPLOAD_TRUSTED) && acc[0] != 0) return GMSX_ERR_NOT_CANONICAL;
    if ((flags & GMSX_UPLOAD_TRUSTED) && (acc[0] & 2)) return GMSX_ERR_NOT_CANONICAL;  // out-of-range ids are never tolerated
    g->alg_elements = acc[1];
    g->max_deg = int32_t(acc[2]);

    // 2. d+ -> exclusive scan -> doff
    int64_t *dplus = nullptr;
    if (int rc = dmalloc(&dplus, n + 1, nullptr)) return rc;
    struct Guard { void *p; ~Guard() { (void)hipFree(p); } } g_dplus{dplus};
    GMSX_HIP(hipMemsetAsync(dplus, 0, size_t(n + 1) * sizeof(int64_t), s));
    if (n > 0) hipLaunchKernelGGL(k_out_degree, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, dplus);
    if (int rc = dmalloc(&g->doff, n + 1, g)) return rc;
    {
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, dplus, g->doff, int64_t(0), size_t(n + 1), rocprim::plus<int64_t>(), s));
        void *tmp = nullptr;
        GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
        Guard g_tmp{tmp};
        GMSX_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, dplus, g->doff, int64_t(0), size_t(n + 1), rocprim::plus<int64_t>(), s));
        GMSX_HIP(hipStreamSynchronize(s));
    }
    int64_t m = 0;
    GMSX_HIP(hipMemcpy(&m, g->doff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    g->m = m;
    if (!(flags & GMSX_UPLOAD_TRUSTED) && m * 2 != g->nnz) return GMSX_ERR_NOT_CANONICAL;

    // 3. DAG rows
    if (int rc = dmalloc(&g->dadj, m + 8, g)) return rc;  // +8: the 16-byte row loads of the count kernels may overrun the last row
    GMSX_HIP(hipMemsetAsync(g->dadj + m, 0xff, 8 * sizeof(int32_t), s));
    if (n > 0) hipLaunchKernelGGL(k_fill_dag, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, g->doff, g->dadj);

    // 4. work-sorted launch order: vertices by decreasing d+, plus the bin boundaries
    if (int rc = dmalloc(&g->order, n, g)) return rc;
    int32_t *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr;
    if (int rc = dmalloc(&keys_in, n, nullptr)) return rc;
    Guard g_ki{keys_in};
    if (int rc = dmalloc(&keys_out, n, nullptr)) return rc;
    Guard g_ko{keys_out};
    if (int rc = dmalloc(&vals_in, n, nullptr)) return rc;
    Guard g_vi{vals_in};
    if (n > 0) {
        hipLaunchKernelGGL(k_sort_keys, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, g->doff, keys_in, vals_in);
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, keys_in, keys_out, vals_in, g->order, size_t(n), 0, 32, s));
        void *tmp = nullptr;
        GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
        Guard g_tmp{tmp};
        GMSX_HIP(rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys_in, keys_out, vals_in, g->order, size_t(n), 0, 32, s));
        int32_t *d_thr = nullptr;
        int64_t *d_out = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_thr), sizeof(kBinThr)));
        Guard g_thr{d_thr};
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(int64_t) * gmsx_graph::kBins));
        Guard g_out{d_out};
        GMSX_HIP(hipMemcpyAsync(d_thr, kBinThr, sizeof(kBinThr), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_bin_bounds, dim3(1), dim3(64), 0, s, n, keys_out, gmsx_graph::kBins, d_thr, d_out);
        GMSX_HIP(hipMemcpyAsync(g->bin_end, d_out, sizeof(int64_t) * gmsx_graph::kBins, hipMemcpyDeviceToHost, s));
        int32_t top = 0;
        GMSX_HIP(hipMemcpyAsync(&top, keys_out, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        g->max_dplus = top;
    }
    GMSX_HIP(hipStreamSynchronize(s));
    GMSX_HIP(hipGetLastError());
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_init(int device) {
    Ctx &c = ctx();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return GMSX_ERR_NO_DEVICE;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
    }
    if (device >= count) return GMSX_ERR_INVALID;
    if (c.device == device) return GMSX_OK;
    if (hipSetDevice(device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
    if (!c.own_stream) {
        if (hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking) != hipSuccess) return GMSX_ERR_NO_DEVICE;
        for (auto &e : c.ev)
            if (hipEventCreate(&e) != hipSuccess) return GMSX_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
    c.compute_units = prop.multiProcessorCount;
    c.stream = c.own_stream;
    c.device = device;
    return GMSX_OK;
}

int gmsx_set_stream(void *hip_stream) {
    if (int rc = ensure_init()) return rc;
    ctx().stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx().own_stream;
    return GMSX_OK;
}

int gmsx_device_info(char *name, size_t name_len, int *compute_units, int64_t *hbm_bytes) {
    if (int rc = ensure_init()) return rc;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx().device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
    if (name && name_len) {
        std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = int64_t(prop.totalGlobalMem);
    return GMSX_OK;
}

int gmsx_graph_upload(int64_t n, const int64_t *offsets, const int32_t *neigh, uint32_t flags, gmsx_graph **out) {
    if (!out || n < 0 || !offsets || n > 0x7fffffffll) return GMSX_ERR_INVALID;
    if (offsets[0] != 0 || offsets[n] < 0 || (offsets[n] > 0 && !neigh)) return GMSX_ERR_INVALID;
    if (int rc = ensure_init()) return rc;
    gmsx_graph *g = new (std::nothrow) gmsx_graph;
    if (!g) return GMSX_ERR_NOMEM;
    g->n = n;
    g->nnz = offsets[n];
    int rc = dmalloc(&g->off, n + 1, g);
    if (!rc) rc = dmalloc(&g->adj, g->nnz, g);
    if (!rc) {
        hipStream_t s = ctx().stream;
        if (hipMemcpyAsync(g->off, offsets, size_t(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess ||
            (g->nnz && hipMemcpyAsync(g->adj, neigh, size_t(g->nnz) * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) ||
            hipStreamSynchronize(s) != hipSuccess) {
            (void)hipGetLastError();
            rc = GMSX_ERR_KERNEL;
        }
    }
    // offsets must be monotone before any kernel walks the rows
    if (!rc) {
        for (int64_t i = 0; i < n; ++i)
            if (offsets[i + 1] < offsets[i]) { rc = GMSX_ERR_INVALID; break; }
    }
    if (!rc) rc = build_device_sets(g, flags);
    if (rc) {
        free_graph(g);
        return rc;
    }
    *out = g;
    return GMSX_OK;
}

int gmsx_graph_upload_csr(const gmsx_csr *h, uint32_t flags, gmsx_graph **out) {
    if (!h) return GMSX_ERR_INVALID;
    if (h->g.directed) return GMSX_ERR_DIRECTED;
    return gmsx_graph_upload(h->g.n, h->g.off.get(), h->g.neigh.get(), flags, out);
}

int gmsx_graph_free(gmsx_graph *g) {
    free_graph(g);
    return GMSX_OK;
}

int64_t gmsx_graph_num_nodes(const gmsx_graph *g) { return g ? g->n : int64_t(GMSX_ERR_INVALID); }
int64_t gmsx_graph_num_edges(const gmsx_graph *g) { return g ? g->m : int64_t(GMSX_ERR_INVALID); }
int64_t gmsx_graph_device_bytes(const gmsx_graph *g) { return g ? g->device_bytes : int64_t(GMSX_ERR_INVALID); }
int32_t gmsx_graph_max_out_degree(const gmsx_graph *g) { return g ? g->max_dplus : int32_t(GMSX_ERR_INVALID); }

int gmsx_version(void) { return GMSX_VERSION; }

const char *gmsx_strerror(int status) {
    switch (status) {
        case GMSX_OK: return "ok";
        case GMSX_ERR_INVALID: return "invalid argument";
        case GMSX_ERR_NOMEM: return "host allocation failed";
        case GMSX_ERR_IO: return "file could not be opened or read";
        case GMSX_ERR_FORMAT: return "unknown suffix or malformed file";
        case GMSX_ERR_DIRECTED: return "directed graph where an undirected one is required";
        case GMSX_ERR_NO_DEVICE: return "no HIP device available (or HIP runtime initialisation failed)";
        case GMSX_ERR_DEVICE_MEM: return "device allocation failed";
        case GMSX_ERR_NOT_CANONICAL: return "CSR rows are not sorted / loop-free / symmetric";
        case GMSX_ERR_OVERFLOW: return "vertex ids do not fit int32";
        case GMSX_ERR_UNSUPPORTED: return "request not supported by this build";
        case GMSX_ERR_KERNEL: return "HIP kernel launch or synchronisation failed";
        default: return "unknown gmsx status";
    }
}

}  // extern "C"
