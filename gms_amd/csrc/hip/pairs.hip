// Full-row set intersection on gfx950: the generic Set::intersect_count over CSR neighbourhoods
//   SortedSetBase::intersect_count   gms/representations/sets/sorted_set.h:176-182 -> sorted_set_operations.h:44-71
//   RoaringSet::intersect_count      gms/representations/sets/roaring_set.h:144-152
// and the two reference loops that are nothing but "one full-row intersect_count per edge":
//   TriangleCount::Par::count_total       triangle_count/parallel/total.h:7-24     (GMSX_TC_FULL: reference-verbatim, total/3)
//   TriangleCount::Par::vertex_count2     triangle_count/parallel/vertex.h:14-27   (counts[u] = Σ_{v∈N(u)} |N(u)∩N(v)|)
//
// One wave per pair: the 64 lanes stream the SHORTER row with coalesced loads and binary-search each id in the
// longer row (wave-cooperative binary search; both rows are sorted), matches are counted with ballot + popcount.
// This is the work-efficient form of the reference's two-pointer merge (Σ min(d_u,d_v)·log max(d_u,d_v) probes instead
// of Σ (d_u+d_v) merge steps); rows stay in L2/MALL because CSR rows are re-used across the pairs of a hub.
#include "device_graph.hpp"

#include <algorithm>

#include <rocprim/device/device_scan.hpp>

namespace gmsx {

// |A ∩ B| for two ascending rows; wave-uniform arguments; returns the wave-uniform count
__device__ __forceinline__ uint32_t wave_intersect_count(const int32_t *__restrict__ a, int64_t la, const int32_t *__restrict__ b,
                                                         int64_t lb, int lane) {
    if (la > lb) {  // stream the shorter, search the longer
        const int32_t *t = a; a = b; b = t;
        const int64_t tl = la; la = lb; lb = tl;
    }
    uint32_t cnt = 0;
    for (int64_t base = 0; base < la; base += 64) {
        const int64_t i = base + lane;
        bool hit = false;
        if (i < la) {
            const int32_t x = a[i];
            int64_t lo = 0, hi = lb;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (b[mid] < x) lo = mid + 1; else hi = mid;
            }
            hit = lo < lb && b[lo] == x;
        }
        cnt += uint32_t(__popcll(__ballot(hit)));
    }
    return cnt;
}

// out[i] = |N(u[i]) ∩ N(v[i])|
__global__ __launch_bounds__(256) void k_pair_batch(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, int64_t n,
                                                    int64_t n_pairs, const int32_t *__restrict__ pu, const int32_t *__restrict__ pv,
                                                    uint32_t *__restrict__ out, unsigned long long *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t p = wave0; p < n_pairs; p += nwaves) {
        const int32_t u = pu[p], v = pv[p];
        if (u < 0 || v < 0 || u >= n || v >= n) {
            if (lane == 0) {
                out[p] = 0;
                atomicOr(&flags[0], 1ull);
            }
            continue;
        }
        const uint32_t c = wave_intersect_count(adj + off[u], off[u + 1] - off[u], adj + off[v], off[v + 1] - off[v], lane);
        if (lane == 0) out[p] = c;
    }
}

// Materialising set operations on full rows (round 5; the base of the LISTING consumers — BK's non-count mode, tomita.h:51-86, the k-clique-star
// output, k_clique_star_list/parallel/output.h:14-68 — which need the sets themselves, not their sizes):
//   SortedSetBase::intersect / difference    gms/representations/sets/sorted_set.h:160-197 -> sorted_set_operations.h:16-42, 73-99
// for a batch of vertex pairs, CSR-shaped: pass 1 counts (k_pair_batch above, |A \ B| = |A| - |A ∩ B|), an exclusive scan places the results,
// pass 2 — this kernel — streams one row again (the shorter one for an intersection: either order gives the same ascending result; N(u) for
// N(u) \ N(v)), binary-searches the other and writes the kept ids behind the wave's running offset by ballot + prefix popcount: ascending, as the
// reference's merges emit them.
template <bool DIFF>
__global__ __launch_bounds__(256) void k_pair_fill(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, int64_t n, int64_t n_pairs,
                                                   const int32_t *__restrict__ pu, const int32_t *__restrict__ pv, const int64_t *__restrict__ out_off,
                                                   int32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t p = wave0; p < n_pairs; p += nwaves) {
        const int32_t u = pu[p], v = pv[p];
        if (u < 0 || v < 0 || u >= n || v >= n) continue;  // (flagged by the count pass: the call fails)
        const int32_t *a = adj + off[u], *b = adj + off[v];
        int64_t la = off[u + 1] - off[u], lb = off[v + 1] - off[v];
        if (!DIFF && la > lb) {  // stream the shorter, search the longer
            const int32_t *t = a; a = b; b = t;
            const int64_t tl = la; la = lb; lb = tl;
        }
        int64_t at = out_off[p];
        for (int64_t base = 0; base < la; base += 64) {
            const int64_t i = base + lane;
            bool keep = false;
            int32_t x = 0;
            if (i < la) {
                x = a[i];
                int64_t lo = 0, hi = lb;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (b[mid] < x) lo = mid + 1; else hi = mid;
                }
                const bool hit = lo < lb && b[lo] == x;
                keep = DIFF ? !hit : hit;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) out[at + __popcll(m & lt)] = x;
            at += __popcll(m);
        }
    }
}
// counts of pass 1 -> sizes of the results (uint32 -> int64, |A \ B| = |A| - |A ∩ B|), scanned in place by the caller
__global__ void k_pair_sizes(const int64_t *__restrict__ off, int64_t n, int64_t n_pairs, const int32_t *__restrict__ pu, const int32_t *__restrict__ pv,
                             const uint32_t *__restrict__ cnt, int diff, int64_t *__restrict__ sizes) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p > n_pairs) return;
    if (p == n_pairs) { sizes[p] = 0; return; }
    const int32_t u = pu[p];
    const bool ok = u >= 0 && u < n && pv[p] >= 0 && pv[p] < n;
    sizes[p] = !ok ? 0 : diff ? (off[u + 1] - off[u]) - int64_t(cnt[p]) : int64_t(cnt[p]);
}

// Vertex similarity (vertex_similarity/vertex_similarity.h:30-222): one wave per pair.  Count-based metrics reuse
// wave_intersect_count; Adamic-Adar / resource allocation add a per-common-neighbour term while intersecting.
__global__ __launch_bounds__(256) void k_pair_similarity(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, int64_t n,
                                                         int metric, int64_t n_pairs, const int32_t *__restrict__ pu,
                                                         const int32_t *__restrict__ pv, double *__restrict__ out,
                                                         unsigned long long *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t p = wave0; p < n_pairs; p += nwaves) {
        const int32_t u = pu[p], v = pv[p];
        if (u < 0 || v < 0 || u >= n || v >= n) {
            if (lane == 0) {
                out[p] = 0.0;
                atomicOr(&flags[0], 1ull);
            }
            continue;
        }
        const int32_t *a = adj + off[u], *b = adj + off[v];
        int64_t la = off[u + 1] - off[u], lb = off[v + 1] - off[v];
        const double ca = double(la), cb = double(lb);
        double r;
        if (metric == GMSX_SIM_ADAMIC_ADAR || metric == GMSX_SIM_RESOURCE) {
            if (la > lb) {
                const int32_t *t = a; a = b; b = t;
                const int64_t tl = la; la = lb; lb = tl;
            }
            double sum = 0.0;
            for (int64_t base = 0; base < la; base += 64) {
                const int64_t i = base + lane;
                if (i < la) {
                    const int32_t x = a[i];
                    int64_t lo = 0, hi = lb;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (b[mid] < x) lo = mid + 1; else hi = mid;
                    }
                    if (lo < lb && b[lo] == x) {
                        const double deg = double(off[x + 1] - off[x]);
                        sum += metric == GMSX_SIM_ADAMIC_ADAR ? 1.0 / log(deg) : 1.0 / deg;
                    }
                }
            }
            for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
            r = sum;
        } else {
            const double cnt = double(wave_intersect_count(a, la, b, lb, lane));
            switch (metric) {
                case GMSX_SIM_JACCARD: r = (la == 0 && lb == 0) ? 1.0 : cnt / (ca + cb + cnt); break;   // sic, vertex_similarity.h:31-36
                case GMSX_SIM_OVERLAP: r = cnt / (ca < cb ? ca : cb); break;                              // :66-68
                case GMSX_SIM_COMMON_NEIGHBORS: r = cnt; break;                                           // :139-143
                case GMSX_SIM_TOTAL_NEIGHBORS: r = ca + cb - cnt; break;                                  // union_count, :155-159
                default: r = ca * cb; break;                                                              // :171-174
            }
        }
        if (lane == 0) out[p] = r;
    }
}

// One wave per CSR entry e = (u -> v).  MODE 0: u < v only, Σ into acc (the reference's count_total sum, /3 on the host).
// MODE 1: every entry, counts[u] += |N(u) ∩ N(v)| (vertex_count2).  Entries [first, end) of the shard.
template <int MODE>
__global__ __launch_bounds__(256) void k_edge_pairs(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, int64_t n,
                                                    int64_t first, int64_t end, unsigned long long *__restrict__ acc,
                                                    unsigned long long *__restrict__ counts) {
    __shared__ unsigned long long red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long total = 0, units = 0;
    for (int64_t e = first + wave0; e < end; e += nwaves) {
        // source vertex of entry e: last u with off[u] <= e
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const int64_t u = lo;
        const int32_t v = adj[e];
        if (MODE == 0 && !(u < v)) continue;
        const uint32_t c = wave_intersect_count(adj + off[u], off[u + 1] - off[u], adj + off[v], off[v + 1] - off[v], lane);
        total += c;
        units += 1;
        if (MODE == 1 && lane == 0 && c) atomicAdd(&counts[u], (unsigned long long)c);
    }
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[(blockIdx.x & 63) * 16], t);
    }
    __syncthreads();
    if (lane == 0) red[wave] = units;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[64 * 16], t);
    }
}

static int run_edge_pairs(const gmsx_graph *g, int mode, int part, int nparts, uint64_t *sum, unsigned long long *d_counts,
                          gmsx_stats *st) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    unsigned long long *acc = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), sizeof(unsigned long long) * (64 * 16 + 1)));
    struct Guard { void *p; ~Guard() { (void)hipFree(p); } } guard{acc};
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * (64 * 16 + 1), s));
    // contiguous entry ranges; entries are shuffled across hubs well enough by the row order for a first cut
    const int64_t first = g->nnz / nparts * part + std::min<int64_t>(part, g->nnz % nparts);
    const int64_t end = first + g->nnz / nparts + (part < g->nnz % nparts ? 1 : 0);
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    if (end > first) {
        const int64_t waves = end - first;
        const int64_t blocks = std::min<int64_t>((waves + 3) / 4, int64_t(c.compute_units > 0 ? c.compute_units : 256) * 32);
        if (mode == 0)
            hipLaunchKernelGGL(k_edge_pairs<0>, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, first, end, acc, d_counts);
        else
            hipLaunchKernelGGL(k_edge_pairs<1>, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, first, end, acc, d_counts);
    }
    GMSX_HIP(hipEventRecord(c.ev[1], s));
    GMSX_HIP(hipGetLastError());
    unsigned long long host[64 * 16 + 1];
    GMSX_HIP(hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    unsigned long long total = 0;
    for (int i = 0; i < 64; ++i) total += host[i * 16];
    if (sum) *sum = total;
    if (st) {
        float ms = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        *st = gmsx_stats{double(ms), 0.0, host[64 * 16], nparts == 1 && mode == 0 ? g->alg_elements : 0, 0, 1, 0};
    }
    return GMSX_OK;
}

int tc_full_partial(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st) {
    return run_edge_pairs(g, 0, part, nparts, partial, nullptr, st);
}

// counts[u] = Σ_{v∈N(u)} |N(u) ∩ N(v)| into a zeroed device array indexed by caller vertex id: the oriented bit-matrix kernels
// (kclique.hip) when the graph fits them, else one full-row intersect_count per CSR entry.  Shared by gmsx_tc_vertex_count2 and
// gmsx_tc_ordering (ordering.hip).
int tc_vertex_counts_device(const gmsx_graph *g, unsigned long long *d_counts, gmsx_stats *stats) {
    hipStream_t s = ctx().stream;
    int rc = kclique_vertex_counts(g, d_counts, stats);
    if (rc == GMSX_ERR_UNSUPPORTED) {
        GMSX_HIP(hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * size_t(std::max<int64_t>(g->n, 1)), s));
        rc = run_edge_pairs(g, 1, 0, 1, nullptr, d_counts, stats);
    }
    return rc;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_tc_vertex_count2(const gmsx_graph *g, int64_t *counts, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !counts) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        hipStream_t s = ctx().stream;
        unsigned long long *d_counts = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_counts), sizeof(unsigned long long) * size_t(std::max<int64_t>(g->n, 1))));
        struct Guard { void *p; ~Guard() { (void)hipFree(p); } } guard{d_counts};
        GMSX_HIP(hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * size_t(std::max<int64_t>(g->n, 1)), s));
        if (int rc = tc_vertex_counts_device(g, d_counts, stats)) return rc;
        if (g->n > 0) GMSX_HIP(hipMemcpy(counts, d_counts, sizeof(int64_t) * size_t(g->n), hipMemcpyDeviceToHost));
        return GMSX_OK;
    });
}

int gmsx_vertex_similarity_batch(const gmsx_graph *g, int metric, int64_t n_pairs, const int32_t *u, const int32_t *v, double *out,
                                 gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || n_pairs < 0 || metric < GMSX_SIM_JACCARD || metric > GMSX_SIM_PREF_ATTACHMENT || (n_pairs > 0 && (!u || !v || !out)))
            return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (n_pairs == 0) {
            if (stats) *stats = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
            return GMSX_OK;
        }
        Ctx &c = ctx();
        hipStream_t s = c.stream;
        int32_t *du = nullptr, *dv = nullptr;
        double *dout = nullptr;
        unsigned long long *flags = nullptr;
        struct Guard { void *p = nullptr; ~Guard() { (void)hipFree(p); } } g1, g2, g3, g4;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&du), size_t(n_pairs) * 4)); g1.p = du;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dv), size_t(n_pairs) * 4)); g2.p = dv;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dout), size_t(n_pairs) * 8)); g3.p = dout;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&flags), 8)); g4.p = flags;
        GMSX_HIP(hipMemcpyAsync(du, u, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemcpyAsync(dv, v, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemsetAsync(flags, 0, 8, s));
        GMSX_HIP(hipEventRecord(c.ev[0], s));
        const int64_t blocks = std::min<int64_t>((n_pairs + 3) / 4, int64_t(c.compute_units > 0 ? c.compute_units : 256) * 32);
        hipLaunchKernelGGL(k_pair_similarity, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, metric, n_pairs, du, dv, dout, flags);
        GMSX_HIP(hipEventRecord(c.ev[1], s));
        GMSX_HIP(hipGetLastError());
        unsigned long long bad = 0;
        GMSX_HIP(hipMemcpyAsync(out, dout, size_t(n_pairs) * 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipMemcpyAsync(&bad, flags, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        if (stats) {
            float ms = 0.f;
            GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
            *stats = gmsx_stats{double(ms), 0.0, uint64_t(n_pairs), 0, 0, 1, 0};
        }
        return bad ? GMSX_ERR_INVALID : GMSX_OK;
    });
}

int gmsx_intersect_count_batch(const gmsx_graph *g, int64_t n_pairs, const int32_t *u, const int32_t *v, uint32_t *out,
                               gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || n_pairs < 0 || (n_pairs > 0 && (!u || !v || !out))) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (n_pairs == 0) {
            if (stats) *stats = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
            return GMSX_OK;
        }
        Ctx &c = ctx();
        hipStream_t s = c.stream;
        int32_t *du = nullptr, *dv = nullptr;
        uint32_t *dout = nullptr;
        unsigned long long *flags = nullptr;
        struct Guard { void *p = nullptr; ~Guard() { (void)hipFree(p); } } g1, g2, g3, g4;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&du), size_t(n_pairs) * 4)); g1.p = du;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dv), size_t(n_pairs) * 4)); g2.p = dv;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dout), size_t(n_pairs) * 4)); g3.p = dout;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&flags), 8)); g4.p = flags;
        GMSX_HIP(hipMemcpyAsync(du, u, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemcpyAsync(dv, v, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemsetAsync(flags, 0, 8, s));
        GMSX_HIP(hipEventRecord(c.ev[0], s));
        const int64_t blocks = std::min<int64_t>((n_pairs + 3) / 4, int64_t(c.compute_units > 0 ? c.compute_units : 256) * 32);
        hipLaunchKernelGGL(k_pair_batch, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, n_pairs, du, dv, dout, flags);
        GMSX_HIP(hipEventRecord(c.ev[1], s));
        GMSX_HIP(hipGetLastError());
        unsigned long long bad = 0;
        GMSX_HIP(hipMemcpyAsync(out, dout, size_t(n_pairs) * 4, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipMemcpyAsync(&bad, flags, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        if (stats) {
            float ms = 0.f;
            GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
            *stats = gmsx_stats{double(ms), 0.0, uint64_t(n_pairs), 0, 0, 1, 0};
        }
        return bad ? GMSX_ERR_INVALID : GMSX_OK;  // a vertex id outside [0, n)
    });
}

int gmsx_set_op_batch(const gmsx_graph *g, int op, int64_t n_pairs, const int32_t *u, const int32_t *v, int64_t *out_offsets, int32_t *out_ids, int64_t out_capacity,
                      gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || n_pairs < 0 || !out_offsets || (op != GMSX_SETOP_INTERSECT && op != GMSX_SETOP_DIFFERENCE) || (n_pairs > 0 && (!u || !v)) || out_capacity < 0)
            return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        out_offsets[0] = 0;
        if (n_pairs == 0) {
            if (stats) *stats = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
            return GMSX_OK;
        }
        Ctx &c = ctx();
        hipStream_t s = c.stream;
        int32_t *du = nullptr, *dv = nullptr, *dout = nullptr;
        uint32_t *dcnt = nullptr;
        int64_t *dsz = nullptr, *doff = nullptr;
        unsigned long long *flags = nullptr;
        void *tmp = nullptr;
        struct Guard { void *p = nullptr; ~Guard() { (void)hipFree(p); } } g1, g2, g3, g4, g5, g6, g7, g8;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&du), size_t(n_pairs) * 4)); g1.p = du;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dv), size_t(n_pairs) * 4)); g2.p = dv;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dcnt), size_t(n_pairs) * 4)); g3.p = dcnt;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dsz), size_t(n_pairs + 1) * 8)); g4.p = dsz;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&doff), size_t(n_pairs + 1) * 8)); g5.p = doff;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&flags), 8)); g6.p = flags;
        GMSX_HIP(hipMemcpyAsync(du, u, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemcpyAsync(dv, v, size_t(n_pairs) * 4, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipMemsetAsync(flags, 0, 8, s));
        GMSX_HIP(hipEventRecord(c.ev[0], s));
        const int64_t blocks = std::min<int64_t>((n_pairs + 3) / 4, int64_t(c.compute_units > 0 ? c.compute_units : 256) * 32);
        hipLaunchKernelGGL(k_pair_batch, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, n_pairs, du, dv, dcnt, flags);
        hipLaunchKernelGGL(k_pair_sizes, dim3(unsigned(n_pairs / 256 + 1)), dim3(256), 0, s, g->off, g->n, n_pairs, du, dv, dcnt, op == GMSX_SETOP_DIFFERENCE ? 1 : 0, dsz);
        size_t tb = 0;
        GMSX_HIP(rocprim::exclusive_scan(nullptr, tb, dsz, doff, int64_t(0), size_t(n_pairs + 1), rocprim::plus<int64_t>(), s));
        GMSX_HIP(hipMalloc(&tmp, tb ? tb : 8)); g7.p = tmp;
        GMSX_HIP(rocprim::exclusive_scan(tmp, tb, dsz, doff, int64_t(0), size_t(n_pairs + 1), rocprim::plus<int64_t>(), s));
        unsigned long long bad = 0;
        GMSX_HIP(hipMemcpyAsync(out_offsets, doff, size_t(n_pairs + 1) * 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipMemcpyAsync(&bad, flags, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        if (bad) return GMSX_ERR_INVALID;  // a vertex id outside [0, n)
        const int64_t total = out_offsets[n_pairs];
        int launches = 3;
        if (out_ids && total > out_capacity) return GMSX_ERR_INVALID;  // the offsets say how much room the result needs
        if (out_ids && total > 0) {
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dout), size_t(total) * 4)); g8.p = dout;
            if (op == GMSX_SETOP_DIFFERENCE)
                hipLaunchKernelGGL(k_pair_fill<true>, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, n_pairs, du, dv, doff, dout);
            else
                hipLaunchKernelGGL(k_pair_fill<false>, dim3(unsigned(blocks)), dim3(256), 0, s, g->off, g->adj, g->n, n_pairs, du, dv, doff, dout);
            ++launches;
            GMSX_HIP(hipEventRecord(c.ev[1], s));
            GMSX_HIP(hipGetLastError());
            GMSX_HIP(hipMemcpyAsync(out_ids, dout, size_t(total) * 4, hipMemcpyDeviceToHost, s));
            GMSX_HIP(hipStreamSynchronize(s));
        } else {
            GMSX_HIP(hipEventRecord(c.ev[1], s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        if (stats) {
            float ms = 0.f;
            GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
            *stats = gmsx_stats{double(ms), 0.0, uint64_t(n_pairs), 0, 0, launches, 0};
        }
        return GMSX_OK;
    });
}

}  // extern "C"
