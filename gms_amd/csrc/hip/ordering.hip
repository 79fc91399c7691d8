// Vertex orderings that consume the path's operators, on gfx950:
//   gmsx_adg_rank     PpParallel::getDegeneracyOrderingApproxSGraph<boundary_function::averageDegree>
//                     (gms/algorithms/preprocessing/parallel/degeneracy_approx_set.h:14-86, boundary_function.h:14-23) — the
//                     preprocessing step in front of the Bron–Kerbosch driver (maximal_clique_enum_bron_kerbosch.cc:36-39)
//   gmsx_tc_ordering  PpParallel::triangleCountOrdering (preprocessing/parallel/triangle_count.h:11-30)
//
// ADG.  The reference peels in rounds: X = { remaining v : deg[v] <= (unsigned)((1+eps) * mean remaining degree) } leaves, and
// every other remaining vertex v PULLs  deg[v] -= |N(v) ∩ X|  (:74-79, one Set::intersect_count per remaining vertex and
// round, i.e. rounds x nnz row traffic).  The device gets the same integers by PUSHing: every x in X walks its own row once
// and decrements the counters of its still-remaining neighbours — each CSR entry is touched once over the whole run (nnz
// atomics in total; integer adds commute, so the counters are exact).  Per round: one reduction (Σ deg, #remaining) whose
// two integers go to the host, which evaluates the reference's double expression for the border bit for bit; one
// select kernel that stamps the round into state[] and emits (deg << 32 | id) keys; a rocPRIM radix sort of the batch
// (ties: vertex id — the reference leaves them to __gnu_parallel); one push kernel.
#include "device_graph.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

namespace gmsx {

namespace {

struct Dev {
    void *p = nullptr;
    ~Dev() { (void)hipFree(p); }
    template <class T> T *as() { return static_cast<T *>(p); }
};
template <class T>
int dalloc(Dev &d, int64_t count) {
    if (hipMalloc(&d.p, size_t(std::max<int64_t>(count, 1)) * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        d.p = nullptr;
        return GMSX_ERR_DEVICE_MEM;
    }
    return GMSX_OK;
}

__global__ void k_adg_init(int64_t n, const int64_t *__restrict__ off, int32_t *__restrict__ deg, int32_t *__restrict__ state) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v < n) {
        deg[v] = int32_t(off[v + 1] - off[v]);  // degreeCounter[i] = out_neigh(i).cardinality()   (:26)
        state[v] = -1;                          // -1 = still in the graph, else the round it left in
    }
}

// acc[0] += Σ deg over remaining vertices, acc[1] += #remaining          (boundary_function.h:16-20)
__global__ __launch_bounds__(256) void k_adg_sum(int64_t n, const int32_t *__restrict__ deg, const int32_t *__restrict__ state,
                                                 unsigned long long *__restrict__ acc) {
    __shared__ unsigned long long red[8];
    unsigned long long s = 0, c = 0;
    for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < n; v += int64_t(gridDim.x) * blockDim.x)
        if (state[v] < 0) {
            s += (unsigned long long)deg[v];
            ++c;
        }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        c += __shfl_down(c, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave * 2] = s;
        red[wave * 2 + 1] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ts = red[0] + red[2] + red[4] + red[6], tc = red[1] + red[3] + red[5] + red[7];
        if (ts) atomicAdd(&acc[0], ts);
        if (tc) atomicAdd(&acc[1], tc);
    }
}

// the partition of :41-51: remaining vertices with deg <= border join the batch (state := round); their sort keys
// (deg << 32 | id) are appended through one wave-aggregated atomic per wave
__global__ __launch_bounds__(256) void k_adg_select(int64_t n, const int32_t *__restrict__ deg, int32_t *__restrict__ state, uint32_t border,
                                                    int32_t round, unsigned long long *__restrict__ keys,
                                                    unsigned long long *__restrict__ batch_count) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t end = ((n + 63) / 64) * 64;  // whole waves stay converged for the ballot
    for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < end; v += stride) {
        const bool take = v < n && state[v] < 0 && uint32_t(deg[v]) <= border;
        const unsigned long long m = __ballot(take);
        if (m == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(batch_count, (unsigned long long)__popcll(m));
        base = (unsigned long long)__shfl((long long)base, 0);
        if (take) {
            state[v] = round;
            keys[base + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)uint32_t(deg[v]) << 32) | (unsigned long long)uint32_t(v);
        }
    }
}

// :64-71 (result in rank or order format) + the PUSH form of :74-79: one wave per batch vertex walks its row
__global__ __launch_bounds__(256) void k_adg_push(int64_t batch, const unsigned long long *__restrict__ sorted_keys, int64_t counter,
                                                  int rank_format, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                  const int32_t *__restrict__ state, int32_t *__restrict__ deg, int32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t i = wave0; i < batch; i += nwaves) {
        const int32_t x = int32_t(uint32_t(sorted_keys[i] & 0xffffffffull));
        if (lane == 0) {
            if (rank_format) out[x] = int32_t(counter + i);
            else out[counter + i] = x;
        }
        for (int64_t j = off[x] + lane; j < off[x + 1]; j += 64) {
            const int32_t w = adj[j];
            if (state[w] < 0) atomicSub(&deg[w], 1);  // w is still in the graph after this round: it loses neighbour x
        }
    }
}

__global__ void k_iota(int64_t n, int32_t *__restrict__ ids) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = int32_t(i);
}

}  // namespace
}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_adg_rank(const gmsx_graph *g, double epsilon, int rank_format, int32_t *out, int32_t *rounds_out, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !out || !(epsilon >= 0.0)) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        Ctx &c = ctx();
        hipStream_t s = c.stream;
        const int64_t n = g->n;
        if (stats) *stats = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
        if (rounds_out) *rounds_out = 0;
        if (n == 0) return GMSX_OK;
        Dev d_deg, d_state, d_keys, d_sorted, d_out, d_acc, d_tmp;
        if (int rc = dalloc<int32_t>(d_deg, n)) return rc;
        if (int rc = dalloc<int32_t>(d_state, n)) return rc;
        if (int rc = dalloc<unsigned long long>(d_keys, n)) return rc;
        if (int rc = dalloc<unsigned long long>(d_sorted, n)) return rc;
        if (int rc = dalloc<int32_t>(d_out, n)) return rc;
        if (int rc = dalloc<unsigned long long>(d_acc, 4)) return rc;
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_sorted.as<unsigned long long>(), size_t(n), 0, 64, s));
        if (int rc = dalloc<char>(d_tmp, int64_t(tmp_bytes))) return rc;

        const unsigned tb = unsigned((n + 255) / 256);
        const int cus = c.compute_units > 0 ? c.compute_units : 256;
        const unsigned sweep = unsigned(std::min<int64_t>((n + 255) / 256, int64_t(cus) * 16));
        GMSX_HIP(hipEventRecord(c.ev[0], s));
        hipLaunchKernelGGL(k_adg_init, dim3(tb), dim3(256), 0, s, n, g->off, d_deg.as<int32_t>(), d_state.as<int32_t>());
        int64_t counter = 0;
        int32_t round = 0;
        int launches = 1;
        while (counter < n) {
            unsigned long long acc[3] = {0, 0, 0};
            GMSX_HIP(hipMemsetAsync(d_acc.p, 0, 3 * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_adg_sum, dim3(sweep), dim3(256), 0, s, n, d_deg.as<int32_t>(), d_state.as<int32_t>(), d_acc.as<unsigned long long>());
            GMSX_HIP(hipMemcpyAsync(acc, d_acc.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            GMSX_HIP(hipStreamSynchronize(s));
            if (int64_t(acc[1]) != n - counter || acc[1] == 0) return GMSX_ERR_KERNEL;
            // boundary_function.h:14-23 + degeneracy_approx_set.h:39: the sum of ints accumulated in a double is exact below 2^53
            const double res = double(acc[0]);
            const unsigned border = unsigned((1 + epsilon) * (res / double(int(acc[1]))));
            hipLaunchKernelGGL(k_adg_select, dim3(sweep), dim3(256), 0, s, n, d_deg.as<int32_t>(), d_state.as<int32_t>(), uint32_t(border), round,
                               d_keys.as<unsigned long long>(), d_acc.as<unsigned long long>() + 2);
            GMSX_HIP(hipMemcpyAsync(&acc[2], d_acc.as<unsigned long long>() + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            GMSX_HIP(hipStreamSynchronize(s));
            const int64_t batch = int64_t(acc[2]);
            if (batch <= 0 || batch > n - counter) return GMSX_ERR_KERNEL;  // the minimum is never above the mean: a round cannot be empty
            // :55-59 sort the batch by remaining degree (ties: id); degrees < 2^31, so 63 key bits carry everything
            GMSX_HIP(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_sorted.as<unsigned long long>(), size_t(batch), 0, 64, s));
            const unsigned pb = unsigned(std::min<int64_t>((batch + 3) / 4, int64_t(cus) * 32));
            hipLaunchKernelGGL(k_adg_push, dim3(pb), dim3(256), 0, s, batch, d_sorted.as<unsigned long long>(), counter, rank_format ? 1 : 0, g->off,
                               g->adj, d_state.as<int32_t>(), d_deg.as<int32_t>(), d_out.as<int32_t>());
            counter += batch;
            ++round;
            launches += 4;
        }
        GMSX_HIP(hipEventRecord(c.ev[1], s));
        GMSX_HIP(hipMemcpyAsync(out, d_out.p, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipGetLastError());
        if (rounds_out) *rounds_out = round;
        if (stats) {
            float ms = 0.f;
            GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
            *stats = gmsx_stats{double(ms), 0.0, uint64_t(n), 0, uint64_t(round), launches, 0};
        }
        return GMSX_OK;
    });
}

int gmsx_tc_ordering(const gmsx_graph *g, int32_t *ordering, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !ordering) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        hipStream_t s = ctx().stream;
        const int64_t n = g->n;
        if (stats) *stats = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
        if (n == 0) return GMSX_OK;
        Dev d_counts, d_counts_sorted, d_ids, d_ids_sorted, d_tmp;
        if (int rc = dalloc<unsigned long long>(d_counts, n)) return rc;
        if (int rc = dalloc<unsigned long long>(d_counts_sorted, n)) return rc;
        if (int rc = dalloc<int32_t>(d_ids, n)) return rc;
        if (int rc = dalloc<int32_t>(d_ids_sorted, n)) return rc;
        GMSX_HIP(hipMemsetAsync(d_counts.p, 0, size_t(n) * sizeof(unsigned long long), s));
        // counts[u] = Σ_{v∈N(u)} |N(u) ∩ N(v)|: CountFn = Par::vertex_count2_once (triangle_count.h:14, parallel/vertex.h:30-49)
        if (int rc = tc_vertex_counts_device(g, d_counts.as<unsigned long long>(), stats)) return rc;
        hipLaunchKernelGGL(k_iota, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, d_ids.as<int32_t>());
        // sort ids by count ascending (:23-29); the radix sort is stable, so equal counts keep ascending ids
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_counts.as<unsigned long long>(), d_counts_sorted.as<unsigned long long>(),
                                           d_ids.as<int32_t>(), d_ids_sorted.as<int32_t>(), size_t(n), 0, 64, s));
        if (int rc = dalloc<char>(d_tmp, int64_t(tmp_bytes))) return rc;
        GMSX_HIP(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_counts.as<unsigned long long>(), d_counts_sorted.as<unsigned long long>(),
                                           d_ids.as<int32_t>(), d_ids_sorted.as<int32_t>(), size_t(n), 0, 64, s));
        GMSX_HIP(hipMemcpyAsync(ordering, d_ids_sorted.p, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipGetLastError());
        return GMSX_OK;
    });
}

}  // extern "C"
