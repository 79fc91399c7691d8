// Triangle counting on gfx950: the device replacement for
//   GMS::TriangleCount::Par::count_total   (gms/algorithms/set_based/triangle_count/parallel/total.h:7-24)
// whose inner operator is Set::intersect_count (representations/sets/sorted_set.h:176-182 ->
// sorted_set_operations.h:44-71).
//
// Formulation (GMSX_TC_ORIENTED).  The reference evaluates, for every undirected edge {u,v}, one
// intersect_count on the full rows and divides the sum by 3.  Here every undirected edge is still one
// intersect_count, but on the degree-oriented rows:  T = Σ_{u} Σ_{v∈N+(u)} |N+(u) ∩ N+(v)|,
// which meets every triangle exactly once, so the returned count is the same integer.
//
// Kernel shape (one pivot vertex u per workgroup or per wave):
//   1. the pivot row N+(u) is staged into LDS as an open-addressing hash set (load <= 0.5, usually <= 0.25);
//   2. the rows N+(v), v ∈ N+(u), are streamed from HBM/L2 with coalesced 64-lane loads;
//   3. every streamed id probes the LDS set; hits are counted per lane, reduced per workgroup, and added to one
//      of 64 spread u64 accumulators (one atomic per workgroup).
// No MFMA: this is integer/indexing work bounded by HBM/L2 row streaming and LDS probe rate.
#include "device_graph.hpp"

#include <cstdio>

namespace gmsx {

static constexpr int kAccSlots = 64;     // spread accumulators, 128 B apart
static constexpr int kAccStride = 16;    // in u64

__device__ __forceinline__ uint32_t hash_slot(int32_t w, int shift) { return (uint32_t(w) * 0x9E3779B1u) >> shift; }

__device__ __forceinline__ void set_insert(int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift);
    while (atomicCAS(&tbl[h], -1, w) != -1) h = (h + 1) & mask;
}

__device__ __forceinline__ unsigned set_contains(const int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift);
    while (true) {
        const int32_t x = tbl[h];
        if (x == w) return 1u;
        if (x == -1) return 0u;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int64_t readlane64(int64_t x, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(uint64_t(x)), l);
    const uint32_t hi = __builtin_amdgcn_readlane(uint32_t(uint64_t(x) >> 32), l);
    return int64_t((uint64_t(hi) << 32) | lo);
}

// 4-byte-aligned 16-byte load: rows start at arbitrary dword offsets, the hardware only needs dword alignment
struct __attribute__((packed, aligned(4))) i4u { int32_t x, y, z, w; };

__device__ __forceinline__ unsigned probe4(const int32_t *tbl, uint32_t mask, int shift, i4u w, int valid) {
    // first-slot reads of the four ids are issued together (four LDS reads in flight); at load <= 0.25 most
    // probes resolve there, the rest continue down the chain
    const uint32_t h0 = hash_slot(w.x, shift), h1 = hash_slot(w.y, shift), h2 = hash_slot(w.z, shift), h3 = hash_slot(w.w, shift);
    const int32_t x0 = tbl[h0], x1 = tbl[h1], x2 = tbl[h2], x3 = tbl[h3];
    unsigned c = 0;
    if (valid > 0) { if (x0 == w.x) c++; else if (x0 != -1) c += set_contains(tbl, mask, shift, w.x); }
    if (valid > 1) { if (x1 == w.y) c++; else if (x1 != -1) c += set_contains(tbl, mask, shift, w.y); }
    if (valid > 2) { if (x2 == w.z) c++; else if (x2 != -1) c += set_contains(tbl, mask, shift, w.z); }
    if (valid > 3) { if (x3 == w.w) c++; else if (x3 != -1) c += set_contains(tbl, mask, shift, w.w); }
    return c;
}

// Streams `rows` rows against the LDS set.  Lane l holds the extent (rb, rl) of row l.  The wave works as four
// 16-lane groups, each streaming its own row with 16-byte loads (64 ids per group step, two steps in flight), so a
// wave keeps up to eight independent 256-byte requests outstanding instead of one.
__device__ __forceinline__ unsigned long long scan_rows(const int32_t *tbl, uint32_t mask, int shift,
                                                        const int32_t *__restrict__ dadj, int64_t rb, int rl, int rows,
                                                        int lane) {
    const int grp = lane >> 4, sub4 = (lane & 15) * 4;
    unsigned long long cnt = 0;
    for (int r0 = 0; r0 < rows; r0 += 4) {
        // extents of rows r0..r0+3 through wave-uniform readlanes (lanes >= rows hold length 0), then a per-group
        // select: a per-lane __shfl here gets sunk under the row-count predicate by the compiler and then reads
        // inactive lanes
        const int m0 = r0 & 63, m1 = (r0 + 1) & 63, m2 = (r0 + 2) & 63, m3 = (r0 + 3) & 63;
        const int64_t b0 = readlane64(rb, m0), b1 = readlane64(rb, m1), b2 = readlane64(rb, m2), b3 = readlane64(rb, m3);
        const int l0 = __builtin_amdgcn_readlane(rl, m0), l1 = __builtin_amdgcn_readlane(rl, m1),
                  l2 = __builtin_amdgcn_readlane(rl, m2), l3 = __builtin_amdgcn_readlane(rl, m3);
        const int64_t b = grp == 0 ? b0 : grp == 1 ? b1 : grp == 2 ? b2 : b3;
        const int l = grp == 0 ? l0 : grp == 1 ? l1 : grp == 2 ? l2 : l3;
        const int32_t *row = dadj + b;
        int j = sub4;
        for (; j + 64 < l; j += 128) {  // two 64-id steps per iteration
            const i4u a = *reinterpret_cast<const i4u *>(row + j);
            const i4u c = *reinterpret_cast<const i4u *>(row + j + 64);
            cnt += probe4(tbl, mask, shift, a, 4);
            cnt += probe4(tbl, mask, shift, c, l - (j + 64));
        }
        if (j < l) {
            const i4u a = *reinterpret_cast<const i4u *>(row + j);
            cnt += probe4(tbl, mask, shift, a, l - j);
        }
    }
    return cnt;
}

// ---------------------------------------------------------------------------------------------
// Workgroup per pivot (d+ >= 64).  256 threads = 4 waves share one LDS set; wave w streams the rows of the
// pivot-list entries w, w+4, ….  Pivot rows longer than half the table are processed in tiles.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tc_oriented_block(const int64_t *__restrict__ doff, const int32_t *__restrict__ dadj,
                                                           const int32_t *__restrict__ order, int64_t first, int64_t end,
                                                           int nparts, int part, int log_tbl,
                                                           unsigned long long *__restrict__ acc) {
    extern __shared__ int32_t tbl[];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pos = first + int64_t(blockIdx.x) * nparts + part;
    if (pos >= end) return;  // uniform per block
    const int32_t u = order[pos];
    const int64_t beg = doff[u];
    const int dp = int(doff[u + 1] - beg);
    const int tbl_size = 1 << log_tbl, shift = 32 - log_tbl;
    const uint32_t mask = uint32_t(tbl_size - 1);
    const int tile = tbl_size >> 1;

    unsigned long long cnt = 0;
    for (int t0 = 0; t0 < dp; t0 += tile) {
        const int tn = min(tile, dp - t0);
        __syncthreads();  // previous tile's probes are done
        for (int i = tid; i < tbl_size; i += 256) tbl[i] = -1;
        __syncthreads();
        for (int i = tid; i < tn; i += 256) set_insert(tbl, mask, shift, dadj[beg + t0 + i]);
        __syncthreads();
        // rows of the pivot list, 256 per batch: lane l of wave w prefetches the extent of row (base + 4l + w)
        for (int base = 0; base < dp; base += 256) {
            const int idx = base + lane * 4 + wave;
            int64_t rb = 0;
            int rl = 0;
            if (idx < dp) {
                const int32_t v = dadj[beg + idx];
                rb = doff[v];
                rl = int(doff[v + 1] - rb);
            }
            const int rows = min(64, (dp - base - wave + 3) >> 2);
            cnt += scan_rows(tbl, mask, shift, dadj, rb, rl, rows, lane);
        }
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[(blockIdx.x & (kAccSlots - 1)) * kAccStride], t);
    }
}

// ---------------------------------------------------------------------------------------------
// Wave per pivot (2 <= d+ < 64).  Each of the 4 waves of a workgroup owns a private 2^LOG-entry LDS set and walks
// its own pivots with a grid stride; one atomic per workgroup at the end.
// ---------------------------------------------------------------------------------------------
template <int LOG>
__global__ __launch_bounds__(256) void k_tc_oriented_wave(const int64_t *__restrict__ doff, const int32_t *__restrict__ dadj,
                                                          const int32_t *__restrict__ order, int64_t first, int64_t end,
                                                          int nparts, int part, unsigned long long *__restrict__ acc) {
    constexpr int SIZE = 1 << LOG, SHIFT = 32 - LOG;
    constexpr uint32_t MASK = SIZE - 1;
    __shared__ int32_t tbl_all[4 * SIZE];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *tbl = tbl_all + wave * SIZE;
    const int64_t nwaves = int64_t(gridDim.x) * 4;
    unsigned long long cnt = 0;
    for (int64_t q = int64_t(blockIdx.x) * 4 + wave;; q += nwaves) {
        const int64_t pos = first + q * nparts + part;
        if (pos >= end) break;  // uniform per wave
        const int32_t u = order[pos];
        const int64_t beg = doff[u];
        const int dp = int(doff[u + 1] - beg);  // < 64
        for (int i = lane; i < SIZE; i += 64) tbl[i] = -1;
        __builtin_amdgcn_wave_barrier();
        int64_t rb = 0;
        int rl = 0;
        if (lane < dp) {
            const int32_t v = dadj[beg + lane];
            set_insert(tbl, MASK, SHIFT, v);
            rb = doff[v];
            rl = int(doff[v + 1] - rb);
        }
        __builtin_amdgcn_wave_barrier();
        cnt += scan_rows(tbl, MASK, SHIFT, dadj, rb, rl, dp, lane);
        __builtin_amdgcn_wave_barrier();
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[(blockIdx.x & (kAccSlots - 1)) * kAccStride], t);
    }
}

// units / probes of a partition (untimed bookkeeping for gmsx_stats): wave per pivot position
__global__ __launch_bounds__(256) void k_tc_oriented_stats(const int64_t *__restrict__ doff, const int32_t *__restrict__ dadj,
                                                           const int32_t *__restrict__ order, int64_t first, int64_t end,
                                                           int nparts, int part, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long units = 0, probes = 0;
    for (int64_t q = wave0;; q += nwaves) {
        const int64_t pos = first + q * nparts + part;
        if (pos >= end) break;
        const int32_t u = order[pos];
        const int64_t b = doff[u], e = doff[u + 1];
        if (lane == 0) units += (unsigned long long)(e - b);
        for (int64_t j = b + lane; j < e; j += 64) {
            const int32_t v = dadj[j];
            probes += (unsigned long long)(doff[v + 1] - doff[v]);
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        units += __shfl_down(units, s);
        probes += __shfl_down(probes, s);
    }
    if (lane == 0) {
        if (units) atomicAdd(&out[0], units);
        if (probes) atomicAdd(&out[1], probes);
    }
}

static int64_t part_count(int64_t first, int64_t end, int nparts, int part) {
    const int64_t span = end - first - part;
    return span <= 0 ? 0 : (span + nparts - 1) / nparts;
}

static int tc_oriented(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    unsigned long long *acc = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), sizeof(unsigned long long) * (kAccSlots * kAccStride + 2)));
    struct Guard { void *p; ~Guard() { (void)hipFree(p); } } guard{acc};
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * (kAccSlots * kAccStride + 2), s));
    GMSX_HIP(hipEventRecord(c.ev[1], s));

    int launches = 0;
    // block-per-pivot bins: d+ >= 64 -> bins 0..4 with table sizes 2^15, 2^15, 2^13, 2^11, 2^9
    static const int kLogTbl[5] = {15, 15, 13, 11, 9};
    static bool attr_set = false;
    if (!attr_set) {
        GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tc_oriented_block),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (1 << 15) * 4));
        attr_set = true;
    }
    int64_t lo = 0;
    for (int b = 0; b < 5; ++b) {
        const int64_t hi = g->bin_end[b];
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (cnt > 0) {
            hipLaunchKernelGGL(k_tc_oriented_block, dim3(unsigned(cnt)), dim3(256), size_t(4) << kLogTbl[b], s, g->doff,
                               g->dadj, g->order, lo, hi, nparts, part, kLogTbl[b], acc);
            ++launches;
        }
        lo = hi;
    }
    const int64_t cap_blocks = int64_t(c.compute_units > 0 ? c.compute_units : 256) * 32;
    {  // 16 <= d+ < 64
        const int64_t hi = g->bin_end[5];
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (cnt > 0) {
            const int64_t blocks = std::min<int64_t>((cnt + 3) / 4, cap_blocks);
            hipLaunchKernelGGL(k_tc_oriented_wave<8>, dim3(unsigned(blocks)), dim3(256), 0, s, g->doff, g->dadj, g->order, lo,
                               hi, nparts, part, acc);
            ++launches;
        }
        lo = hi;
    }
    {  // 2 <= d+ < 16
        const int64_t hi = g->bin_end[6];
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (cnt > 0) {
            const int64_t blocks = std::min<int64_t>((cnt + 3) / 4, cap_blocks);
            hipLaunchKernelGGL(k_tc_oriented_wave<6>, dim3(unsigned(blocks)), dim3(256), 0, s, g->doff, g->dadj, g->order, lo,
                               hi, nparts, part, acc);
            ++launches;
        }
    }
    GMSX_HIP(hipEventRecord(c.ev[2], s));
    GMSX_HIP(hipGetLastError());

    if (st) {  // untimed bookkeeping
        const int64_t hi = g->n;  // every oriented edge is a unit, also those of pivots with d+ < 2 (no work)
        const int64_t cnt = part_count(0, hi, nparts, part);
        if (cnt > 0) {
            const int64_t blocks = std::min<int64_t>((cnt + 3) / 4, cap_blocks);
            hipLaunchKernelGGL(k_tc_oriented_stats, dim3(unsigned(blocks)), dim3(256), 0, s, g->doff, g->dadj, g->order,
                               int64_t(0), hi, nparts, part, acc + kAccSlots * kAccStride);
        }
    }
    unsigned long long host[kAccSlots * kAccStride + 2];
    GMSX_HIP(hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    unsigned long long total = 0;
    for (int i = 0; i < kAccSlots; ++i) total += host[i * kAccStride];
    *partial = total;
    if (st) {
        float ms_setup = 0.f, ms_kernel = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms_setup, c.ev[0], c.ev[1]));
        GMSX_HIP(hipEventElapsedTime(&ms_kernel, c.ev[1], c.ev[2]));
        st->kernel_ms = ms_kernel;
        st->setup_ms = ms_setup;
        st->units = host[kAccSlots * kAccStride];
        st->probes = host[kAccSlots * kAccStride + 1];
        st->alg_elements = nparts == 1 ? g->alg_elements : 0;
        st->launches = launches;
        st->reserved = 0;
    }
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_tc_divisor(int algo) { return algo == GMSX_TC_FULL ? 3 : 1; }

int gmsx_tc_partial(const gmsx_graph *g, int algo, int part, int nparts, uint64_t *partial, gmsx_stats *stats) {
    if (!g || !partial || nparts < 1 || part < 0 || part >= nparts) return GMSX_ERR_INVALID;
    if (algo != GMSX_TC_AUTO && algo != GMSX_TC_ORIENTED && algo != GMSX_TC_FULL) return GMSX_ERR_INVALID;
    if (int rc = ensure_init()) return rc;
    if (algo == GMSX_TC_FULL) return GMSX_ERR_UNSUPPORTED;
    return tc_oriented(g, part, nparts, partial, stats);
}

int gmsx_tc_total(const gmsx_graph *g, int algo, uint64_t *triangles, gmsx_stats *stats) {
    if (!triangles) return GMSX_ERR_INVALID;
    uint64_t partial = 0;
    if (int rc = gmsx_tc_partial(g, algo, 0, 1, &partial, stats)) return rc;
    const uint64_t div = uint64_t(gmsx_tc_divisor(algo));
    if (partial % div != 0) return GMSX_ERR_KERNEL;  // the reference asserts total % 3 == 0 (parallel/total.h:22)
    *triangles = partial / div;
    return GMSX_OK;
}

}  // extern "C"
