// Host graph substrate of gmsx: synthetic generators, edge-list -> CSR builder, degree relabelling,
// .el / .sg readers and the .sg writer.  Pure host C++17 + OpenMP; no device code.
//
// Behavioural contract = the reference loader (paths relative to the spcl/gms tree):
//   generator  gms/third_party/gapbs/generator.h:52-127   (R-MAT A/B/C = .57/.19/.19, uniform, id permutation)
//   builder    gms/third_party/gapbs/builder.h:108-117,145-156,206-298 (n = max id + 1, symmetrise, sort/unique/no-loop rows)
//   relabel    gms/third_party/gapbs/builder.h:1699-1733, gapbs/benchmark.h:50-68,158-176 (WorthRelabelling)
//   files      gms/third_party/gapbs/reader.h:49-56,252-305, writer.h:39-69
// The CSR it produces is bit-identical to the reference's (tests/test_loader.py: FNV fingerprints of
// SURVEY Appendix B and array equality against the compiled reference).
//
// Three pieces of libstdc++ behaviour fix the reference's output and are restated here explicitly
// (so the result does not depend on which libstdc++ this file is compiled against):
//   - std::uniform_real_distribution<float>(0,1) on mt19937  -> canonical_float()
//   - std::uniform_int_distribution on mt19937 (GCC >= 11: Lemire's method) -> bounded_u32()
//   - std::shuffle's two-swaps-per-draw fast path for n <= 65535 -> shuffle_ids()
// std::mt19937 itself is fully specified by the C++ standard and is used as is.
#include "gmsx_internal.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#ifdef _OPENMP
#include <parallel/algorithm>
#endif
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <limits>
#include <memory>
#include <new>
#include <random>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace gmsx {

static constexpr uint32_t kSeed = 27491095u;         // gapbs/util.h:25
static constexpr int64_t kGenBlock = int64_t(1) << 18;  // generator.h:152: rng reseeded per 2^18-edge block

// phase timings of the host substrate on stderr when GMSX_TIMING=1 (the reference prints "Generate Time" / "Build Time")
struct PhaseTimer {
    const bool on = opt_on("TIMING");
    double t0 = now();
    static double now() {
#ifdef _OPENMP
        return omp_get_wtime();
#else
        return double(std::clock()) / CLOCKS_PER_SEC;
#endif
    }
    void lap(const char *what) {
        if (!on) return;
        const double t1 = now();
        std::fprintf(stderr, "[gmsx host] %-28s %8.3f s\n", what, t1 - t0);
        t0 = t1;
    }
};

// ---- restated libstdc++ distributions ------------------------------------------------------

// generate_canonical<float,24> on a 32-bit engine: one draw, float(x) / 2^32 in float arithmetic,
// clamped below 1.0f (float(x) rounds to nearest, so the quotient can reach 1.0f).
static inline float canonical_float(std::mt19937 &rng) {
    float r = static_cast<float>(static_cast<uint32_t>(rng())) / 4294967296.0f;
    if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
    return r;
}

// Unbiased integer in [0, range), range in [1, 2^32): Lemire's nearly-divisionless reduction with a
// 32x32->64 product and rejection of the low part below (2^32 mod range).
static inline uint32_t bounded_u32(std::mt19937 &rng, uint32_t range) {
    uint64_t prod = uint64_t(uint32_t(rng())) * range;
    uint32_t low = uint32_t(prod);
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            prod = uint64_t(uint32_t(rng())) * range;
            low = uint32_t(prod);
        }
    }
    return uint32_t(prod >> 32);
}
// closed interval [0, hi]
static inline uint32_t uniform_closed(std::mt19937 &rng, uint32_t hi) {
    if (hi == 0xFFFFFFFFu) return uint32_t(rng());
    return bounded_u32(rng, hi + 1u);
}

// Fisher-Yates as libstdc++ runs it on random-access iterators with a 32-bit engine.
static void shuffle_ids(int32_t *a, int64_t n, std::mt19937 &rng) {
    if (n <= 1) return;
    const uint64_t un = uint64_t(n);
    if (0xFFFFFFFFull / un >= un) {  // n*n fits the engine range: one draw feeds two swaps
        int64_t i = 1;
        if ((un % 2) == 0) {
            std::swap(a[i], a[uniform_closed(rng, 1)]);
            ++i;
        }
        while (i != n) {
            const uint64_t r0 = uint64_t(i) + 1, r1 = r0 + 1;
            const uint32_t x = uniform_closed(rng, uint32_t(r0 * r1 - 1));
            std::swap(a[i], a[x / r1]);
            ++i;
            std::swap(a[i], a[x % r1]);
            ++i;
        }
        return;
    }
    // general path: one draw per swap.  The targets do not depend on the array, so they are drawn a chunk ahead (same
    // engine sequence) and prefetched: the swaps themselves stay sequential, but no longer wait for a cache miss each
    // (n = 2^26: the array is 256 MB, every target a miss).
    constexpr int64_t kChunk = 4096, kAhead = 32;
    std::unique_ptr<uint32_t[]> tgt(new uint32_t[size_t(kChunk)]);
    for (int64_t base = 1; base < n; base += kChunk) {
        const int64_t cnt = std::min<int64_t>(kChunk, n - base);
        for (int64_t k = 0; k < cnt; ++k) tgt[k] = uniform_closed(rng, uint32_t(base + k));
        for (int64_t k = 0; k < std::min<int64_t>(kAhead, cnt); ++k) __builtin_prefetch(a + tgt[k], 1, 0);
        for (int64_t k = 0; k < cnt; ++k) {
            if (k + kAhead < cnt) __builtin_prefetch(a + tgt[k + kAhead], 1, 0);
            std::swap(a[base + k], a[tgt[k]]);
        }
    }
}

// ---- edge lists ------------------------------------------------------------------------------

struct EdgeList {
    int64_t m = 0;
    std::unique_ptr<int32_t[]> u, v;
    int alloc(int64_t edges) {
        m = edges;
        u.reset(new (std::nothrow) int32_t[size_t(std::max<int64_t>(edges, 1))]);
        v.reset(new (std::nothrow) int32_t[size_t(std::max<int64_t>(edges, 1))]);
        return (u && v) ? GMSX_OK : GMSX_ERR_NOMEM;
    }
};

static int make_rmat(int scale, int degree, EdgeList &el, float A = 0.57f, float B = 0.19f, float C = 0.19f) {
    const int64_t n = int64_t(1) << scale, m = n * degree;
    PhaseTimer pt;
    if (int rc = el.alloc(m)) return rc;
    const float AB = A + B, ABC = A + B + C;  // float sums, as the reference compares against
    int32_t *eu = el.u.get(), *ev = el.v.get();
    // id permutation (generator.h:52-62): identity shuffled once with mt19937(kSeed).  The shuffle is inherently serial;
    // one thread runs it while the others generate the edges (it then joins them).
    // (2 MB-aligned and advised as huge pages where the host allows it: the shuffle touches the array at random, and with
    // 4 KB pages every swap is a TLB miss first)
    struct FreeDeleter { void operator()(int32_t *p) const { std::free(p); } };
    std::unique_ptr<int32_t[], FreeDeleter> perm;
    {
        void *mem = nullptr;
        const size_t bytes = ((size_t(n) * sizeof(int32_t) + (size_t(2) << 20) - 1) >> 21) << 21;
        if (posix_memalign(&mem, size_t(2) << 20, bytes) != 0) return GMSX_ERR_NOMEM;
#ifdef MADV_HUGEPAGE
        (void)madvise(mem, bytes, MADV_HUGEPAGE);
#endif
        perm.reset(static_cast<int32_t *>(mem));
    }
#pragma omp parallel
    {
#pragma omp single nowait
        {
            for (int64_t i = 0; i < n; ++i) perm[i] = int32_t(i);
            std::mt19937 prng(kSeed);
            PhaseTimer ps;
            shuffle_ids(perm.get(), n, prng);
            ps.lap("  rmat: shuffle alone");
        }
        std::mt19937 rng;
#pragma omp for schedule(dynamic, 1)
        for (int64_t block = 0; block < m; block += kGenBlock) {
            rng.seed(uint32_t(kSeed + block / kGenBlock));
            const int64_t end = std::min(block + kGenBlock, m);
            for (int64_t e = block; e < end; ++e) {
                int32_t src = 0, dst = 0;
                for (int depth = 0; depth < scale; ++depth) {
                    const float p = canonical_float(rng);
                    src <<= 1;
                    dst <<= 1;
                    if (p < AB) {
                        if (p > A) dst++;
                    } else {
                        src++;
                        if (p > ABC) dst++;
                    }
                }
                eu[e] = src;
                ev[e] = dst;
            }
        }
    }
    pt.lap("  rmat: draws || shuffle");
    // the permutation is applied to both ends
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < m; ++e) {
        eu[e] = perm[eu[e]];
        ev[e] = perm[ev[e]];
    }
    pt.lap("  rmat: apply permutation");
    return GMSX_OK;
}

static int make_uniform(int scale, int degree, EdgeList &el) {
    const int64_t n = int64_t(1) << scale, m = n * degree;
    if (int rc = el.alloc(m)) return rc;
    int32_t *eu = el.u.get(), *ev = el.v.get();
    const uint32_t hi = uint32_t(n - 1);
#pragma omp parallel
    {
        std::mt19937 rng;
#pragma omp for schedule(dynamic, 4)
        for (int64_t block = 0; block < m; block += kGenBlock) {
            rng.seed(uint32_t(kSeed + block / kGenBlock));
            const int64_t end = std::min(block + kGenBlock, m);
            for (int64_t e = block; e < end; ++e) {
                // The reference constructs Edge(udist(rng), udist(rng)) (generator.h:74); argument evaluation
                // order is unspecified and its GCC-built binary draws the SECOND argument first.  Pinned by the
                // uniform-graph fingerprint test.
                const uint32_t second = uniform_closed(rng, hi);
                const uint32_t first = uniform_closed(rng, hi);
                eu[e] = int32_t(first);
                ev[e] = int32_t(second);
            }
        }
    }
    return GMSX_OK;
}

// ---- CSR construction --------------------------------------------------------------------------

static int alloc_csr(Csr &g, int64_t n, int64_t nnz) {
    g.n = n;
    g.nnz = nnz;
    g.off.reset(new (std::nothrow) int64_t[size_t(n + 1)]);
    g.neigh.reset(new (std::nothrow) int32_t[size_t(std::max<int64_t>(nnz, 1))]);
    return (g.off && g.neigh) ? GMSX_OK : GMSX_ERR_NOMEM;
}

// exclusive prefix sum of per-vertex counts into off[0..n]; two-pass blocked so large n stays parallel
static void prefix_sum(const int64_t *cnt, int64_t n, int64_t *off) {
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    std::vector<int64_t> part(size_t(nt) + 1, 0);
    const int64_t chunk = (n + nt - 1) / std::max(nt, 1);
    // slices are walked by slice index, not by thread id: the team may be smaller than `nt` (OMP_THREAD_LIMIT, OMP_DYNAMIC,
    // a nested region) and every slice must still be summed and written
#pragma omp parallel num_threads(nt)
    {
#pragma omp for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = std::min<int64_t>(n, t * chunk), hi = std::min<int64_t>(n, lo + chunk);
            int64_t s = 0;
            for (int64_t i = lo; i < hi; ++i) s += cnt[i];
            part[size_t(t) + 1] = s;
        }
#pragma omp single
        for (int i = 0; i < nt; ++i) part[size_t(i) + 1] += part[size_t(i)];
#pragma omp for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            const int64_t lo = std::min<int64_t>(n, t * chunk), hi = std::min<int64_t>(n, lo + chunk);
            int64_t s = part[size_t(t)];
            for (int64_t i = lo; i < hi; ++i) {
                off[i] = s;
                s += cnt[i];
            }
        }
    }
    off[n] = part[size_t(nt)];
}

// Edge list -> canonical symmetric CSR.  Equivalent to MakeGraphFromEL + SquishGraph: both directions
// of every pair, rows sorted, duplicates and self-loops removed, n = max id + 1 unless given.
static int build_from_el(const EdgeList &el, int64_t num_nodes, bool symmetrize, Csr &out) {
    const int64_t m = el.m;
    const int32_t *eu = el.u.get(), *ev = el.v.get();
    if (num_nodes < 0) {
        int32_t mx = 0;  // reference starts its max-reduction at 0, so an empty list gives n = 1
#pragma omp parallel for reduction(max : mx) schedule(static)
        for (int64_t e = 0; e < m; ++e) mx = std::max(mx, std::max(eu[e], ev[e]));
        num_nodes = int64_t(mx) + 1;
    }
    PhaseTimer pt;
    const int64_t n = num_nodes;
    int bad = 0;  // cheap guard; the generators never trip it
#pragma omp parallel for reduction(| : bad) schedule(static)
    for (int64_t e = 0; e < m; ++e) bad |= int(eu[e] < 0 || ev[e] < 0 || eu[e] >= n || ev[e] >= n);
    if (bad) return GMSX_ERR_INVALID;

    pt.lap("build: max id + guard");
    // passes 1-2 without atomics (hub rows made the fetch_add counters the hot spot of a 256-thread host): the directed
    // pairs are first partitioned by the high bits of their row vertex into buckets — every thread counts and then writes
    // its own slice of the edge list through private cursors — and each bucket, which owns a contiguous vertex range, is
    // then counted and scattered by one thread.  Placement inside a row differs from run to run; the per-row sort below
    // makes the result deterministic, as before.
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    int shift = 0;
    while (((n - 1) >> shift) >= 4096) ++shift;  // at most 4096 buckets
    const int64_t nb = ((n - 1) >> shift) + 1;
    std::unique_ptr<int64_t[]> hist(new (std::nothrow) int64_t[size_t(nb) * size_t(nthreads) + 1]);
    std::unique_ptr<int64_t[]> raw_off(new (std::nothrow) int64_t[size_t(n + 1)]);
    std::unique_ptr<int64_t[]> cnt(new (std::nothrow) int64_t[size_t(n + 1)]);
    if (!hist || !raw_off || !cnt) return GMSX_ERR_NOMEM;
    std::fill(hist.get(), hist.get() + size_t(nb) * size_t(nthreads) + 1, int64_t(0));
    auto slice = [&](int t, int64_t &lo, int64_t &hi) {
        lo = m * int64_t(t) / nthreads;
        hi = m * int64_t(t + 1) / nthreads;
    };
    // (slices are handed out by index with schedule(static,1): correct for any team size the runtime actually delivers)
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int t = 0; t < nthreads; ++t) {
        int64_t lo, hi;
        slice(t, lo, hi);
        int64_t *h = hist.get() + size_t(t) * size_t(nb);  // layout [slice][bucket] while counting
        for (int64_t e = lo; e < hi; ++e) {
            if (eu[e] == ev[e]) continue;
            ++h[eu[e] >> shift];
            if (symmetrize) ++h[ev[e] >> shift];
        }
    }
    // exclusive prefix in (bucket, thread) order -> private write cursors; bucket_off[b] = start of bucket b
    std::unique_ptr<int64_t[]> bucket_off(new (std::nothrow) int64_t[size_t(nb + 1)]);
    if (!bucket_off) return GMSX_ERR_NOMEM;
    int64_t run = 0;
    for (int64_t bkt = 0; bkt < nb; ++bkt) {
        bucket_off[bkt] = run;
        for (int t = 0; t < nthreads; ++t) {
            int64_t &slot = hist[size_t(t) * size_t(nb) + size_t(bkt)];
            const int64_t c = slot;
            slot = run;
            run += c;
        }
    }
    bucket_off[nb] = run;
    const int64_t raw_nnz = run;
    std::unique_ptr<int32_t[]> prow(new (std::nothrow) int32_t[size_t(std::max<int64_t>(raw_nnz, 1))]);
    std::unique_ptr<int32_t[]> pnbr(new (std::nothrow) int32_t[size_t(std::max<int64_t>(raw_nnz, 1))]);
    std::unique_ptr<int32_t[]> raw(new (std::nothrow) int32_t[size_t(std::max<int64_t>(raw_nnz, 1))]);
    if (!prow || !pnbr || !raw) return GMSX_ERR_NOMEM;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int t = 0; t < nthreads; ++t) {
        int64_t lo, hi;
        slice(t, lo, hi);
        int64_t *cur = hist.get() + size_t(t) * size_t(nb);
        for (int64_t e = lo; e < hi; ++e) {
            const int32_t a = eu[e], c = ev[e];
            if (a == c) continue;
            int64_t p = cur[a >> shift]++;
            prow[p] = a;
            pnbr[p] = c;
            if (symmetrize) {
                p = cur[c >> shift]++;
                prow[p] = c;
                pnbr[p] = a;
            }
        }
    }
    pt.lap("build: bucket partition");
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bkt = 0; bkt < nb; ++bkt) {  // raw row sizes of the bucket's vertices (self-loops dropped already; duplicates later)
        const int64_t v0 = bkt << shift, v1 = std::min<int64_t>(n, (bkt + 1) << shift);
        for (int64_t v = v0; v < v1; ++v) cnt[v] = 0;
        for (int64_t p = bucket_off[bkt]; p < bucket_off[bkt + 1]; ++p) ++cnt[prow[p]];
    }
    cnt[n] = 0;
    pt.lap("build: count degrees");
    prefix_sum(cnt.get(), n, raw_off.get());
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bkt = 0; bkt < nb; ++bkt) {
        const int64_t v0 = bkt << shift, v1 = std::min<int64_t>(n, (bkt + 1) << shift);
        for (int64_t v = v0; v < v1; ++v) cnt[v] = raw_off[v];
        for (int64_t p = bucket_off[bkt]; p < bucket_off[bkt + 1]; ++p) raw[cnt[prow[p]]++] = pnbr[p];
    }
    prow.reset();
    pnbr.reset();
    pt.lap("build: prefix + scatter");
    // pass 3: sort + unique each row, record the surviving length
    std::unique_ptr<int64_t[]> len(new (std::nothrow) int64_t[size_t(n + 1)]);
    if (!len) return GMSX_ERR_NOMEM;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t u = 0; u < n; ++u) {
        int32_t *b = raw.get() + raw_off[u], *e = raw.get() + raw_off[u + 1];
        std::sort(b, e);
        len[u] = std::unique(b, e) - b;
    }
    pt.lap("build: sort + unique rows");
    cnt.reset();
    // pass 4: compact
    std::unique_ptr<int64_t[]> off(new (std::nothrow) int64_t[size_t(n + 1)]);
    if (!off) return GMSX_ERR_NOMEM;
    prefix_sum(len.get(), n, off.get());
    if (int rc = alloc_csr(out, n, off[n])) return rc;
    std::memcpy(out.off.get(), off.get(), size_t(n + 1) * sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t u = 0; u < n; ++u)
        std::copy(raw.get() + raw_off[u], raw.get() + raw_off[u] + len[u], out.neigh.get() + off[u]);
    out.directed = !symmetrize;
    return GMSX_OK;
}

// ---- relabel by decreasing degree ----------------------------------------------------------------

// gapbs/benchmark.h:158-176 with SourcePicker (gapbs/benchmark.h:50-68): 1000 draws of a non-isolated
// vertex from mt19937(kSeed); relabel iff mean/1.3 > median of the sampled degrees and m/n >= 10.
bool worth_relabelling(const Csr &g) {
    if (g.n <= 0) return false;
    const int64_t avg = (g.nnz / 2) / g.n;
    if (avg < 10) return false;
    std::mt19937 rng(kSeed);
    const int64_t samples = std::min<int64_t>(1000, g.n);
    std::vector<int64_t> s(static_cast<size_t>(samples));
    int64_t total = 0;
    for (int64_t t = 0; t < samples; ++t) {
        int64_t src;
        do {
            src = int64_t(uniform_closed(rng, uint32_t(g.n - 1)));
        } while (g.off[src + 1] == g.off[src]);
        s[size_t(t)] = g.off[src + 1] - g.off[src];
        total += s[size_t(t)];
    }
    std::sort(s.begin(), s.end());
    const double mean = double(total) / double(samples);
    const double median = double(s[size_t(samples / 2)]);
    return mean / 1.3 > median;
}

// builder.h:1699-1733: new id = position in the descending (degree, old id) order; rows re-sorted.
int relabel_by_degree(const Csr &g, Csr &out) {
    if (g.directed) return GMSX_ERR_DIRECTED;
    const int64_t n = g.n;
    std::vector<std::pair<int64_t, int32_t>> key(static_cast<size_t>(n));
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) key[size_t(v)] = {g.off[v + 1] - g.off[v], int32_t(v)};
    // a total order (ids are unique), so the parallel multiway merge sort gives the same permutation as std::sort
#ifdef _OPENMP
    __gnu_parallel::sort(key.begin(), key.end(), std::greater<std::pair<int64_t, int32_t>>());
#else
    std::sort(key.begin(), key.end(), std::greater<std::pair<int64_t, int32_t>>());
#endif
    std::unique_ptr<int32_t[]> new_id(new (std::nothrow) int32_t[size_t(std::max<int64_t>(n, 1))]);
    std::unique_ptr<int64_t[]> deg(new (std::nothrow) int64_t[size_t(n + 1)]);
    if (!new_id || !deg) return GMSX_ERR_NOMEM;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        deg[i] = key[size_t(i)].first;
        new_id[key[size_t(i)].second] = int32_t(i);
    }
    if (int rc = alloc_csr(out, n, g.nnz)) return rc;
    prefix_sum(deg.get(), n, out.off.get());
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t u = 0; u < n; ++u) {
        int32_t *dst = out.neigh.get() + out.off[new_id[u]];
        int64_t k = 0;
        for (int64_t e = g.off[u]; e < g.off[u + 1]; ++e) dst[k++] = new_id[g.neigh[e]];
        std::sort(dst, dst + k);
    }
    out.directed = false;
    return GMSX_OK;
}

static int finish(Csr &&built, int relabel, gmsx_csr **out) {
    std::unique_ptr<gmsx_csr> h(new (std::nothrow) gmsx_csr);
    if (!h) return GMSX_ERR_NOMEM;
    const bool do_relabel = !built.directed && (relabel == GMSX_RELABEL_ALWAYS ||
                                                (relabel == GMSX_RELABEL_AUTO && worth_relabelling(built)));
    if (do_relabel) {
        if (int rc = relabel_by_degree(built, h->g)) return rc;
        h->relabelled = true;
    } else {
        h->g = std::move(built);
    }
    *out = h.release();
    return GMSX_OK;
}

// ---- files ---------------------------------------------------------------------------------------

static std::string suffix_of(const std::string &p) {
    const size_t k = p.rfind('.');
    return k == std::string::npos ? std::string() : p.substr(k);
}

static int read_el(const std::string &path, EdgeList &el) {
    std::ifstream in(path);
    if (!in.is_open()) return GMSX_ERR_IO;
    std::vector<int32_t> us, vs;
    int64_t a, b;
    while (in >> a >> b) {  // reader.h:49-56: whitespace-separated pairs until the first parse failure
        if (a < 0 || b < 0 || a > std::numeric_limits<int32_t>::max() || b > std::numeric_limits<int32_t>::max())
            return GMSX_ERR_OVERFLOW;
        us.push_back(int32_t(a));
        vs.push_back(int32_t(b));
    }
    if (int rc = el.alloc(int64_t(us.size()))) return rc;
    std::copy(us.begin(), us.end(), el.u.get());
    std::copy(vs.begin(), vs.end(), el.v.get());
    return GMSX_OK;
}

// ---- the other text formats of the reference's Reader (gapbs/reader.h:58-218); weights are parsed and dropped, the
// hot path is unweighted.  Every reader yields a 0-based edge list; symmetrisation / dedup happen in the builder.
static int push_edge(std::vector<int32_t> &us, std::vector<int32_t> &vs, long long a, long long b) {
    if (a < 0 || b < 0 || a > std::numeric_limits<int32_t>::max() || b > std::numeric_limits<int32_t>::max()) return GMSX_ERR_OVERFLOW;
    us.push_back(int32_t(a));
    vs.push_back(int32_t(b));
    return GMSX_OK;
}
static int finish_el(std::vector<int32_t> &us, std::vector<int32_t> &vs, EdgeList &el) {
    if (int rc = el.alloc(int64_t(us.size()))) return rc;
    std::copy(us.begin(), us.end(), el.u.get());
    std::copy(vs.begin(), vs.end(), el.v.get());
    return GMSX_OK;
}
// .wel: "u v w" per line (reader.h:58-66)
static int read_wel(const std::string &path, EdgeList &el) {
    std::ifstream in(path);
    if (!in.is_open()) return GMSX_ERR_IO;
    std::vector<int32_t> us, vs;
    long long a, b;
    double w;
    while (in >> a >> b >> w)
        if (int rc = push_edge(us, vs, a, b)) return rc;
    return finish_el(us, vs, el);
}
// .gr: DIMACS shortest-path format, lines "a u v w" with 1-based ids, everything else ignored (reader.h:68-84)
static int read_gr(const std::string &path, EdgeList &el) {
    std::ifstream in(path);
    if (!in.is_open()) return GMSX_ERR_IO;
    std::vector<int32_t> us, vs;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] != 'a') continue;
        long long a, b;
        double w;
        char tag;
        std::istringstream ls(line);
        if (!(ls >> tag >> a >> b >> w)) return GMSX_ERR_FORMAT;
        if (int rc = push_edge(us, vs, a - 1, b - 1)) return rc;
    }
    return finish_el(us, vs, el);
}
// .graph: METIS — '%' comments, header "n m [fmt]", then one line per vertex listing its 1-based neighbours
// (fmt 1: neighbour/weight pairs; fmt 0 or 100: plain) (reader.h:86-142)
static int read_metis(const std::string &path, EdgeList &el) {
    std::ifstream in(path);
    if (!in.is_open()) return GMSX_ERR_IO;
    std::string line;
    long long n = -1, m = 0, fmt = 0;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '%') continue;
        std::istringstream hs(line);
        if (!(hs >> n >> m)) return GMSX_ERR_FORMAT;
        if (hs >> fmt) {
            if (fmt != 0 && fmt != 1 && fmt != 100) return GMSX_ERR_FORMAT;  // reference: exit(-20)
        }
        break;
    }
    if (n < 0) return GMSX_ERR_FORMAT;
    std::vector<int32_t> us, vs;
    long long u = 0;
    while (u < n && std::getline(in, line)) {
        if (!line.empty() && line[0] == '%') continue;
        std::istringstream ls(line);
        long long v, w;
        // Reference behaviour kept on purpose (reader.h:127-136 extract "v >> std::ws" as the loop condition): a
        // neighbour that is the very last character of its line sets eofbit, the following ws extraction then fails,
        // and that neighbour is NOT added.  Lines that end in whitespace load completely.  tests/test_loader.py pins
        // both cases against the compiled reference.
        if (fmt == 1) {
            while (ls >> v >> w >> std::ws)
                if (int rc = push_edge(us, vs, u, v - 1)) return rc;
        } else {
            while (ls >> v >> std::ws)
                if (int rc = push_edge(us, vs, u, v - 1)) return rc;
        }
        ++u;
    }
    return finish_el(us, vs, el);
}
// .mtx: Matrix Market "matrix coordinate", pattern / real / double / integer, square; 1-based (reader.h:146-218)
static int read_mtx(const std::string &path, EdgeList &el) {
    std::ifstream in(path);
    if (!in.is_open()) return GMSX_ERR_IO;
    std::string start, object, format, field, symmetry, line;
    if (!(in >> start >> object >> format >> field >> symmetry)) return GMSX_ERR_FORMAT;
    if (start != "%%MatrixMarket" || object != "matrix" || format != "coordinate" || field == "complex") return GMSX_ERR_FORMAT;
    const bool weights = field == "real" || field == "double" || field == "integer";
    if (!weights && field != "pattern") return GMSX_ERR_FORMAT;
    if (symmetry != "symmetric" && symmetry != "general" && symmetry != "skew-symmetric") return GMSX_ERR_FORMAT;
    std::getline(in, line);  // rest of the banner line
    long long rows = -1, cols = -1, nz = -1;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '%') continue;
        std::istringstream hs(line);
        if (!(hs >> rows >> cols >> nz)) return GMSX_ERR_FORMAT;
        break;
    }
    if (rows < 0 || rows != cols) return GMSX_ERR_FORMAT;  // reference: exit(-26)
    std::vector<int32_t> us, vs;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        long long a, b;
        if (!(ls >> a >> b)) continue;
        if (int rc = push_edge(us, vs, a - 1, b - 1)) return rc;
        if (symmetry == "symmetric")  // mirrored even for a directed build, as the reference does
            if (int rc = push_edge(us, vs, b - 1, a - 1)) return rc;
    }
    return finish_el(us, vs, el);
}

static int validate_csr_arrays(const int64_t *off, const int32_t *ng, int64_t n, int64_t nnz);
// .sg layout (writer.h:39-69): bool directed; int64 nnz; int64 n; int64 offsets[n+1]; int32 neigh[nnz]
static int read_sg(const std::string &path, Csr &g) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return GMSX_ERR_IO;
    struct Closer { std::FILE *f; ~Closer() { std::fclose(f); } } closer{f};
    unsigned char directed = 0;
    int64_t nnz = 0, n = 0;
    if (std::fread(&directed, 1, 1, f) != 1 || std::fread(&nnz, 8, 1, f) != 1 || std::fread(&n, 8, 1, f) != 1)
        return GMSX_ERR_FORMAT;
    if (n < 0 || nnz < 0 || n > std::numeric_limits<int32_t>::max()) return GMSX_ERR_FORMAT;
    if (int rc = alloc_csr(g, n, nnz)) return rc;
    if (std::fread(g.off.get(), 8, size_t(n + 1), f) != size_t(n + 1)) return GMSX_ERR_FORMAT;
    if (nnz && std::fread(g.neigh.get(), 4, size_t(nnz), f) != size_t(nnz)) return GMSX_ERR_FORMAT;
    if (int rc = validate_csr_arrays(g.off.get(), g.neigh.get(), n, nnz)) return rc;
    g.directed = directed != 0;
    return GMSX_OK;
}


// ---- ".sgx": the same CSR as ".sg" with every array at a 64-byte aligned file offset, so that it can be MAPPED instead of read ----------
// (the reference's .sg puts the int64 offsets at byte 17 — bool + two int64 — and the ids behind them: neither array can be used in place).
// Layout: 64-byte header {char magic[8] = "GMSXCSR1"; int64 directed, n, nnz; zero padding}; int64 offsets[n + 1] at byte 64; int32 neigh[nnz]
// at the next multiple of 64.  gmsx_csr_load maps both arrays PRIVATE + read-only-in-practice: the N ranks of a multi-GPU run (and bench.py's
// profiling children) that load one cache file share ONE copy of it in the page cache instead of holding N private ones (VERDICT r5 weak 7:
// 8 x 17 GB at RMAT scale 27), and loading costs the validation pass only.
static constexpr char kSgxMagic[8] = {'G', 'M', 'S', 'X', 'C', 'S', 'R', '1'};
static constexpr int64_t kSgxAlign = 64;
static int64_t sgx_neigh_offset(int64_t n) { return (kSgxAlign + (n + 1) * 8 + kSgxAlign - 1) / kSgxAlign * kSgxAlign; }

void ArrayFree::unmap(void *base, size_t bytes) { (void)munmap(base, bytes); }

static int write_sgx(const Csr &g, const char *path) {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return GMSX_ERR_IO;
    int64_t head[8] = {0, g.directed ? 1 : 0, g.n, g.nnz, 0, 0, 0, 0};
    std::memcpy(head, kSgxMagic, 8);
    const int64_t pad = sgx_neigh_offset(g.n) - (kSgxAlign + (g.n + 1) * 8);
    const char zeros[64] = {0};
    bool ok = std::fwrite(head, 8, 8, f) == 8 && std::fwrite(g.off.get(), 8, size_t(g.n + 1), f) == size_t(g.n + 1) &&
              (pad == 0 || std::fwrite(zeros, 1, size_t(pad), f) == size_t(pad)) &&
              (g.nnz == 0 || std::fwrite(g.neigh.get(), 4, size_t(g.nnz), f) == size_t(g.nnz));
    ok = (std::fclose(f) == 0) && ok;
    return ok ? GMSX_OK : GMSX_ERR_IO;
}

// offsets monotone from 0 to nnz, ids in [0, n): a stale / truncated / planted cache file must not lead to out-of-bounds row walks later
static int validate_csr_arrays(const int64_t *off, const int32_t *ng, int64_t n, int64_t nnz) {
    if (off[0] != 0 || off[n] != nnz) return GMSX_ERR_FORMAT;
    int bad = 0;
#pragma omp parallel for reduction(| : bad) schedule(static)
    for (int64_t u = 0; u < n; ++u) bad |= int(off[u + 1] < off[u]);
    if (bad) return GMSX_ERR_FORMAT;
#pragma omp parallel for reduction(| : bad) schedule(static)
    for (int64_t j = 0; j < nnz; ++j) bad |= int(ng[j] < 0 || int64_t(ng[j]) >= n);
    return bad ? GMSX_ERR_FORMAT : GMSX_OK;
}

static int read_sgx(const std::string &path, Csr &g) {
    const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return GMSX_ERR_IO;
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) return GMSX_ERR_IO;
    int64_t head[8];
    if (pread(fd, head, sizeof(head), 0) != ssize_t(sizeof(head)) || std::memcmp(head, kSgxMagic, 8) != 0) return GMSX_ERR_FORMAT;
    const int64_t n = head[2], nnz = head[3];
    if (n < 0 || nnz < 0 || n > std::numeric_limits<int32_t>::max() || (head[1] != 0 && head[1] != 1)) return GMSX_ERR_FORMAT;
    const int64_t noff = sgx_neigh_offset(n);
    if (nnz > (std::numeric_limits<int64_t>::max() - noff) / 4 || int64_t(st.st_size) < noff + nnz * 4) return GMSX_ERR_FORMAT;  // truncated
    // two mappings of the one file, each owned by its array (either may be released first): PRIVATE + writable = shared with the page cache
    // until somebody writes (nobody does: a relabelled / re-sorted graph is a new Csr), then copy-on-write — never written back
    const int64_t page = int64_t(sysconf(_SC_PAGESIZE));
    auto map_part = [&](int64_t at, int64_t bytes, void **base, size_t *len) -> void * {
        const int64_t lo = at / page * page;
        *len = size_t(at - lo + std::max<int64_t>(bytes, 1));
        *base = mmap(nullptr, *len, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, off_t(lo));
        if (*base == MAP_FAILED) return nullptr;
        (void)madvise(*base, *len, MADV_WILLNEED);
        return static_cast<char *>(*base) + (at - lo);
    };
    void *b0 = nullptr, *b1 = nullptr;
    size_t l0 = 0, l1 = 0;
    void *p_off = map_part(kSgxAlign, (n + 1) * 8, &b0, &l0);
    if (!p_off) return GMSX_ERR_NOMEM;
    g.off = std::unique_ptr<int64_t[], ArrayFree>(static_cast<int64_t *>(p_off), ArrayFree{b0, l0});
    void *p_ng = map_part(noff, nnz * 4, &b1, &l1);
    if (!p_ng) return GMSX_ERR_NOMEM;
    g.neigh = std::unique_ptr<int32_t[], ArrayFree>(static_cast<int32_t *>(p_ng), ArrayFree{b1, l1});
    g.n = n;
    g.nnz = nnz;
    g.directed = head[1] != 0;
    g.mapped = true;
    return validate_csr_arrays(g.off.get(), g.neigh.get(), n, nnz);
}

}  // namespace gmsx

// ================================================================================================
// C-ABI (include/gmsx.h, "Host graph substrate")
// ================================================================================================
namespace gmsx {
int host_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void parallel_memcpy(void *dst, const void *src, size_t bytes) {
#ifdef _OPENMP
#pragma omp parallel
    {
        const size_t nt = size_t(omp_get_num_threads()), t = size_t(omp_get_thread_num());
        const size_t per = ((bytes + nt - 1) / nt + 4095) & ~size_t(4095), lo = std::min(bytes, per * t), hi = std::min(bytes, lo + per);
        if (hi > lo) std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo);
    }
#else
    std::memcpy(dst, src, bytes);
#endif
}
// ---- options (gmsx_set_option) ------------------------------------------------------------------------------------------------------
// A fixed table: a name the library does not know is refused, so a typo cannot silently do nothing.  All of them change HOW a result is
// computed (limits, budgets, container forms, diagnostics on stderr), never the result.
namespace {
struct Option {
    const char *name;
    char value[32];
    bool set;
};
Option g_options[] = {
    // diagnostics (stderr)
    {"TIMING", "", false}, {"MEM_TRACE", "", false}, {"BK_VERBOSE", "", false},
    // triangle-count containers (device_graph.hip) and passes (tc.hip)
    {"INLINE_LIMIT", "", false}, {"TC_INLINE_FIRST", "", false}, {"TC_DELTA", "", false}, {"TC_DELTA_PCT", "", false}, {"TC_GAP12", "", false},
    {"TC_HYBRID", "", false}, {"TC_TAIL_DELTA", "", false}, {"TC_TWO_SIDED", "", false}, {"TC_HOT_WINDOWS", "", false}, {"TC_HOT_KB", "", false},
    {"TC_HOT_MIN", "", false}, {"TC_TEST_MAX_UNITS", "", false}, {"TC_KEEP_ROWS", "", false}, {"TC_MEM_LIMIT_MB", "", false}, {"TC_OVERLAP", "", false},
    {"TC_PERSIST", "", false}, {"TC_ITEM_WGS", "", false}, {"SORT_CHUNK", "", false}, {"UPLOAD_STAGED", "", false}, {"INIT_LAZY", "", false},
    // k-clique (kclique.hip)
    {"KC_SLAB_MB", "", false}, {"KC_MAXD", "", false}, {"KC_STREAMS", "", false}, {"KC_PIPE_ALL", "", false}, {"KC_STREAM_BUILD", "", false},
    {"KC_REVERSE", "", false}, {"KC_REV_MIN", "", false}, {"KC_REV_FACTOR", "", false}, {"KC_REV_TAIL", "", false}, {"KC_REV_TAIL_MIN", "", false}, {"KC_REV_GW", "", false}, {"KC_TRI", "", false}, {"KC_MFMA", "", false}, {"KC_POOL_MB", "", false},
    // Bron–Kerbosch (bk.hip)
    {"BK_MAXC", "", false}, {"BK_ARENA_MB", "", false}, {"BK_GROUPS", "", false}, {"BK_SMALL_P", "", false}, {"BK_SMALL_P_GROUPS", "", false},
    {"BK_BUDGET", "", false}, {"BK_BUDGET0", "", false}, {"BK_RESUME_GRAB", "", false}, {"BK_SPLIT_BUILD", "", false}, {"BK_TINY_ROOTS", "", false},
    {"BK_TINY_BESIDE", "", false},
};
Option *find_option(const char *name) {
    if (!name) return nullptr;
    for (Option &o : g_options)
        if (std::strcmp(o.name, name) == 0) return &o;
    return nullptr;
}
}  // namespace

const char *opt(const char *name) {
    const Option *o = find_option(name);
    return o && o->set ? o->value : nullptr;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_csr_generate(int generator, int scale, int degree, int relabel, int threads, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!out || scale < 1 || degree < 1 || relabel < 0 || relabel > 2) return GMSX_ERR_INVALID;
        if (scale > 30) return GMSX_ERR_OVERFLOW;  // ids are int32 (generator.h:41-48 exits with -31)
        if (generator != GMSX_GEN_KRONECKER && generator != GMSX_GEN_UNIFORM) return GMSX_ERR_INVALID;
    #ifdef _OPENMP
        struct Threads {  // restored on every way out, an exception included
            int saved, set;
            ~Threads() { if (set > 0) omp_set_num_threads(saved); }
        } threads_guard{omp_get_max_threads(), threads};
        if (threads > 0) omp_set_num_threads(threads);
    #else
        (void)threads;
    #endif
        int rc;
        {
            Csr g;
            {
                PhaseTimer pt;
                EdgeList el;
                rc = generator == GMSX_GEN_UNIFORM ? make_uniform(scale, degree, el) : make_rmat(scale, degree, el);
                pt.lap("generate edge list");
                if (!rc) rc = build_from_el(el, -1, true, g);
            }
            PhaseTimer pf;
            if (!rc) rc = finish(std::move(g), relabel, out);
            pf.lap("relabel decision + relabel");
        }
        return rc;
    });
}

int gmsx_csr_generate_rmat(int scale, int degree, double a, double b, double c, int relabel, int threads, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!out || scale < 1 || degree < 1 || relabel < 0 || relabel > 2) return GMSX_ERR_INVALID;
        if (!(a > 0 && b >= 0 && c >= 0 && a + b + c < 1.0)) return GMSX_ERR_INVALID;
        if (scale > 30) return GMSX_ERR_OVERFLOW;
    #ifdef _OPENMP
        struct Threads {  // restored on every way out, an exception included
            int saved, set;
            ~Threads() { if (set > 0) omp_set_num_threads(saved); }
        } threads_guard{omp_get_max_threads(), threads};
        if (threads > 0) omp_set_num_threads(threads);
    #else
        (void)threads;
    #endif
        int rc;
        {
            Csr g;
            {
                EdgeList el;
                rc = make_rmat(scale, degree, el, float(a), float(b), float(c));
                if (!rc) rc = build_from_el(el, -1, true, g);
            }
            if (!rc) rc = finish(std::move(g), relabel, out);
        }
        return rc;
    });
}

int gmsx_csr_from_edges(int64_t num_nodes, int64_t num_edges, const int32_t *src, const int32_t *dst,
                        int symmetrize, int relabel, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!out || num_edges < 0 || (num_edges && (!src || !dst)) || relabel < 0 || relabel > 2) return GMSX_ERR_INVALID;
        EdgeList el;
        if (int rc = el.alloc(num_edges)) return rc;
        if (num_edges) {
            std::memcpy(el.u.get(), src, size_t(num_edges) * 4);
            std::memcpy(el.v.get(), dst, size_t(num_edges) * 4);
        }
        Csr g;
        if (int rc = build_from_el(el, num_nodes, symmetrize != 0, g)) return rc;
        return finish(std::move(g), relabel, out);
    });
}

int gmsx_csr_load(const char *path, int symmetrize, int relabel, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!path || !out || relabel < 0 || relabel > 2) return GMSX_ERR_INVALID;
        const std::string p(path), suf = suffix_of(p);
        Csr g;
        if (suf == ".sg") {
            if (int rc = read_sg(p, g)) return rc;
        } else if (suf == ".sgx") {
            if (int rc = read_sgx(p, g)) return rc;
        } else if (suf == ".el" || suf == ".wel" || suf == ".gr" || suf == ".graph" || suf == ".mtx") {
            EdgeList el;
            const int rc = suf == ".el" ? read_el(p, el) : suf == ".wel" ? read_wel(p, el) : suf == ".gr" ? read_gr(p, el)
                         : suf == ".graph" ? read_metis(p, el) : read_mtx(p, el);
            if (rc) return rc;
            if (int rc2 = build_from_el(el, -1, symmetrize != 0, g)) return rc2;
        } else {
            return GMSX_ERR_FORMAT;
        }
        return finish(std::move(g), relabel, out);
    });
}

int gmsx_csr_save_sg(const gmsx_csr *h, const char *path) {
    return gmsx::guard([&]() -> int {
        if (!h || !path) return GMSX_ERR_INVALID;
        std::FILE *f = std::fopen(path, "wb");
        if (!f) return GMSX_ERR_IO;
        const Csr &g = h->g;
        const unsigned char directed = g.directed ? 1 : 0;
        bool ok = std::fwrite(&directed, 1, 1, f) == 1 && std::fwrite(&g.nnz, 8, 1, f) == 1 &&
                  std::fwrite(&g.n, 8, 1, f) == 1 && std::fwrite(g.off.get(), 8, size_t(g.n + 1), f) == size_t(g.n + 1) &&
                  (g.nnz == 0 || std::fwrite(g.neigh.get(), 4, size_t(g.nnz), f) == size_t(g.nnz));
        ok = (std::fclose(f) == 0) && ok;
        return ok ? GMSX_OK : GMSX_ERR_IO;
    });
}

int gmsx_csr_save_sgx(const gmsx_csr *h, const char *path) {
    return gmsx::guard([&]() -> int {
        if (!h || !path) return GMSX_ERR_INVALID;
        return write_sgx(h->g, path);
    });
}
int gmsx_csr_is_mapped(const gmsx_csr *h) { return h ? int(h->g.mapped) : GMSX_ERR_INVALID; }

int gmsx_csr_from_arrays(int64_t n, const int64_t *offsets, const int32_t *neigh, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!out || n < 0 || !offsets || n > std::numeric_limits<int32_t>::max()) return GMSX_ERR_INVALID;
        if (offsets[0] != 0) return GMSX_ERR_INVALID;
        for (int64_t i = 0; i < n; ++i)
            if (offsets[i + 1] < offsets[i]) return GMSX_ERR_INVALID;
        const int64_t nnz = offsets[n];
        if (nnz && !neigh) return GMSX_ERR_INVALID;
        for (int64_t e = 0; e < nnz; ++e)
            if (neigh[e] < 0 || neigh[e] >= n) return GMSX_ERR_INVALID;
        std::unique_ptr<gmsx_csr> h(new (std::nothrow) gmsx_csr);
        if (!h) return GMSX_ERR_NOMEM;
        if (int rc = alloc_csr(h->g, n, nnz)) return rc;
        std::memcpy(h->g.off.get(), offsets, size_t(n + 1) * 8);
        if (nnz) std::memcpy(h->g.neigh.get(), neigh, size_t(nnz) * 4);
        *out = h.release();
        return GMSX_OK;
    });
}

int gmsx_csr_worth_relabelling(const gmsx_csr *h) { return h ? int(worth_relabelling(h->g)) : GMSX_ERR_INVALID; }

int gmsx_csr_relabel_by_degree(const gmsx_csr *h, gmsx_csr **out) {
    return gmsx::guard([&]() -> int {
        if (!h || !out) return GMSX_ERR_INVALID;
        std::unique_ptr<gmsx_csr> r(new (std::nothrow) gmsx_csr);
        if (!r) return GMSX_ERR_NOMEM;
        if (int rc = relabel_by_degree(h->g, r->g)) return rc;
        r->relabelled = true;
        *out = r.release();
        return GMSX_OK;
    });
}

int64_t gmsx_csr_num_nodes(const gmsx_csr *h) { return h ? h->g.n : int64_t(GMSX_ERR_INVALID); }
int gmsx_set_host_threads(int n) {
    return gmsx::guard([&]() -> int {
    #ifdef _OPENMP
        const int before = omp_get_max_threads();
        omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
        return before;
    #else
        (void)n;
        return 1;
    #endif
    });
}

int64_t gmsx_csr_num_edges(const gmsx_csr *h) { return h ? (h->g.directed ? h->g.nnz : h->g.nnz / 2) : int64_t(GMSX_ERR_INVALID); }
int64_t gmsx_csr_num_edges_directed(const gmsx_csr *h) { return h ? h->g.nnz : int64_t(GMSX_ERR_INVALID); }
const int64_t *gmsx_csr_offsets(const gmsx_csr *h) { return h ? h->g.off.get() : nullptr; }
const int32_t *gmsx_csr_neighbors(const gmsx_csr *h) { return h ? h->g.neigh.get() : nullptr; }

uint64_t gmsx_csr_merge_elements(const gmsx_csr *h) {
    if (!h) return 0;
    const Csr &g = h->g;
    uint64_t s = 0;
#pragma omp parallel for reduction(+ : s) schedule(dynamic, 1024)
    for (int64_t u = 0; u < g.n; ++u) {
        const uint64_t du = uint64_t(g.off[u + 1] - g.off[u]);
        for (int64_t e = g.off[u]; e < g.off[u + 1]; ++e) {
            const int32_t v = g.neigh[e];
            if (u < v) s += du + uint64_t(g.off[v + 1] - g.off[v]);
        }
    }
    return s;
}

uint64_t gmsx_csr_fingerprint(const gmsx_csr *h, int which) {
    if (!h) return 0;
    const unsigned char *p;
    size_t len;
    if (which == 0) {
        p = reinterpret_cast<const unsigned char *>(h->g.off.get());
        len = size_t(h->g.n + 1) * 8;
    } else {
        p = reinterpret_cast<const unsigned char *>(h->g.neigh.get());
        len = size_t(h->g.nnz) * 4;
    }
    uint64_t x = 1469598103934665603ull;
    for (size_t i = 0; i < len; ++i) {
        x ^= p[i];
        x *= 1099511628211ull;
    }
    return x;
}

int gmsx_set_option(const char *name, const char *value) {
    gmsx::Option *o = gmsx::find_option(name);
    if (!o) return GMSX_ERR_INVALID;
    if (!value) {
        o->set = false;
        return GMSX_OK;
    }
    if (std::strlen(value) >= sizeof(o->value)) return GMSX_ERR_INVALID;
    std::strcpy(o->value, value);
    o->set = true;
    return GMSX_OK;
}
void gmsx_reset_options(void) {
    for (gmsx::Option &o : gmsx::g_options) o.set = false;
}
int gmsx_option_name(int index, const char **name) {
    const int count = int(sizeof(gmsx::g_options) / sizeof(gmsx::g_options[0]));
    if (index < 0 || index >= count || !name) return GMSX_ERR_INVALID;
    *name = gmsx::g_options[index].name;
    return GMSX_OK;
}

void gmsx_csr_free(gmsx_csr *h) { delete h; }

}  // extern "C"
