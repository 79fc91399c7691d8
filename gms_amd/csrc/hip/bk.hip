// Bron–Kerbosch maximal-clique COUNT on gfx950: the device replacement for
//   BkEppsteinPar::mceBench   gms/algorithms/set_based/maximal_clique_enum/parallel/eppsteinPAR.h:18-53
//   BkTomita::expand/findPivot gms/algorithms/set_based/maximal_clique_enum/sequential/tomita.h:12-86
// compiled with -DBK_COUNT (the reference's count-only build: BK_CLIQUE_COUNTER, helper.h:15).
//
// Eppstein outer loop: every vertex v starts one search with cand = higher-ranked neighbours, fini = lower-ranked
// ones.  The count does not depend on the rank, so the device uses its own degree rank: cand = the
// oriented row N+(v) (|cand| <= d+max), fini = the in-neighbours N-(v).  Inside a search all sets are BITMAPS over the
// local universe and every set operation of the reference becomes a word-wise AND ("RoaringSet bitmap-AND"):
//   cand.intersect(N(q))  ->  P  & Cadj[q]      (c-bit rows, one 32-bit word per lane)
//   fini.intersect(N(q))  ->  Xc & Cadj[q]  (finished candidates)   and   Xf & XT[q]  (the in-neighbours, x-bit rows)
//   cand.difference(N(p)) ->  P & ~Cadj[p]
//   findPivot             ->  argmax_u popc(P & Cadj[u]) over u in P ∪ Xc  (any pivot choice yields the same count)
// Round 0: start vertices whose structures fit 8 KB of LDS are built and searched by one wave each (k_bk_wave<true>, eight per queue ticket);
// the others are BUILT by workgroups — k_bk_block, a piece of 2048 row jobs of a start vertex per work item, Cadj | XT straight into an arena
// — and leave a root record.  Rounds >= 1 (k_bk_resume): one wave per record runs the recursion with an explicit stack in a global slab and,
// past its node budget, re-splits what is left into records for the next round.
#include "device_graph.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace gmsx {

static constexpr int kLdsSlabWords = 2048;  // 8 KB per wave: tasks of at most ~90 candidates; with the trimmed scratch below 17 waves fit a CU
static constexpr unsigned long long kEmptySlot = ~0ull;
static constexpr unsigned long long kWideTask = 1ull << 62;  // task key flag: more than 2048 candidates
static constexpr int kBkMaxCand = 16384;  // widest bit rows of the register-resident search kernels: eight words per lane; start
                                          // vertices with more candidates run on the memory-resident search (k_bk_wave<false, 0>)

// ---- k_bk_resume4 (below): searches of 4 / 8 / 16 lanes
static constexpr int kBkGroupMaxC = 512;
static constexpr int kBkSlot = 8;                   // words per lane of a saved level: P, Xc, ext, xfn | list index, list word, -, -
#ifndef GMSX_BK_PARENT_REGS
#define GMSX_BK_PARENT_REGS 1  // the level above the current one stays in registers (k_bk_resume4): -1 … 2 ms on configs[3]
#endif
// lanes per search by candidate count: 16 lanes hold 512 candidates, 8 hold 256, 4 hold 128 (GMSX_BK_GROUP_MIN: the narrowest group compiled in)
#ifndef GMSX_BK_GROUP_MIN
#define GMSX_BK_GROUP_MIN 16  // measured on configs[3] (all records <= 512 candidates, 60 % of the nodes <= 256): 16 -> 151 ms, 8 -> 155-167, 4 -> 180-190: more searches
                              // per wave put every block of the step machine on every trip and eight waits behind one memory walk
#endif
__host__ __device__ inline int bk_group_lanes(long long c) { return c <= 128 && GMSX_BK_GROUP_MIN <= 4 ? 4 : c <= 256 && GMSX_BK_GROUP_MIN <= 8 ? 8 : 16; }
__host__ __device__ inline int bk_group_class(long long c) { const int g = bk_group_lanes(c); return g == 16 ? 0 : g == 8 ? 1 : 2; }
// a saved level: kBkSlot words per lane of the group + 2 words per possible list pair; a search may go c levels deep
__host__ __device__ inline unsigned long long bk_group_level_words(int lanes, long long xw) { return (unsigned long long)(lanes * kBkSlot + ((2 * xw + 3) & ~3ll)); }
__host__ __device__ inline unsigned long long bk_group_slab_words(long long c, long long xw) { return (unsigned long long)(c + 1) * bk_group_level_words(bk_group_lanes(c), xw); }

__host__ __device__ inline uint32_t bk_map_size(int c) {
    uint32_t s = 64;
    while (s < 2u * uint32_t(c)) s <<= 1;
    return s;
}
__host__ __device__ inline unsigned long long bk_slab_words(int c, long long x) {
    const unsigned long long cw = (unsigned long long)((c + 31) / 32), xw = (unsigned long long)((x + 31) / 32);
    return 2ull * bk_map_size(c) + (unsigned long long)c * cw + (unsigned long long)c * xw +
           (unsigned long long)(c + 1) * (3 * cw + xw + 1);
}

// slab of a memory-resident search: the usual structures + the pivot-candidate list (c words) + one Xf flag byte per level
__host__ __device__ inline unsigned long long bk_slab_words_mem(int c, long long x) {
    return bk_slab_words(c, x) + (unsigned long long)c + (unsigned long long)(c + 8) / 4ull + 8ull;
}

// per start vertex: slab requirement (0 = no search needed); isolated vertices are counted right here
__global__ void k_bk_tasks(int64_t n, const int64_t *__restrict__ off, const int32_t *__restrict__ oldid,
                           const int32_t *__restrict__ dplus, unsigned long long *__restrict__ keys, int32_t *__restrict__ vals,
                           int max_c, int32_t *__restrict__ giant, unsigned long long giant_cap,
                           unsigned long long *__restrict__ acc /* [0] isolated count, [1] giant tasks, [2] their largest slab, [7] tasks with > 2048 candidates */) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int32_t o = oldid[v];
    const long long deg = off[o + 1] - off[o];
    const int c = dplus[v];
    const long long x = deg - c;
    unsigned long long w = 0;
    if (c == 0) {
        if (deg == 0) atomicAdd(&acc[0], 1ull);  // an isolated vertex is a maximal clique (eppsteinPAR.h:32-47, tomita.h:73-78)
    } else {
        w = bk_slab_words(c, x);
        if (c > max_c) {  // too wide for the register-resident search: its own list, run after the rounds by the memory-resident search
            const unsigned long long at = atomicAdd(&acc[1], 1ull);
            if (at < giant_cap) giant[at] = int32_t(v);
            atomicMax(&acc[2], bk_slab_words_mem(c, x));
            w = 0;
        } else if (c > 2048) {  // more than one word per lane: these tasks sort first and run on the WPL = 2 / 4 / 8 kernels
            w |= kWideTask;
            atomicAdd(&acc[7], 1ull);
        }
    }
    keys[v] = w;
    vals[v] = int32_t(v);
}

__device__ __forceinline__ int64_t bk_readlane64(int64_t x, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(uint64_t(x)), l);
    const uint32_t hi = __builtin_amdgcn_readlane(uint32_t(uint64_t(x) >> 32), l);
    return int64_t((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ uint32_t bk_hash(int32_t w, uint32_t mask) { return (uint32_t(w) * 0x9E3779B1u >> 7) & mask; }

__device__ __forceinline__ int bk_find(const unsigned long long *map, uint32_t mask, int32_t w) {
    uint32_t h = bk_hash(w, mask);
    while (true) {
        const unsigned long long s = __hip_atomic_load(&map[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == kEmptySlot) return -1;
        if (int32_t(s >> 32) == w) return int(s & 0xffffffffull);
        h = (h + 1) & mask;
    }
}

// scan both oriented containers of rank id a; for every id that is a member of C call f(local index)
template <class F>
__device__ __forceinline__ void bk_scan_row(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                            const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, int32_t a,
                                            const unsigned long long *map, uint32_t mask, int start, int stride, F f) {
    const int64_t hb = hoff[a], he = hoff[a + 1];
    for (int64_t j = hb + start; j < he; j += stride) {
        const uint32_t w = hadj[j];
        if (w != 0xFFFFu) {
            const int k = bk_find(map, mask, int32_t(w));
            if (k >= 0) f(k);
        }
    }
    const int64_t tb = toff[a], te = toff[a + 1];
    for (int64_t j = tb + start; j < te; j += stride) {
        const int k = bk_find(map, mask, tadj[j]);
        if (k >= 0) f(k);
    }
}

// two packed 16-bit ids of a 16-byte hub-container load
struct __attribute__((packed, aligned(4))) bk_u4 { uint32_t x, y, z, w; };
template <class F>
__device__ __forceinline__ void bk_probe2(const unsigned long long *map, uint32_t mask, uint32_t pair, bool valid, F &f) {
    const uint32_t w0 = pair & 0xffffu, w1 = pair >> 16;
    if (valid && w0 != 0xFFFFu) { const int k = bk_find(map, mask, int32_t(w0)); if (k >= 0) f(k); }
    if (valid && w1 != 0xFFFFu) { const int k = bk_find(map, mask, int32_t(w1)); if (k >= 0) f(k); }
}
// 16-lane-group scan of the oriented row of rank id a: 16-byte loads of the hub container (8 ids per lane), the tail
// container with a stride of 16
template <class F>
__device__ __forceinline__ void bk_scan_row_group(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, int32_t a,
                                                  const unsigned long long *map, uint32_t mask, int sub, F f) {
    const int64_t hb = hoff[a], he = hoff[a + 1];
    const int64_t tb = toff[a], te = toff[a + 1];
    for (int64_t j = hb + sub * 8; j < he; j += 128) {
        const bk_u4 cur = *reinterpret_cast<const bk_u4 *>(hadj + j);
        const int64_t left = he - j;  // even, >= 2
        bk_probe2(map, mask, cur.x, true, f);
        bk_probe2(map, mask, cur.y, left > 2, f);
        bk_probe2(map, mask, cur.z, left > 4, f);
        bk_probe2(map, mask, cur.w, left > 6, f);
    }
    // the tail container with 16-byte loads, two in flight per lane: 128 ids per trip of the group.  (One 4-byte load per lane and trip —
    // 16 ids — made a candidate row of a few hundred ids a chain of a dozen dependent round trips: most of the LDS tasks' build time.)
    for (int64_t j = tb + sub * 4; j < te; j += 128) {
        const bk_u4 cur = *reinterpret_cast<const bk_u4 *>(tadj + j);  // tadj is padded by four ids
        const bool two = j + 64 < te;
        bk_u4 nxt{0u, 0u, 0u, 0u};
        if (two) nxt = *reinterpret_cast<const bk_u4 *>(tadj + j + 64);
        const int64_t left = te - j;  // >= 1
        { const int k = bk_find(map, mask, int32_t(cur.x)); if (k >= 0) f(k); }
        if (left > 1) { const int k = bk_find(map, mask, int32_t(cur.y)); if (k >= 0) f(k); }
        if (left > 2) { const int k = bk_find(map, mask, int32_t(cur.z)); if (k >= 0) f(k); }
        if (left > 3) { const int k = bk_find(map, mask, int32_t(cur.w)); if (k >= 0) f(k); }
        if (two) {
            { const int k = bk_find(map, mask, int32_t(nxt.x)); if (k >= 0) f(k); }
            if (left > 65) { const int k = bk_find(map, mask, int32_t(nxt.y)); if (k >= 0) f(k); }
            if (left > 66) { const int k = bk_find(map, mask, int32_t(nxt.z)); if (k >= 0) f(k); }
            if (left > 67) { const int k = bk_find(map, mask, int32_t(nxt.w)); if (k >= 0) f(k); }
        }
    }
}

// 8-lane-group scan of the oriented row of rank id a (rows of in-neighbours: ~d+avg ids): 16-byte loads of both
// containers — 8 hub ids / 4 tail ids per lane and step — so the lanes of a group share one row instead of each
// walking its own (a wave of single-row lanes idles behind its longest row)
template <class F>
__device__ __forceinline__ void bk_scan_row_group8(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                   const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, int32_t a,
                                                   const unsigned long long *map, uint32_t mask, int sub, F f) {
    const int64_t hb = hoff[a], he = hoff[a + 1];
    const int64_t tb = toff[a], te = toff[a + 1];
    for (int64_t j = hb + sub * 8; j < he; j += 64) {
        const bk_u4 cur = *reinterpret_cast<const bk_u4 *>(hadj + j);
        const int64_t left = he - j;  // even, >= 2
        bk_probe2(map, mask, cur.x, true, f);
        bk_probe2(map, mask, cur.y, left > 2, f);
        bk_probe2(map, mask, cur.z, left > 4, f);
        bk_probe2(map, mask, cur.w, left > 6, f);
    }
    for (int64_t j = tb + sub * 4; j < te; j += 64) {  // two loads in flight per lane: 64 ids per trip of the group
        const bk_u4 cur = *reinterpret_cast<const bk_u4 *>(tadj + j);  // tadj is padded by four ids
        const bool two = j + 32 < te;
        bk_u4 nxt{0u, 0u, 0u, 0u};
        if (two) nxt = *reinterpret_cast<const bk_u4 *>(tadj + j + 32);
        const int64_t left = te - j;  // >= 1
        { const int k = bk_find(map, mask, int32_t(cur.x)); if (k >= 0) f(k); }
        if (left > 1) { const int k = bk_find(map, mask, int32_t(cur.y)); if (k >= 0) f(k); }
        if (left > 2) { const int k = bk_find(map, mask, int32_t(cur.z)); if (k >= 0) f(k); }
        if (left > 3) { const int k = bk_find(map, mask, int32_t(cur.w)); if (k >= 0) f(k); }
        if (two) {
            { const int k = bk_find(map, mask, int32_t(nxt.x)); if (k >= 0) f(k); }
            if (left > 33) { const int k = bk_find(map, mask, int32_t(nxt.y)); if (k >= 0) f(k); }
            if (left > 34) { const int k = bk_find(map, mask, int32_t(nxt.z)); if (k >= 0) f(k); }
            if (left > 35) { const int k = bk_find(map, mask, int32_t(nxt.w)); if (k >= 0) f(k); }
        }
    }
}

__device__ __forceinline__ int wave_sum(int x) {
    for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s);
    return x;
}

// ---- load balancing ------------------------------------------------------------------------------------------
// A search that exceeds its node budget is not finished by its wave: the wave copies the start vertex's read-only
// structures (Cadj, XT) to a persistent ARENA, writes every stack level that still has unexplored branches as a
// resumable RECORD into the next round's pool, and moves on.  The host launches rounds until the pool is empty, so a
// huge search tree is re-split level by level across thousands of waves without any inter-wave synchronisation
// inside a kernel.  If the arena or the pool is full the wave simply keeps searching (correctness never depends on it).
#ifdef GMSX_BK_STATS
__device__ unsigned long long g_bk_nodes;  // profiling build: nodes of the register-resident searches since the last read
// profiling build: [0..7] entered nodes by candidate width c of the start vertex (<= 32, 64, 128, 256, 512, 1024, 2048, more); [8..15] entered nodes by
// |P| (0, 1, 2-6, 7-16, 17-32, 33-64, 65-256, more); [16] entered with Xf non-empty; [17] branch steps; [18] leaf fast paths; [19] one-candidate-child
// fast paths; [20] pushes; [21] pivot-scored nodes; [22] sum of |P u Xc| over them; [23] deepest level; [24] Xf words ANDed; [25] one-candidate nodes
// [26] Xf checks of the fast paths, [27] their words, [28] cycles in Xf loops, [29] cycles in bk_search, [30] cycles in pivot scoring, [31] non-zero words of the child Xf of a push
__device__ unsigned long long g_bk_hist[32];
__device__ unsigned long long g_bkg_stat[20];  // k_bk_resume4's trip statistics (see there)
#define BK_STAT(i, v) do { if (lane == 0) atomicAdd(&g_bk_hist[i], (unsigned long long)(v)); } while (0)
#define BK_T0() const long long t0_ = clock64()
#define BK_T1(i) BK_STAT(i, clock64() - t0_)
#else
#define BK_T0() do { } while (0)
#define BK_T1(i) do { } while (0)
#define BK_STAT(i, v) do { } while (0)
#endif
struct BkShared {
    uint32_t *arena;
    unsigned long long arena_cap;       // words
    unsigned long long *arena_head;     // words used
    uint32_t *pool;                     // next round's records
    unsigned long long pool_cap;        // words
    unsigned long long *pool_head;
    unsigned long long *dir;            // word offsets of the records in `pool`
    unsigned long long dir_cap;
    unsigned long long *dir_count;
    unsigned long long *max_stack;      // max (c+1)*lvl over the dumped records
    unsigned budget;                    // nodes per task before it is split
    int small_p;                        // nodes with at most this many candidates take their first candidate as the pivot (no scoring)
    int small_p_groups;                 // the same for k_bk_resume4, where such a pivot costs nothing (its row is the first branch's row)
    const int64_t *bmoff;               // bitset containers of the hub rows (device_graph.hpp)
    const uint32_t *bmpool;
    int32_t dense_limit;
};
#ifndef GMSX_BK_SPLIT
#define GMSX_BK_SPLIT 8
#endif
static constexpr int kBkSplit = GMSX_BK_SPLIT;    // records a level with pending branches is cut into when a search is split
static constexpr int kRecHeader = 8;  // v, c, x, xf_ne, arena offset (2 words), 2 spare
static constexpr unsigned long long kNoArena = ~0ull;

// Iterative Tomita recursion on bitmaps, resumable.  Registers: this lane's WPL words of P / Xc / ext (word lane + 64 h;
// words >= cw hold 0; WPL = 1 up to 2048 candidates, 2 up to 4096); Xf levels and the saved words of the ancestors live in
// `stack` (level l at stack + l*lvl: P, Xc, ext, Xf).
// `structs`/`struct_words`: where Cadj|XT currently live (for the copy to the arena); arena_off: kNoArena until copied.
template <int WPL>
__device__ __forceinline__ void bk_search(const uint32_t *Cadj, const uint32_t *XT, uint32_t *stack, unsigned char *xfne_stack,
                                          int32_t v, int c, int x, uint32_t (&P)[WPL], uint32_t (&Xc)[WPL], uint32_t (&ext)[WPL], int xf_ne,
                                          bool entering, int lane, unsigned long long &cnt, const BkShared &sh, unsigned long long arena_off,
                                          bool global_structs, uint32_t *piv_P /* LDS, 64*WPL words: the current P for all lanes */,
                                          unsigned short *piv_list /* LDS, 2048*WPL entries: members of P ∪ Xc */,
                                          unsigned long long &node_words /* += nodes visited x words of a Cadj row: gmsx_stats.stream_bytes */) {
    const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
    const int lvl = 3 * cw + xw + 1;
    unsigned budget = sh.budget, nodes = 0;
    int depth = 0;
#ifdef GMSX_BK_STATS
    const long long t_all_ = clock64();
    struct TAll { long long t; int lane; __device__ ~TAll() { if (lane == 0) atomicAdd(&g_bk_hist[29], (unsigned long long)(clock64() - t)); } } t_all_guard{t_all_, lane};
#endif
    auto any_of = [](const uint32_t (&a)[WPL]) {
        uint32_t r = a[0];
#pragma unroll
        for (int h = 1; h < WPL; ++h) r |= a[h];
        return r;
    };
    auto pop_level = [&](const uint32_t *lv) {
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            const int w = lane + 64 * h;
            P[h] = w < cw ? lv[w] : 0u;
            Xc[h] = w < cw ? lv[cw + w] : 0u;
            ext[h] = w < cw ? lv[2 * cw + w] : 0u;
        }
    };
    while (true) {
        if (entering) {
            ++nodes;
#ifdef GMSX_BK_STATS
            {
                int pc_ = 0;
#pragma unroll
                for (int h = 0; h < WPL; ++h) pc_ += __popc(P[h]);
                pc_ = wave_sum(pc_);
                BK_STAT(c <= 32 ? 0 : c <= 64 ? 1 : c <= 128 ? 2 : c <= 256 ? 3 : c <= 512 ? 4 : c <= 1024 ? 5 : c <= 2048 ? 6 : 7, 1);
                BK_STAT(8 + (pc_ == 0 ? 0 : pc_ == 1 ? 1 : pc_ <= 6 ? 2 : pc_ <= 16 ? 3 : pc_ <= 32 ? 4 : pc_ <= 64 ? 5 : pc_ <= 256 ? 6 : 7), 1);
                if (xf_ne) BK_STAT(16, 1);
                if (lane == 0) atomicMax(&g_bk_hist[23], (unsigned long long)depth);
            }
#endif
            if (__ballot(any_of(P) != 0) == 0) {
                if (__ballot(any_of(Xc) != 0) == 0 && !xf_ne) cnt++;
                entering = false;
                if (depth == 0) break;
                --depth;  // pop
                pop_level(stack + size_t(depth) * lvl);
                xf_ne = int(xfne_stack[depth]);
                continue;
            }
            // FAST PATH: one candidate left (the commonest node of the whole search tree).  Whatever the pivot, the node has at most the
            // branch q = that candidate, whose child has P' = {} — so the node yields one maximal clique iff no finished vertex is
            // adjacent to q (Xc ∩ N(q) = Xf ∩ N(q) = {}), and nothing otherwise (a pivot adjacent to q removes the branch; tomita.h:12-40,
            // 51-86 reaches the same two outcomes through findPivot + one more expand).  No pivot scoring, no push / pop of a level.
            {
                int lanes_nz = 0;
#pragma unroll
                for (int h = 0; h < WPL; ++h) lanes_nz += __popcll(__ballot(P[h] != 0));
                if (lanes_nz == 1) {  // wave-uniform
                    int hsel = 0;
#pragma unroll
                    for (int h = 0; h < WPL; ++h)
                        if (__ballot(P[h] != 0)) hsel = h;
                    unsigned long long nzl = 0;
                    uint32_t pw = 0;
#pragma unroll
                    for (int h = 0; h < WPL; ++h)
                        if (h == hsel) {
                            nzl = __ballot(P[h] != 0);
                            pw = __builtin_amdgcn_readlane(P[h], __ffsll((long long)nzl) - 1);
                        }
                    if ((pw & (pw - 1u)) == 0u) {
                        BK_STAT(25, 1);
                        const int q = ((__ffsll((long long)nzl) - 1 + 64 * hsel) << 5) + __ffs(pw) - 1;
                        uint32_t any = 0;
#pragma unroll
                        for (int h = 0; h < WPL; ++h) {
                            const int w = lane + 64 * h;
                            any |= w < cw ? (Xc[h] & Cadj[size_t(q) * cw + w]) : 0u;
                        }
                        if (xf_ne) {
                            BK_T0();
                            BK_STAT(26, 1); BK_STAT(27, xw);
                            const uint32_t *lvx = stack + size_t(depth) * lvl + 3 * cw;
                            const uint32_t *xt = XT + size_t(q) * xw;
                            for (int w = lane; w < xw; w += 64) any |= lvx[w] & xt[w];
                            BK_T1(28);
                        }
                        if (__ballot(any != 0) == 0) cnt++;
                        entering = false;
                        if (depth == 0) break;
                        --depth;  // pop
                        pop_level(stack + size_t(depth) * lvl);
                        xf_ne = int(xfne_stack[depth]);
                        continue;
                    }
                }
            }
            // SMALL NODES: with two or three candidates left the best pivot saves at most a branch or two, while finding it costs the
            // expansion of P ∪ Xc, a scored row fetch per member and a wave argmax.  Any pivot yields the same count (tomita.h:12-40 is a
            // heuristic): such a node takes its FIRST candidate.
            if (sh.small_p > 1) {
                int pc = 0;
#pragma unroll
                for (int h = 0; h < WPL; ++h) pc += __popc(P[h]);
                pc = wave_sum(pc);
                if (pc <= sh.small_p) {  // wave-uniform
                    int hsel = 0;
#pragma unroll
                    for (int h = WPL - 1; h >= 0; --h)
                        if (__ballot(P[h] != 0)) hsel = h;
                    uint32_t pw = 0;
                    int L0 = 0;
#pragma unroll
                    for (int h = 0; h < WPL; ++h)
                        if (h == hsel) {
                            L0 = __ffsll((long long)__ballot(P[h] != 0)) - 1;
                            pw = __builtin_amdgcn_readlane(P[h], L0);
                        }
                    const int best0 = ((L0 + 64 * hsel) << 5) + __ffs(pw) - 1;
#pragma unroll
                    for (int h = 0; h < WPL; ++h) {
                        const int w = lane + 64 * h;
                        const uint32_t prow = w < cw ? Cadj[size_t(best0) * cw + w] : 0u;
                        ext[h] = P[h] & ~prow;
                    }
                    entering = false;
                }
            }
            if (entering) {
            // pivot: argmax over u in P ∪ Xc of |P ∩ N(u)|.  One LANE per candidate: P is parked in LDS, the members of
            // P ∪ Xc are expanded into an LDS list (wave prefix sum of the per-word popcounts), then every lane scores its
            // own candidates with independent row loads (64 rows in flight instead of one dependent load per candidate).
            BK_T0();
            uint32_t U[WPL];
            int mine = 0;
#pragma unroll
            for (int h = 0; h < WPL; ++h) {
                U[h] = P[h] | Xc[h];
                mine += __popc(U[h]);
                if (lane + 64 * h < cw) piv_P[lane + 64 * h] = P[h];
            }
            int pre = mine;
            for (int sft = 1; sft < 64; sft <<= 1) {
                const int o = __shfl_up(pre, sft);
                if (lane >= sft) pre += o;
            }
            const int ncand = __builtin_amdgcn_readlane(pre, 63);
            BK_STAT(21, 1);
            BK_STAT(22, ncand);
            {
                int at = pre - mine;
#pragma unroll
                for (int h = 0; h < WPL; ++h) {
                    uint32_t bits = U[h];
                    while (bits) {
                        piv_list[at++] = (unsigned short)(((lane + 64 * h) << 5) + __ffs(bits) - 1);
                        bits &= bits - 1;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            int best = 0x7fffffff, best_score = -1;
            for (int k = lane; k < ncand; k += 64) {
                const int u = int(piv_list[k]);
                const uint32_t *row = Cadj + size_t(u) * cw;
                int sc = 0;
                for (int w = 0; w < cw; ++w) sc += __popc(piv_P[w] & row[w]);
                if (sc > best_score || (sc == best_score && u < best)) {
                    best_score = sc;
                    best = u;
                }
            }
            for (int sft = 32; sft > 0; sft >>= 1) {  // wave argmax (ties -> smallest index, so every lane agrees)
                const int os = __shfl_xor(best_score, sft), ob = __shfl_xor(best, sft);
                if (os > best_score || (os == best_score && ob < best)) {
                    best_score = os;
                    best = ob;
                }
            }
            __builtin_amdgcn_wave_barrier();
            best = uni32(best);  // every lane holds the same winner
#pragma unroll
            for (int h = 0; h < WPL; ++h) {
                const int w = lane + 64 * h;
                const uint32_t prow = w < cw ? Cadj[size_t(best) * cw + w] : 0u;
                ext[h] = P[h] & ~prow;
            }
            entering = false;
            BK_T1(30);
            }
        }
        // next branch vertex q of this node
        unsigned long long nzh[WPL];
        unsigned long long nz_any = 0;
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            nzh[h] = __ballot(ext[h] != 0);
            nz_any |= nzh[h];
        }
        if (!nz_any) {
            if (depth == 0) break;
            --depth;  // pop
            pop_level(stack + size_t(depth) * lvl);
            xf_ne = int(xfne_stack[depth]);
            continue;
        }
        if (nodes >= budget) {
            // ---- split: hand every level that still has branches to the next round --------------------------------
            bool ok = true;
            if (arena_off == kNoArena) {  // first split of this start vertex: persist Cadj | XT
                const unsigned long long need = ((unsigned long long)c * cw + (unsigned long long)c * xw + 3ull) & ~3ull;
                unsigned long long off0 = 0;
                if (lane == 0) off0 = atomicAdd(sh.arena_head, need);
                off0 = uni64(off0);
                if (off0 + need > sh.arena_cap) ok = false;
                else {
                    for (unsigned long long i = lane; i < (unsigned long long)c * cw + (unsigned long long)c * xw; i += 64)
                        sh.arena[off0 + i] = Cadj[i];  // XT follows Cadj in the slab
                    arena_off = off0;
                }
            }
            int nrec = 0;
            if (ok) {
                // levels 0..depth-1 are in the stack, level `depth` is in registers: spill it so all look alike
                uint32_t *cur = stack + size_t(depth) * lvl;
#pragma unroll
                for (int h = 0; h < WPL; ++h) {
                    const int w = lane + 64 * h;
                    if (w < cw) {
                        cur[w] = P[h];
                        cur[cw + w] = Xc[h];
                        cur[2 * cw + w] = ext[h];
                    }
                }
                if (lane == 0) xfne_stack[depth] = (unsigned char)xf_ne;
                if (global_structs) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (readers: this wave; an agent-scope fence also flushes the XCD's L2)
                __builtin_amdgcn_wave_barrier();
                // A level with b pending branch vertices becomes min(b, kBkSplit) records: the branches in ascending order, cut into
                // equal runs; run j starts from the state its predecessors leave behind (their vertices moved from P to X —
                // tomita.h:68-70 — which needs no search, only the bitmaps).  One record per level made the siblings of a wide node
                // run one after the other, a budget at a time: the config-4 graph needed 118 rounds whatever the budget.
                auto level_bits = [&](int l) {
                    int b = 0;
                    for (int w = lane; w < cw; w += 64) b += __popc(stack[size_t(l) * lvl + 2 * cw + w]);
                    for (int sft = 32; sft > 0; sft >>= 1) b += __shfl_xor(b, sft);
                    return b;
                };
                for (int l = 0; l <= depth; ++l) nrec += min(level_bits(l), kBkSplit);
                const unsigned long long rec_words = (unsigned long long)(kRecHeader + 3 * cw + xw);
                unsigned long long p0 = 0, d0 = sh.dir_cap;
                if (lane == 0) {
                    p0 = atomicAdd(sh.pool_head, rec_words * nrec);
                    if (p0 + rec_words * nrec <= sh.pool_cap) d0 = atomicAdd(sh.dir_count, (unsigned long long)nrec);
                }
                p0 = uni64(p0);
                d0 = uni64(d0);
                if (p0 + rec_words * nrec > sh.pool_cap || d0 + nrec > sh.dir_cap) {
                    ok = false;  // directory slots that were claimed but not written keep their ~0 fill and are skipped
                } else {
                    auto lowest = [](uint32_t x, int k) -> uint32_t {  // the k lowest set bits of x
                        if (k <= 0) return 0u;
                        if (k >= __popc(x)) return x;
                        uint32_t r = 0;
                        while (k--) {
                            const uint32_t b = x & (0u - x);
                            r |= b;
                            x ^= b;
                        }
                        return r;
                    };
                    int r = 0;
                    for (int l = 0; l <= depth; ++l) {
                        const uint32_t *lv = stack + size_t(l) * lvl;
                        const int nb = level_bits(l);
                        if (nb == 0) continue;
                        const int parts = min(nb, kBkSplit);
                        for (int j = 0; j < parts; ++j) {
                            const int a = int((long long)nb * j / parts), b = int((long long)nb * (j + 1) / parts);  // ranks [a, b) of the pending branches
                            uint32_t *rec = sh.pool + p0 + rec_words * r;
                            if (lane == 0) {
                                rec[0] = uint32_t(v);
                                rec[1] = uint32_t(c);
                                rec[2] = uint32_t(x);
                                rec[3] = uint32_t(xfne_stack[l]);
                                rec[4] = uint32_t(arena_off & 0xffffffffull);
                                rec[5] = uint32_t(arena_off >> 32);
                                rec[6] = rec[7] = 0;
                                sh.dir[d0 + r] = p0 + rec_words * r;
                            }
                            int base = 0;  // pending branches in the words before this block of 64
                            for (int w0 = 0; w0 < cw; w0 += 64) {
                                const int w = w0 + lane;
                                const uint32_t e = w < cw ? lv[2 * cw + w] : 0u;
                                const int mine = __popc(e);
                                int pre = mine;
                                for (int sft = 1; sft < 64; sft <<= 1) {
                                    const int o = __shfl_up(pre, sft);
                                    if (lane >= sft) pre += o;
                                }
                                const int rank0 = base + pre - mine;  // rank of this word's first pending branch
                                base += __builtin_amdgcn_readlane(pre, 63);
                                if (w < cw) {
                                    const uint32_t before = lowest(e, a - rank0);        // branches of the runs in front of this one
                                    const uint32_t run = lowest(e, b - rank0) & ~before;  // this run
                                    rec[kRecHeader + w] = lv[w] & ~before;                // P
                                    rec[kRecHeader + cw + w] = lv[cw + w] | before;       // Xc
                                    rec[kRecHeader + 2 * cw + w] = run;                   // ext
                                }
                            }
                            for (int w = lane; w < xw; w += 64) rec[kRecHeader + 3 * cw + w] = lv[3 * cw + w];
                            ++r;
                        }
                    }
                    if (lane == 0) {
                        atomicMax(sh.max_stack, (unsigned long long)(c + 1) * (unsigned long long)lvl);
                        if (c <= 512) atomicMax(sh.max_stack + 1 + bk_group_class(c), bk_group_slab_words(c, xw));  // k_bk_resume4's need
                    }
                }
            }
#ifdef GMSX_BK_STATS
            if (ok && lane == 0) atomicAdd(&g_bk_nodes, (unsigned long long)nodes);
#endif
            if (ok) {                // the rest of this search belongs to the next round
                node_words += (unsigned long long)nodes * (unsigned long long)cw;
                return;
            }
            budget = 0xffffffffu;    // no room: finish it here
        }
        int hsel = 0;
#pragma unroll
        for (int h = WPL - 1; h >= 0; --h)
            if (nzh[h]) hsel = h;  // wave-uniform: the first word group with a branch vertex
        const int L = __ffsll((long long)nzh[hsel]) - 1;
        uint32_t word = 0;
#pragma unroll
        for (int h = 0; h < WPL; ++h)
            if (h == hsel) word = __builtin_amdgcn_readlane(ext[h], L);
        const int bit = __ffs(word) - 1;
        const int q = ((L + 64 * hsel) << 5) + bit;
        BK_STAT(17, 1);
        uint32_t Pn[WPL], Xcn[WPL];
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            const int w = lane + 64 * h;
            const uint32_t qrow = w < cw ? Cadj[size_t(q) * cw + w] : 0u;
            Pn[h] = P[h] & qrow;
            Xcn[h] = Xc[h] & qrow;
        }
        uint32_t *lv = stack + size_t(depth) * lvl;
        if (__ballot(any_of(Pn) != 0) == 0) {
            // FAST PATH: the child has no candidate — it is a leaf: one maximal clique iff no finished vertex is adjacent to all of R ∪ {q}.
            // Decided right here (no level pushed, entered and popped again); the node goes on with q moved from cand to fini.
            uint32_t any = any_of(Xcn);
            if (xf_ne && __ballot(any != 0) == 0) {
                BK_T0();
                BK_STAT(26, 1); BK_STAT(27, xw);
                const uint32_t *xt = XT + size_t(q) * xw;
                for (int w = lane; w < xw; w += 64) any |= lv[3 * cw + w] & xt[w];
                BK_T1(28);
            }
            if (__ballot(any != 0) == 0) cnt++;
            ++nodes;
            BK_STAT(18, 1);
#pragma unroll
            for (int h = 0; h < WPL; ++h)
                if (h == hsel && lane == L) {
                    ext[h] &= ~(1u << bit);
                    P[h] &= ~(1u << bit);
                    Xc[h] |= 1u << bit;
                }
            continue;
        }
        {
            // FAST PATH: the child has ONE candidate q2 — the child is decided like a one-candidate node above, with its Xf = Xf ∩ N(q) formed
            // on the fly: one maximal clique iff nothing finished is adjacent to q2 as well.  No level pushed.
            int lanes_nz = 0;
#pragma unroll
            for (int h = 0; h < WPL; ++h) lanes_nz += __popcll(__ballot(Pn[h] != 0));
            if (lanes_nz == 1) {
                int h2 = 0;
#pragma unroll
                for (int h = 0; h < WPL; ++h)
                    if (__ballot(Pn[h] != 0)) h2 = h;
                unsigned long long nzl = 0;
                uint32_t pw = 0;
#pragma unroll
                for (int h = 0; h < WPL; ++h)
                    if (h == h2) {
                        nzl = __ballot(Pn[h] != 0);
                        pw = __builtin_amdgcn_readlane(Pn[h], __ffsll((long long)nzl) - 1);
                    }
                if ((pw & (pw - 1u)) == 0u) {
                    const int q2 = ((__ffsll((long long)nzl) - 1 + 64 * h2) << 5) + __ffs(pw) - 1;
                    uint32_t any = 0;
#pragma unroll
                    for (int h = 0; h < WPL; ++h) {
                        const int w = lane + 64 * h;
                        any |= w < cw ? (Xcn[h] & Cadj[size_t(q2) * cw + w]) : 0u;
                    }
                    if (xf_ne && __ballot(any != 0) == 0) {
                        BK_T0();
                        BK_STAT(26, 1); BK_STAT(27, xw);
                        const uint32_t *xt = XT + size_t(q) * xw, *xt2 = XT + size_t(q2) * xw;
                        for (int w = lane; w < xw; w += 64) any |= lv[3 * cw + w] & xt[w] & xt2[w];
                        BK_T1(28);
                    }
                    if (__ballot(any != 0) == 0) cnt++;
                    nodes += 2;
                    BK_STAT(19, 1);
#pragma unroll
                    for (int h = 0; h < WPL; ++h)
                        if (h == hsel && lane == L) {
                            ext[h] &= ~(1u << bit);
                            P[h] &= ~(1u << bit);
                            Xc[h] |= 1u << bit;
                        }
                    continue;
                }
            }
        }
        uint32_t *nx = lv + lvl;
        int child_ne = 0;
        BK_STAT(20, 1);
        if (xf_ne) {
            BK_STAT(24, xw);
            BK_T0();
            uint32_t any = 0;
            const uint32_t *xt = XT + size_t(q) * xw;
#ifdef GMSX_BK_STATS
            int nzw_ = 0;
#endif
            for (int w = lane; w < xw; w += 64) {
                const uint32_t t = lv[3 * cw + w] & xt[w];
                nx[3 * cw + w] = t;
                any |= t;
#ifdef GMSX_BK_STATS
                nzw_ += t != 0;
#endif
            }
            child_ne = __ballot(any != 0) != 0 ? 1 : 0;
            BK_T1(28);
#ifdef GMSX_BK_STATS
            nzw_ = wave_sum(nzw_);
            BK_STAT(31, nzw_);
#endif
        }
        // this node continues with q moved from cand to fini (tomita.h:68-70)
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            if (h == hsel && lane == L) {
                ext[h] &= ~(1u << bit);
                P[h] &= ~(1u << bit);
                Xc[h] |= 1u << bit;
            }
            const int w = lane + 64 * h;
            if (w < cw) {
                lv[w] = P[h];
                lv[cw + w] = Xc[h];
                lv[2 * cw + w] = ext[h];
            }
        }
        if (lane == 0) xfne_stack[depth] = (unsigned char)xf_ne;
        __builtin_amdgcn_wave_barrier();
        ++depth;
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            P[h] = Pn[h];
            Xc[h] = Xcn[h];
        }
        xf_ne = child_ne;
        entering = true;
    }
#ifdef GMSX_BK_STATS
    if (lane == 0) atomicAdd(&g_bk_nodes, (unsigned long long)nodes);
#endif
    node_words += (unsigned long long)nodes * (unsigned long long)cw;
}

// Memory-resident Tomita search for start vertices with more candidates than the register-resident kernels hold (c > 16384, or
// whatever GMSX_BK_MAXC says): the same recursion with P / Xc / ext of EVERY level in the slab (level l at stack + l*lvl: P, Xc,
// ext, Xf), the pivot-candidate list and the per-level Xf flags behind the stack, every set operation a lane-strided loop over the
// cw words.  No node budget, no re-split: one wave finishes its start vertex.  An order of magnitude slower per node than the
// register kernels; it exists so that no graph is refused for the width of a neighbourhood.
__device__ __forceinline__ void bk_search_mem(const uint32_t *Cadj, const uint32_t *XT, uint32_t *stack, int c, int x, int xf_ne, int lane,
                                              unsigned long long &cnt) {
    const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
    const int lvl = 3 * cw + xw + 1;
    uint32_t *piv_list = stack + size_t(c + 1) * lvl;                                  // c entries
    unsigned char *xfne = reinterpret_cast<unsigned char *>(piv_list + c);              // c + 1 entries
    auto wave_or = [&](const uint32_t *a, int nw) {
        uint32_t r = 0;
        for (int w = lane; w < nw; w += 64) r |= a[w];
        return __ballot(r != 0) != 0;
    };
    int depth = 0;
    bool entering = true;
    while (true) {
        uint32_t *lv = stack + size_t(depth) * lvl;
        uint32_t *P = lv, *Xc = lv + cw, *ext = lv + 2 * cw;
        if (entering) {
            if (!wave_or(P, cw)) {
                if (!wave_or(Xc, cw) && !xf_ne) cnt++;
                if (depth == 0) break;
                --depth;
                xf_ne = int(xfne[depth]);
                entering = false;
                continue;
            }
            // pivot: argmax over u in P ∪ Xc of |P ∩ N(u)| (ties -> smallest index); candidates listed by a wave prefix sum per 64-word chunk
            int ncand = 0;
            for (int w0 = 0; w0 < cw; w0 += 64) {
                const int w = w0 + lane;
                uint32_t bits = w < cw ? (P[w] | Xc[w]) : 0u;
                const int mine = __popc(bits);
                int pre = mine;
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const int o = __shfl_up(pre, sft);
                    if (lane >= sft) pre += o;
                }
                int at = ncand + pre - mine;
                while (bits) {
                    piv_list[at++] = uint32_t((w << 5) + __ffs(bits) - 1);
                    bits &= bits - 1;
                }
                ncand += __builtin_amdgcn_readlane(pre, 63);
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            int best = 0x7fffffff, best_score = -1;
            for (int k = lane; k < ncand; k += 64) {
                const int u = int(piv_list[k]);
                const uint32_t *row = Cadj + size_t(u) * cw;
                int sc = 0;
                for (int w = 0; w < cw; ++w) sc += __popc(P[w] & row[w]);
                if (sc > best_score || (sc == best_score && u < best)) {
                    best_score = sc;
                    best = u;
                }
            }
            for (int sft = 32; sft > 0; sft >>= 1) {
                const int os = __shfl_xor(best_score, sft), ob = __shfl_xor(best, sft);
                if (os > best_score || (os == best_score && ob < best)) {
                    best_score = os;
                    best = ob;
                }
            }
            best = uni32(best);
            const uint32_t *prow = Cadj + size_t(best) * cw;
            for (int w = lane; w < cw; w += 64) ext[w] = P[w] & ~prow[w];
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            entering = false;
        }
        // next branch vertex q of this node: the lowest set bit of ext
        int q = -1;
        for (int w0 = 0; w0 < cw && q < 0; w0 += 64) {
            const int w = w0 + lane;
            const uint32_t e = w < cw ? ext[w] : 0u;
            const unsigned long long nz = __ballot(e != 0);
            if (nz) {
                const int L = __ffsll((long long)nz) - 1;
                const uint32_t word = __builtin_amdgcn_readlane(e, L);
                q = ((w0 + L) << 5) + __ffs(word) - 1;
            }
        }
        if (q < 0) {
            if (depth == 0) break;
            --depth;
            xf_ne = int(xfne[depth]);
            continue;
        }
        uint32_t *nx = lv + lvl;
        const uint32_t *qrow = Cadj + size_t(q) * cw;
        for (int w = lane; w < cw; w += 64) {
            nx[w] = P[w] & qrow[w];
            nx[cw + w] = Xc[w] & qrow[w];
        }
        int child_ne = 0;
        if (xf_ne) {
            uint32_t any = 0;
            const uint32_t *xt = XT + size_t(q) * xw;
            for (int w = lane; w < xw; w += 64) {
                const uint32_t t = lv[3 * cw + w] & xt[w];
                nx[3 * cw + w] = t;
                any |= t;
            }
            child_ne = __ballot(any != 0) != 0 ? 1 : 0;
        }
        // this node continues with q moved from cand to fini (tomita.h:68-70)
        if (lane == 0) {
            const uint32_t bit = 1u << (q & 31);
            ext[q >> 5] &= ~bit;
            P[q >> 5] &= ~bit;
            Xc[q >> 5] |= bit;
            xfne[depth] = (unsigned char)xf_ne;
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        ++depth;
        xf_ne = child_ne;
        entering = true;
    }
}

// BUILD of one start vertex (one wave): the id -> index map of its candidates, Cadj (symmetric closure of the DAG rows inside C) and
// XT (C x X0 adjacency) — into LDS (tiny tasks), a per-wave slab (k_bk_wave) or straight into the arena (k_bk_build).
template <bool LDS_SLAB>
__device__ __forceinline__ void bk_build(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, const int32_t *__restrict__ newid,
                                         const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff,
                                         const int32_t *__restrict__ tadj, const BkShared &sh, int32_t v, int c, int x, int64_t ob, int64_t oe,
                                         unsigned long long *map, uint32_t msize, uint32_t *Cadj, uint32_t *XT, int32_t *in_stage, int lane) {
    const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
    const uint32_t mmask = msize - 1;
    (void)off;
            // ---- build: map, Cadj (symmetric closure of the DAG rows inside C), XT (C x X0 adjacency) -------------------
            for (uint32_t i = lane; i < msize; i += 64) map[i] = kEmptySlot;
            {
                const size_t nz = size_t(c) * cw + size_t(c) * xw;
                size_t head = (16 - (reinterpret_cast<uintptr_t>(Cadj) & 15)) & 15;  // bytes to the next 16-byte boundary
                head = min(nz, head / 4);
                for (size_t i = lane; i < head; i += 64) Cadj[i] = 0;
                uint4 *z4 = reinterpret_cast<uint4 *>(Cadj + head);
                const size_t n4 = (nz - head) / 4;
                for (size_t i = lane; i < n4; i += 64) z4[i] = make_uint4(0u, 0u, 0u, 0u);
                for (size_t i = head + n4 * 4 + lane; i < nz; i += 64) Cadj[i] = 0;
            }
            if (!LDS_SLAB) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int64_t hb = uni64(hoff[v]), tb = uni64(toff[v]);
            int hc = int(uni64(hoff[v + 1]) - hb);
            if (hc > 0 && uni32(uint32_t(hadj[hb + hc - 1])) == 0xFFFFu) --hc;
            for (int i = lane; i < c; i += 64) {
                const int32_t a = i < hc ? int32_t(hadj[hb + i]) : tadj[tb + (i - hc)];
                uint32_t h = bk_hash(a, mmask);
                const unsigned long long packed = ((unsigned long long)uint32_t(a) << 32) | (unsigned long long)uint32_t(i);
                while (atomicCAS(&map[h], kEmptySlot, packed) != kEmptySlot) h = (h + 1) & mmask;
            }
            if (!LDS_SLAB) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // rows of the hub candidates (all of them have a bitset container): lane k asks "is candidate k in N+(a_i)?" with
            // one word gather; the candidate ids of a 64-chunk are loaded once, the rows go eight at a time (eight gathers
            // in flight).  Candidates ascend with their index, so only rows i > k can hit.
            {
                for (int k0 = 0; k0 < hc; k0 += 64) {
                    const int k = k0 + lane;
                    const uint32_t w = k < hc ? uint32_t(hadj[hb + k]) : 0xFFFFFFFFu;
                    for (int i0 = k0; i0 < hc; i0 += 64) {  // 64 rows: ids and bitset offsets lane-parallel, then by readlane
                        int32_t ai = 0;
                        int64_t rbi = 0;
                        if (i0 + lane < hc) {
                            ai = int32_t(hadj[hb + i0 + lane]);
                            rbi = sh.bmoff[ai];
                        }
                        const int nrow = min(64, hc - i0);
                        for (int r0 = 0; r0 < nrow; r0 += 8) {
                            uint32_t wd[8];
    #pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const int32_t a = __builtin_amdgcn_readlane(ai, (r0 + r) & 63);  // 0 beyond the last row: never > w
                                const int64_t rb = bk_readlane64(rbi, (r0 + r) & 63);
                                wd[r] = (w < uint32_t(a)) ? sh.bmpool[rb + (w >> 5)] : 0u;
                            }
    #pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                if ((wd[r] >> (w & 31u)) & 1u) {
                                    const int i = i0 + r0 + r;
                                    atomicOr(&Cadj[size_t(i) * cw + (k >> 5)], 1u << (k & 31));
                                    atomicOr(&Cadj[size_t(k) * cw + (i >> 5)], 1u << (i & 31));
                                }
                            }
                        }
                    }
                }
                // tail candidates: stream their containers through the map, four rows per trip (one per 16-lane group)
#if defined(GMSX_BK_AB) && GMSX_BK_AB == 5  // A/B builds (wrong counts): 4 = no in-neighbour rows, 5 = no tail-candidate rows, 6 = neither
                if (c < 0)
#elif defined(GMSX_BK_AB) && GMSX_BK_AB == 6
                if (c < 0)
#endif
                for (int i0 = hc; i0 < c; i0 += 4) {
                    const int i = i0 + (lane >> 4);
                    if (i < c) {
                        const int32_t a = tadj[tb + (i - hc)];
                        bk_scan_row_group(hoff, hadj, toff, tadj, a, map, mmask, lane & 15, [&](int k) {
                            atomicOr(&Cadj[size_t(i) * cw + (k >> 5)], 1u << (k & 31));
                            atomicOr(&Cadj[size_t(k) * cw + (i >> 5)], 1u << (i & 31));
                        });
                    }
                }
            }
            // rows of the in-neighbours, 64 CSR entries per batch; t = index of the in-neighbour in X0
            int xbase = 0;
#if defined(GMSX_BK_AB) && (GMSX_BK_AB == 4 || GMSX_BK_AB == 6)
            if (c < 0)
#endif
            for (int64_t e0 = ob; e0 < oe; e0 += 64) {
                const int64_t e = e0 + lane;
                int32_t nw = -1;
                bool keep = false;
                if (e < oe) {
                    nw = newid[adj[e]];
                    keep = nw > v;
                }
                const unsigned long long m = __ballot(keep);
                const int kept = __popcll(m);
                // compact the kept in-neighbours of this batch, then eight rows per step, one per 8-lane group
                __builtin_amdgcn_wave_barrier();
                if (keep) in_stage[__popcll(m & ((1ull << lane) - 1ull))] = nw;
                __builtin_amdgcn_wave_barrier();
                for (int r0 = 0; r0 < kept; r0 += 8) {
                    const int r = r0 + (lane >> 3);
                    if (r < kept) {
                        const int t = xbase + r;
                        bk_scan_row_group8(hoff, hadj, toff, tadj, in_stage[r], map, mmask, lane & 7,
                                           [&](int k) { atomicOr(&XT[size_t(k) * xw + (t >> 5)], 1u << (t & 31)); });
                    }
                }
                xbase += kept;
            }
            if (!LDS_SLAB) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop L1 lines of the slab cached for an earlier task
            }
            __builtin_amdgcn_wave_barrier();

}

// Round 0: one wave per start vertex.  LDS_SLAB: every structure of the search lives in this wave's LDS slab (tasks of
// at most kLdsSlabWords words); otherwise in slabs[block * slab_words].
#ifndef GMSX_BK_GRAB
#define GMSX_BK_GRAB 8
#endif
template <bool LDS_SLAB, int WPL>
__global__ __launch_bounds__(64) void k_bk_wave(const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                const int32_t *__restrict__ newid, const int32_t *__restrict__ oldid,
                                                const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                const int32_t *__restrict__ dplus, const int32_t *__restrict__ task_v,
                                                int64_t first, int64_t end, int nparts, int part,
                                                unsigned long long *__restrict__ queue, uint32_t *__restrict__ slabs,
                                                unsigned long long slab_words, unsigned long long *__restrict__ acc, BkShared sh,
                                                // LDS-slab tasks only (round 5, GMSX_BK_TINY_ROOTS): not searched here — Cadj | XT go to the arena and a ROOT
                                                // RECORD to the pool, at offsets the layout scans computed per task, and k_bk_resume4 searches them four to a
                                                // wave (a search of at most ~90 candidates keeps three of this kernel's 64 lanes busy)
                                                const int64_t *__restrict__ emit_aoff = nullptr, const int64_t *__restrict__ emit_roff = nullptr,
                                                unsigned long long emit_abase = 0, unsigned long long emit_rbase = 0, unsigned long long emit_dbase = 0) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_slab[LDS_SLAB ? kLdsSlabWords : 4];
    // per level: is Xf non-empty (written by one lane, read by all); an LDS-slab task has < 256 candidates (its slab would not fit otherwise)
    constexpr int WR = WPL > 0 ? WPL : 1;  // WPL = 0: the memory-resident search keeps nothing of the search in LDS
    __shared__ unsigned char xfne_stack[LDS_SLAB ? 256 : 2052 * WR];
    __shared__ int32_t in_stage[64];             // build: the kept in-neighbours of one 64-entry batch, compacted
    // global-slab variant: the id -> index map of the build phase lives in LDS whenever it fits (c <= 512); the probes
    // of the in-neighbour rows are the long dependent chains of the build
    constexpr uint32_t kLdsMapSlots = 1024;
    // one 8.25 KB LDS work area: the build's id -> index map (global-slab variant), then the search's pivot scratch
    static_assert(!LDS_SLAB || WPL == 1, "LDS-slab tasks are tiny");
    constexpr uint32_t kWorkWords = LDS_SLAB ? 128 : (2 * kLdsMapSlots > 1024u * WR ? 2 * kLdsMapSlots : 1024u * WR);  // map area, or piv_list: 2048*WPL u16 (256 u16 for an LDS-slab task)
    __shared__ __attribute__((aligned(16))) uint32_t lds_work[kWorkWords + 64 * WR];
    unsigned long long *lds_map = reinterpret_cast<unsigned long long *>(lds_work);
    uint32_t *piv_P = lds_work + kWorkWords;
    unsigned short *piv_list = reinterpret_cast<unsigned short *>(lds_work);
    const int lane = threadIdx.x;
    uint32_t *slab = LDS_SLAB ? lds_slab : slabs + size_t(blockIdx.x) * slab_words;
    unsigned long long cnt = 0, node_words = 0;
    // the LDS tasks take kGrab consecutive start vertices per queue ticket: 1.65 M tickets on ONE address (a device-scope atomic is resolved
    // behind the L2s of the eight XCDs) were a serial resource of the kernel
    constexpr int64_t kGrab = LDS_SLAB ? GMSX_BK_GRAB : 1;
    int64_t q_next = 0, q_end = 0;
    while (true) {
        if (q_next == q_end) {
            unsigned long long q0 = 0;
            if (lane == 0) q0 = atomicAdd(queue, (unsigned long long)kGrab);
            q_next = int64_t(uni64(q0));
            q_end = q_next + kGrab;
        }
        const int64_t qi = q_next++;
        const int64_t pos = first + qi * nparts + part;
        if (pos >= end) break;
        const int32_t v = uni32(task_v[pos]);
        const int32_t vo = uni32(oldid[v]);
        const int c = uni32(dplus[v]);
        const int64_t ob = uni64(off[vo]), oe = uni64(off[vo + 1]);
        const int x = int(oe - ob) - c;
        const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
        const uint32_t msize = bk_map_size(c);
        unsigned long long *map = (!LDS_SLAB && msize <= kLdsMapSlots) ? lds_map : reinterpret_cast<unsigned long long *>(slab);
        uint32_t *Cadj = slab + 2 * size_t(msize);
        uint32_t *XT = Cadj + size_t(c) * cw;
        uint32_t *stack = XT + size_t(c) * xw;

        bk_build<LDS_SLAB>(off, adj, newid, hoff, hadj, toff, tadj, sh, v, c, x, ob, oe, map, msize, Cadj, XT, in_stage, lane);

        // ---- search from the root: P = C, Xc = {}, Xf = X0 ------------------------------------------------------
        for (int w = lane; w < xw; w += 64) {
            const int bits = x - w * 32;
            stack[3 * cw + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
        }
        if constexpr (WPL == 0) {
            for (int w = lane; w < cw; w += 64) {
                const int bits = c - w * 32;
                stack[w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
                stack[cw + w] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            bk_search_mem(Cadj, XT, stack, c, x, x > 0 ? 1 : 0, lane, cnt);
        } else {
            uint32_t P[WR], Xc[WR], ext[WR];
#pragma unroll
            for (int h = 0; h < WR; ++h) {
                const int w = lane + 64 * h;
                P[h] = Xc[h] = ext[h] = 0u;
                if (w < cw) {
                    const int bits = c - w * 32;
                    P[h] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
#if defined(GMSX_BK_AB) && GMSX_BK_AB >= 3  // A/B build (wrong counts): the LDS tasks' build without their search
            if (LDS_SLAB) { cnt += Cadj[0] & 1u; continue; }
#endif
            if (LDS_SLAB && emit_aoff) {
                const unsigned long long a0 = emit_abase + (unsigned long long)emit_aoff[qi], r0 = emit_rbase + (unsigned long long)emit_roff[qi];
                const int words = c * cw + c * xw;  // Cadj, then XT: contiguous in the slab
                uint32_t *dst = sh.arena + a0;
                for (int i = lane; i < words; i += 64) dst[i] = Cadj[i];
                uint32_t *rec = sh.pool + r0;
                if (lane == 0) {
                    rec[0] = uint32_t(v);
                    rec[1] = uint32_t(c);
                    rec[2] = uint32_t(x);
                    rec[3] = x > 0 ? 1u : 0u;
                    rec[4] = uint32_t(a0 & 0xffffffffull);
                    rec[5] = uint32_t(a0 >> 32);
                    rec[6] = 1u;  // root: enter the node
                    rec[7] = 0u;
                    sh.dir[emit_dbase + (unsigned long long)qi] = r0;
                }
                for (int w = lane; w < cw; w += 64) {
                    const int bits = c - w * 32;
                    rec[kRecHeader + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);  // P = every candidate
                    rec[kRecHeader + cw + w] = 0u;                                          // Xc
                    rec[kRecHeader + 2 * cw + w] = 0u;                                      // ext (set by the pivot step)
                }
                for (int w = lane; w < xw; w += 64) {
                    const int bits = x - w * 32;
                    rec[kRecHeader + 3 * cw + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);  // Xf = every in-neighbour
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            bk_search<WR>(Cadj, XT, stack, xfne_stack, v, c, x, P, Xc, ext, x > 0 ? 1 : 0, true, lane, cnt, sh, kNoArena, !LDS_SLAB, piv_P, piv_list, node_words);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && cnt) atomicAdd(&acc[(blockIdx.x & 63) * 16], cnt);
    if (lane == 0 && node_words) atomicAdd(&acc[(blockIdx.x & 63) * 16 + 15], node_words);
}

// ---- round 0 of the start vertices too big for an LDS slab: BUILD and SEARCH are separate kernels -------------------------------------
// k_bk_wave builds Cadj | XT of a start vertex and searches it in the same wave: ~110 VGPRs and 11 KB of LDS, 3.5 waves per SIMD — for a
// build that is bound by the latency of dependent row fetches (27 % VALU issue, 64 % of the wave cycles waiting, round 2).  The build
// alone needs neither the search's registers nor its stack, so it runs as its own kernel at twice the occupancy and writes Cadj | XT
// STRAIGHT INTO THE ARENA (offsets from a prefix sum over the chunk: no slab, no copy when the search is split later) plus a ROOT RECORD
// per start vertex (P = all candidates, Xc = {}, Xf = X0, flag "entering"); the search of the roots is then the ordinary resume kernel.
// k_bk_layout: per task of a chunk the arena words, the record words, the pieces of the workgroup build, and the maxima the launch needs.
#ifndef GMSX_BK_PIECE_JOBS
#define GMSX_BK_PIECE_JOBS 2048
#endif
static constexpr int kBkPieceJobs = GMSX_BK_PIECE_JOBS;
#ifndef GMSX_BK_BLOCK_GRAB
#define GMSX_BK_BLOCK_GRAB 4
#endif
static constexpr int kBkBlockGrab = GMSX_BK_BLOCK_GRAB;  // pieces per queue ticket  // row jobs (candidate rows + CSR positions) of one k_bk_block work item: 128 trips of 16 rows
__global__ void k_bk_layout(int64_t lo, int64_t cnt, int nparts, int part, const int32_t *__restrict__ task_v, const int64_t *__restrict__ off,
                            const int32_t *__restrict__ oldid, const int32_t *__restrict__ dplus, int x_is_degree, int64_t *__restrict__ need_a,
                            int64_t *__restrict__ need_r, int64_t *__restrict__ need_p,
                            unsigned long long *__restrict__ maxima /* [0] (c+1)*lvl  [1] global map words  [2..4] slab words of a 16- / 8- / 4-lane search */) {
    const int64_t qi = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (qi > cnt) return;
    if (qi == cnt) { need_a[qi] = 0; need_r[qi] = 0; need_p[qi] = 0; return; }
    const int32_t v = task_v[lo + qi * nparts + part];
    const int32_t vo = oldid[v];
    const long long c = dplus[v], x = (off[vo + 1] - off[vo]) - (x_is_degree ? 0 : c);  // k_bk_block: one XT column per CSR position of v's row
    const long long cw = (c + 31) >> 5, xw = (x + 31) >> 5;
    need_a[qi] = (c * cw + c * xw + 3) & ~3ll;
    need_r[qi] = kRecHeader + 3 * cw + xw;
    need_p[qi] = (c + (off[vo + 1] - off[vo]) + kBkPieceJobs - 1) / kBkPieceJobs;  // k_bk_block: pieces of kBkPieceJobs row jobs (>= 1: c > 0)
    atomicMax(&maxima[0], (unsigned long long)((c + 1) * (3 * cw + xw + 1)));
    if (c <= 512) atomicMax(&maxima[2 + bk_group_class(c)], bk_group_slab_words(c, xw));  // k_bk_resume4's need
    const unsigned long long msize = bk_map_size(int(c));
    if (msize > 1024) atomicMax(&maxima[1], 2ull * msize);
}
// first index i in [0, cnt] whose prefix exceeds either budget (prefixes relative to index `start`)
__global__ void k_bk_chunk_end(int64_t start, int64_t cnt, const int64_t *__restrict__ aoff, const int64_t *__restrict__ roff, int64_t a_cap, int64_t r_cap,
                               int64_t max_tasks, int64_t *__restrict__ out) {
    int64_t lo = start + 1, hi = min(cnt, start + max_tasks);  // at least one task per chunk
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (aoff[mid] - aoff[start] <= a_cap && roff[mid] - roff[start] <= r_cap) lo = mid; else hi = mid - 1;
    }
    out[0] = lo;
}
// one entry per piece: task index << 20 | piece index (a start vertex of degree 2^31 has 2^20 pieces)
__global__ void k_bk_pieces(int64_t cnt, const int64_t *__restrict__ poff, unsigned long long *__restrict__ pieces) {
    const int64_t qi = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (qi >= cnt) return;
    const int64_t b = poff[qi], e = poff[qi + 1];
    for (int64_t i = b; i < e; ++i) pieces[i] = ((unsigned long long)qi << 20) | (unsigned long long)(i - b);
}
// k_bk_block: the same build by a WORKGROUP per start vertex, shaped like the triangle kernels — the candidates C = N+(v) staged in LDS as
// the 65536-bit hub bitmap + the index of the first candidate of every bitmap word (candidates ascend, so local index = that + popcount
// of the lower bits) and a 32768-bit filter in front of the ascending tail-candidate list; then every row that can hold an edge into C
// is STREAMED with 16-byte loads by a group of GMSX_BK_BLOCK_GROUP lanes (8; 16 in rounds 3-4): the rows N+(a_i) of the candidates (hits -> Cadj, both directions) and the rows
// N+(t) of the in-neighbours t (hits -> XT).  A streamed id costs a bitmap probe (one LDS read + bit test) instead of a hash-table walk,
// sixteen rows are in flight per workgroup with the ids and extents of the next two batches already loading, and no wave idles behind a
// dependent chain.  XT has one column per CSR POSITION of v's row (x = degree; the positions of out-neighbours stay empty), so no
// compaction or scan over the in-neighbours is needed; the root record's Xf is the mask of the in-neighbour positions.
// A work item is a PIECE of a start vertex: kBkPieceJobs of its row jobs.  A hub late in the order has few candidates but 10^5..10^6 CSR
// positions, and one workgroup walking them sixteen at a time WAS the kernel's duration (measured: 94 of 129 ms remained with the row scans
// compiled out); the LDS sets cost only the candidates to rebuild, every hit is an atomic OR, so the pieces of one vertex run anywhere.
// Cadj | XT are zeroed by the host (one fill of the chunk's arena span) before the launch; piece 0 writes the root record.
struct BkRowJob {
    int32_t a;         // rank id whose oriented row is streamed; < 0: nothing (an out-neighbour position, or past the end)
    int64_t hs, he, ts, te;
};
#ifndef GMSX_BK_BLOCK_WAVES
#define GMSX_BK_BLOCK_WAVES 5
#endif
#ifndef GMSX_BK_BLOCK_GROUP
#define GMSX_BK_BLOCK_GROUP 8  // lanes per row job of k_bk_block (round 5; 16 before).  configs[3]: 16 / 8 / 4 lanes = 62.9 / 58.1 / 56.9 ms for the kernel, 147.7 / 142.8 / 141.5 for the call — most rows of a start vertex's neighbours are a unit or two, and a group has ONE row in flight; 8: the long rows of hub candidates still move 128 bytes per step
#endif
#ifndef GMSX_BK_ROWSCAN
#define GMSX_BK_ROWSCAN 0  // 0: one 16-byte load in flight per lane and row part; 1: two.  configs[3]: alone 37.6 / 36.7 ms, the whole call 212.3 / 214.0 ms
                           // (beside the LDS tasks the extra registers and instructions cost more than the second load gains)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GMSX_BK_BLOCK_WAVES))) void k_bk_block(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, const int32_t *__restrict__ newid,
                                                  const int32_t *__restrict__ oldid, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, const int32_t *__restrict__ dplus,
                                                  const int32_t *__restrict__ task_v, int64_t lo, int nparts, int part, int64_t q0,
                                                  const unsigned long long *__restrict__ pieces, int64_t p0, int64_t p1,
                                                  const int64_t *__restrict__ aoff, const int64_t *__restrict__ roff, unsigned long long *__restrict__ queue,
                                                  BkShared sh) {
    __shared__ __attribute__((aligned(16))) uint32_t bm[2048];
    __shared__ unsigned short pre[2048];
    __shared__ __attribute__((aligned(16))) uint32_t flt[1024];
    constexpr int kTailLds = 1024;
    __shared__ int32_t tcand[kTailLds];  // the ascending tail candidates (when they fit): the filter lets ~tc / 32768 of the streamed tail ids
                                          // through, and a binary search in GLOBAL memory behind it stalled the whole wave for ~8 dependent round trips
    __shared__ long long s_task;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int GL = GMSX_BK_BLOCK_GROUP, NG = 256 / GL;  // lanes per row job, row jobs in flight per workgroup
    const int grp = tid / GL, sub = tid % GL;
    long long t_next = 0;
    int t_have = 0;  // pieces left of this workgroup's queue ticket (kBkBlockGrab per ticket)
    while (true) {
        __syncthreads();
        if (t_have == 0) {
            if (tid == 0) s_task = (long long)atomicAdd(queue, (unsigned long long)kBkBlockGrab);
            __syncthreads();
            t_next = s_task;
            t_have = kBkBlockGrab;
        }
        const long long t_cur = t_next++;
        --t_have;
        if (p0 + t_cur >= p1) break;
        const unsigned long long piece = pieces[p0 + t_cur];
        const int64_t qi = int64_t(piece >> 20);
        const int piece_i = int(piece & 0xfffffull);
        const int32_t v = task_v[lo + qi * nparts + part];
        const int32_t vo = oldid[v];
        const int c = dplus[v];
        const int64_t ob = off[vo], oe = off[vo + 1];
        const int deg = int(oe - ob);
        const int cw = (c + 31) >> 5, xw = (deg + 31) >> 5;
        const int64_t vhb = hoff[v], vtb = toff[v];
        const int tc = int(toff[v + 1] - vtb), hc = c - tc;  // hub candidates (the 0xFFFF pad excluded), tail candidates
        const unsigned long long a0 = (unsigned long long)(aoff[qi] - aoff[q0]);
        uint32_t *Cadj = sh.arena + a0;
        uint32_t *XT = Cadj + size_t(c) * cw;
        uint32_t *rec = sh.pool + (unsigned long long)(roff[qi] - roff[q0]);
        for (int i = tid; i < 512; i += 256) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4 *>(flt)[tid] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int i = tid; i < hc; i += 256) {
            const uint32_t id = hadj[vhb + i];
            atomicOr(&bm[id >> 5], 1u << (id & 31u));
            if (i == 0 || (uint32_t(hadj[vhb + i - 1]) >> 5) != (id >> 5)) pre[id >> 5] = (unsigned short)i;  // candidates ascend
        }
        for (int i = tid; i < tc; i += 256) {
            const uint32_t id = uint32_t(tadj[vtb + i]);
            atomicOr(&flt[(id >> 5) & 1023u], 1u << (id & 31u));
            if (i < kTailLds) tcand[i] = int32_t(id);
        }
        // the record (piece 0): header, P = C, Xc = ext = {}, Xf = the in-neighbour positions of v's CSR row
        if (piece_i == 0) {
        if (tid == 0) {
            rec[0] = uint32_t(v);
            rec[1] = uint32_t(c);
            rec[2] = uint32_t(deg);
            rec[3] = deg > c ? 1u : 0u;
            rec[4] = uint32_t(a0 & 0xffffffffull);
            rec[5] = uint32_t(a0 >> 32);
            rec[6] = 1u;  // root: enter the node
            rec[7] = 0u;
            sh.dir[qi - q0] = (unsigned long long)(roff[qi] - roff[q0]);
        }
        for (int w = tid; w < cw; w += 256) {
            const int bits = c - w * 32;
            rec[kRecHeader + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
            rec[kRecHeader + cw + w] = 0u;
            rec[kRecHeader + 2 * cw + w] = 0u;
        }
        for (int pb = wave * 64; pb < deg; pb += 256) {
            const int p = pb + lane;
            const bool keep = p < deg && newid[adj[ob + p]] > v;
            const unsigned long long m = __ballot(keep);
            if (lane == 0) {
                rec[kRecHeader + 3 * cw + (pb >> 5)] = uint32_t(m);
                if ((pb >> 5) + 1 < xw) rec[kRecHeader + 3 * cw + (pb >> 5) + 1] = uint32_t(m >> 32);
            }
        }
        }
        __syncthreads();
        // ---- the row jobs: j < c -> candidate i = j (hits into Cadj); else CSR position p = j - c (in-neighbours only; hits into XT column p)
        const int njobs = int(min(int64_t(c) + deg, (int64_t(piece_i) + 1) * kBkPieceJobs));  // this piece: jobs [piece_i * kBkPieceJobs, njobs)
        // THE JOB PIPELINE.  A job's row sits behind a chain of dependent fetches: CSR entry -> rank id (newid) -> extents (hoff / toff)
        // -> the row.  Three stages run a trip apart per lane group — trip t issues  raw(t+3) | newid(raw(t+2)) | extents(id(t+1))  and
        // streams the rows of job t — under two rules without which hipcc serialises them again (rounds 2-3: 33 of the kernel's 42 waits
        // were vmcnt(0), the kernel 76 % waiting):
        //   * every load is UNCONDITIONAL (clamped index, the value dropped afterwards, written as selects — behind an `if` the compiler
        //     sinks the load into the branch): a load under a branch cannot be counted, and every later wait becomes vmcnt(0);
        //   * a loaded value is first TOUCHED one trip later (the stage registers hold raw loaded words; the selects that turn them into a
        //     job id / an extent run at the top of the next trip): a use right behind the load is a wait right behind the load.
        const int j0 = piece_i * kBkPieceJobs + grp;  // this group's jobs: j0, j0 + NG, … < njobs
        struct RawLoad { int32_t xh, xt, xa; };
        struct ExtLoad { int64_t hs, he, ts, te; };
        auto load_raw = [&](int j) -> RawLoad {
            const int jj = max(min(j, njobs - 1), 0);
            RawLoad r;
            r.xh = int32_t(hadj[vhb + min(jj, max(hc - 1, 0))]);  // (hadj / tadj end in padding: an empty part reads it)
            r.xt = tadj[vtb + min(max(jj - hc, 0), max(tc - 1, 0))];
            r.xa = adj[ob + min(max(jj - c, 0), max(deg - 1, 0))];
            return r;
        };
        auto fin_raw = [&](const RawLoad &r, int j) -> int32_t {  // >= 0: old id of CSR position j - c; <= -2: candidate job, rank id = -(x) - 2; -1: none
            const int32_t cand = -(j < hc ? r.xh : r.xt) - 2;
            const int32_t x = j < c ? cand : r.xa;
            return (j < j0 || j >= njobs) ? -1 : x;
        };
        auto fin_id = [&](int32_t nw, int32_t raw) -> int32_t {  // the rank id whose row the job streams, -1 = none
            const int32_t in = nw > v ? nw : -1;
            return raw >= 0 ? in : (raw == -1 ? -1 : -(raw + 2));
        };
        auto load_ext = [&](int32_t a) -> ExtLoad {
            const int32_t aa = max(a, 0);
            return ExtLoad{hoff[aa], hoff[aa + 1], toff[aa], toff[aa + 1]};
        };
        RawLoad l_raw = load_raw(j0 - NG);        // (jobs in front of j0 do not exist: fin_raw drops them)
        int32_t raw_n = -1, l_nw = 0;             // the raw job whose newid is in flight, and that word
        int32_t id_e = -1;                        // the job id whose extents are in flight
        ExtLoad l_ext = load_ext(-1);
        for (int j = j0 - 3 * NG; j < njobs; j += NG) {
            // top of the trip: what the previous trip loaded becomes values
            const int32_t raw2 = fin_raw(l_raw, j + 2 * NG);
            const int32_t id1 = fin_id(l_nw, raw_n);
            const BkRowJob cur{id_e, id_e >= 0 ? l_ext.hs : 0, id_e >= 0 ? l_ext.he : 0, id_e >= 0 ? l_ext.ts : 0, id_e >= 0 ? l_ext.te : 0};
            // … and the next round of fetches goes out before this trip's rows
            l_raw = load_raw(j + 3 * NG);
            l_nw = newid[max(raw2, 0)];
            raw_n = raw2;
            l_ext = load_ext(id1);
            id_e = id1;
#if defined(GMSX_BK_AB) && GMSX_BK_AB == 1  // A/B builds (wrong counts): 1 = the job pipeline alone, 2 = + the row loads without their probes
            if (cur.a == -12345) {
#else
            if (cur.a >= 0) {
#endif
                const bool is_cand = j < c;
                const int col = j - c;
                auto hit = [&](int k) {
#ifdef GMSX_BK_NO_HIT_ATOMICS  // A/B build (wrong counts): the build without its global atomics — 100 of 103 ms remain: they are not its limit
                    if (k < 0) Cadj[0] = 1;
                    return;
#endif
                    if (is_cand) {
                        atomicOr(&Cadj[size_t(j) * cw + (k >> 5)], 1u << (k & 31));
                        atomicOr(&Cadj[size_t(k) * cw + (j >> 5)], 1u << (j & 31));
                    } else {
                        atomicOr(&XT[size_t(k) * xw + (col >> 5)], 1u << (col & 31));
                    }
                };
                auto probe_hub = [&](const bk_u4 &p4, int left) {  // eight 16-bit ids against the bitmap; the 0xFFFF pad is never in it
                    const uint32_t wds[4] = {p4.x, p4.y, p4.z, p4.w};
#if defined(GMSX_BK_AB) && GMSX_BK_AB == 2
                    if ((wds[0] ^ wds[1] ^ wds[2] ^ wds[3]) == 0x12345678u && left == 7) hit(0);
                    return;
#endif
                    uint32_t mask = 0;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t id = (wds[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
                        mask |= ((bm[id >> 5] >> (id & 31u)) & 1u) << t;
                    }
                    mask &= left >= 8 ? 0xffu : ((1u << max(left, 0)) - 1u);
                    while (mask) {
                        const int t = __ffs(mask) - 1;
                        mask &= mask - 1;
                        const uint32_t id = (wds[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
                        const uint32_t word = bm[id >> 5];
                        hit(int(pre[id >> 5]) + __popc(word & ((1u << (id & 31u)) - 1u)));
                    }
                };
                auto probe_tail = [&](const bk_u4 &p4, int left) {  // four 32-bit ids against the filter, the few that pass against the ascending list
                    const uint32_t wds[4] = {p4.x, p4.y, p4.z, p4.w};
#if defined(GMSX_BK_AB) && GMSX_BK_AB == 2
                    if ((wds[0] ^ wds[1] ^ wds[2] ^ wds[3]) == 0x12345678u && left == 7) hit(0);
                    return;
#endif
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t id = wds[t];
                        if (t < left && ((flt[(id >> 5) & 1023u] >> (id & 31u)) & 1u)) {
                            int lo2 = 0, hi2 = tc;
                            if (tc <= kTailLds) {
                                while (lo2 < hi2) {
                                    const int mid = (lo2 + hi2) >> 1;
                                    if (uint32_t(tcand[mid]) < id) lo2 = mid + 1; else hi2 = mid;
                                }
                                if (lo2 < tc && uint32_t(tcand[lo2]) == id) hit(hc + lo2);
                            } else {
                                while (lo2 < hi2) {
                                    const int mid = (lo2 + hi2) >> 1;
                                    if (uint32_t(tadj[vtb + mid]) < id) lo2 = mid + 1; else hi2 = mid;
                                }
                                if (lo2 < tc && uint32_t(tadj[vtb + lo2]) == id) hit(hc + lo2);
                            }
                        }
                    }
                };
                const int64_t te = tc > 0 ? cur.te : cur.ts;  // no tail candidates: the tail part of the row cannot hit
#if GMSX_BK_ROWSCAN == 0
                for (int64_t q = cur.hs + sub * 8; q < cur.he; q += 8 * GL) probe_hub(*reinterpret_cast<const bk_u4 *>(hadj + q), int(min(int64_t(8), cur.he - q)));
                for (int64_t q = cur.ts + sub * 4; q < te; q += 4 * GL) probe_tail(*reinterpret_cast<const bk_u4 *>(tadj + q), int(min(int64_t(4), te - q)));
#else
                // two loads in flight per lane and part; a probe is issued only where a load was (most rows of a late start vertex have no hub
                // part at all: probing the zeros of an absent load cost a third of the kernel's VALU instructions)
                for (int64_t q = cur.hs + sub * 8; q < cur.he; q += 16 * GL) {
                    const bk_u4 h0 = *reinterpret_cast<const bk_u4 *>(hadj + q);
                    const bool two = q + 8 * GL < cur.he;
                    bk_u4 h1{0u, 0u, 0u, 0u};
                    if (two) h1 = *reinterpret_cast<const bk_u4 *>(hadj + q + 8 * GL);
                    probe_hub(h0, int(min(int64_t(8), cur.he - q)));
                    if (two) probe_hub(h1, int(min(int64_t(8), cur.he - q - 8 * GL)));
                }
                for (int64_t q = cur.ts + sub * 4; q < te; q += 8 * GL) {
                    const bk_u4 t0 = *reinterpret_cast<const bk_u4 *>(tadj + q);  // tadj is padded by four ids
                    const bool two = q + 4 * GL < te;
                    bk_u4 t1{0u, 0u, 0u, 0u};
                    if (two) t1 = *reinterpret_cast<const bk_u4 *>(tadj + q + 4 * GL);
                    probe_tail(t0, int(min(int64_t(4), te - q)));
                    if (two) probe_tail(t1, int(min(int64_t(4), te - q - 4 * GL)));
                }
#endif
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_bk_build(const int64_t *__restrict__ off, const int32_t *__restrict__ adj, const int32_t *__restrict__ newid,
                                                 const int32_t *__restrict__ oldid, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                 const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, const int32_t *__restrict__ dplus,
                                                 const int32_t *__restrict__ task_v, int64_t lo, int nparts, int part, int64_t q0, int64_t q1,
                                                 const int64_t *__restrict__ aoff, const int64_t *__restrict__ roff, unsigned long long *__restrict__ queue,
                                                 uint32_t *__restrict__ map_scratch, unsigned long long map_words, BkShared sh) {
    constexpr uint32_t kLdsMapSlots = 1024;
    __shared__ __attribute__((aligned(16))) unsigned long long lds_map[kLdsMapSlots];
    __shared__ int32_t in_stage[64];
    const int lane = threadIdx.x;
    while (true) {
        unsigned long long t0 = 0;
        if (lane == 0) t0 = atomicAdd(queue, 1ull);
        const int64_t qi = q0 + int64_t(uni64(t0));
        if (qi >= q1) break;
        const int32_t v = uni32(task_v[lo + qi * nparts + part]);
        const int32_t vo = uni32(oldid[v]);
        const int c = uni32(dplus[v]);
        const int64_t ob = uni64(off[vo]), oe = uni64(off[vo + 1]);
        const int x = int(oe - ob) - c;
        const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
        const uint32_t msize = bk_map_size(c);
        unsigned long long *map = msize <= kLdsMapSlots ? lds_map : reinterpret_cast<unsigned long long *>(map_scratch + size_t(blockIdx.x) * map_words);
        const unsigned long long a0 = (unsigned long long)(uni64(aoff[qi]) - uni64(aoff[q0]));
        uint32_t *Cadj = sh.arena + a0;
        uint32_t *XT = Cadj + size_t(c) * cw;
        bk_build<false>(off, adj, newid, hoff, hadj, toff, tadj, sh, v, c, x, ob, oe, map, msize, Cadj, XT, in_stage, lane);
        // the root record: P = C, Xc = {}, ext unused (the search enters the node and picks its pivot), Xf = X0
        const unsigned long long r0 = (unsigned long long)(uni64(roff[qi]) - uni64(roff[q0]));
        uint32_t *rec = sh.pool + r0;
        if (lane == 0) {
            rec[0] = uint32_t(v);
            rec[1] = uint32_t(c);
            rec[2] = uint32_t(x);
            rec[3] = x > 0 ? 1u : 0u;
            rec[4] = uint32_t(a0 & 0xffffffffull);
            rec[5] = uint32_t(a0 >> 32);
            rec[6] = 1u;  // root: enter the node
            rec[7] = 0u;
            sh.dir[qi - q0] = r0;
        }
        for (int w = lane; w < cw; w += 64) {
            const int bits = c - w * 32;
            rec[kRecHeader + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
            rec[kRecHeader + cw + w] = 0u;
            rec[kRecHeader + 2 * cw + w] = 0u;
        }
        for (int w = lane; w < xw; w += 64) {
            const int bits = x - w * 32;
            rec[kRecHeader + 3 * cw + w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#ifndef GMSX_BK_RESUME_WAVES
#define GMSX_BK_RESUME_WAVES 5  // waves per SIMD the one-word-per-lane resume kernel is compiled for: at 6 (80 VGPRs) the fast paths of the search
                                // spill 40 bytes per lane; 5 (102 VGPRs, no scratch) is 5 ms faster on the configs[3] graph
#endif
// Rounds >= 1: one wave per resumable record; Cadj | XT are read from the arena, the stack lives in this wave's slab.
template <int WPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPL == 1 ? GMSX_BK_RESUME_WAVES : WPL == 2 ? 3 : 1))) void k_bk_resume(const uint32_t *__restrict__ pool_in, const unsigned long long *__restrict__ dir_in,
                                                  unsigned long long n_records, unsigned long long *__restrict__ queue, unsigned grab,
                                                  uint32_t *__restrict__ slabs, unsigned long long slab_words,
                                                  unsigned long long *__restrict__ acc, BkShared sh, int min_c /* records with fewer candidates belong to k_bk_resume4 */) {
    __shared__ unsigned char xfne_stack[2052 * WPL];
    __shared__ uint32_t piv_P[64 * WPL];
    __shared__ unsigned short piv_list[2048 * WPL];
    const int lane = threadIdx.x;
    uint32_t *stack = slabs + size_t(blockIdx.x) * slab_words;
    unsigned long long cnt = 0, node_words = 0;
    unsigned long long q_next = 0, q_end = 0;  // `grab` records per queue ticket (tickets on one address are a serial resource: ~10 ns each)
    while (true) {
        if (q_next == q_end) {
            unsigned long long t0 = 0;
            if (lane == 0) t0 = atomicAdd(queue, (unsigned long long)grab);
            q_next = uni64(t0);
            q_end = q_next + grab;
        }
        const unsigned long long q0 = q_next++;
        if (q0 >= n_records) break;
        const unsigned long long roff = uni64(dir_in[q0]);
        if (roff == ~0ull) continue;  // a claimed-but-unwritten directory slot (its wave kept the search)
        const uint32_t *rec = pool_in + roff;
        const int32_t v = uni32(int32_t(rec[0]));
        const int c = uni32(int(rec[1])), x = uni32(int(rec[2])), xf_ne = uni32(int(rec[3]));
        if (c < min_c) continue;
        const bool root = uni32(rec[6]) != 0;  // written by k_bk_build: the node is entered (pivot choice), not resumed
        const int pairs = uni32(int(rec[7]));  // Xf as a list of (word index, word) pairs (a record split off by k_bk_resume4), 0 = dense words
        const unsigned long long aoff = (unsigned long long)uni32(rec[4]) | ((unsigned long long)uni32(rec[5]) << 32);
        const int cw = (c + 31) >> 5, xw = (x + 31) >> 5;
        const uint32_t *Cadj = sh.arena + aoff;
        const uint32_t *XT = Cadj + size_t(c) * cw;
#ifdef GMSX_BK_STATS  // profiling build: records and their Xf words by candidate-count bucket, in the unused slots beside the accumulators
        if (lane == 0) {
            const int b = c <= 32 ? 0 : c <= 64 ? 1 : c <= 128 ? 2 : c <= 256 ? 3 : c <= 512 ? 4 : c <= 1024 ? 5 : 6;
            atomicAdd(&acc[(blockIdx.x & 63) * 16 + 1 + b], 1ull);
            atomicAdd(&acc[(blockIdx.x & 63) * 16 + 8 + b], (unsigned long long)xw);
        }
#endif
        uint32_t P[WPL], Xc[WPL], ext[WPL];
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
            const int w = lane + 64 * h;
            P[h] = w < cw ? rec[kRecHeader + w] : 0u;
            Xc[h] = w < cw ? rec[kRecHeader + cw + w] : 0u;
            ext[h] = w < cw ? rec[kRecHeader + 2 * cw + w] : 0u;
        }
        if (pairs == 0) {
            for (int w = lane; w < xw; w += 64) stack[3 * cw + w] = rec[kRecHeader + 3 * cw + w];
        } else {
            for (int w = lane; w < xw; w += 64) stack[3 * cw + w] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (readers: this wave; an agent-scope fence also flushes the XCD's L2)
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < pairs; i += 64) stack[3 * cw + rec[kRecHeader + 3 * cw + 2 * i]] = rec[kRecHeader + 3 * cw + 2 * i + 1];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (readers: this wave; an agent-scope fence also flushes the XCD's L2)
        }
        __builtin_amdgcn_wave_barrier();
        bk_search<WPL>(Cadj, XT, stack, xfne_stack, v, c, x, P, Xc, ext, xf_ne, root, lane, cnt, sh, aoff, true, piv_P, piv_list, node_words);
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && cnt) atomicAdd(&acc[(blockIdx.x & 63) * 16], cnt);
    if (lane == 0 && node_words) atomicAdd(&acc[(blockIdx.x & 63) * 16 + 15], node_words);
}

// ---- rounds >= 1 for records with at most 512 candidates: FOUR searches per wave -------------------------------------------------------
// k_bk_resume runs one search per wave: with c <= 512 (every record of the BASELINE configs[3] graph) at most 16 of its 64 lanes hold a
// word of P, every search node is a chain of ~3 dependent global loads, and the node's ~100 wave-instructions are issued for those 16 lanes
// (round 4: 61 % of the wave cycles parked, VALU 39 % busy).  Here a search is a 16-LANE GROUP (lane `sub` holds word `sub` of P / Xc / ext),
// a wave runs four of them in lock-step, and the search is a STATE MACHINE with one step per trip of the wave's loop:
//   * every step consumes ONE row of Cadj — the pivot's (ext = P & ~row), the branch vertex's (child P' = P & row) or the only candidate's
//     of a one-candidate child — or one saved level of the stack (pop).  Which row the next step needs is decided at the END of a step, so
//     the loads of all four groups are issued together, unconditionally placed at the top of the trip, and waited for once: four
//     independent dependent-load chains per wave instead of one;
//   * the finished in-neighbours Xf (x bits, one per CSR position of the start vertex: hundreds of words, nearly all zero below the root —
//     measured: 2.8 non-zero words per pushed child) are a LIST of (word index, word) pairs below level 0.  A list of at most 16 pairs lives
//     in registers, one pair per lane, and its gather XT[q][index] is one more load of the same hoisted batch: the Xf checks of the leaf fast
//     paths and the child Xf of a push cost no extra round trip.  Longer lists and the dense level 0 of a root record are walked in memory.
// Records are those of k_bk_resume (same pool, same header; rec[7] = pairs of a list-form Xf, 0 = dense words), so the two kernels share
// the rounds: this one takes c <= 512, k_bk_resume the wider ones.  Tomita's recursion is unchanged (tomita.h:12-86): pivot = argmax
// |P ∩ N(u)| over P ∪ Xc, scored by the lanes of the group; nodes with <= small_p candidates take their first candidate.
struct __attribute__((packed, aligned(4))) bk_u2 { uint32_t x, y; };
// reductions over a group of G = 4 / 8 / 16 consecutive lanes by DPP: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror — every lane ends with the result
template <int CTRL> __device__ __forceinline__ int bkg_dpp(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, false); }
template <int G> __device__ __forceinline__ int bkg_sum(int x) {
    x += bkg_dpp<0xB1>(x); x += bkg_dpp<0x4E>(x);
    if (G >= 8) x += bkg_dpp<0x141>(x);
    if (G >= 16) x += bkg_dpp<0x140>(x);
    return x;
}
template <int G> __device__ __forceinline__ int bkg_max(int x) {
    x = max(x, bkg_dpp<0xB1>(x)); x = max(x, bkg_dpp<0x4E>(x));
    if (G >= 8) x = max(x, bkg_dpp<0x141>(x));
    if (G >= 16) x = max(x, bkg_dpp<0x140>(x));
    return x;
}
template <int G> __device__ __forceinline__ int bkg_min(int x) {
    x = min(x, bkg_dpp<0xB1>(x)); x = min(x, bkg_dpp<0x4E>(x));
    if (G >= 8) x = min(x, bkg_dpp<0x141>(x));
    if (G >= 16) x = min(x, bkg_dpp<0x140>(x));
    return x;
}
template <int G> __device__ __forceinline__ int bkg_first(uint32_t w, int sub) { return bkg_min<G>(w != 0u ? (sub << 5) + __ffs(w) - 1 : 0xffff); }  // lowest set bit of the group's set
template <int G> __device__ __forceinline__ uint32_t bkg_ballot(bool p, int gsh) { return uint32_t(__ballot(p) >> gsh) & ((1u << G) - 1u); }  // the group's bits of the wave ballot
template <int G> __device__ __forceinline__ int bkg_scan(int x, int sub) {  // inclusive prefix sum over the lanes of the group (rare paths only)
    for (int d = 1; d < G; d <<= 1) {
        const int t = __shfl_up(x, d, G);
        if (sub >= d) x += t;
    }
    return x;
}

// Xf of a level held in memory — dense words (n < 0: the xw words at src) or a list of n (word index, word) pairs — against row(s) of XT.
// MODE 0: the child list Xf ∩ N(q) -> dst, returns its length; 1: is Xf ∩ N(q) non-empty; 2: is Xf ∩ N(q) ∩ N(q2) non-empty.
template <int MODE, int G>
__device__ __forceinline__ int bkg_xf_mem(const uint32_t *src, int n, int xw, const uint32_t *xt, const uint32_t *xt2, uint32_t *dst, int sub, int gsh) {
    const uint32_t lt = (1u << sub) - 1u;
    int out = 0;
    if (n < 0) {
        for (int w0 = 0; w0 < xw; w0 += 4 * G) {
            const int w = w0 + 4 * sub;
            const int wl = min(w, (xw - 1) & ~3);  // lanes behind the row load its last unit again (unclamped they read up to 4 G - 1 words past a record or the last XT row — the arena and the pools carry 64 bytes of slack)
            const bk_u4 a = *reinterpret_cast<const bk_u4 *>(src + wl), b = *reinterpret_cast<const bk_u4 *>(xt + wl);  // at most three words past the row: masked below
            uint32_t t[4] = {a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w};
            if (MODE == 2) {
                const bk_u4 b2 = *reinterpret_cast<const bk_u4 *>(xt2 + wl);
                t[0] &= b2.x; t[1] &= b2.y; t[2] &= b2.z; t[3] &= b2.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (w + i >= xw) t[i] = 0u;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t m = bkg_ballot<G>(t[i] != 0u, gsh);
                    if (t[i] != 0u) *reinterpret_cast<bk_u2 *>(dst + 2 * (out + __popc(m & lt))) = bk_u2{uint32_t(w + i), t[i]};
                    out += __popc(m);
                }
            } else {
                if (bkg_ballot<G>((t[0] | t[1] | t[2] | t[3]) != 0u, gsh)) return 1;
            }
        }
    } else {
        for (int i0 = 0; i0 < n; i0 += G) {
            const int i = i0 + sub;
            bk_u2 pr{0u, 0u};
            if (i < n) pr = *reinterpret_cast<const bk_u2 *>(src + 2 * i);
            uint32_t t = pr.y & xt[pr.x];
            if (MODE == 2) t &= xt2[pr.x];
            if (MODE == 0) {
                const uint32_t m = bkg_ballot<G>(t != 0u, gsh);
                if (t != 0u) *reinterpret_cast<bk_u2 *>(dst + 2 * (out + __popc(m & lt))) = bk_u2{pr.x, t};
                out += __popc(m);
            } else {
                if (bkg_ballot<G>(t != 0u, gsh)) return 1;
            }
        }
    }
    return out;
}

#ifdef GMSX_BK_STATS
#define BKG_ST(i, v) st_[i] += (unsigned long long)(v)
#define BKG_GROUPS(i, pred) st_[i] += (unsigned long long)(__popcll(__ballot(pred)) / G)
#else
#define BKG_ST(i, v) do { } while (0)
#define BKG_GROUPS(i, pred) do { } while (0)
#endif

// the searches of one class of records (G lanes per search, 64 / G searches per wave) until the class's queue is empty
template <int G>
__device__ __forceinline__ void bk_group_searches(const uint32_t *__restrict__ pool_in, const unsigned long long *__restrict__ dir_in, unsigned long long n_records,
                                                  unsigned long long *__restrict__ queue, unsigned grab, uint32_t *__restrict__ wave_slab,
                                                  unsigned long long slab_words /* per search */, const BkShared &sh, unsigned short *piv_list /* LDS, 2048 */,
                                                  uint32_t *piv_P /* LDS, 64 */, unsigned long long &cnt_out, unsigned long long &node_words_out,
                                                  unsigned long long *st_) {
    // what the step at the top of the next trip consumes: M_PIVOT the pivot's row (ext = P & ~row), M_PIVOTB the same when the pivot is a candidate —
    // then it is also the first branch vertex and its row serves both (a node's branches may be taken in any order) —, M_BRANCH the row of branch
    // vertex q, M_CHILD1 the row of the only candidate of q's child, M_POP a saved level; M_ENTER / M_NEXT: a record was fetched (root / resumed)
    enum : int { M_REC = 0, M_ENTER, M_NEXT, M_PIVOT, M_PIVOTB, M_BRANCH, M_CHILD1, M_POP, M_DONE };
    constexpr int kMaxC = 32 * G, kMinC = G == GMSX_BK_GROUP_MIN ? 0 : 16 * G;  // this class: kMinC < c <= kMaxC
    constexpr int kFixed = G * kBkSlot;
    (void)st_;
    const int lane = threadIdx.x, sub = lane & (G - 1), grp = lane / G, gsh = lane & ~(G - 1);
    const uint32_t lt = (1u << sub) - 1u;
    uint32_t *const slab = wave_slab + size_t(grp) * slab_words;
    unsigned short *const my_list = piv_list + grp * kMaxC;
    uint32_t *const my_P = piv_P + grp * G;
    unsigned long long cnt = 0, node_words = 0;
    // the group's search (every lane of the group holds the same value of what is not a bitmap word)
    const uint32_t *rec = pool_in, *Cadj = pool_in, *XT = pool_in, *xf0 = pool_in;
    uint32_t c = 0, cw = 0, xw = 0, depth = 0, lvl = kFixed;
    int mode = M_REC;
    int xfn = 0;                   // Xf of the current level: 0 empty, -1 the dense words of level 0, n > 0 a list of n pairs (n <= G: in ridx / rval)
    uint32_t P = 0, Xc = 0, ext = 0, Xcn = 0, ridx = 0, rval = 0, tq = 0;
    // the level above the current one, while it is in registers (GMSX_BK_PARENT_REGS): a node whose children are all decided in place — most nodes — comes
    // back to its parent without a step of its own, and the parent is written to its slot only when a grandchild is entered
    bool has_par = false;
    uint32_t pP = 0, pXc = 0, pext = 0, pridx = 0, prval = 0;
    int pxfn = 0;
    uint32_t q = 0, rowreq = 0;
    unsigned nodes = 0;
    bool nosplit = false;
    unsigned long long q_next = 0, q_end = 0;
    auto level = [&](uint32_t l) { return slab + (unsigned long long)l * lvl; };
    auto xf_src = [&](uint32_t l) { return l == 0 ? xf0 : level(l) + kFixed; };
    while (true) {
        // ---- a group without a search takes the next record of its class ---------------------------------------------------------------
        BKG_ST(1, __ballot(mode == M_REC) != 0);
        while (mode == M_REC) {
            if (q_next == q_end) {
                unsigned long long t0 = 0;
                if (sub == 0) t0 = atomicAdd(queue, (unsigned long long)grab);
                q_next = (unsigned long long)__shfl((long long)t0, gsh);
                q_end = q_next + grab;
            }
            const unsigned long long qi = q_next++;
            if (qi >= n_records) { mode = M_DONE; break; }
            const unsigned long long roff = dir_in[qi];
            if (roff == ~0ull) continue;  // a claimed-but-unwritten directory slot (its search was kept by the wave that claimed it)
            rec = pool_in + roff;
            c = rec[1];
            if (c > uint32_t(kMaxC) || c <= uint32_t(kMinC)) continue;  // another class's (beyond 512: k_bk_resume's)
            const uint32_t x = rec[2];
            const unsigned long long aoff = (unsigned long long)rec[4] | ((unsigned long long)rec[5] << 32);
            const uint32_t xf_ne = rec[3], root = rec[6], pairs = rec[7];
            cw = (c + 31) >> 5;
            xw = (x + 31) >> 5;
            lvl = uint32_t(bk_group_level_words(G, xw));
            Cadj = sh.arena + aoff;
            XT = Cadj + size_t(c) * cw;
            P = sub < cw ? rec[kRecHeader + sub] : 0u;
            Xc = sub < cw ? rec[kRecHeader + cw + sub] : 0u;
            ext = sub < cw ? rec[kRecHeader + 2 * cw + sub] : 0u;
            xf0 = rec + kRecHeader + 3 * cw;
            xfn = xf_ne == 0u ? 0 : pairs != 0u ? int(pairs) : -1;
            ridx = rval = 0u;
            if (xfn > 0 && xfn <= G && sub < uint32_t(xfn)) {
                const bk_u2 pr = *reinterpret_cast<const bk_u2 *>(xf0 + 2 * sub);
                ridx = pr.x;
                rval = pr.y;
            }
            asm volatile("" : "+v"(ridx), "+v"(rval), "+v"(P), "+v"(Xc), "+v"(ext));  // waited for HERE, not by a vmcnt(0) in front of the next trip's loads
            depth = 0;
            nodes = 0;
            rowreq = 0;
            nosplit = false;
            has_par = false;
            if (root != 0u) { ++nodes; mode = M_ENTER; } else mode = M_NEXT;
        }
        if (__ballot(mode != M_DONE) == 0) break;
        BKG_ST(0, 1);
        BKG_GROUPS(4, mode == M_PIVOT); BKG_GROUPS(5, mode == M_PIVOTB); BKG_GROUPS(6, mode == M_BRANCH); BKG_GROUPS(7, mode == M_CHILD1);
        BKG_GROUPS(8, mode == M_POP); BKG_GROUPS(9, mode == M_DONE);
        // ---- the loads of this step: four per lane, UNCONDITIONAL (every address is valid whatever the mode: a stale row index, the slot of the
        //      current level), issued back to back for all groups, consumed below behind counted waits ------------------------------------------
        uint32_t *const lv = slab + (unsigned long long)depth * lvl + sub * kBkSlot;
        const uint32_t row = Cadj[rowreq * cw + sub];  // lanes past cw read the next row: their words of P / Xc are zero
        uint32_t xtw = 0u;
        uint4 sv{0u, 0u, 0u, 0u};
        uint2 sl{0u, 0u};
        // (wave-uniform branches: a trip in which no group pops / no group holds a register list leaves the loads out — they were a third of the
        //  kernel's traffic beyond the L2 as unconditional dummies — and the waits stay counted: everything is consumed at one point below)
        if (__ballot(xfn > 0 && xfn <= G && (mode == M_BRANCH || mode == M_CHILD1 || mode == M_PIVOTB)) != 0) xtw = XT[(unsigned long long)rowreq * xw + ridx];
        if (__ballot(mode == M_POP) != 0) {
            sv = *reinterpret_cast<const uint4 *>(lv);
            sl = *reinterpret_cast<const uint2 *>(lv + 4);
        }
        bool next = mode == M_NEXT || mode == M_POP || mode == M_PIVOT;
        bool need_pivot = mode == M_ENTER;
        int pc = 0, first = 0;  // of the node whose pivot is chosen below
        {   // selects, not branches: a load whose only use sits under a branch is sunk into it by the compiler, and waited for right there
            const bool pop = mode == M_POP, piv = mode == M_PIVOT || mode == M_PIVOTB;
            P = pop ? sv.x : P;
            Xc = pop ? sv.y : Xc;
            ext = pop ? sv.z : ext;
            xfn = pop ? int(sv.w) : xfn;
            ridx = pop ? sl.x : ridx;
            rval = pop ? sl.y : rval;
            ext = piv ? P & ~row : ext;
            asm volatile("" : "+v"(xtw));  // (used by one branch below: this keeps its load up here with the others)
        }
        const bool reglist = xfn > 0 && xfn <= G;
        if (mode == M_BRANCH || mode == M_CHILD1 || mode == M_PIVOTB) {
            // ---- the child of branch vertex q (M_CHILD1: the one-candidate child of q, through the row of its candidate) -------------------
            const bool c1 = mode == M_CHILD1;
            const uint32_t Pn = c1 ? 0u : P & row, Xq = (c1 ? Xcn : Xc) & row, t = (c1 ? tq : rval) & xtw;
            if (!c1 && uint32_t(sub) == (q >> 5)) {  // this node continues with q moved from cand to fini (tomita.h:68-70)
                const uint32_t b = 1u << (q & 31);
                ext &= ~b; P &= ~b; Xc |= b;
            }
            const int pcn = bkg_sum<G>(__popc(Pn));
            const int fn = bkg_first<G>(Pn, sub);
            if (pcn == 0) {
                // no candidate: a leaf, decided here — one maximal clique iff no finished vertex is adjacent to the whole clique
                bool blocked = bkg_ballot<G>(Xq != 0u, gsh) != 0u;
                if (!blocked && xfn != 0) {
                    if (reglist) blocked = bkg_ballot<G>(t != 0u, gsh) != 0u;
                    else {
                        BKG_ST(2, 1);
                        blocked = c1 ? bkg_xf_mem<2, G>(xf_src(depth), xfn, int(xw), XT + size_t(q) * xw, XT + size_t(rowreq) * xw, nullptr, sub, gsh) != 0
                                     : bkg_xf_mem<1, G>(xf_src(depth), xfn, int(xw), XT + size_t(q) * xw, nullptr, nullptr, sub, gsh) != 0;
                    }
                }
                if (!blocked && sub == 0) cnt++;
                nodes += c1 ? 2u : 1u;
                next = true;
            } else if (pcn == 1) {
                // ONE candidate: decided by the next step from that candidate's row, no level pushed
                Xcn = Xq;
                tq = t;
                rowreq = uint32_t(fn);
                mode = M_CHILD1;
            } else {
                // enter the child.  The level is kept only if it has another branch left (else nothing would come back to it): in registers, the
                // level those held before goes to its slot
                const bool last = bkg_ballot<G>(ext != 0u, gsh) == 0u && (xfn == 0 || reglist);
                BKG_GROUPS(10, true); BKG_GROUPS(11, last);
#if GMSX_BK_PARENT_REGS
                if (!last) {
                    if (has_par) {
                        uint32_t *pl = lv - lvl;  // the slot of level depth - 1
                        *reinterpret_cast<uint4 *>(pl) = uint4{pP, pXc, pext, uint32_t(pxfn)};
                        *reinterpret_cast<uint2 *>(pl + 4) = uint2{pridx, prval};
                    }
                    pP = P; pXc = Xc; pext = ext; pxfn = xfn; pridx = ridx; prval = rval;
                    has_par = true;
                }
#else
                if (!last) {
                    *reinterpret_cast<uint4 *>(lv) = uint4{P, Xc, ext, uint32_t(xfn)};
                    *reinterpret_cast<uint2 *>(lv + 4) = uint2{ridx, rval};
                }
#endif
                int nxf = 0;
                if (xfn != 0) {
                    if (reglist) {
                        const uint32_t mm = bkg_ballot<G>(t != 0u, gsh);
                        nxf = __popc(mm);
                        const int dst = t != 0u ? __popc(mm & lt) : G - 1;  // lanes without a pair send a zero to a lane no pair goes to
                        const uint32_t si = t != 0u ? ridx : 0u;
                        ridx = uint32_t(__builtin_amdgcn_ds_permute((gsh + dst) << 2, int(si)));
                        rval = uint32_t(__builtin_amdgcn_ds_permute((gsh + dst) << 2, int(t)));
                    } else {
                        BKG_ST(2, 1);
                        uint32_t *dl = level(depth + 1) + kFixed;
                        nxf = bkg_xf_mem<0, G>(xf_src(depth), xfn, int(xw), XT + size_t(q) * xw, nullptr, dl, sub, gsh);
                        ridx = rval = 0u;
                        if (nxf <= G && sub < nxf) {
                            const bk_u2 pr = *reinterpret_cast<const bk_u2 *>(dl + 2 * sub);
                            ridx = pr.x;
                            rval = pr.y;
                        }
                        asm volatile("" : "+v"(ridx), "+v"(rval));  // (as in the fetch above)
                    }
                }
                if (!last) ++depth;
                P = Pn; Xc = Xq; xfn = nxf;
                ++nodes;
                need_pivot = true;
                pc = pcn;
                first = fn;
            }
        }
        // ---- pivot of an entered node: argmax over u in P ∪ Xc of |P ∩ N(u)| (tomita.h:12-40); its row is the next step's ----------------
        if (need_pivot) {
            if (mode == M_ENTER) {
                pc = bkg_sum<G>(__popc(P));
                first = bkg_first<G>(P, sub);
            }
            int best = first;
            bool in_p = true;
            if (pc > sh.small_p_groups) {
                BKG_ST(3, 1);
                const uint32_t U = P | Xc;
                const int mine = __popc(U);
                const int incl = bkg_scan<G>(mine, sub);
                const int ncand = __shfl(incl, gsh + G - 1);
                my_P[sub] = P;
                {
                    int at = incl - mine;
                    uint32_t bits = U;
                    while (bits) {
                        my_list[at++] = (unsigned short)((sub << 5) + __ffs(bits) - 1);
                        bits &= bits - 1u;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int key = -1;
                const int cw4 = int(cw + 3) >> 2;
                for (int k = sub; k < ncand; k += G) {
                    const int u = int(my_list[k]);
                    const uint32_t *r = Cadj + size_t(u) * cw;
                    int sc = 0;
                    for (int w4 = 0; w4 < cw4; ++w4) {  // words past cw belong to the next row: P's words there are zero
                        const bk_u4 rw = *reinterpret_cast<const bk_u4 *>(r + 4 * w4);
                        sc += __popc(my_P[4 * w4] & rw.x) + __popc(my_P[4 * w4 + 1] & rw.y) + __popc(my_P[4 * w4 + 2] & rw.z) + __popc(my_P[4 * w4 + 3] & rw.w);
                    }
                    key = max(key, (sc << 16) | (0xffff - u));  // ties -> the smallest index
                }
                key = bkg_max<G>(key);
                best = 0xffff - (key & 0xffff);
                in_p = ((my_P[best >> 5] >> (best & 31)) & 1u) != 0u;
                __builtin_amdgcn_wave_barrier();
            }
            rowreq = uint32_t(best);
            q = uint32_t(best);  // (read by M_PIVOTB only)
            mode = in_p ? M_PIVOTB : M_PIVOT;
        }
        // ---- the next branch vertex of this node, or the way back up ----------------------------------------------------------------------
        if (next) {
#if GMSX_BK_PARENT_REGS
            {   // a finished node whose parent is in registers: the parent is the current node again, right here (it has a branch left: it was kept)
                const bool back = has_par && bkg_ballot<G>(ext != 0u, gsh) == 0u;
                P = back ? pP : P;
                Xc = back ? pXc : Xc;
                ext = back ? pext : ext;
                xfn = back ? pxfn : xfn;
                ridx = back ? pridx : ridx;
                rval = back ? prval : rval;
                depth -= back ? 1u : 0u;
                has_par = has_par && !back;
            }
#endif
            const int fq = bkg_first<G>(ext, sub);
            if (fq == 0xffff) {
                if (depth == 0) {
                    mode = M_REC;
                    node_words += (unsigned long long)nodes * cw;  // the search is through
                } else { --depth; mode = M_POP; }
            } else {
                bool split = false;
                if (nodes >= sh.budget && !nosplit) {
                    // ---- split: every level that still has branches becomes up to kBkSplit records of the next round (as in bk_search) ----
                    uint32_t *cur = level(depth) + sub * kBkSlot;
                    *reinterpret_cast<uint4 *>(cur) = uint4{P, Xc, ext, uint32_t(xfn)};
                    *reinterpret_cast<uint2 *>(cur + 4) = uint2{ridx, rval};
                    if (has_par) {
                        uint32_t *pl = cur - lvl;
                        *reinterpret_cast<uint4 *>(pl) = uint4{pP, pXc, pext, uint32_t(pxfn)};
                        *reinterpret_cast<uint2 *>(pl + 4) = uint2{pridx, prval};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (readers: this wave; an agent-scope fence also flushes the XCD's L2)
                    int nrec = 0;
                    for (uint32_t l = 0; l <= depth; ++l) nrec += min(bkg_sum<G>(__popc(level(l)[sub * kBkSlot + 2])), kBkSplit);
                    const unsigned long long rec_words = (unsigned long long)(kRecHeader + 3 * cw + xw);
                    unsigned long long p0 = 0, d0 = sh.dir_cap;
                    if (sub == 0) {
                        p0 = atomicAdd(sh.pool_head, rec_words * nrec);
                        if (p0 + rec_words * nrec <= sh.pool_cap) d0 = atomicAdd(sh.dir_count, (unsigned long long)nrec);
                    }
                    p0 = (unsigned long long)__shfl((long long)p0, gsh);
                    d0 = (unsigned long long)__shfl((long long)d0, gsh);
                    if (p0 + rec_words * nrec > sh.pool_cap || d0 + nrec > sh.dir_cap) {
                        nosplit = true;  // no room: this search is finished here (claimed directory slots keep their ~0 fill)
                    } else {
                        auto lowest = [](uint32_t x, int k) -> uint32_t {  // the k lowest set bits of x
                            if (k <= 0) return 0u;
                            if (k >= __popc(x)) return x;
                            uint32_t r = 0;
                            while (k--) {
                                const uint32_t b = x & (0u - x);
                                r |= b;
                                x ^= b;
                            }
                            return r;
                        };
                        int r = 0;
                        for (uint32_t l = 0; l <= depth; ++l) {
                            const uint32_t *ls = level(l) + sub * kBkSlot;
                            const uint32_t lP = ls[0], lXc = ls[1], e = ls[2];
                            const int ln = int(ls[3]);
                            const uint32_t lidx = ls[4], lval = ls[5];
                            const int mine = __popc(e);
                            const int incl = bkg_scan<G>(mine, sub);
                            const int nb = __shfl(incl, gsh + G - 1);
                            if (nb == 0) continue;
                            const int rank0 = incl - mine;  // rank of this word's first pending branch
                            const int parts = min(nb, kBkSplit);
                            const bool in_regs = ln >= 0 && ln <= G;  // the level's Xf: empty or a register list (saved in the slot) — else in memory
                            const uint32_t *src = xf_src(l);            // (memory form at level 0: the Xf of the record this search came from)
                            for (int j = 0; j < parts; ++j) {
                                const int a = int((long long)nb * j / parts), b = int((long long)nb * (j + 1) / parts);  // ranks [a, b) of the pending branches
                                uint32_t *out = sh.pool + p0 + rec_words * r;
                                // Xf of the level: as a list when it fits the dense area of the record, else scattered into dense words
                                uint32_t o3 = 0u, o7 = 0u;
                                uint32_t *ox = out + kRecHeader + 3 * cw;
                                if (ln < 0) {  // the dense words of level 0
                                    o3 = 1u;
                                    for (uint32_t w = sub; w < xw; w += G) ox[w] = src[w];
                                } else if (ln > 0) {
                                    o3 = 1u;
                                    if (2u * uint32_t(ln) <= xw) {
                                        o7 = uint32_t(ln);
                                        if (in_regs) { if (sub < ln) *reinterpret_cast<bk_u2 *>(ox + 2 * sub) = bk_u2{lidx, lval}; }
                                        else for (uint32_t i = sub; i < 2u * uint32_t(ln); i += G) ox[i] = src[i];
                                    } else {
                                        for (uint32_t w = sub; w < xw; w += G) ox[w] = 0u;
                                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (readers: this wave; an agent-scope fence also flushes the XCD's L2)
                                        if (in_regs) { if (sub < ln) ox[lidx] = lval; }
                                        else for (uint32_t i = sub; i < uint32_t(ln); i += G) ox[src[2 * i]] = src[2 * i + 1];
                                    }
                                }
                                if (sub == 0) {
                                    out[0] = rec[0];
                                    out[1] = c;
                                    out[2] = rec[2];
                                    out[3] = o3;
                                    out[4] = rec[4];
                                    out[5] = rec[5];
                                    out[6] = 0u;
                                    out[7] = o7;
                                    sh.dir[d0 + r] = p0 + rec_words * r;
                                }
                                if (uint32_t(sub) < cw) {
                                    const uint32_t before = lowest(e, a - rank0);        // branches of the runs in front of this one
                                    const uint32_t run = lowest(e, b - rank0) & ~before;  // this run
                                    out[kRecHeader + sub] = lP & ~before;                 // P
                                    out[kRecHeader + cw + sub] = lXc | before;            // Xc
                                    out[kRecHeader + 2 * cw + sub] = run;                 // ext
                                }
                                ++r;
                            }
                        }
                        if (sub == 0) {
                            atomicMax(sh.max_stack, (unsigned long long)(c + 1) * (unsigned long long)(3 * cw + xw + 1));
                            atomicMax(sh.max_stack + 1 + bk_group_class(c), bk_group_slab_words(c, xw));
                        }
                        split = true;
                    }
                }
                if (split) {
                    mode = M_REC;
                    node_words += (unsigned long long)nodes * cw;
                } else {
                    q = uint32_t(fq);
                    rowreq = q;
                    mode = M_BRANCH;
                }
            }
        }
    }
    if (sub == 0) {
        cnt_out += cnt;
        node_words_out += node_words;
    }
}

#ifndef GMSX_BK_GROUP_WAVES
#define GMSX_BK_GROUP_WAVES 4  // 3 (168 VGPRs): 191-202 ms on configs[3], 5 (102, spills in the step): 265
#endif
// every wave drains the queues of all classes one after the other (odd workgroups the widest class first), so that no wave idles while any class has records
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GMSX_BK_GROUP_WAVES))) void k_bk_resume4(
    const uint32_t *__restrict__ pool_in, const unsigned long long *__restrict__ dir_in, unsigned long long n_records, unsigned long long *__restrict__ queues /* [3] */,
    unsigned grab, uint32_t *__restrict__ slabs, unsigned long long wave_slab_words, unsigned long long slab16, unsigned long long slab8, unsigned long long slab4,
    unsigned long long *__restrict__ acc, BkShared sh) {
    __shared__ unsigned short piv_list[2048];
    __shared__ uint32_t piv_P[64];
    uint32_t *const wave_slab = slabs + size_t(blockIdx.x) * wave_slab_words;
    unsigned long long cnt = 0, node_words = 0;
#ifdef GMSX_BK_STATS
    // profiling build: [0] trips of the wave loop, [1] trips in which a group fetched, [2] with a memory walk of Xf, [3] with pivot scoring; group steps in
    // [4] PIVOT [5] PIVOTB [6] BRANCH [7] CHILD1 [8] POP, [9] idle (DONE) group-trips, [10] pushes, [11] of them without a kept level (lane 0's groups only)
    unsigned long long st_[12] = {};
#else
    unsigned long long *st_ = nullptr;
#endif
    (void)slab4;
    const bool wide_first = (blockIdx.x & 1) != 0;
    if (wide_first) bk_group_searches<16>(pool_in, dir_in, n_records, queues + 0, grab, wave_slab, slab16, sh, piv_list, piv_P, cnt, node_words, st_);
#if GMSX_BK_GROUP_MIN <= 4
    bk_group_searches<4>(pool_in, dir_in, n_records, queues + 2, grab, wave_slab, slab4, sh, piv_list, piv_P, cnt, node_words, st_);
#endif
#if GMSX_BK_GROUP_MIN <= 8
    bk_group_searches<8>(pool_in, dir_in, n_records, queues + 1, grab, wave_slab, slab8, sh, piv_list, piv_P, cnt, node_words, st_);
#endif
    if (!wide_first) bk_group_searches<16>(pool_in, dir_in, n_records, queues + 0, grab, wave_slab, slab16, sh, piv_list, piv_P, cnt, node_words, st_);
    if (cnt) atomicAdd(&acc[(blockIdx.x & 63) * 16], cnt);
    if (node_words) atomicAdd(&acc[(blockIdx.x & 63) * 16 + 15], node_words);
#ifdef GMSX_BK_STATS
    if (threadIdx.x == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_bkg_stat[i], st_[i]);
#endif
}

// gmsx_stats.stream_bytes of a Bron-Kerbosch call, build part: per start vertex of the shard every row that can hold an edge into its candidate set — the
// oriented rows of ALL its neighbours (candidates -> Cadj, in-neighbours -> XT), 2 bytes per hub id + 4 per tail id: what k_bk_block streams.  One wave
// per start vertex (task order, the shard's stride).
__global__ __launch_bounds__(256) void k_stat_bk_bytes(int64_t n_tasks, int nparts, int part, const int32_t *__restrict__ task_v, const int64_t *__restrict__ off,
                                                     const int32_t *__restrict__ adj, const int32_t *__restrict__ newid, const int32_t *__restrict__ oldid,
                                                     const int64_t *__restrict__ hoff, const int64_t *__restrict__ toff, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long b = 0;
    for (int64_t qi = wave0;; qi += nwaves) {
        const int64_t pos = qi * nparts + part;
        if (pos >= n_tasks) break;
        const int32_t vo = oldid[task_v[pos]];
        const int64_t ob = off[vo], oe = off[vo + 1];
        for (int64_t j = ob + lane; j < oe; j += 64) {
            const int32_t w = newid[adj[j]];
            b += 2ull * (unsigned long long)(hoff[w + 1] - hoff[w]) + 4ull * (unsigned long long)(toff[w + 1] - toff[w]);
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) b += __shfl_xor(b, sft);
    if (lane == 0 && b) atomicAdd(out, b);
}

static int64_t part_count(int64_t first, int64_t end, int nparts, int part) {
    const int64_t span = end - first - part;
    return span <= 0 ? 0 : (span + nparts - 1) / nparts;
}

static int bk_partial(const gmsx_graph *g, int part, int nparts, uint64_t *out, gmsx_stats *st) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    const int64_t n = g->n;
    struct Guard { void *p = nullptr; ~Guard() { (void)hipFree(p); } } g_acc, g_ki, g_ko, g_vi, g_vo, g_tmp, g_arena, g_pool0, g_pool1, g_dir0, g_dir1;
    unsigned long long *acc = nullptr;
    // control words after the 64 spread accumulators: [0] isolated vertices [1] giant tasks [2] queue [3] arena head [4] pool head [5] directory count
    // [6] stack words of k_bk_resume [7..9] slab words of a 16- / 8- / 4-lane search [10] build queue [11] LDS-task queue [12..14] k_bk_resume4's queues [16..20] layout maxima
    constexpr int kCtl = 64 * 16;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), sizeof(unsigned long long) * (kCtl + 32)));
    g_acc.p = acc;
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * (kCtl + 32), s));
    if (n == 0) {
        *out = 0;
        if (st) *st = gmsx_stats{0.0, 0.0, 0, 0, 0, 0, 0};
        return GMSX_OK;
    }
    // ---- tasks: slab requirement per start vertex, heavy first (untimed setup, like the reference's preprocessing step)
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    int32_t *v_in = nullptr, *v_out = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&k_in), size_t(n) * 8)); g_ki.p = k_in;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&k_out), size_t(n) * 8)); g_ko.p = k_out;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&v_in), size_t(n) * 4)); g_vi.p = v_in;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&v_out), size_t(n) * 4)); g_vo.p = v_out;
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    int max_c = kBkMaxCand;
    if (const char *e = opt("BK_MAXC")) {  // test hook: a lower width limit sends more start vertices through the memory-resident search
        const int v = std::atoi(e);
        if (v >= 1 && v < kBkMaxCand) max_c = v;
    }
    const unsigned long long giant_cap = (unsigned long long)std::min<int64_t>(n, int64_t(1) << 20);
    int32_t *giant = nullptr;
    Guard g_giant;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&giant), size_t(giant_cap) * 4));
    g_giant.p = giant;
    hipLaunchKernelGGL(k_bk_tasks, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, g->off, g->oldid, g->dplus, k_in, v_in, max_c, giant, giant_cap,
                       acc + kCtl);
    size_t tmp_bytes = 0;
    GMSX_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, size_t(n), 0, 64, s));
    void *tmp = nullptr;
    GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    g_tmp.p = tmp;
    GMSX_HIP(rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, k_in, k_out, v_in, v_out, size_t(n), 0, 64, s));
    std::unique_ptr<unsigned long long[]> words_mem(new (std::nothrow) unsigned long long[static_cast<size_t>(n > 0 ? n : 1)]);  // 1 GB at scale 27
    if (!words_mem) return GMSX_ERR_NOMEM;
    unsigned long long *const words = words_mem.get();
    unsigned long long head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GMSX_HIP(hipMemcpyAsync(words, k_out, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipMemcpyAsync(head, acc + kCtl, sizeof(head), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    const int64_t n_giant = int64_t(head[1]);    // start vertices beyond the register-resident width: memory-resident search after the rounds
    const unsigned long long giant_slab_w = (head[2] + 3ull) & ~3ull;
    if (n_giant > int64_t(giant_cap)) return GMSX_ERR_UNSUPPORTED;  // more than a million of them
    const int64_t n_wide = int64_t(head[7]);     // tasks with more than 2048 candidates: sorted first (kWideTask), WPL = 2 / 4 / 8 kernels
    const int widest = std::min(g->max_dplus, max_c);  // candidates of a start vertex = its d+
    const int wpl_wide = widest <= 4096 ? 2 : widest <= 8192 ? 4 : 8;
    for (int64_t i = 0; i < n_wide; ++i) words[size_t(i)] &= ~kWideTask;
    GMSX_HIP(hipMemsetAsync(acc + kCtl + 7, 0, 8, s));  // ([7] held the number of wide tasks until here)

    // ---- arena + record pools of the load balancer
    const int cu = c.compute_units > 0 ? c.compute_units : 256;
    size_t free_b = 0, total_b = 0;
    GMSX_HIP(hipMemGetInfo(&free_b, &total_b));
    const unsigned long long budget_bytes = std::min<unsigned long long>(free_b / 4, 16ull << 30);
    BkShared sh{};
    // the arena (Cadj | XT of the start vertices of a chunk + of the LDS-slab searches that split) is sized by NEED once the layout of the
    // start vertices is known — a fixed 48 GB allocation per call cost seconds of first-touch time now and then
    unsigned long long arena_hard_cap = std::min<unsigned long long>(free_b / 4, 48ull << 30) / 4;
    if (const char *e = opt("BK_ARENA_MB")) {  // test hook: a small arena makes small graphs build their roots in several chunks
        const long v = std::atol(e);
        if (v >= 1) arena_hard_cap = std::min<unsigned long long>(arena_hard_cap, ((unsigned long long)v << 20) / 4);
    }
    auto alloc_arena = [&](unsigned long long want_words) -> int {
        sh.arena_cap = std::min(arena_hard_cap, std::max<unsigned long long>(want_words, (256ull << 20) / 4));
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&sh.arena), sh.arena_cap * 4 + 64));  // + 64: 16-byte loads may run 12 bytes past a row
        g_arena.p = sh.arena;
        return GMSX_OK;
    };
    // 1 = records with at most 512 candidates are searched four to a wave (k_bk_resume4, default); 0 = every record by k_bk_resume (round 4)
    const bool use_groups = [] { const char *e = opt("BK_GROUPS"); return !e || std::atoi(e) != 0; }();
    sh.pool_cap = std::min<unsigned long long>(free_b / 16, 2ull << 30) / 4;
    sh.dir_cap = 8ull << 20;
    uint32_t *pools[2] = {nullptr, nullptr};
    unsigned long long *dirs[2] = {nullptr, nullptr};
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&pools[0]), sh.pool_cap * 4 + 64)); g_pool0.p = pools[0];  // + 64: 16-byte loads may run 12 bytes past a record
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&pools[1]), sh.pool_cap * 4 + 64)); g_pool1.p = pools[1];
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dirs[0]), sh.dir_cap * 8)); g_dir0.p = dirs[0];
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&dirs[1]), sh.dir_cap * 8)); g_dir1.p = dirs[1];
    unsigned long long *queue = acc + kCtl + 2, *gqueue = acc + kCtl + 12;  // tickets of k_bk_resume / k_bk_resume4
    sh.arena_head = acc + kCtl + 3;
    sh.pool_head = acc + kCtl + 4;
    sh.dir_count = acc + kCtl + 5;
    sh.max_stack = acc + kCtl + 6;
    // nodes before a search is re-split.  Round 3 swept 128 … 8192 on the config-4 graph and Kronecker scale 14 and took 512; with round 4's build
    // pipeline the configs[3] graph takes 248.6 / 209.5 / 203.1 / 200.0 / 204.0 ms at 256 / 512 / 1024 / 2048 / 4096, RMAT 18 ef 64 (a = .45)
    // 275.9 -> 260.1 ms at 2048, the sparse graphs (one round) are indifferent
    sh.budget = 2048;
    sh.small_p = [] { const char *e = opt("BK_SMALL_P"); return e ? std::atoi(e) : 6; }();  // swept on the configs[3] graph: 0 (off) 316 ms, 2 304, 3 ~300, 4 295, 6 293, 8 295, 16 303
    sh.small_p_groups = [] { const char *e = opt("BK_SMALL_P_GROUPS"); return e ? std::atoi(e) : 12; }();
    sh.bmoff = g->bmoff;
    sh.bmpool = g->bmpool;
    sh.dense_limit = g->dense_limit;
    if (const char *e = opt("BK_BUDGET")) {  // tuning knob: nodes a search may visit before it is re-split
        const long v = std::atol(e);
        if (v >= 16 && v <= (1l << 30)) sh.budget = unsigned(v);
    }
    unsigned budget0 = sh.budget;  // round 0 (start vertices: build + first stretch of the search)
    if (const char *e = opt("BK_BUDGET0")) {
        const long v = std::atol(e);
        if (v >= 16 && v <= (1l << 30)) budget0 = unsigned(v);
    }
    const unsigned budget_resume = sh.budget;
    sh.budget = budget0;
    int cur = 0;
    sh.pool = pools[cur];
    sh.dir = dirs[cur];
    GMSX_HIP(hipMemsetAsync(sh.dir, 0xff, sh.dir_cap * 8, s));
    // 2 (default): k_bk_block, a workgroup per start vertex, rows streamed against an LDS bitmap; 1: k_bk_build, a wave per start vertex with
    // the hash-map build of k_bk_wave; 0: round 2's combined build + search bins
    const int64_t resume_grab = [] { const char *e = opt("BK_RESUME_GRAB"); return e ? std::max(1, std::atoi(e)) : 1; }();  // measured on configs[3]: 8 costs 4 ms (the records of a round differ in cost; their queue is not the limit)
    const int split_build = [] { const char *e = opt("BK_SPLIT_BUILD"); return e ? std::atoi(e) : 2; }();
    int64_t n_tasks = 0;
    while (n_tasks < n && words[size_t(n_tasks)] > 0) ++n_tasks;
    int64_t n_glob = 0;  // tasks beyond an LDS slab: sorted first (the wide ones, > 2048 candidates, at the very front)
    while (n_glob < n_tasks && (n_glob < n_wide || words[size_t(n_glob)] > (unsigned long long)kLdsSlabWords)) ++n_glob;
    // LAYOUT of the start vertices that get their own build kernel (arena and record offsets by prefix sums) and the arena itself: setup like
    // the task sort above — allocations of gigabytes now and then stall for a second, they are not part of the kernels' time
    const int64_t cnt_glob = split_build ? part_count(0, n_glob, nparts, part) : 0;
    int64_t need_total = 0;  // arena words of the start vertices built by k_bk_block
    const int64_t cnt_tiny = part_count(n_glob, n_tasks, nparts, part);
    const bool tiny_roots = use_groups && split_build >= 2 && cnt_glob > 0 && [] { const char *e = opt("BK_TINY_ROOTS"); return e && std::atoi(e) != 0; }();
    int64_t *t_need_a = nullptr, *t_need_r = nullptr, *t_need_p = nullptr, *t_aoff = nullptr, *t_roff = nullptr, tiny_a = 0, tiny_r = 0;
    Guard g_tna, g_tnr, g_tnp, g_tao, g_tro;
    int64_t *need_a = nullptr, *need_r = nullptr, *aoff = nullptr, *roff = nullptr, *d_end = nullptr, *need_p = nullptr, *poff = nullptr;
    unsigned long long *pieces = nullptr;
    Guard g_na, g_nr, g_ao, g_ro, g_de, g_map, g_np, g_po, g_pc;
    unsigned long long *maxima = acc + kCtl + 16;  // [0] stack words, [1] global map words, [2..4] slab words of a 16- / 8- / 4-lane search
    unsigned long long mx[5] = {0, 0, 0, 0, 0}, map_words = 0;  // [0] stack words of k_bk_resume, [1] global map words, [2..4] slab words of a 16- / 8- / 4-lane search
    int64_t build_waves = 0;
    uint32_t *map_scratch = nullptr;
    if (cnt_glob > 0) {
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&need_a), size_t(cnt_glob + 1) * 8)); g_na.p = need_a;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&need_r), size_t(cnt_glob + 1) * 8)); g_nr.p = need_r;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&aoff), size_t(cnt_glob + 1) * 8)); g_ao.p = aoff;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&roff), size_t(cnt_glob + 1) * 8)); g_ro.p = roff;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_end), 8)); g_de.p = d_end;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&need_p), size_t(cnt_glob + 1) * 8)); g_np.p = need_p;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&poff), size_t(cnt_glob + 1) * 8)); g_po.p = poff;
        GMSX_HIP(hipMemsetAsync(maxima, 0, 40, s));
        hipLaunchKernelGGL(k_bk_layout, dim3(unsigned(cnt_glob / 256 + 1)), dim3(256), 0, s, int64_t(0), cnt_glob, nparts, part, v_out, g->off, g->oldid, g->dplus,
                           split_build >= 2 ? 1 : 0, need_a, need_r, need_p, maxima);
        {
            size_t scan_bytes = 0;
            GMSX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, need_a, aoff, int64_t(0), size_t(cnt_glob + 1), rocprim::plus<int64_t>(), s));
            void *scan_tmp = nullptr;
            GMSX_HIP(hipMalloc(&scan_tmp, scan_bytes ? scan_bytes : 8));
            Guard g_scan;
            g_scan.p = scan_tmp;
            GMSX_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, need_a, aoff, int64_t(0), size_t(cnt_glob + 1), rocprim::plus<int64_t>(), s));
            GMSX_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, need_r, roff, int64_t(0), size_t(cnt_glob + 1), rocprim::plus<int64_t>(), s));
            GMSX_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, need_p, poff, int64_t(0), size_t(cnt_glob + 1), rocprim::plus<int64_t>(), s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        if (split_build >= 2) {  // the work items of k_bk_block
            int64_t n_pieces = 0;
            GMSX_HIP(hipMemcpy(&n_pieces, poff + cnt_glob, 8, hipMemcpyDeviceToHost));
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&pieces), size_t(n_pieces + 1) * 8)); g_pc.p = pieces;
            hipLaunchKernelGGL(k_bk_pieces, dim3(unsigned(cnt_glob / 256 + 1)), dim3(256), 0, s, cnt_glob, poff, pieces);
        }
        // GMSX_BK_TINY_ROOTS=1 (off by default): the LDS-slab tasks are BUILT by k_bk_wave<true> and SEARCHED by k_bk_resume4 — they leave Cadj | XT in the arena
        // and a root record in the pool, at offsets from the same kind of layout scans (their x = in-neighbours only: the wave build compacts them).  Measured
        // on configs[3]: round 0 62 -> 55 ms (the build alone beside k_bk_block), round 1 32.8 -> 48.8 ms (1.65 M searches of ~33 nodes: a record fetch and a
        // root pivot each) — 156 against 147.4 ms: their search in LDS, hidden beside the build, is the cheaper place
        if (tiny_roots && cnt_tiny > 0) {
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&t_need_a), size_t(cnt_tiny + 1) * 8)); g_tna.p = t_need_a;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&t_need_r), size_t(cnt_tiny + 1) * 8)); g_tnr.p = t_need_r;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&t_need_p), size_t(cnt_tiny + 1) * 8)); g_tnp.p = t_need_p;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&t_aoff), size_t(cnt_tiny + 1) * 8)); g_tao.p = t_aoff;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&t_roff), size_t(cnt_tiny + 1) * 8)); g_tro.p = t_roff;
            hipLaunchKernelGGL(k_bk_layout, dim3(unsigned(cnt_tiny / 256 + 1)), dim3(256), 0, s, n_glob, cnt_tiny, nparts, part, v_out, g->off, g->oldid, g->dplus, 0,
                               t_need_a, t_need_r, t_need_p, maxima);
            size_t scan_bytes = 0;
            GMSX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, t_need_a, t_aoff, int64_t(0), size_t(cnt_tiny + 1), rocprim::plus<int64_t>(), s));
            void *scan_tmp = nullptr;
            GMSX_HIP(hipMalloc(&scan_tmp, scan_bytes ? scan_bytes : 8));
            Guard g_scan;
            g_scan.p = scan_tmp;
            GMSX_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, t_need_a, t_aoff, int64_t(0), size_t(cnt_tiny + 1), rocprim::plus<int64_t>(), s));
            GMSX_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, t_need_r, t_roff, int64_t(0), size_t(cnt_tiny + 1), rocprim::plus<int64_t>(), s));
            GMSX_HIP(hipStreamSynchronize(s));
            GMSX_HIP(hipMemcpy(&tiny_a, t_aoff + cnt_tiny, 8, hipMemcpyDeviceToHost));
            GMSX_HIP(hipMemcpy(&tiny_r, t_roff + cnt_tiny, 8, hipMemcpyDeviceToHost));
        }
        GMSX_HIP(hipMemcpy(mx, maxima, sizeof(mx), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(&need_total, aoff + cnt_glob, 8, hipMemcpyDeviceToHost));
        // the roots may take 3/4 of the arena (below): everything in one chunk when the device allows, + room for the LDS-slab searches that split
        if (int rc = alloc_arena((unsigned long long)need_total / 3 * 4 + (unsigned long long)tiny_a + (512ull << 20) / 4)) return rc;
        if (opt("BK_VERBOSE"))
            std::fprintf(stderr, "[gmsx bk] start vertices %lld: %lld built in the arena (%lld words), %lld in LDS slabs\n", (long long)n_tasks, (long long)cnt_glob,
                         (long long)need_total, (long long)(n_tasks - n_glob));
        map_words = split_build >= 2 ? 0ull : (mx[1] + 3ull) & ~3ull;
        build_waves = std::min<int64_t>(cnt_glob, int64_t(cu) * 24);
        if (map_words > 0) {
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&map_scratch), size_t(build_waves) * map_words * 4));
            g_map.p = map_scratch;
        }
    }
    // the stack slabs of the resume kernel (grow-only, reused by every round): for the root round their size is known from the layout
    Guard g_rslab, g_gslab;
    size_t resume_cap = 0, group_cap = 0;
    const bool any_wide_records = !use_groups || widest > kBkGroupMaxC;  // k_bk_resume has records to search
    if (cnt_glob > 0 && mx[0] > 0 && any_wide_records) {
        const unsigned long long slab_bytes = ((mx[0] + 3ull) & ~3ull) * 4ull;
        if (slab_bytes <= budget_bytes) {
            const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({cnt_glob, int64_t(cu) * 24, int64_t(budget_bytes / slab_bytes)}));
            resume_cap = size_t(waves) * slab_bytes;
            GMSX_HIP(hipMalloc(&g_rslab.p, resume_cap));
        }
    }
    const int group_waves_per_cu = 4 * GMSX_BK_GROUP_WAVES;
    // words of a wave's slab area in k_bk_resume4: 4 searches of 16 lanes, 8 of 8 or 16 of 4, whichever class needs most
    auto group_wave_words = [](unsigned long long w16, unsigned long long w8, unsigned long long w4) {
        return std::max({4ull * ((w16 + 3ull) & ~3ull), 8ull * ((w8 + 3ull) & ~3ull), 16ull * ((w4 + 3ull) & ~3ull)});
    };
    if (cnt_glob > 0 && (mx[2] | mx[3] | mx[4]) != 0 && use_groups) {
        const unsigned long long slab_bytes = group_wave_words(mx[2], mx[3], mx[4]) * 4ull;
        if (slab_bytes <= budget_bytes) {
            const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({(cnt_glob + (tiny_roots ? cnt_tiny : 0) + 3) / 4, int64_t(cu) * group_waves_per_cu, int64_t(budget_bytes / slab_bytes)}));
            group_cap = size_t(waves) * slab_bytes;
            GMSX_HIP(hipMalloc(&g_gslab.p, group_cap));
        }
    }
    GMSX_HIP(hipStreamSynchronize(s));
    GMSX_HIP(hipEventRecord(c.ev[1], s));

    // ---- round 0.  Start vertices whose structures fit an LDS slab (<= kLdsSlabWords): one wave builds and searches (k_bk_wave<true>).
    //      The others (GMSX_BK_SPLIT_BUILD=0 restores round 2's one-kernel bins): k_bk_build writes Cadj | XT into the arena and a root
    //      record per start vertex, in chunks that fit the arena and the record pool; the resume rounds below search them.
    int launches = 0;
    // ---- rounds >= 1: resume the split searches (and search the root records of k_bk_build) until no record is left
    int rounds = 0;
    // GMSX_BK_TINY_BESIDE=2: the LDS-slab tasks run beside the FIRST RESUME ROUND (what they split off joins that round's output) instead of beside the
    // build: hooks called around the launch of a round's kernels, and before the next round reads the counters
    std::function<int()> round_pre, round_post, round_join;
    auto run_rounds = [&]() -> int {
    while (true) {
        if (round_join) {
            if (int rc = round_join()) return rc;
            round_join = nullptr;
        }
        unsigned long long ctl[6] = {0, 0, 0, 0, 0, 0};  // pool_head, dir_count, max_stack, slab words of a 16- / 8- / 4-lane search
        GMSX_HIP(hipMemcpyAsync(ctl, sh.pool_head, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        // a wave whose allocation overshot a capacity kept its search: only fully written records are below the caps
        const unsigned long long n_rec = std::min(ctl[1], sh.dir_cap);
        if (n_rec == 0) break;
        const unsigned long long stack_w = (ctl[2] + 3ull) & ~3ull, group_w = group_wave_words(ctl[3], ctl[4], ctl[5]);
        const uint32_t *pool_in = pools[cur];
        const unsigned long long *dir_in = dirs[cur];
        cur ^= 1;
        sh.pool = pools[cur];
        sh.dir = dirs[cur];
        GMSX_HIP(hipMemsetAsync(queue, 0, 8, s));
        GMSX_HIP(hipMemsetAsync(gqueue, 0, 24, s));
        GMSX_HIP(hipMemsetAsync(sh.pool_head, 0, 6 * sizeof(unsigned long long), s));
        GMSX_HIP(hipMemsetAsync(sh.dir, 0xff, sh.dir_cap * 8, s));
        // few records left: split sooner so that the idle waves get work (the tail rounds are latency-, not throughput-bound)
        const int64_t full = int64_t(cu) * 24 * 2;
        sh.budget = int64_t(n_rec) >= full ? budget_resume : unsigned(std::max<int64_t>(128, int64_t(budget_resume) * int64_t(n_rec) / full));
        // records per queue ticket (GMSX_BK_RESUME_GRAB, default 1: eight cost 4 ms on configs[3] — the records of a round differ in cost, their queue is not the limit)
        const unsigned grab = unsigned(std::max<int64_t>(1, std::min<int64_t>(resume_grab, int64_t(n_rec) / (int64_t(cu) * 24 * 16))));
        if (round_pre) {
            if (int rc = round_pre()) return rc;
            round_pre = nullptr;
        }
        if (use_groups && group_w > 0) {  // records with at most 512 candidates: four searches per wave
            const unsigned long long slab_bytes = group_w * 4ull;
            if (slab_bytes > budget_bytes) return GMSX_ERR_DEVICE_MEM;
            const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({int64_t((n_rec + 3) / 4), int64_t(cu) * group_waves_per_cu, int64_t(budget_bytes / slab_bytes)}));
            if (size_t(waves) * slab_bytes > group_cap) {  // grow-only, reused by every round
                (void)hipFree(g_gslab.p);
                g_gslab.p = nullptr;
                group_cap = size_t(waves) * slab_bytes;
                GMSX_HIP(hipMalloc(&g_gslab.p, group_cap));
            }
            if (opt("BK_VERBOSE"))
                std::fprintf(stderr, "[gmsx bk] round %d: %llu records (%llu pool words), %lld waves of 4 / 8 / 16 searches, %llu slab words each, budget %u\n", rounds + 1,
                             n_rec, ctl[0], (long long)waves, group_w, sh.budget);
            hipLaunchKernelGGL(k_bk_resume4, dim3(unsigned(waves)), dim3(64), 0, s, pool_in, dir_in, n_rec, gqueue, grab, static_cast<uint32_t *>(g_gslab.p), group_w,
                               (ctl[3] + 3ull) & ~3ull, (ctl[4] + 3ull) & ~3ull, (ctl[5] + 3ull) & ~3ull, acc, sh);
            ++launches;
        }
        if (!use_groups || widest > kBkGroupMaxC) {  // the others (every record with GMSX_BK_GROUPS=0): one search per wave
            const int min_c = use_groups ? kBkGroupMaxC + 1 : 0;
            const unsigned long long slab_bytes = std::max<unsigned long long>(stack_w * 4ull, 16);
            if (slab_bytes > budget_bytes) return GMSX_ERR_DEVICE_MEM;
            const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({int64_t(n_rec), int64_t(cu) * 24, int64_t(budget_bytes / slab_bytes)}));
            if (size_t(waves) * slab_bytes > resume_cap) {  // grow-only stack slabs, reused by every round
                (void)hipFree(g_rslab.p);
                g_rslab.p = nullptr;
                resume_cap = size_t(waves) * slab_bytes;
                GMSX_HIP(hipMalloc(&g_rslab.p, resume_cap));
            }
            if (opt("BK_VERBOSE"))
                std::fprintf(stderr, "[gmsx bk] round %d: %llu records (%llu pool words), %lld waves, stack %llu words, budget %u\n", rounds + 1, n_rec, ctl[0],
                             (long long)waves, stack_w, sh.budget);
            if (n_wide > 0 && wpl_wide == 2)  // records of wide tasks may be anywhere in the pool
                hipLaunchKernelGGL(k_bk_resume<2>, dim3(unsigned(waves)), dim3(64), 0, s, pool_in, dir_in, n_rec, queue, grab,
                                   static_cast<uint32_t *>(g_rslab.p), stack_w, acc, sh, min_c);
            else if (n_wide > 0 && wpl_wide == 4)
                hipLaunchKernelGGL(k_bk_resume<4>, dim3(unsigned(waves)), dim3(64), 0, s, pool_in, dir_in, n_rec, queue, grab,
                                   static_cast<uint32_t *>(g_rslab.p), stack_w, acc, sh, min_c);
            else if (n_wide > 0)
                hipLaunchKernelGGL(k_bk_resume<8>, dim3(unsigned(waves)), dim3(64), 0, s, pool_in, dir_in, n_rec, queue, grab,
                                   static_cast<uint32_t *>(g_rslab.p), stack_w, acc, sh, min_c);
            else
                hipLaunchKernelGGL(k_bk_resume<1>, dim3(unsigned(waves)), dim3(64), 0, s, pool_in, dir_in, n_rec, queue, grab,
                                   static_cast<uint32_t *>(g_rslab.p), stack_w, acc, sh, min_c);
            ++launches;
        }
        if (round_post) {
            if (int rc = round_post()) return rc;
            round_post = nullptr;
        }
        if (++rounds > 100000) return GMSX_ERR_KERNEL;
    }
    return GMSX_OK;
    };

    // the LDS-slab tasks: their own queue word; `beside` = on a side stream next to the build kernel that was just launched on s (the build
    // streams rows at 8 workgroups per CU, the tiny searches live in LDS: each leaves what the other needs), joined before the rounds
    unsigned long long *tqueue = acc + kCtl + 11;
    bool tiny_beside = false;
    struct TinyJoin {  // every way out joins the side stream again (error returns free buffers the kernel beside may still use)
        Ctx &c;
        hipStream_t s;
        bool &armed;
        ~TinyJoin() {
            if (armed && hipEventRecord(c.ev_join[0], c.side[0]) == hipSuccess) {
                (void)hipStreamWaitEvent(s, c.ev_join[0], 0);
                (void)hipStreamSynchronize(s);
            }
        }
    } tiny_join{c, s, tiny_beside};
    unsigned long long emit_abase = 0, emit_rbase = 0, emit_dbase = 0;
    bool emit = false;  // set by the first chunk when the LDS-slab tasks' roots fit behind its own
    auto launch_tiny = [&](int64_t lo, int64_t hi, bool beside) -> int {
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (cnt <= 0) return GMSX_OK;
        GMSX_HIP(hipMemsetAsync(tqueue, 0, 8, s));
        hipStream_t st = s;
        if (beside && c.side[0] && c.ev_fork && c.ev_join[0]) {
            st = c.side[0];
            GMSX_HIP(hipEventRecord(c.ev_fork, s));
            GMSX_HIP(hipStreamWaitEvent(st, c.ev_fork, 0));
        }
        const int64_t waves = std::min<int64_t>(cnt, int64_t(cu) * 16);
        hipLaunchKernelGGL((k_bk_wave<true, 1>), dim3(unsigned(waves)), dim3(64), 0, st, g->off, g->adj, g->newid, g->oldid, g->hoff, g->hadj, g->toff, g->tadj,
                           g->dplus, v_out, lo, hi, nparts, part, tqueue, static_cast<uint32_t *>(nullptr), 0ull, acc, sh,
                           emit ? t_aoff : static_cast<const int64_t *>(nullptr), emit ? t_roff : static_cast<const int64_t *>(nullptr), emit_abase, emit_rbase, emit_dbase);
        ++launches;
        tiny_beside = st != s;
        return GMSX_OK;
    };
    auto join_tiny = [&]() -> int {  // after the kernel that runs beside it was launched on s
        if (tiny_beside) {
            tiny_beside = false;
            GMSX_HIP(hipEventRecord(c.ev_join[0], c.side[0]));
            GMSX_HIP(hipStreamWaitEvent(s, c.ev_join[0], 0));
        }
        return GMSX_OK;
    };
    if (split_build) {
        bool tiny_done = false;
        if (cnt_glob > 0) {
            unsigned long long *bqueue = acc + kCtl + 10;
            // the roots may take at most 3/4 of the arena and half of the pool: searches of the LDS-slab tasks that split need room too
            const int64_t a_cap = int64_t(sh.arena_cap / 4 * 3), r_cap = int64_t(sh.pool_cap / 2), max_tasks = int64_t(sh.dir_cap / 2);
            for (int64_t q0 = 0; q0 < cnt_glob;) {
                hipLaunchKernelGGL(k_bk_chunk_end, dim3(1), dim3(1), 0, s, q0, cnt_glob, aoff, roff, a_cap, r_cap, max_tasks, d_end);
                int64_t q1 = 0, span[4] = {0, 0, 0, 0};
                GMSX_HIP(hipMemcpyAsync(&q1, d_end, 8, hipMemcpyDeviceToHost, s));
                GMSX_HIP(hipStreamSynchronize(s));
                GMSX_HIP(hipMemcpy(&span[0], aoff + q0, 8, hipMemcpyDeviceToHost));
                GMSX_HIP(hipMemcpy(&span[1], aoff + q1, 8, hipMemcpyDeviceToHost));
                GMSX_HIP(hipMemcpy(&span[2], roff + q0, 8, hipMemcpyDeviceToHost));
                GMSX_HIP(hipMemcpy(&span[3], roff + q1, 8, hipMemcpyDeviceToHost));
                if (uint64_t(span[1] - span[0]) > sh.arena_cap || uint64_t(span[3] - span[2]) > sh.pool_cap) return GMSX_ERR_DEVICE_MEM;  // one start vertex beyond the arena
                // pool_head / dir_count / max_stack as if the roots had been split off by an earlier round; arena_head behind their structures
                unsigned long long ctl0[6] = {(unsigned long long)(span[3] - span[2]), (unsigned long long)(q1 - q0), mx[0], mx[2], mx[3], mx[4]};
                unsigned long long ah = (unsigned long long)(span[1] - span[0]);
                const int tiny_mode = [] { const char *e = opt("BK_TINY_BESIDE"); return e ? std::atoi(e) : 1; }();
                if (!tiny_done && tiny_roots && cnt_tiny > 0 && tiny_mode != 2 && ah + (unsigned long long)tiny_a + (64ull << 20) / 4 <= sh.arena_cap &&
                    ctl0[0] + (unsigned long long)tiny_r <= sh.pool_cap / 4 * 3 && ctl0[1] + (unsigned long long)cnt_tiny <= sh.dir_cap) {
                    emit = true;  // the LDS-slab tasks' structures and root records go behind this chunk's: round 1 searches both
                    emit_abase = ah;
                    emit_rbase = ctl0[0];
                    emit_dbase = ctl0[1];
                    ah += (unsigned long long)tiny_a;
                    ctl0[0] += (unsigned long long)tiny_r;
                    ctl0[1] += (unsigned long long)cnt_tiny;
                }
                const unsigned long long zero_words = (unsigned long long)(span[1] - span[0]);  // Cadj | XT the pieces of k_bk_block OR into
                GMSX_HIP(hipMemcpyAsync(sh.pool_head, ctl0, sizeof(ctl0), hipMemcpyHostToDevice, s));
                GMSX_HIP(hipMemcpyAsync(sh.arena_head, &ah, 8, hipMemcpyHostToDevice, s));
                GMSX_HIP(hipMemsetAsync(bqueue, 0, 8, s));
                int64_t pspan[2] = {0, 0};
                if (split_build >= 2) {  // Cadj | XT of the chunk start empty: the pieces of a start vertex only OR into them
                    GMSX_HIP(hipMemsetAsync(sh.arena, 0, size_t(zero_words) * 4, s));
                    GMSX_HIP(hipMemcpyAsync(&pspan[0], poff + q0, 8, hipMemcpyDeviceToHost, s));
                    GMSX_HIP(hipMemcpyAsync(&pspan[1], poff + q1, 8, hipMemcpyDeviceToHost, s));
                }
                GMSX_HIP(hipStreamSynchronize(s));  // ctl0 / ah / pspan are stack variables
                if (!tiny_done) {  // the LDS-slab tasks run beside the first chunk's build; what they split off joins its records
                    const int beside = [] { const char *e = opt("BK_TINY_BESIDE"); return e ? std::atoi(e) : 1; }();  // 0: one after the other (profiling)
                    if (beside == 2 && c.side[0] && c.ev_fork && c.ev_join[0] && part_count(n_glob, n_tasks, nparts, part) > 0) {
                        round_pre = [&]() -> int {
                            GMSX_HIP(hipMemsetAsync(tqueue, 0, 8, s));
                            GMSX_HIP(hipEventRecord(c.ev_fork, s));
                            return GMSX_OK;
                        };
                        round_post = [&]() -> int {
                            const int64_t cnt = part_count(n_glob, n_tasks, nparts, part);
                            GMSX_HIP(hipStreamWaitEvent(c.side[0], c.ev_fork, 0));
                            const int64_t waves = std::min<int64_t>(cnt, int64_t(cu) * 16);
                            hipLaunchKernelGGL((k_bk_wave<true, 1>), dim3(unsigned(waves)), dim3(64), 0, c.side[0], g->off, g->adj, g->newid, g->oldid, g->hoff, g->hadj, g->toff,
                                               g->tadj, g->dplus, v_out, n_glob, n_tasks, nparts, part, tqueue, static_cast<uint32_t *>(nullptr), 0ull, acc, sh);
                            ++launches;
                            tiny_beside = true;
                            round_join = [&]() -> int { return join_tiny(); };
                            return GMSX_OK;
                        };
                    } else if (int rc = launch_tiny(n_glob, n_tasks, beside != 0)) return rc;
                    tiny_done = true;
                }
                if (split_build >= 2)
                    hipLaunchKernelGGL(k_bk_block, dim3(unsigned(std::min<int64_t>(pspan[1] - pspan[0], int64_t(cu) * 8))), dim3(256), 0, s, g->off, g->adj, g->newid,
                                       g->oldid, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, v_out, int64_t(0), nparts, part, q0, pieces, pspan[0], pspan[1], aoff, roff,
                                       bqueue, sh);
                else
                    hipLaunchKernelGGL(k_bk_build, dim3(unsigned(std::min<int64_t>(q1 - q0, build_waves))), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid, g->hoff,
                                       g->hadj, g->toff, g->tadj, g->dplus, v_out, int64_t(0), nparts, part, q0, q1, aoff, roff, bqueue, map_scratch, map_words, sh);
                ++launches;
                if (int rc = join_tiny()) return rc;
                if (int rc = run_rounds()) return rc;
                // the next chunk starts from empty pools again
                sh.pool = pools[cur];
                sh.dir = dirs[cur];
                GMSX_HIP(hipMemsetAsync(sh.dir, 0xff, sh.dir_cap * 8, s));
                GMSX_HIP(hipMemsetAsync(sh.pool_head, 0, 6 * sizeof(unsigned long long), s));
                GMSX_HIP(hipMemsetAsync(sh.arena_head, 0, 8, s));
                q0 = q1;
            }
        }
        if (!tiny_done) {
            if (!sh.arena)
                if (int rc = alloc_arena(std::min<unsigned long long>(free_b / 8, 8ull << 30) / 4)) return rc;
            if (int rc = launch_tiny(n_glob, n_tasks, false)) return rc;
            if (int rc = run_rounds()) return rc;
        }
    } else {
        if (int rc = alloc_arena(std::min<unsigned long long>(free_b / 8, 8ull << 30) / 4)) return rc;
        int64_t lo = 0;
        while (lo < n_tasks) {
            const bool wide = lo < n_wide;
            const int64_t bin_end = wide ? n_wide : n_tasks;  // the wide tasks form their own bins
            const unsigned long long top = words[size_t(lo)];
            const bool lds = !wide && top <= (unsigned long long)kLdsSlabWords;
            int64_t hi = lo;
            while (hi < bin_end && (lds || words[size_t(hi)] * 4 > top)) ++hi;
            const int64_t cnt = part_count(lo, hi, nparts, part);
            if (cnt > 0) {
                GMSX_HIP(hipMemsetAsync(queue, 0, 8, s));
                if (lds) {
                    const int64_t waves = std::min<int64_t>(cnt, int64_t(cu) * 16);
                    hipLaunchKernelGGL((k_bk_wave<true, 1>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid, g->hoff,
                                       g->hadj, g->toff, g->tadj, g->dplus, v_out, lo, hi, nparts, part, queue,
                                       static_cast<uint32_t *>(nullptr), 0ull, acc, sh);
                } else {
                    const unsigned long long slab_w = (top + 3ull) & ~3ull;  // 16-byte aligned slabs (64-bit map slots)
                    const unsigned long long slab_bytes = slab_w * 4ull;
                    if (slab_bytes > budget_bytes) return GMSX_ERR_DEVICE_MEM;
                    const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({cnt, int64_t(cu) * 16, int64_t(budget_bytes / slab_bytes)}));
                    uint32_t *slabs = nullptr;
                    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&slabs), size_t(waves) * slab_bytes));
                    Guard g_slabs;  // freed on every path out of this bin, error returns included
                    g_slabs.p = slabs;
                    if (wide && wpl_wide == 2)
                        hipLaunchKernelGGL((k_bk_wave<false, 2>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid,
                                           g->hoff, g->hadj, g->toff, g->tadj, g->dplus, v_out, lo, hi, nparts, part, queue, slabs, slab_w, acc, sh);
                    else if (wide && wpl_wide == 4)
                        hipLaunchKernelGGL((k_bk_wave<false, 4>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid,
                                           g->hoff, g->hadj, g->toff, g->tadj, g->dplus, v_out, lo, hi, nparts, part, queue, slabs, slab_w, acc, sh);
                    else if (wide)
                        hipLaunchKernelGGL((k_bk_wave<false, 8>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid,
                                           g->hoff, g->hadj, g->toff, g->tadj, g->dplus, v_out, lo, hi, nparts, part, queue, slabs, slab_w, acc, sh);
                    else
                        hipLaunchKernelGGL((k_bk_wave<false, 1>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid,
                                           g->hoff, g->hadj, g->toff, g->tadj, g->dplus, v_out, lo, hi, nparts, part, queue, slabs, slab_w, acc, sh);
                    GMSX_HIP(hipStreamSynchronize(s));
                }
                ++launches;
            }
            lo = hi;
        }
        if (int rc = run_rounds()) return rc;
    }
    // ---- start vertices too wide for the register-resident search: one wave each, the whole search in its global slab
    if (n_giant > 0) {
        const int64_t cnt = part_count(0, n_giant, nparts, part);
        if (cnt > 0) {
            {   // the list was appended with atomics: sort it, so that every rank shards the same sequence
                int32_t *sorted = nullptr;
                GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&sorted), size_t(n_giant) * 4));
                Guard g_sorted;
                g_sorted.p = sorted;
                size_t sort_bytes = 0;
                GMSX_HIP(rocprim::radix_sort_keys(nullptr, sort_bytes, giant, sorted, size_t(n_giant), 0, 32, s));
                void *sort_tmp = nullptr;
                GMSX_HIP(hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 8));
                Guard g_st;
                g_st.p = sort_tmp;
                GMSX_HIP(rocprim::radix_sort_keys(sort_tmp, sort_bytes, giant, sorted, size_t(n_giant), 0, 32, s));
                GMSX_HIP(hipMemcpyAsync(giant, sorted, size_t(n_giant) * 4, hipMemcpyDeviceToDevice, s));
                GMSX_HIP(hipStreamSynchronize(s));
            }
            const unsigned long long slab_bytes = giant_slab_w * 4ull;
            if (slab_bytes > budget_bytes) return GMSX_ERR_DEVICE_MEM;
            const int64_t waves = std::max<int64_t>(1, std::min<int64_t>({cnt, int64_t(cu) * 8, int64_t(budget_bytes / slab_bytes)}));
            uint32_t *slabs = nullptr;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&slabs), size_t(waves) * slab_bytes));
            Guard g_slabs;
            g_slabs.p = slabs;
            GMSX_HIP(hipMemsetAsync(queue, 0, 8, s));
            hipLaunchKernelGGL((k_bk_wave<false, 0>), dim3(unsigned(waves)), dim3(64), 0, s, g->off, g->adj, g->newid, g->oldid, g->hoff, g->hadj,
                               g->toff, g->tadj, g->dplus, giant, int64_t(0), n_giant, nparts, part, queue, slabs, giant_slab_w, acc, sh);
            GMSX_HIP(hipStreamSynchronize(s));
            ++launches;
        }
    }
    GMSX_HIP(hipEventRecord(c.ev[2], s));
    GMSX_HIP(hipGetLastError());
    unsigned long long host[kCtl + 32];
    GMSX_HIP(hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    unsigned long long total = 0;
    for (int i = 0; i < 64; ++i) total += host[i * 16];
#ifdef GMSX_BK_STATS
    {
        unsigned long long nodes = 0, zero = 0;
        (void)hipMemcpyFromSymbol(&nodes, HIP_SYMBOL(g_bk_nodes), 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bk_nodes), &zero, 8);
        std::fprintf(stderr, "[gmsx bk] search nodes %llu, maximal cliques %llu\n", nodes, total);
        unsigned long long hist[32], zeros[32] = {};
        (void)hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_bk_hist), sizeof(hist));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bk_hist), zeros, sizeof(zeros));
        std::fprintf(stderr, "[gmsx bk] entered nodes by c <= 32/64/128/256/512/1024/2048/more:");
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %llu", hist[i]);
        std::fprintf(stderr, "\n[gmsx bk] entered nodes by |P| = 0 / 1 / 2-6 / 7-16 / 17-32 / 33-64 / 65-256 / more:");
        for (int i = 8; i < 16; ++i) std::fprintf(stderr, " %llu", hist[i]);
        std::fprintf(stderr, "\n[gmsx bk] entered with Xf %llu, one-candidate nodes %llu, branch steps %llu, leaf fast paths %llu, one-candidate children %llu, pushes %llu (Xf words %llu), "
                     "pivot-scored nodes %llu (sum |P u Xc| %llu), deepest level %llu\n", hist[16], hist[25], hist[17], hist[18], hist[19], hist[20], hist[24], hist[21], hist[22], hist[23]);
        {
            unsigned long long gs[20], gz[20] = {};
            (void)hipMemcpyFromSymbol(gs, HIP_SYMBOL(g_bkg_stat), sizeof(gs));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bkg_stat), gz, sizeof(gz));
            std::fprintf(stderr, "[gmsx bk] k_bk_resume4: trips %llu (with a fetch %llu, a memory walk of Xf %llu, pivot scoring %llu); group steps PIVOT %llu PIVOT+BRANCH %llu BRANCH %llu "
                         "CHILD1 %llu POP %llu, idle group-trips %llu; pushes %llu, of them without a saved level %llu\n",
                         gs[0], gs[1], gs[2], gs[3], gs[4], gs[5], gs[6], gs[7], gs[8], gs[9], gs[10], gs[11]);
        }
        std::fprintf(stderr, "[gmsx bk] Xf checks of the fast paths %llu (%llu words), non-zero words of the pushed child Xf %llu; wave cycles: search %llu, of them Xf loops %llu, pivot scoring %llu\n",
                     hist[26], hist[27], hist[31], hist[29], hist[28], hist[30]);
    }
    for (int b = 0; b < 7; ++b) {
        unsigned long long nrec = 0, xws = 0;
        for (int i = 0; i < 64; ++i) { nrec += host[i * 16 + 1 + b]; xws += host[i * 16 + 8 + b]; }
        std::fprintf(stderr, "[gmsx bk] resumed records with c <= %d: %llu, mean Xf words %.1f\n", b < 6 ? 32 << b : 1 << 30, nrec, nrec ? double(xws) / double(nrec) : 0.0);
    }
#endif
    if (part == 0) total += host[kCtl];  // isolated vertices, counted once
    *out = total;
    if (st) {
        float ms_setup = 0.f, ms = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms_setup, c.ev[0], c.ev[1]));
        GMSX_HIP(hipEventElapsedTime(&ms, c.ev[1], c.ev[2]));
        // ALGORITHMIC bytes of this formulation (no cache assumed), outside the timed region: the rows the builds walk + Cadj | XT of the start vertices
        // built in the arena, written once + one Cadj row (cw words) per search-tree node — the operand of the reference's cand.intersect(N(q))
        unsigned long long node_words = 0;
        for (int i = 0; i < 64; ++i) node_words += host[i * 16 + 15];
        unsigned long long build_bytes = 0;
        if (n_tasks > 0) {
            GMSX_HIP(hipMemsetAsync(acc + kCtl + 30, 0, 8, s));
            hipLaunchKernelGGL(k_stat_bk_bytes, dim3(unsigned(cu * 8)), dim3(256), 0, s, n_tasks, nparts, part, v_out, g->off, g->adj, g->newid, g->oldid, g->hoff, g->toff,
                               acc + kCtl + 30);
            GMSX_HIP(hipMemcpyAsync(&build_bytes, acc + kCtl + 30, 8, hipMemcpyDeviceToHost, s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        const uint64_t alg = uint64_t(build_bytes) + 4ull * uint64_t(need_total) + 4ull * uint64_t(node_words);
        *st = gmsx_stats{double(ms), double(ms_setup), uint64_t(part_count(0, n, nparts, part)), 0, uint64_t(rounds), launches, 0, alg};
    }
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_bk_partial(const gmsx_graph *g, const int32_t *rank, int part, int nparts, uint64_t *partial, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !partial || nparts < 1 || part < 0 || part >= nparts) return GMSX_ERR_INVALID;
        // `rank` is what the reference's drivers hand from the preprocessing step to mceBench(graph, ordering).  The number of
        // maximal cliques does not depend on it and the device splits by its own degree rank, so it is validated
        // (a permutation of 0..n-1, as every rank-format ordering is) and otherwise not needed.
        // No exception may cross the C ABI: the scratch bitmap is a nothrow allocation.  Validated on EVERY call (an O(n) pass next to an
        // enumeration): a memo keyed on the pointer would accept an array that was changed, or another one at the same address.
        if (rank) {
            const int64_t n = g->n;
            const size_t words = size_t((n + 63) / 64 + 1);
            uint64_t *seen = new (std::nothrow) uint64_t[words]();
            if (!seen) return GMSX_ERR_NOMEM;
            bool ok = true;
            for (int64_t i = 0; i < n && ok; ++i) {
                const int64_t r = rank[i];
                if (r < 0 || r >= n) { ok = false; break; }
                uint64_t &w = seen[size_t(r >> 6)];
                const uint64_t bit = 1ull << (r & 63);
                if (w & bit) ok = false;
                w |= bit;
            }
            delete[] seen;
            if (!ok) return GMSX_ERR_INVALID;
        }
        if (int rc = ensure_init()) return rc;
        return bk_partial(g, part, nparts, partial, stats);
    });
}

int gmsx_bk_count(const gmsx_graph *g, const int32_t *rank, uint64_t *maximal_cliques, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        return gmsx_bk_partial(g, rank, 0, 1, maximal_cliques, stats);
    });
}

}  // extern "C"
