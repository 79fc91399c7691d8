// Device-resident graph of libgmsx (gfx950).  Shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gmsx.h"
#include "gmsx_internal.hpp"  // gmsx::guard: no exception crosses the C ABI

// The A/B switches of the kernel sources ("wrong counts" builds that compile a part of a kernel out to time the rest, GMSX_TC_ONLY)
// exist in DEVELOPMENT builds only: `make` never defines GMSX_DEV_HOOKS, tools/ab_lib.sh does.  A shipped libgmsx.so cannot be
// steered into a miscount — neither by a macro that slipped into EXTRA nor by the environment (tests/test_capi_symbols.py checks the strings).
#if !defined(GMSX_DEV_HOOKS) &&                                                                                                               \
    (defined(GMSX_KC_NO_PROBE) || defined(GMSX_KC_NO_HITS) || defined(GMSX_KC_NO_TAIL) || defined(GMSX_KC_NO_INNER) || defined(GMSX_KC_CELLS_ONLY) ||   \
     defined(GMSX_KC_NO_ROWS) || defined(GMSX_KC_BUILD_ONLY) || defined(GMSX_KC_NO_SLAB_CALL) || defined(GMSX_KC_NO_PAIRS) || defined(GMSX_KC_SLAB_ROW_SCAN) || \
     defined(GMSX_BK_AB) || defined(GMSX_BK_NO_HIT_ATOMICS) || defined(GMSX_BK_STATS) || defined(GMSX_TC_NO_PROBE) || defined(GMSX_TC_STAGING_ONLY))
#error "A/B switches need -DGMSX_DEV_HOOKS (tools/ab_lib.sh sets it); the default build of libgmsx.so carries none of them"
#endif

// Layout in HBM (all arrays hipMalloc'd once at upload, read-only afterwards):
//
//   off  int64[n+1], adj int32[nnz]   the caller's symmetric CSR, rows ascending (the reference's CSRGraph rows)
//
//   Internal vertex numbering ("rank ids"): vertices sorted by DEcreasing (degree, id); rank id 0 is the biggest
//   hub.  newid[old] / oldid[rank] convert.  Edges are oriented from the lower- to the higher-degree endpoint,
//   i.e. towards the SMALLER rank id:  N+(u) = { v in N(u) : rank(v) < rank(u) }, so every undirected edge appears
//   exactly once and d+ = O(sqrt(m)).
//
//   The oriented rows are stored Roaring-style in two containers per vertex, indexed by rank id:
//     hub part   hoff int64[n+1], hadj uint16[...]   targets with rank id < kHub (=65535), 2 bytes each; every row
//                                                     is padded to an even count with 0xFFFF (never a valid target)
//     tail part  toff int64[n+1], tadj int32[...]    targets with rank id >= kHub, 4 bytes each
//   On power-law graphs >85 % of all entries and >95 % of the streamed ids are hub entries.
//     bitset part bmoff int64[H+1], bmpool uint32[...]  every hub row (rank id < H = min(n, kHub)) additionally as a bitmap over [0, v)
//                                                     (Roaring's bitset container, <= 268 MB in total): the dense stream-row form of the
//                                                     triangle kernel (AND + popcount, 128 ids per 16-byte unit) and the edge test of the
//                                                     k-clique recursion.
//
//   dplus int32[n]                     true out-degree (hub + tail) per rank id
//   order int32[n]                     rank ids by decreasing d+ (work-sorted launch order, heavy first);
// WORK ITEM of the triangle kernels: up to kTaskChunk consecutive entries of ONE pivot's hub-entry list (htask) or tail-entry list
// (ttask).  A self-contained 64-byte record — everything a persistent workgroup needs to stage the item without a dependent lookup:
// bc = first entry (40 bits) | entries << 40; cont = where the pivot's own container part lies (first id in hadj / tadj, 40 bits) | its
// ids << 40; pivot = rank id whose row is staged in LDS; pos = the pivot's position in `order` (what the shard rule of a multi-GPU run
// is evaluated on).  (The lists are class-sorted — form, then length step: an item is a few runs of equally formed rows, which is all the
// kernels' one step stream per item needs.)
struct __attribute__((aligned(64))) gmsx_tc_item {
    uint64_t bc;
    uint64_t cont;
    int32_t pivot;
    int32_t pos;
    uint32_t kind;       // 0 = hub item, 1 = tail item
    uint16_t fbeg[4];    // fbeg[f] = entries of the item whose form is below f (the lists are sorted by form first): where form f begins
    uint32_t reserved[7];
};
static_assert(sizeof(gmsx_tc_item) == 64, "work item record = one 64-byte line");
// TASK LIST: the stream-row descriptors of all receivers, 6 bytes per entry in two arrays (round 5; 8-byte words before: −1.9 GB at scale
// 26).  An entry is the 64-bit descriptor first unit << 24 | form << 22 | units squeezed into 48 bits — lo = (first unit & 0xffff) << 16 |
// form | units, hi = first unit >> 16 — and expanded again when a workgroup stages its item in LDS (two coalesced loads per entry instead of
// one; the scan path reads the 8-byte LDS copy as before).  Hub lists: 2 form bits + 14 unit bits (a hub row is <= 65 535 ids = 8 192
// list units, and another form is only taken when it is shorter); tail lists: 1 form bit (list / 16-bit delta) + 15 unit bits (131 068
// ids in the 32-bit form).  The build refuses a graph whose rows exceed the fields (k_task_limits: GMSX_ERR_UNSUPPORTED; pools of >= 2^32 units: GMSX_ERR_DEVICE_MEM, i.e. built a share of the pivots at a time).
constexpr uint32_t kTaskHubUnitsMax = (1u << 14) - 1, kTaskTailUnitsMax = (1u << 15) - 1;
struct TaskList {
    uint32_t *lo = nullptr;
    uint16_t *hi = nullptr;
    bool tail = false;
    __host__ __device__ TaskList at(int64_t first) const { return TaskList{lo + first, hi + first, tail}; }
    __device__ __forceinline__ unsigned long long get(int64_t i) const {
        const uint32_t l = lo[i], h = hi[i];
        const unsigned long long first = ((unsigned long long)h << 16) | (l >> 16);
        const uint32_t form = tail ? ((l >> 15) & 1u) << 1 : (l >> 14) & 3u, units = tail ? l & kTaskTailUnitsMax : l & kTaskHubUnitsMax;
        return (first << 24) | ((unsigned long long)form << 22) | units;
    }
    __device__ __forceinline__ void put(int64_t i, unsigned long long d) const {
        const uint32_t first = uint32_t(d >> 24), form = (uint32_t(d) >> 22) & 3u, units = uint32_t(d) & 0x3fffffu;
        lo[i] = (first << 16) | (tail ? (form >> 1) << 15 : form << 14) | units;
        hi[i] = uint16_t(first >> 16);
    }
};
struct gmsx_graph {
    int64_t n = 0, nnz = 0, m = 0;
    int64_t *off = nullptr;
    int32_t *adj = nullptr;
    int32_t *newid = nullptr;
    int32_t *oldid = nullptr;
    int64_t *hoff = nullptr;
    uint16_t *hadj = nullptr;
    int64_t *toff = nullptr;
    int32_t *tadj = nullptr;
    int64_t *bmoff = nullptr;   // [dense_limit + 1] word offsets into bmpool (multiples of 4); equal neighbours = not dense
    uint32_t *bmpool = nullptr; // bitset containers of the dense hub rows
    int32_t dense_limit = 0;    // = min(n, hub limit): hub rank ids; their rows have a bitset AND only hub entries
    int32_t bitset_limit = 0;   // = dense_limit: every rank id below it has a bitset container over [0, v) in bmpool
    // STREAM ROWS (triangle kernels): the hub part of every row once more, in the form that is cheapest to stream,
    // every row a whole number of 16-byte units (padded with neutral fillers, so the scanners need no tail handling) at a 16-byte
    // aligned offset of ONE pool; srow[v] packs (offset / 16) << 24 | form << 22 | units into 8 bytes = one load per row fetch.
    //   form 0  16-bit list      8 ids per unit, filler 0xFFFF (never in a pivot bitmap)
    //   form 1  bitset           128 ids per unit over [0, v), filler 0 (hub rows only, when smaller than the list)
    //   form 2  byte-delta       unit = 16-bit base id + count byte + 13 gap bytes; a gap above 255 ends the unit early.
    //                            14 ids per unit when gaps < 256: 1.14 B/id against 2 B/id
    //   form 3  12-bit gaps      unit = 16-bit base id + 4-bit count + nine 12-bit gaps; a gap above 4095 ends the unit early.
    //                            10 ids per unit = 1.6 B/id: the sparse rows (gaps of 256 … 4095) that would otherwise stay lists
    unsigned long long *srow = nullptr;  // [n]
    unsigned long long *srow2 = nullptr; // [n] second descriptor of a HYBRID row (0 otherwise): srow = prefix bitmap over [0, B), srow2 = the ids from B on
    int32_t *ksplit = nullptr;           // [n] ids of the row below B (0 = not hybrid)
    uint32_t *spool = nullptr;           // 16-byte units
    int64_t spool_units = 0;
    // … and the TAIL part of every row the same way (trow / tpool): form 0 = 32-bit ids, 4 per unit, filler -2 (never a key of a
    // pivot's tail set); form 2 = 16-bit delta: 32-bit base, count, five 16-bit gaps — 6 ids per unit, 2.67 B/id against 4 B/id
    // TASK LISTS (tc.hip).  |N+(u) ∩ N+(v)| of an oriented edge (u,v) can be counted with either endpoint as the pivot (its row as bitmap +
    // tail set in LDS) and the other one streamed; the pass streams the SMALLER row (fewer 16-byte units).  So every receiving vertex w
    // owns two lists of stream-row descriptors (6 bytes each: TaskList) — HUB entries (rows probed against w's bitmap) and TAIL entries (rows probed against
    // w's tail set) — one descriptor per non-empty row part of every edge it is the pivot of: members v of its own row whose rows are the
    // smaller ones ("forward"), in-neighbours u that hand their edge over because the part of their row below w is smaller ("reverse", cut
    // at w's id; only towards heavy w), and the 64-unit chunks of w's inline rows.  Each list is laid out CLASS BY CLASS — class = (form,
    // ceil(log2 units)) — by counting and bucketing at build time (no sort), so the kernels run compile-time-shaped loops over runs of
    // equally formed, similarly long rows: 4-, 8- or 16-lane groups for rows of <= 4, <= 8, more units.  The order INSIDE a class is the
    // arrival order of atomic cursors — it differs from process to process, which is why a multi-GPU shard is a set of whole PIVOTS.
    TaskList htask{nullptr, nullptr, false};  // hub entries of all receivers, receiver by receiver (in `order`), class by class
    TaskList ttask{nullptr, nullptr, true};   // tail entries likewise
    int64_t htask_entries = 0, ttask_entries = 0;
    struct gmsx_tc_item *hitem = nullptr, *titem = nullptr;  // work items over htask / ttask
    int64_t hitems = 0, titems = 0;
    int64_t inline_hentries = 0, inline_tentries = 0;  // of which: chunks of inline rows
    // gmsx_tc_partial(part, nparts) on a FULL upload: the work items of that shard, compacted on demand (cached for the last
    // (part, nparts)), so that a shard walks its own items only
    mutable struct gmsx_tc_item *shard_hitem = nullptr, *shard_titem = nullptr;
    mutable int64_t shard_hitems = 0, shard_titems = 0;
    mutable int shard_idx_part = -1, shard_idx_nparts = -1;
    int32_t *tunits = nullptr;           // [heavy pivots, by position in `order`] oriented edges whose entries live at this pivot (forward + reverse): the bookkeeping of gmsx_stats.units
    int64_t task_reverse = 0;            // edges handed over to the other endpoint
    // LIGHT EDGES (round 4, k_tc_light): the oriented edges (u, v) no work item covers — u light (2 <= d+ < kHeavy), v a FAR LIGHT member of it
    // (rank id >= inline_limit, d+ < kHeavy) — as self-contained 32-byte records: where the hub part and the tail part of u's row and of
    // v's row lie in hadj / tadj (first id, 40 bits | ids << 40; of u's tail part only the ids in front of v: nothing else can be in N+(v)).
    // Both rows are short (< 64 ids), so the edge is ONE all-pairs comparison in registers: no LDS, no per-pivot state, every edge
    // independent.  Edge e of the (deterministic) list belongs to shard shard_of(e, nparts) — the pivots' rule; a sharded upload keeps its own at slot e / nparts.
    uint4 *ledge = nullptr;
    int64_t n_ledge = 0, ledge_total = 0;
    // HOT WINDOWS (round 4): the hub-entry list of a receiver is laid out phase by phase — first the entries whose stream row starts in
    // window 0 of the pool ([0, tc_window_units) 16-byte units: the rows of the biggest hubs, the most often streamed bytes of the graph),
    // then window 1 … and last everything else — and the work items of phase 0 of ALL receivers come first, then phase 1 …: while a phase
    // runs, the rows streamed fit the 4 MB L2 of every XCD instead of passing through it.  0 windows = one phase (rounds 1-3 order).
    int tc_hot_windows = 0;
    uint32_t tc_window_units = 1;
    int tc_hot_min = 0;                  // a receiver with fewer entries than this in a window streams them with its next phase
    int inline_first = 64;               // a HEAVY pivot hands the edges to its first inline_first members (<= 64) over inline rows as well
    int32_t inline_limit = 0;            // light pivots hand their edges to members of rank id < inline_limit (and to heavy ones) as INLINE ROWS
    int64_t inline_units = 0;            // 16-byte units of all inline rows (inside spool / tpool)
    unsigned long long *trow = nullptr;
    uint32_t *tpool = nullptr;
    int64_t tpool_units = 0;
    int32_t *tsplit = nullptr;  // int32[n]: position in the tail row of the first target >= inline_limit (= tail length if none)
    int64_t dense_rows = 0, bmpool_words = 0;
    bool rows_sorted = false;   // both containers of every row ascending (always: sorted in vertex ranges at upload)
    int32_t *dplus = nullptr;
    int32_t *order = nullptr;
    int64_t hub_entries = 0, tail_entries = 0;
    int32_t max_dplus = 0;
    int32_t max_deg = 0;
    // k-clique REVERSE ROWS (round 6; kclique.hip, ensure_kc_reverse — built at the first k-clique call, like the triangle-count task lists).  The BUILD
    // needs rows[i] = N+(v_i) ∩ N+(u) for every member v_i of every pivot u.  Forward it streams N+(v_i) against u's bitmap; but nothing of N+(v_i)
    // above v_i's predecessors can hit, and for a HUB member w = v_i the same bits come from the other side: the i members below w — a prefix of u's
    // own hub list, 2 i bytes — probed against w's bitset container.  An edge is handed to its receiver w when that is at least twice cheaper; the
    // receivers then run first (k_kc_reverse: w's bitset staged in LDS, the prefixes of all its in-neighbours streamed through it, hit bits written
    // as finished matrix rows into an arena) and the pivots copy those rows instead of streaming the member.
    mutable bool kc_rev_tried = false;       // built (or found not worth building) already
    mutable uint32_t *kc_rel = nullptr;      // [hoff[n]] per hub-entry position: word offset of that member's row inside its pivot's arena span; ~0u = streamed forward
    mutable int64_t *kc_aoff = nullptr;      // [n + 1] word offset of a pivot's span in the arena (by rank id)
    mutable uint32_t *kc_arena = nullptr;    // the reverse rows of one call
    mutable ulonglong2 *kc_rec = nullptr;    // [kc_recs] one record per reverse edge, receiver by receiver: x = first id of the pivot's hub list in hadj (40 bits) | i << 40,
                                             //           y = arena word of the row (36 bits) | position of the pivot in `order` << 36 (what a shard is decided on)
    mutable uint4 *kc_item = nullptr;        // [kc_items] work items: x = receiver, y = entries, z | w << 32 = first record
    // TAIL receivers (round 6b): a tail member w (rank id >= 65 535: no bitset container) takes its edges too — k_kc_reverse_tail builds w's bitmap and tail set
    // in LDS from its two lists and streams the pivot's whole hub list and its tail members below w through them.  (With the matrix-core count the forward
    // rows of the tail members were 206 of the 470 ms of a k = 4 call at scale 26.)
    mutable uint32_t *kc_relt = nullptr;     // [toff[n]] per tail-entry position, as kc_rel
    mutable ulonglong2 *kc_rect = nullptr;   // [2 kc_recst] two 16-byte halves per record: {hadj offset | i << 40, arena word | position << 36}, {tadj offset | hc << 40, 0}
    mutable uint4 *kc_itemt = nullptr;       // [kc_itemst] work items of tail receivers
    mutable int64_t kc_recst = 0, kc_itemst = 0;
    mutable int64_t kc_recs = 0, kc_items = 0, kc_arena_words = 0, kc_rev_bytes = 0;
    mutable double kc_rev_build_ms = 0.0;    // what building the lists took (reported once, in gmsx_stats.setup_ms of the call that built them)
    unsigned long long *scratch = nullptr;  // device: a few u64 accumulators
    unsigned long long *acc = nullptr;      // device: 64 spread u64 accumulators (128 B apart) + a few control words, reused by every call
    // host-side memo of read-only facts about the immutable graph (filled lazily; handles are single-threaded)
    mutable int32_t ge_thr[40] = {0};
    mutable int64_t ge_cnt[40] = {0};
    mutable int ge_used = 0;
    mutable int stats_part = -1, stats_nparts = -1;  // shard whose TC bookkeeping (units, probes) is cached below
    mutable uint64_t stats_units = 0, stats_probes = 0, stats_bytes = 0;
    // the triangle-count containers (stream rows, inline rows, task lists …) are built on demand: ensure_tc()
    int shard_part = 0, shard_nparts = 1;   // gmsx_graph_upload_shard: the triangle-count containers hold this rank's pivots only
    // FALLBACK when the triangle-count containers of the whole graph do not fit the device: they are built for 1/tc_passes of the pivots at
    // a time (the sharded-upload machinery, shards = passes) and a call walks the passes, rebuilding between them — slower, never refused
    int tc_passes = 1;
    int64_t tc_limit_bytes = 0, tc_base_bytes = 0;  // GMSX_TC_MEM_LIMIT_MB (test hook): budget of the containers; device_bytes before them
    bool tc_building = false;
    bool tc_ready = false;
    int64_t tc_bytes = 0;                   // their share of device_bytes
    int hub_limit = 0;                      // hub id range this graph was built with (kHub unless the test hook shrank it)
    uint32_t upload_flags = 0;
    uint64_t alg_elements = 0;              // Σ_{u<v}(d_u+d_v), computed on the device at upload
    int64_t device_bytes = 0;
};

namespace gmsx {

static constexpr int kHub = 65535;         // rank ids below this live in the 16-bit hub containers
static constexpr int kBitmapWords = 2048;  // 65536-bit LDS bitmap over the hub id range
static constexpr int kAccWords = 64 * 16 + 16;
static constexpr int kFormList = 0, kFormBitset = 1, kFormDelta = 2, kFormGap12 = 3;
static constexpr int kPoolSlack = 64;    // 16-byte units of padding behind spool / tpool: a lane group loads up to 15 units past the end of a row (tc.hip, scan_run)
static constexpr int kTaskChunk = 512;   // entries per work item (4 KB of descriptors: two such buffers per workgroup, the next item's arriving while this one is scanned)
// entry classes: hub = form * kLenClasses + length class, tail = kHubClasses + (delta ? kLenClasses : 0) + length class.  Length classes
// in steps of ~sqrt(2): <= 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, more units — the groups of a wave work on rows of one class side
// by side, and the wave is done when its longest row is.
static constexpr int kLenClasses = 12;
static constexpr int kHubClasses = 4 * kLenClasses, kTailClasses = 2 * kLenClasses;
static constexpr int kMaxHotWindows = 8;
// hub classes exist once per PHASE of the hub lists (gmsx_graph::tc_hot_windows): phase = the window of the pool the entry's row starts in,
// the last phase everything behind the windows
struct TcClasses {
    int hot_windows;        // 0 … kMaxHotWindows
    uint32_t window_units;  // 16-byte units per window (> 0)
    int hot_min;
    __host__ __device__ int hub_phases() const { return hot_windows + 1; }
    __host__ __device__ int tail_base() const { return hub_phases() * kHubClasses; }
    __host__ __device__ int count() const { return tail_base() + kTailClasses; }
    __host__ __device__ int phase(unsigned long long d) const {
        if (hot_windows <= 0) return 0;
        const unsigned long long w = (d >> 24) / window_units;
        return w < (unsigned long long)hot_windows ? int(w) : hot_windows;
    }
};
__host__ __device__ inline int length_class(uint32_t u) {
    return u <= 16 ? (u <= 4 ? 0 : u <= 8 ? 1 : u <= 12 ? 2 : 3) : u <= 64 ? (u <= 24 ? 4 : u <= 32 ? 5 : u <= 48 ? 6 : 7) : (u <= 96 ? 8 : u <= 128 ? 9 : u <= 192 ? 10 : 11);
}
__host__ __device__ inline int hub_class(const TcClasses &cc, unsigned long long d) {
    return cc.phase(d) * kHubClasses + int((uint32_t(d) >> 22) & 3u) * kLenClasses + length_class(uint32_t(d) & 0x3fffffu);
}
__host__ __device__ inline int tail_class(const TcClasses &cc, unsigned long long d) {
    return cc.tail_base() + (((uint32_t(d) >> 22) & 3u) == 2u ? kLenClasses : 0) + length_class(uint32_t(d) & 0x3fffffu);
}
// the part [*b, *e) of a receiver's hub list (len entries; c = its class offsets) that phase ph's work items cover: its window's entries
// plus those of the windows in front that were too few (< hot_min) to be worth items of their own; the last phase takes what is left
__host__ __device__ inline void hub_phase_range(const TcClasses &cc, const uint32_t *c, int64_t len, int ph, int64_t *b, int64_t *e) {
    int64_t start = 0;
    for (int q = 0;; ++q) {
        const int64_t end = q < cc.hot_windows ? int64_t(c[(q + 1) * kHubClasses]) : len;
        const bool own = q == cc.hot_windows || end - start >= cc.hot_min;
        if (q == ph) {
            *b = start;
            *e = own ? end : start;
            return;
        }
        if (own) start = end;
    }
}
// which rank of a multi-GPU run owns the pivot at position `pos` of `order`: stripes of nparts positions, every other one reversed (the
// order is by decreasing d+, so plain striding would always hand the costlier pivot of a stripe to the lower rank)
__host__ __device__ inline int shard_of(int64_t pos, int nparts) {
    const int j = int(pos % nparts);
    return ((pos / nparts) & 1) ? nparts - 1 - j : j;
}
static constexpr int kHeavy = 64;        // d+ from which a pivot runs on the workgroup kernel
static constexpr int kDeltaIds = 14;  // ids per full 16-byte delta unit  // size of gmsx_graph::acc in u64

// words of the bitset container of hub rank id v (covers ids [0, v)), rounded to 16 bytes
__host__ __device__ inline int64_t bitset_words(int32_t v) { return ((int64_t(v) + 31) / 32 + 3) & ~int64_t(3); }

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;       // stream all launches go to
    hipStream_t own_stream = nullptr;   // created by gmsx_init
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t side[2] = {nullptr, nullptr};  // co-scheduling: the latency-bound light-pivot kernels run beside the bandwidth-bound one
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int compute_units = 0;
};
Ctx &ctx();
int ensure_init();
// number of vertices with d+ >= threshold (= position in `order` where d+ drops below it)
int kclique_vertex_counts(const gmsx_graph *g, unsigned long long *d_counts, gmsx_stats *st);  // kclique.hip
int count_dplus_ge(const gmsx_graph *g, int32_t threshold, int64_t *out);
int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t count, hipStream_t s);  // device_graph.hip (rocPRIM)
// builds the triangle-count containers of the (otherwise immutable) graph if they are not there yet (device_graph.hip)
int ensure_tc(const gmsx_graph *g);
// … for one shard / pass (frees and rebuilds when another one is resident)
int ensure_tc_shard(const gmsx_graph *g, int part, int nparts);
// fills g->shard_hitem / shard_titem for (part, nparts) (device_graph.hip)
int tc_shard_items(const gmsx_graph *g, int part, int nparts);
// counts[u] = Σ_{v∈N(u)} |N(u)∩N(v)| into a zeroed device array (pairs.hip); shared by the per-vertex count and the TC ordering
int tc_vertex_counts_device(const gmsx_graph *g, unsigned long long *d_counts, gmsx_stats *st);
// GMSX_TC_FULL: every edge u<v intersects the FULL rows (pairs.hip); returns the un-divided sum of the shard
int tc_full_partial(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st);

// A value every lane of the wave loaded from the same address is uniform, but the compiler cannot know: pinning it with
// v_readfirstlane moves it (and every offset / pointer derived from it) from vector to scalar registers.
__device__ __forceinline__ int uni32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }
__device__ __forceinline__ int64_t uni64(int64_t x) {
    const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(x)))));
    const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(x) >> 32))));
    return int64_t((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ unsigned long long uni64(unsigned long long x) { return (unsigned long long)uni64(int64_t(x)); }

#define GMSX_HIP(call)                                                    \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) {                                           \
            (void)hipGetLastError();                                      \
            return e_ == hipErrorOutOfMemory ? GMSX_ERR_DEVICE_MEM : GMSX_ERR_KERNEL; \
        }                                                                 \
    } while (0)

}  // namespace gmsx
