// Triangle counting on gfx950: the device replacement for
//   GMS::TriangleCount::Par::count_total   (gms/algorithms/set_based/triangle_count/parallel/total.h:7-24)
// whose inner operator is Set::intersect_count (representations/sets/sorted_set.h:176-182 ->
// sorted_set_operations.h:44-71; roaring_set.h:144-152 -> roaring.c:10090-10118 for the RoaringSet flavour).
//
// Formulation (GMSX_TC_ORIENTED).  The reference evaluates, for every undirected edge {u,v}, one intersect_count
// on the full rows and divides the sum by 3.  Here every undirected edge is still exactly one intersect_count, but
// on the degree-oriented rows:  T = Σ_(u,v) |N+(u) ∩ N+(v)| over the oriented edges (v ∈ N+(u), rank(v) < rank(u)), which meets
// every triangle once, so the returned integer is the same.
//
// One intersect_count = one row staged in LDS (the PIVOT), the other STREAMED through it.  Either endpoint can be the pivot, and
// only the ids of N+(u) below v can be in N+(v).  The upload decides per edge (device_graph.hip, task lists):
//   * u heavy (d+ >= 64): the edge stays with u — v's stream rows are named in u's task list — unless v is heavy too and the
//     part of u's rows below v is shorter than v's rows: then u's rows, CUT at v, are named in v's list;
//   * u light: the members of u below v are COPIED into v's inline rows (v heavy or a popular target, rank id < inline_limit) —
//     a 20-byte row behind a pointer would cost a 128-byte line per fetch, inline it is streamed; only far light members v stay
//     with u and the light-pivot kernel.
// Kernel shape, Roaring-style sets:
//   1. every receiving vertex owns a HUB-entry list and a TAIL-entry list of 8-byte stream-row descriptors, laid out class by class
//      (form x length step) at build time; a WORK ITEM is <= 1024 consecutive entries of one list, one workgroup each.  k_tc_block runs
//      the hub items — the pivot's hub part (rank ids < 65535) staged as a 65536-bit BITMAP (8 KB of LDS) — and k_tc_tail the tail items —
//      the pivot's tail part as an open-addressing hash set fronted by a filter bitmap, both sized per pivot;
//   2. the rows the entries name are streamed from HBM with coalesced 16-byte loads as STREAM ROWS (device_graph.hpp): whole
//      16-byte units of the cheapest form per row — bitset (AND + popcount, 128 ids per unit), 16-bit list (8 ids), byte-delta (base +
//      count + 13 gaps: 14 ids), 12-bit gaps (10 ids; off) — for the hub part, 32-bit ids or 16-bit delta units (6 ids) for the tail part.
//      The workgroup copies its descriptors to LDS, finds the runs (form x {<= 4, <= 8, more units}) and executes one compile-time-shaped
//      loop per run: groups of 4 / 8 / 16 lanes, a row per group, the first unit of the group's next row already loading;
//   3. every streamed hub id is one LDS word read + bit test (no collisions, no branches); a tail id is one filter-bit test and,
//      for the few that pass, a table probe; hits are counted per lane, reduced per workgroup, added to one of 64 spread u64
//      accumulators (one atomic per workgroup);
//   4. k_tc_light — the edges between two light vertices that no work item covers: one 16-lane group per edge, all-pairs in registers.
//   The three kernels run side by side on three streams (hub items: bandwidth + VALU; tail items: short rows; light pivots: latency).
// No MFMA: integer/indexing work bounded by row streaming (HBM) and VALU issue of the decode + probe sequence.
#include "device_graph.hpp"

#include <algorithm>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace gmsx {

static constexpr int kAccSlots = 64;   // spread accumulators, 128 B apart
static constexpr int kAccStride = 16;  // in u64

// 4-byte-aligned 16-byte load: container rows start at arbitrary dword offsets; the hardware needs dword alignment only
struct __attribute__((packed, aligned(4))) u4u { uint32_t x, y, z, w; };

__device__ __forceinline__ int64_t readlane64(int64_t x, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(uint64_t(x)), l);
    const uint32_t hi = __builtin_amdgcn_readlane(uint32_t(uint64_t(x) >> 32), l);
    return int64_t((uint64_t(hi) << 32) | lo);
}

// ---- hub side: 65536-bit bitmap in LDS ---------------------------------------------------------------
__device__ __forceinline__ uint32_t bit_lo(const uint32_t *bm, uint32_t p) {  // id = p & 0xffff
    const uint32_t word = bm[(p >> 5) & 0x7ffu];
    return __builtin_amdgcn_ubfe(word, p, 1u);  // v_bfe_u32 uses offset[4:0] only: no "& 31" needed
}
__device__ __forceinline__ uint32_t bit_hi(const uint32_t *bm, uint32_t p) {  // id = p >> 16
    const uint32_t q = p >> 16;
    const uint32_t word = bm[q >> 5];
    return __builtin_amdgcn_ubfe(word, q & 31u, 1u);
}
__device__ __forceinline__ uint32_t hub_hits8(const uint32_t *bm, u4u p) {
    return bit_lo(bm, p.x) + bit_hi(bm, p.x) + bit_lo(bm, p.y) + bit_hi(bm, p.y) + bit_lo(bm, p.z) + bit_hi(bm, p.z) +
           bit_lo(bm, p.w) + bit_hi(bm, p.w);
}

// ---- stream rows (device_graph.hpp): whole 16-byte units at 16-byte aligned offsets, three forms, no tail handling -------------
// one unit of the byte-delta form: 16-bit base id, count byte (1 … 14 ids), 13 gap bytes.  All fourteen running ids first (a chain
// of byte adds), THEN the fourteen LDS probes back to back — written the other way round the compiler waits for every probe before
// it issues the next — and the count cuts the unused slots off.  A running id never exceeds 65535 + 13*255: the probe of an unused
// slot may read past the 8 KB bitmap into the workgroup's next LDS array, which is harmless (its bit is masked out).
__device__ __forceinline__ uint32_t delta_unit_hits(const uint32_t *bm, uint4 p) {
    uint32_t id[14];
    id[0] = p.x & 0xffffu;
    id[1] = id[0] + (p.x >> 24);
    id[2] = id[1] + (p.y & 0xffu);
    id[3] = id[2] + ((p.y >> 8) & 0xffu);
    id[4] = id[3] + ((p.y >> 16) & 0xffu);
    id[5] = id[4] + (p.y >> 24);
    id[6] = id[5] + (p.z & 0xffu);
    id[7] = id[6] + ((p.z >> 8) & 0xffu);
    id[8] = id[7] + ((p.z >> 16) & 0xffu);
    id[9] = id[8] + (p.z >> 24);
    id[10] = id[9] + (p.w & 0xffu);
    id[11] = id[10] + ((p.w >> 8) & 0xffu);
    id[12] = id[11] + ((p.w >> 16) & 0xffu);
    id[13] = id[12] + (p.w >> 24);
    uint32_t w[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) w[k] = bm[id[k] >> 5];
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < 14; ++k) hits |= __builtin_amdgcn_ubfe(w[k], id[k], 1u) << k;
    const uint32_t n = (p.x >> 16) & 0xffu;
    return uint32_t(__popc(hits & ((1u << n) - 1u)));
}

// one unit of the 12-bit-gap form: 16-bit base, 4-bit count (1 … 10), nine 12-bit gaps from bit 20 on (two of them straddle a dword:
// v_alignbit).  Unused gaps are 0 — the running id repeats — and the count cuts them off.
__device__ __forceinline__ uint32_t gap12_unit_hits(const uint32_t *bm, uint4 p) {
    uint32_t id[10];
    id[0] = p.x & 0xffffu;
    id[1] = id[0] + (p.x >> 20);
    id[2] = id[1] + (p.y & 0xfffu);
    id[3] = id[2] + __builtin_amdgcn_ubfe(p.y, 12u, 12u);
    id[4] = id[3] + (__builtin_amdgcn_alignbit(p.z, p.y, 24u) & 0xfffu);
    id[5] = id[4] + __builtin_amdgcn_ubfe(p.z, 4u, 12u);
    id[6] = id[5] + __builtin_amdgcn_ubfe(p.z, 16u, 12u);
    id[7] = id[6] + (__builtin_amdgcn_alignbit(p.w, p.z, 28u) & 0xfffu);
    id[8] = id[7] + __builtin_amdgcn_ubfe(p.w, 8u, 12u);
    id[9] = id[8] + (p.w >> 20);
    uint32_t w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = bm[id[k] >> 5];
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) hits |= __builtin_amdgcn_ubfe(w[k], id[k], 1u) << k;
    const uint32_t n = (p.x >> 16) & 0xfu;
    return uint32_t(__popc(hits & ((1u << n) - 1u)));
}

// ---- tail side: open-addressing hash set in LDS (keys are rank ids >= kHub; -1 = empty) ------------------------
__device__ __forceinline__ uint32_t hash_slot(int32_t w, int shift) { return (uint32_t(w) * 0x9E3779B1u) >> shift; }

__device__ __forceinline__ void set_insert(int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift);
    while (atomicCAS(&tbl[h], -1, w) != -1) h = (h + 1) & mask;
}
// (one exit condition per trip: with the two early returns of the obvious loop the compiler builds two nested exec-mask regions per probe,
// ~20 scalar + vector instructions a trip; this is ~9)
__device__ __forceinline__ uint32_t set_contains(const int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift), found = 0u;
    bool go;
    do {
        const int32_t x = tbl[h];
        found |= x == w ? 1u : 0u;
        go = x != w && x != -1;
        h = (h + 1) & mask;
    } while (go);
    return found;
}
// Tail stream rows (trow / tpool: 32-bit ids, 4 per 16-byte unit, filler -2 — or 16-bit delta units, 6 ids each) against the pivot's
// tail set; same shape as scan_srows.
// Almost every streamed tail id is a miss (scale 24: 0.7 M of 10.3 G triangles close through a tail id), so the set is fronted by a
// FILTER: a bitmap of the low id bits, 32 bits per table slot (32768 bits for a full 512-key tile).  One LDS word read + bit test answers "no" for all but tl/32768 of the ids; only
// the lanes with a positive walk the open-addressing table (whose divergent probe loops were 47 of k_tc_block's 165 ms at scale 26).
// The filter is ALWAYS 32768 bits (round 4), whatever the table: a wave probes 64 x 6 ids per step, and the exact-probe branch runs for the
// whole wave when ANY of them passes.  Sized with the table (32 bits per slot, 2 slots per key: one false positive per 64 probes, rounds
// 2-3) that was nearly every step, and the walk of the table for a handful of lanes cost three times the filter pass itself (scale 26:
// 180 wave-instructions per step against ~50; the scalar unit 60 % busy with the branches).  A few dozen keys in 32768 bits: ~0.1 %.
static constexpr int kFilterWords = 1024;
// FAST PATH: the filter words of all ids of the unit, each shifted by its id's bit position, ORed — bit 0 of the result says "some id
// may be a key" (2 VALU instructions per id behind the LDS read; v_lshrrev takes the low five bits of the id by itself).  Only then the
// SLOW PATH: which ids (bit tests, the unit's count cut off), and one exact probe per set bit, lane by lane (ffs loop: the lanes that
// got here are few, and six exec-masked blocks — rounds 2-3 — cost 90 instructions for the wave whoever had a bit).
__device__ __forceinline__ uint32_t flt_word(const uint32_t *flt, uint32_t id) { return flt[(id >> 5) & uint32_t(kFilterWords - 1)] >> (id & 31u); }
// y where c (0 / 1) is set, else x — arithmetic, so that the select tree below stays free of branches (written with ?: the compiler turned
// the six-way choice into nested exec-mask regions: ~25 instructions, most of them scalar, per probed id)
__device__ __forceinline__ uint32_t pick(uint32_t c, uint32_t x, uint32_t y) { return x ^ ((x ^ y) & (0u - c)); }
__device__ __forceinline__ uint32_t probe_set_bits(const int32_t *tbl, uint32_t mask, int shift, uint32_t m, const uint32_t (&id)[6]) {
    uint32_t c = 0;
    while (m) {
        const uint32_t k = uint32_t(__ffs(int(m)) - 1);
        m &= m - 1u;
        const uint32_t b0 = k & 1u, b1 = (k >> 1) & 1u, b2 = k >> 2;
        const uint32_t w = pick(b2, pick(b1, pick(b0, id[0], id[1]), pick(b0, id[2], id[3])), pick(b0, id[4], id[5]));
        c += set_contains(tbl, mask, shift, int32_t(w));
    }
    return c;
}
__device__ __forceinline__ uint32_t tail_unit_hits(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p) {
    const uint32_t w0 = flt_word(flt, p.x), w1 = flt_word(flt, p.y), w2 = flt_word(flt, p.z), w3 = flt_word(flt, p.w);
    if (((w0 | w1 | w2 | w3) & 1u) == 0) return 0u;
    const uint32_t id[6] = {p.x, p.y, p.z, p.w, 0u, 0u};  // (the filler -2 may pass the filter, it is never a key)
    return probe_set_bits(tbl, mask, shift, (w0 & 1u) | ((w1 & 1u) << 1) | ((w2 & 1u) << 2) | ((w3 & 1u) << 3), id);
}
// one unit of the 16-bit delta form: 32-bit base, count (low half of word 1), five 16-bit gaps (unused gaps are 0: the id repeats)
__device__ __forceinline__ uint32_t tail_delta_unit_hits(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p) {
    const uint32_t id0 = p.x, id1 = id0 + (p.y >> 16), id2 = id1 + (p.z & 0xffffu), id3 = id2 + (p.z >> 16), id4 = id3 + (p.w & 0xffffu),
                   id5 = id4 + (p.w >> 16);
    const uint32_t w0 = flt_word(flt, id0), w1 = flt_word(flt, id1), w2 = flt_word(flt, id2), w3 = flt_word(flt, id3), w4 = flt_word(flt, id4),
                   w5 = flt_word(flt, id5);
    if (((w0 | w1 | w2 | w3 | w4 | w5) & 1u) == 0) return 0u;
    const uint32_t n = p.y & 0xffu;
    const uint32_t m = ((w0 & 1u) | ((w1 & 1u) << 1) | ((w2 & 1u) << 2) | ((w3 & 1u) << 3) | ((w4 & 1u) << 4) | ((w5 & 1u) << 5)) & ((1u << n) - 1u);
    const uint32_t id[6] = {id0, id1, id2, id3, id4, id5};
    return probe_set_bits(tbl, mask, shift, m, id);
}
// ---------------------------------------------------------------------------------------------
// Work items (device_graph.hpp): one workgroup per item = up to kTaskChunk consecutive entries of ONE pivot's hub-entry list (k_tc_block:
// the pivot's hub part staged as the 65536-bit bitmap, 8 KB) or tail-entry list (k_tc_tail: the pivot's tail part as filter + hash set,
// tiled if longer than half the table).  An entry is one stream-row descriptor (6 bytes, TaskList) — a member's row, the cut row of an in-neighbour
// that handed its edge over, a 64-unit chunk of an inline row: the kernels cannot tell and need not.
// ---------------------------------------------------------------------------------------------
static constexpr int kBlockLog = 10;
template <int FORM>
__device__ __forceinline__ uint32_t hub_unit_hits(const uint32_t *bm, uint4 p, int j) {
    if (FORM == kFormBitset) {
        const uint4 q = *reinterpret_cast<const uint4 *>(bm + 4 * j);
        return uint32_t(__popc(p.x & q.x) + __popc(p.y & q.y) + __popc(p.z & q.z) + __popc(p.w & q.w));
    }
    if (FORM == kFormDelta) return delta_unit_hits(bm, p);
    if (FORM == kFormGap12) return gap12_unit_hits(bm, p);
    return hub_hits8(bm, u4u{p.x, p.y, p.z, p.w});
}
template <int FORM>
__device__ __forceinline__ uint32_t tail_unit_hits_f(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p) {
    return FORM == kFormDelta ? tail_delta_unit_hits(flt, tbl, mask, shift, p) : tail_unit_hits(flt, tbl, mask, shift, p);
}
// entries [lo, hi) of the item (descriptors in LDS), one form, rows of similar length: HIT(p, j) = hits of unit j of a row.  ONE_STEP: the
// classes of the run guarantee units <= W (one load per lane, no loop).
#ifndef GMSX_TC_TAIL_DEPTH
#define GMSX_TC_TAIL_DEPTH 1  // steps in the ring besides the one being issued: ONE is enough once the loads are counted (scale 26, tail items alone: depth 1 / 2 / 3 / 4 = 16.6 / 19.0 / 20.6 / 23.3 ms; hub items 47.9 / 49.0 / 50.2) — the registers of a deeper ring cost the eighth wave per SIMD or spill
#endif
#ifndef GMSX_TC_HUB_GROUP
#define GMSX_TC_HUB_GROUP 16  // lanes per row: one width for all rows — with the step stream a group moves from row to row on its own (8-lane groups: 57 vs 51 ms
#endif                        // for the hub rows — 128-byte requests against 256-byte ones —, no difference for the tail rows)
#ifndef GMSX_TC_TAIL_GROUP
#define GMSX_TC_TAIL_GROUP 16
#endif
#ifndef GMSX_TC_HUB_DEPTH
#define GMSX_TC_HUB_DEPTH 1   // … of the hub scans (the decode of a byte-delta unit wants 30 registers of its own)
#endif
// Entries [lo, hi) of an item (descriptors in LDS), one form, rows of one width class, as a STREAM OF STEPS: a group of W lanes works on
// one row, a step = W consecutive units of it (one 16-byte load per lane), and a lane keeps D steps in flight across row boundaries.
// Every load is UNCONDITIONAL — a lane without a unit in the step re-reads the last unit of its row, a group past its last entry unit 0
// of the pool — because a load under a divergent branch cannot be counted: hipcc then waits with vmcnt(0) at every use, i.e. for the load it
// issued a moment ago, and a wave never has more than one load instruction in flight (rounds 2-3: that, times 32 waves per CU, was the
// 3.7 TB/s of the tail items — 0.9 KB per wave and memory latency).  With counted waits the D loads overlap.
// HIT(p, j) = hits of unit j of a row.
// An item's task entries as they lie in LDS: the two halves of the 6-byte entries (TaskList) as they come from HBM — the entry is taken
// apart when a group opens its row (one ds_read_b32 + one ds_read_u16 and an alignbit instead of one ds_read_b64 and a 64-bit shift).
struct StagedTasks {
    uint32_t lo[kTaskChunk];
    uint32_t hi[kTaskChunk / 2 + 2];  // 16 bits per entry; the direct-to-LDS copy moves whole dwords: an item that begins at an odd entry has one foreign half in front
};
template <int W, int D, bool TAIL>
struct StepStream {
    static constexpr int G = 256 / W;
    const uint32_t *slo;
    const uint16_t *shi;
    const uint4 *pool4;
    int hi, sub;
    // cursor of the step to ISSUE: entry e (first unit of its row, form, units), unit j of this lane.  Per row: its
    // first unit's address, the index of its last unit (a lane without a unit in the step re-reads that one: same line as its
    // neighbours'), and j's bound for "the row is through"
    int e, units, last, lim, j, form;
    const uint4 *row;
    int pending;  // steps of real entries in the ring
    // D == 1 keeps TWO slots and alternates between them (probe slot 0 while slot 1 loads, then the other way round): the same one step
    // in flight, but no register-to-register copies of the unit being probed (five v_mov per step in the rotating form)
    static constexpr int S = D == 1 ? 2 : D;
    uint4 p[S];
    int pj[S];    // unit index of the lane in that step | form << 24, -1 = none
    __device__ __forceinline__ void open_row() {
        int ee = e;
        asm volatile("" : "+v"(ee));  // (both LDS addresses from the entry index each time: as two running addresses they cost two registers for the whole item)
        const uint32_t l = e < hi ? slo[ee] : 0u, h = e < hi ? uint32_t(shi[ee]) : 0u;
        row = pool4 + __builtin_amdgcn_alignbit(h, l, 16);  // first unit = hi << 16 | lo >> 16
        units = int(l & (TAIL ? kTaskTailUnitsMax : kTaskHubUnitsMax));
        form = TAIL ? int((l >> 15) & 1u) << 25 : int((l >> 14) & 3u) << 24;  // (tail lists keep one form bit: list 0 / delta 2)
        last = max(units - 1, 0);
        lim = units + sub;
        j = sub;
    }
    __device__ __forceinline__ void issue(int k) {
        p[k] = row[uint32_t(min(j, last))];
        pj[k] = j < units ? (j | form) : -1;
        pending += e < hi ? 1 : 0;
        j += W;
        if (j >= lim) {  // the row is through (uniform per group): next entry of the group
            e += G;
            open_row();
        }
    }
    // entries [lo, hi_) of the item: the first D steps go out here — BEFORE the pivot's bitmap / table is rebuilt, so that they are in
    // flight while it is (the item kernel), not after
    __device__ __forceinline__ void start(const StagedTasks &sd, int odd, const uint32_t *__restrict__ pool, int lo, int hi_, int tid) {
        slo = sd.lo;
        shi = reinterpret_cast<const uint16_t *>(sd.hi) + odd;
        pool4 = reinterpret_cast<const uint4 *>(pool);
        hi = hi_;
        sub = tid % W;
        e = lo + tid / W;
        pending = 0;
        open_row();
#pragma unroll
        for (int k = 0; k < D; ++k) issue(k);
    }
    // HIT(p, j, form) = hits of unit j of a row of that form.  The entries of an item are sorted by form, so the groups of a wave disagree
    // about it only where two runs meet: ONE stream — one fill, one drain — serves the whole item (rounds 2-3: a loop per form and width).
    template <class Hit>
    __device__ __forceinline__ uint32_t run(Hit hit) {
        uint32_t cnt = 0;
        if constexpr (D == 1) {
            auto step = [&](int cur, int nxt) {
                pending -= 1;
                issue(nxt);
#ifdef GMSX_TC_NO_PROBE
                if (pj[cur] >= 0) cnt += p[cur].x & 1u;
#else
                if (pj[cur] >= 0) cnt += hit(p[cur], pj[cur] & 0xffffff, pj[cur] >> 24);
#endif
            };
            while (pending > 0) {
                step(0, 1);
                if (!(pending > 0)) break;
                step(1, 0);
            }
            return cnt;
        }
        while (pending > 0) {  // the groups of a wave differ by the lengths of their rows
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const uint4 pc = p[k];
                const int jc = pj[k];
                pending -= 1;  // (slots of dead entries push it below zero: the loop ends at the first check after the last real step)
                issue(k);
#ifdef GMSX_TC_NO_PROBE  // A/B build (WRONG counts): every unit is loaded, nothing is probed — what the memory side alone takes
                if (jc >= 0) cnt += pc.x & 1u;
#else
                if (jc >= 0) cnt += hit(pc, jc & 0xffffff, jc >> 24);
#endif
            }
        }
        return cnt;
    }
};
// the hub forms against the pivot bitmap / the tail forms against filter + table, dispatched per step (divergent only where two runs meet)
__device__ __forceinline__ uint32_t hub_hits_by_form(const uint32_t *bm, uint4 p, int j, int f) {
    if (f == kFormList) return hub_unit_hits<kFormList>(bm, p, j);
    if (f == kFormDelta) return hub_unit_hits<kFormDelta>(bm, p, j);
    if (f == kFormBitset) return hub_unit_hits<kFormBitset>(bm, p, j);
    return hub_unit_hits<kFormGap12>(bm, p, j);
}
__device__ __forceinline__ uint32_t tail_hits_by_form(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p, int f) {
    return f == kFormDelta ? tail_delta_unit_hits(flt, tbl, mask, shift, p) : tail_unit_hits(flt, tbl, mask, shift, p);
}
// copies the item's descriptors to LDS
__device__ __forceinline__ void stage_item(const TaskList ent, int ne, int tid, StagedTasks &sd) {
    for (int i = tid; i < ne; i += 256) {
        sd.lo[i] = ent.lo[i];
        reinterpret_cast<uint16_t *>(sd.hi)[i] = ent.hi[i];
    }
    __syncthreads();
}
__device__ __forceinline__ void block_add(unsigned long long cnt, unsigned long long *red, int lane, int wave, int tid, unsigned long long *__restrict__ acc) {
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[(blockIdx.x & (kAccSlots - 1)) * kAccStride], t);
    }
}


__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tc_block(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const uint32_t *__restrict__ spool, const TaskList htask,
                                                  const gmsx_tc_item *__restrict__ items, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) uint32_t bm[kBitmapWords + 128];  // + slack: the delta probes of unused slots read up to 104 words past the bitmap
    __shared__ StagedTasks sdesc;
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gmsx_tc_item &it = items[blockIdx.x];
    const int32_t u = it.pivot;
    const int64_t hb = hoff[u];
    const int hl = int(hoff[u + 1] - hb);
    const int ne = int(it.bc >> 40);
    for (int i = tid; i < (kBitmapWords + 128) / 4; i += 256) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    stage_item(htask.at(int64_t(it.bc & 0xffffffffffull)), ne, tid, sdesc);
    for (int i = tid; i < hl; i += 256) {
        const uint32_t id = hadj[hb + i];
        if (id != 0xFFFFu) atomicOr(&bm[id >> 5], 1u << (id & 31u));
    }
    __syncthreads();
    uint32_t cnt = 0;
#ifndef GMSX_TC_STAGING_ONLY  // (A/B build: what the per-item fixed cost alone takes)
    StepStream<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH, false> st;
    st.start(sdesc, 0, spool, 0, ne, tid);
    cnt += st.run([](uint4 p, int j, int f) { return hub_hits_by_form(bm, p, j, f); });
#endif
    block_add(cnt, red, lane, wave, tid, acc);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tc_tail(const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                 const uint32_t *__restrict__ tpool, const TaskList ttask,
                                                 const gmsx_tc_item *__restrict__ items, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) int32_t tbl[1 << kBlockLog];
    __shared__ __attribute__((aligned(16))) uint32_t flt[kFilterWords];
    __shared__ StagedTasks sdesc;
    __shared__ unsigned long long red[4];
    constexpr int TILE = (1 << kBlockLog) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gmsx_tc_item &it = items[blockIdx.x];
    const int32_t u = it.pivot;
    const int64_t tb = toff[u];
    const int tl = int(toff[u + 1] - tb);
    const int ne = int(it.bc >> 40);
    stage_item(ttask.at(int64_t(it.bc & 0xffffffffffull)), ne, tid, sdesc);
    // the table sized for THIS pivot (most tail parts are a few dozen ids): 2^log slots >= 2 x keys
    int log = 6;
    while ((1 << log) < 2 * min(tl, TILE)) ++log;
    const int size = 1 << log, shift = 32 - log;
    const uint32_t mask = uint32_t(size - 1);
    uint32_t cnt = 0;
    for (int t0 = 0; t0 < tl; t0 += TILE) {  // the pivot's tail part, a tile at a time
        const int tn = min(TILE, tl - t0);
        __syncthreads();
        for (int i = tid; i < size; i += 256) tbl[i] = -1;
        reinterpret_cast<uint4 *>(flt)[tid] = make_uint4(0u, 0u, 0u, 0u);  // kFilterWords = 4 x 256
        __syncthreads();
        for (int i = tid; i < tn; i += 256) {
            const int32_t t = tadj[tb + t0 + i];
            set_insert(tbl, mask, shift, t);
            atomicOr(&flt[(uint32_t(t) >> 5) & uint32_t(kFilterWords - 1)], 1u << (uint32_t(t) & 31u));
        }
        __syncthreads();
#ifndef GMSX_TC_STAGING_ONLY
        StepStream<GMSX_TC_TAIL_GROUP, GMSX_TC_TAIL_DEPTH, true> st;
        st.start(sdesc, 0, tpool, 0, ne, tid);
        cnt += st.run([mask, shift](uint4 p, int, int f) { return tail_hits_by_form(flt, tbl, mask, shift, p, f); });
#endif
    }
    block_add(cnt, red, lane, wave, tid, acc);
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT work-item kernels (round 4): what k_tc_block / k_tc_tail did with one workgroup per item — 5.4 M workgroups per pass at scale
// 26, each opening with a chain of dependent round trips (item -> offsets -> descriptors / pivot ids -> LDS) before its first row load
// — as one launch per queue (hub items, tail items) of as many workgroups as the chip holds.  A workgroup draws tickets of kGrab
// consecutive items (fetched one chunk ahead) and keeps a three-deep pipeline over them:
//   item i+2   its 64-byte record is on its way into LDS (LDS-DMA, 16 lanes of wave 0)
//   item i+1   its descriptors and the pivot's ids are on their way into the OTHER half of the LDS staging buffers by LDS-DMA
//              (global_load_lds: no registers, no instructions at arrival)
//   item i     is scanned: ONE step stream over all its entries (StepStream above), opened before the bitmap is brought up to date.
// Records are self-contained (device_graph.hpp: entries, the pivot's container part), so nothing in the chain depends on a second lookup.
// The pivot bitmap is MAINTAINED instead of rebuilt: consecutive items of one pivot share it; otherwise the ids of the previous pivot (still
// in the other staging half) are XORed out and the new ones XORed in — both commute, so no barrier separates them and nothing clears 8 KB
// per item.  Tail items use the same LDS as hash set (sized per pivot) + 32768-bit filter.
// Two launches, one per queue (hub items, then tail items): see k_tc_items below.
// ---------------------------------------------------------------------------------------------
static constexpr int kGrab = 8;        // items per queue ticket
static constexpr int kIdStage = 256;   // dwords per id staging half: 512 hub ids / 256 tail ids; longer containers read the rest from memory
static constexpr int kQueueStride = 16;  // the two queue heads, 64 bytes apart (unsigned int)

// LDS-DMA, one dword per lane: lane l's source is its own, the destination is lds + 4 l (wave-uniform base in M0).  Invisible to the
// compiler's s_waitcnt bookkeeping: the issuing wave drains it with an explicit vmcnt(0) before the barrier that publishes the data.
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return uint32_t(uintptr_t((const __attribute__((address_space(3))) void *)p)); }
__device__ __forceinline__ void glds_dword(const uint32_t *src, uint32_t lds_byte_addr) {  // LDS addresses as 32-bit numbers: no 64-bit generic pointers per lane
    unsigned keep;
    const unsigned dst = uni32(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void stage_dwords(const uint32_t *__restrict__ src, uint32_t lds_byte_addr, int ndw, int lane, int wave) {
    asm volatile("" : "+v"(lane));  // (the per-lane source address is formed here, not at the top of the item and carried — or spilled — across its scans)
    for (int base = wave * 64; base < ndw; base += 256)  // uniform per wave
        if (base + lane < ndw) glds_dword(src + base + lane, lds_byte_addr + 4u * uint32_t(base));
}
// LDS of a k_tc_items workgroup (18.9 KB: eight per CU)
struct ItemLds {
    uint32_t bm[kBitmapWords + 128];  // hub items: the pivot bitmap (+ slack: the delta probes of unused slots read up to 104 words past it); tail items: table | filter
    StagedTasks sdesc[2];
    uint32_t idst[2][kIdStage];
    uint32_t rec[3][16];  // records of the items A (scanned), B (staging), C (arriving)
    unsigned long long red[4];
    int chunk[2];  // first item of the chunk behind the one the cursor walks (-1: the queue is dry)
};
// One queue (hub items or tail items) until it is dry.
template <bool TAIL>
__device__ __forceinline__ void item_loop(ItemLds &L, const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ pool,
                                          const TaskList task, const gmsx_tc_item *__restrict__ items, int n_items,
                                          unsigned int *__restrict__ qhead, unsigned long long &total) {
    constexpr int TILE = (1 << kBlockLog) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *bm = L.bm;
    const uint32_t a_slo = lds_addr(&L.sdesc[0].lo[0]), a_shi = lds_addr(&L.sdesc[0].hi[0]), a_idst = lds_addr(&L.idst[0][0]), a_rec = lds_addr(&L.rec[0][0]);
    __syncthreads();  // the previous phase is over: its LDS is free
    if (tid == 0) {
        const unsigned a = atomicAdd(qhead, unsigned(kGrab));
        const unsigned b = a < unsigned(n_items) ? atomicAdd(qhead, unsigned(kGrab)) : a;
        L.chunk[0] = a < unsigned(n_items) ? int(a) : -1;
        L.chunk[1] = b < unsigned(n_items) ? int(b) : -1;
    }
    if (!TAIL)
        for (int i = tid; i < (kBitmapWords + 128) / 4; i += 256) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // cursor over the item stream of this workgroup (uniform): item ci of the chunk [.., ce); -1 = none
    int ci = uni32(L.chunk[0]);
    int ce = ci < 0 ? 0 : min(ci + kGrab, n_items);
    auto advance = [&]() -> bool {  // to the next item; true when it entered the next chunk (thread 0 then fetches the one after it)
        if (ci < 0) return false;
        if (++ci < ce) return false;
        ci = uni32(L.chunk[1]);
        ce = ci < 0 ? 0 : min(ci + kGrab, n_items);
        return true;
    };
    auto fetch_rec = [&](int slot, int lane, int wave) {  // the cursor's record -> LDS (wave 0, lanes 0 … 15, asynchronous)
        if (wave == 0 && lane < 16) glds_dword(reinterpret_cast<const uint32_t *>(items + ci) + lane, a_rec + 64u * uint32_t(slot));
    };
    auto stage = [&](const uint32_t *rec, int b, int lane, int wave) {  // descriptors and pivot ids of a record -> staging half b (asynchronous)
        const uint32_t bc_lo = uni32(rec[0]), bc_hi = uni32(rec[1]), c_lo = uni32(rec[2]), c_hi = uni32(rec[3]);
        const uint64_t first = (uint64_t(bc_hi & 0xffu) << 32) | bc_lo, cb = (uint64_t(c_hi & 0xffu) << 32) | c_lo;
        const int ne = int(bc_hi >> 8), cn = int(c_hi >> 8);
        stage_dwords(task.lo + first, a_slo + uint32_t(b) * uint32_t(sizeof(StagedTasks)), ne, lane, wave);
        stage_dwords(reinterpret_cast<const uint32_t *>(task.hi + (first & ~uint64_t(1))), a_shi + uint32_t(b) * uint32_t(sizeof(StagedTasks)), (int(first & 1u) + ne + 1) / 2, lane, wave);
        if (TAIL) stage_dwords(reinterpret_cast<const uint32_t *>(tadj + cb), a_idst + uint32_t(b) * uint32_t(sizeof(L.idst[0])), min(cn, kIdStage), lane, wave);
        else stage_dwords(reinterpret_cast<const uint32_t *>(hadj + cb), a_idst + uint32_t(b) * uint32_t(sizeof(L.idst[0])), min(cn, 2 * kIdStage) / 2, lane, wave);
    };
    auto flip2 = [&](uint32_t w) {  // two 16-bit hub ids: toggle their bits (0xFFFF = row padding)
        const uint32_t lo = w & 0xffffu, hi = w >> 16;
        if (lo != 0xFFFFu) atomicXor(&bm[lo >> 5], 1u << (lo & 31u));
        if (hi != 0xFFFFu) atomicXor(&bm[hi >> 5], 1u << (hi & 31u));
    };
    // prologue: records A and B, then A's staging
    bool vA = ci >= 0, vB = false;
    if (vA) fetch_rec(0, lane, wave);
    if (advance()) {  // (a first chunk of one item: the cursor is in the second chunk already)
        __syncthreads();  // everybody has read chunk[1]
        if (tid == 0 && ci >= 0) {
            const unsigned x = atomicAdd(qhead, unsigned(kGrab));
            L.chunk[1] = x < unsigned(n_items) ? int(x) : -1;
        }
    }
    vB = ci >= 0;
    if (vB) fetch_rec(1, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (vA) stage(L.rec[0], 0, lane, wave);
    int ra = 0, buf = 0;
    bool bm_valid = true, bm_small = true;  // the bitmap holds exactly the hub ids of bm_pivot; all of them are in the other staging half
    int bm_pivot = -1, bm_n = 0;
    const int tid0 = tid;
    while (vA) {
        // per-lane address parts (pool + 16 * sub for every group width, staging offsets …) are loop-invariant, and hoisted out of THIS loop
        // they cost 40 VGPRs for the whole kernel (spills at the 64 that eight waves per SIMD allow): an opaque copy of the thread id per item
        // keeps them inside the item, where they cost a handful of VALU instructions
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int rb = ra == 2 ? 0 : ra + 1, rc = rb == 2 ? 0 : rb + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of item A's staging (and of record B) has landed …
        __syncthreads();                                   // (1) … everybody's has, and nobody reads the previous item's LDS any more
        const bool crossed = advance();
        const bool vC = ci >= 0;
        if (vC) fetch_rec(rc, lane, wave);
        unsigned pend = 0;
        const bool fetch = tid == 0 && crossed && vC;
        if (fetch) pend = atomicAdd(qhead, unsigned(kGrab));  // consumed after the scan: the round trip stays off the path
        const uint32_t *recA = L.rec[ra];
        const uint32_t c_lo = uni32(recA[2]), c_hi = uni32(recA[3]);
        const int64_t cb = int64_t((uint64_t(c_hi & 0xffu) << 32) | c_lo);
        const int cn = int(c_hi >> 8);
        const StagedTasks &sd = L.sdesc[buf];
        const int neA = int(uni32(recA[1]) >> 8), oddA = int(uni32(recA[0]) & 1u);  // (an odd first entry: its 16-bit half is the second of the first dword staged)
        uint32_t c = 0;
        // the item's step stream opens HERE: its first loads are in flight while the pivot's bitmap / table is brought up to date
        StepStream<TAIL ? GMSX_TC_TAIL_GROUP : GMSX_TC_HUB_GROUP, TAIL ? GMSX_TC_TAIL_DEPTH : GMSX_TC_HUB_DEPTH, TAIL> st;
#ifndef GMSX_TC_STAGING_ONLY
        if (!TAIL) st.start(sd, oddA, pool, 0, neA, tid);  // (the tail items open theirs behind the table build: carried across it, the stream's state spills)
#endif
        if (!TAIL) {
            const int pivotA = int(uni32(recA[4]));
            if (!(bm_valid && bm_pivot == pivotA)) {
                if (bm_valid && bm_small) {
                    for (int i = tid; i < bm_n / 2; i += 256) flip2(L.idst[buf ^ 1][i]);  // the previous pivot's ids out …
                } else {
                    for (int i = tid; i < (kBitmapWords + 128) / 4; i += 256) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
                    __syncthreads();
                }
                const int ns = min(cn, 2 * kIdStage);
                for (int i = tid; i < ns / 2; i += 256) flip2(L.idst[buf][i]);  // … this one's in (XOR commutes: no barrier in between)
                for (int i = 2 * kIdStage + 2 * tid; i < cn; i += 512) flip2(*reinterpret_cast<const uint32_t *>(hadj + cb + i));
                bm_valid = true;
                bm_pivot = pivotA;
                bm_n = ns;
                bm_small = cn <= 2 * kIdStage;
            }
            __syncthreads();  // (2)
            if (vB) stage(L.rec[rb], buf ^ 1, lane, wave);
            if (fetch) L.chunk[1] = pend < unsigned(n_items) ? int(pend) : -1;  // the chunk behind the one the cursor just entered (read behind the next barrier (1))
#ifndef GMSX_TC_STAGING_ONLY
            c += st.run([bm](uint4 p, int j, int f) { return hub_hits_by_form(bm, p, j, f); });
#endif
        } else {
            int32_t *tbl = reinterpret_cast<int32_t *>(bm);
            uint32_t *flt = bm + (1 << kBlockLog);
            // the table sized for THIS pivot: 2^log slots >= 2 x keys of a tile
            int log = 6;
            while ((1 << log) < 2 * min(cn, TILE)) ++log;
            const int size = 1 << log, shift = 32 - log;
            const uint32_t mask = uint32_t(size - 1);
            bool staged = false;
            for (int t0 = 0; t0 < cn; t0 += TILE) {  // the pivot's tail part, a tile at a time (one tile for all but a handful of pivots)
                const int tn = min(TILE, cn - t0);
                if (t0 > 0) __syncthreads();
                for (int i = tid; i < size; i += 256) tbl[i] = -1;
                reinterpret_cast<uint4 *>(flt)[tid] = make_uint4(0u, 0u, 0u, 0u);  // kFilterWords = 4 x 256
                __syncthreads();
                int tin = tid0;
                asm volatile("" : "+v"(tin));  // (per-lane addresses of this loop stay inside the tile: held across the scans they spill)
                for (int i = tin; i < tn; i += 256) {
                    const int32_t t = t0 + i < kIdStage ? int32_t(L.idst[buf][t0 + i]) : tadj[cb + t0 + i];
                    set_insert(tbl, mask, shift, t);
                    atomicOr(&flt[(uint32_t(t) >> 5) & uint32_t(kFilterWords - 1)], 1u << (uint32_t(t) & 31u));
                }
                __syncthreads();  // (2)
                if (!staged && vB) stage(L.rec[rb], buf ^ 1, lane, wave);
                if (!staged && fetch) L.chunk[1] = pend < unsigned(n_items) ? int(pend) : -1;
                staged = true;
#ifndef GMSX_TC_STAGING_ONLY
                // (two forms, 32-bit list and 16-bit delta: a run each with its decode compiled in — dispatched per step, as the four hub forms
                //  are, the tail items took 19.6 ms instead of 17.8)
                const int nl = int(uni32(recA[8]) & 0xffffu);  // fbeg[2]: the list entries come first
                if (nl > 0) {
                    st.start(sd, oddA, pool, 0, nl, tid);
                    c += st.run([=](uint4 p, int, int) { return tail_unit_hits(flt, tbl, mask, shift, p); });
                }
                if (neA > nl) {
                    st.start(sd, oddA, pool, nl, neA, tid);
                    c += st.run([=](uint4 p, int, int) { return tail_delta_unit_hits(flt, tbl, mask, shift, p); });
                }
#endif
            }
            if (!staged) {  // (a pivot without tail ids: nothing can match)
                __syncthreads();
                if (vB) stage(L.rec[rb], buf ^ 1, lane, wave);
                if (fetch) L.chunk[1] = pend < unsigned(n_items) ? int(pend) : -1;
            }
        }
        total += c;
        ra = rb;
        vA = vB;
        vB = vC;
        buf ^= 1;
    }
}

// One kernel per queue (hub items, tail items), launched one after the other: mixed in one launch the two kinds took the SUM of their
// times anyway (both are bound by the same memory system, §5.1 of DESIGN.md), and a kernel that holds both loops also holds both
// argument sets and both scans' registers — it spilled at every occupancy above five waves per SIMD.
// Waves per SIMD the two are compiled for.  With a ring of one step both fit 64 registers without a spill (eight waves); deeper rings need
// 72 / 80 registers or spill — scratch traffic on the path of every item, and a reload is a vmcnt(0) in the middle of the stream.
#ifndef GMSX_TC_HUB_MIN_WAVES
#define GMSX_TC_HUB_MIN_WAVES 8
#endif
#ifndef GMSX_TC_TAIL_MIN_WAVES
#define GMSX_TC_TAIL_MIN_WAVES 8
#endif
template <bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TAIL ? GMSX_TC_TAIL_MIN_WAVES : GMSX_TC_HUB_MIN_WAVES, 8))) void k_tc_items(
    const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ pool, const TaskList task,
    const gmsx_tc_item *__restrict__ items, int n_items, unsigned int *__restrict__ qhead, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) ItemLds L;
    const int tid = threadIdx.x;
    unsigned long long total = 0;
    item_loop<TAIL>(L, hadj, tadj, pool, task, items, n_items, qhead, total);
    block_add(total, L.red, tid & 63, tid >> 6, tid, acc);
}

// ---------------------------------------------------------------------------------------------
// LIGHT EDGES (round 4; device_graph.hpp): the edges (u, v) between two light vertices that no work item covers, one 16-lane group per
// edge, no LDS.  Both rows are short (hub part + tail part < 64 ids), so |N+(u) ∩ N+(v)| is an ALL-PAIRS comparison in registers: the
// ids of a part sit interleaved over the 16 lanes (id i in lane i & 15, register i >> 4), v's registers are rotated through the group with
// DPP row_ror, and every lane compares what passes by with its own ids of u — 16 rotations x (registers of u) x (registers of v), the
// register counts being wave-uniform maxima (typically 1 x 1 or 2 x 2).  Every load is unconditional (clamped index, sentinel afterwards)
// and the next edge's record and rows are in flight while this one is compared: k_tc_wave — a wave per pivot, an 8 KB bitmap per wave,
// 16 waves per CU, one load in flight per wave behind a chain of dependent loads — took 10.6 ms for 14 GB at scale 26 (1.3 TB/s).
// ---------------------------------------------------------------------------------------------
// Does a lane of the 16-lane row hold an x equal to this lane's a?  x as the lane K places further along the row sees it (DPP row_ror:K),
// K = 0 … 15, every rotation reading the ORIGINAL register (independent instructions; round 4's first version rotated one register fifteen
// times: a dependent chain with a DPP hazard nop per link, 3.8 ms for the light edges of scale 26).  The ids of a row are distinct, so an
// a meets at most ONE equal x in all of v's registers: the compares are OR-ed as lane masks (v_cmp into SGPRs + s_or on the scalar unit)
// and counted once per a — two vector instructions per rotation instead of four.
template <int K>
__device__ __forceinline__ uint32_t ror_in_row(uint32_t x) {
    if constexpr (K == 0) return x;
    else return uint32_t(__builtin_amdgcn_mov_dpp(int(x), 0x120 + K, 0xf, 0xf, false));
}
__device__ __forceinline__ bool row_has(uint32_t a, uint32_t x) {
    bool m = false;
#define GMSX_ROT(K) m |= a == ror_in_row<K>(x);
    GMSX_ROT(0) GMSX_ROT(1) GMSX_ROT(2) GMSX_ROT(3) GMSX_ROT(4) GMSX_ROT(5) GMSX_ROT(6) GMSX_ROT(7)
    GMSX_ROT(8) GMSX_ROT(9) GMSX_ROT(10) GMSX_ROT(11) GMSX_ROT(12) GMSX_ROT(13) GMSX_ROT(14) GMSX_ROT(15)
#undef GMSX_ROT
    return m;
}
struct LightRows {
    uint32_t ah[4], at[4], bh[4], bt[4];  // u's hub / tail ids, v's hub / tail ids (interleaved over the 16 lanes)
};
struct LightRec {
    uint4 r0, r1;
};
// The k-th edge of shard `part` of `nparts` in a FULL list is edge k * nparts + (k odd ? nparts - 1 - part : part) — shard_of() of the edge's
// index, the rule the pivots are sharded by (a sharded upload holds its edges densely: nparts = 1 here).
__global__ __launch_bounds__(256) void k_tc_light(const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint4 *__restrict__ ledge, int nparts,
                                                  int part, int64_t count, unsigned long long *__restrict__ acc) {
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15;
    const int64_t ngroups = int64_t(gridDim.x) * 16, g0 = int64_t(blockIdx.x) * 16 + (tid >> 4);
    const int64_t trips = (count + ngroups - 1) / ngroups;  // the same for every group (uniform loop): a group without an edge runs on empty rows
    auto load_rec = [&](int64_t t) -> LightRec {
        const int64_t k = g0 + t * ngroups;
        const int64_t kk = k < count ? k : 0;  // (count > 0 here)
        const int64_t e = nparts <= 1 ? kk : kk * nparts + ((kk & 1) ? nparts - 1 - part : part);
        LightRec r;
        r.r0 = ledge[2 * e];
        r.r1 = ledge[2 * e + 1];
        if (k >= count) {  // lengths 0: everything below degenerates to sentinels
            r.r0.y &= 0xffu;
            r.r0.w &= 0xffu;
            r.r1.y &= 0xffu;
            r.r1.w &= 0xffu;
        }
        return r;
    };
    // registers a part of `len` ids needs, the maximum over the four groups of the wave (uniform)
    auto wave_max = [](int x) {
        return max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
    };
    auto load_rows = [&](const LightRec &r) -> LightRows {
        const int64_t hu = int64_t((uint64_t(r.r0.y & 0xffu) << 32) | r.r0.x), tu = int64_t((uint64_t(r.r0.w & 0xffu) << 32) | r.r0.z);
        const int64_t hv = int64_t((uint64_t(r.r1.y & 0xffu) << 32) | r.r1.x), tv = int64_t((uint64_t(r.r1.w & 0xffu) << 32) | r.r1.z);
        const int hlu = int(r.r0.y >> 8), tlu = int(r.r0.w >> 8), hlv = int(r.r1.y >> 8), tlv = int(r.r1.w >> 8);
        // the first register of every part unconditionally (countable loads); the second … fourth only when a group of the wave has that
        // many ids (a scalar branch: most light rows have fewer than 16 ids per part)
        const int more = wave_max((max(max(hlu, hlv), max(tlu, tlv)) + 15) >> 4);
        LightRows w;
        w.ah[0] = hadj[hlu ? hu + min(sub, hlu - 1) : 0];
        w.bh[0] = hadj[hlv ? hv + min(sub, hlv - 1) : 0];
        w.at[0] = uint32_t(tadj[tlu ? tu + min(sub, tlu - 1) : 0]);
        w.bt[0] = uint32_t(tadj[tlv ? tv + min(sub, tlv - 1) : 0]);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            w.ah[k] = w.bh[k] = w.at[k] = w.bt[k] = 0;
            if (k < more) {
                const int i = sub + 16 * k;
                w.ah[k] = hadj[hlu ? hu + min(i, hlu - 1) : 0];
                w.bh[k] = hadj[hlv ? hv + min(i, hlv - 1) : 0];
                w.at[k] = uint32_t(tadj[tlu ? tu + min(i, tlu - 1) : 0]);
                w.bt[k] = uint32_t(tadj[tlv ? tv + min(i, tlv - 1) : 0]);
            }
        }
        return w;
    };
    unsigned long long total = 0;
    if (trips > 0) {
        LightRec rec = load_rec(0);
        LightRows rows = load_rows(rec);
        LightRec rec_n = load_rec(1);
        for (int64_t t = 0; t < trips; ++t) {
            const LightRows rows_n = load_rows(rec_n);  // the next edge's rows and the record behind it are in flight while this edge is compared
            const LightRec rec_nn = load_rec(t + 2);
            const int hlu = int(rec.r0.y >> 8), tlu = int(rec.r0.w >> 8), hlv = int(rec.r1.y >> 8), tlv = int(rec.r1.w >> 8);
            // registers in use, the maximum over the four groups of the wave (uniform)
            const int nah = wave_max((hlu + 15) >> 4), nbh = wave_max((hlv + 15) >> 4), nat = wave_max((tlu + 15) >> 4), nbt = wave_max((tlv + 15) >> 4);
            uint32_t c = 0;
#pragma unroll
            for (int ka = 0; ka < 4; ++ka) {
                if (ka >= nah) break;
                const int ia = sub + 16 * ka;
                const uint32_t a = (ia < hlu && rows.ah[ka] != 0xFFFFu) ? rows.ah[ka] : 0xFFFFFFFFu;  // 0xFFFF = row padding
                bool m = false;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb >= nbh) break;
                    const int ib = sub + 16 * kb;
                    const uint32_t x = (ib < hlv && rows.bh[kb] != 0xFFFFu) ? rows.bh[kb] : 0xFFFFFFFEu;
                    m |= row_has(a, x);
                }
                c += m ? 1u : 0u;
            }
#pragma unroll
            for (int ka = 0; ka < 4; ++ka) {
                if (ka >= nat) break;
                const int ia = sub + 16 * ka;
                const uint32_t a = ia < tlu ? rows.at[ka] : 0xFFFFFFFFu;
                bool m = false;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb >= nbt) break;
                    const int ib = sub + 16 * kb;
                    const uint32_t x = ib < tlv ? rows.bt[kb] : 0xFFFFFFFEu;
                    m |= row_has(a, x);
                }
                c += m ? 1u : 0u;
            }
            total += c;
            rec = rec_n;
            rows = rows_n;
            rec_n = rec_nn;
        }
    }
    block_add(total, red, lane, wave, tid, acc);
}

// units / probes / algorithmic stream bytes of a shard (untimed bookkeeping for gmsx_stats).  out[2] follows what the count kernels
// read, byte for byte, assuming no on-chip reuse:
//   light pivot u (2 <= d+ < 64; k_tc_stats, wave per pivot position): per far light member v (rank id >= inline_limit, d+ < 64) the
//       32-byte edge record and the hub / tail parts of both rows as k_tc_light reads them (u's tail part up to v);
//   work item (k_tc_item_stats, wave per item): the pivot's container (hub part for a hub item, tail part for a tail item) once; per
//       entry 6 bytes of descriptor and the stream row it describes (whole 16-byte units) — members' rows, cut rows and inline chunks alike.
// out[0] = oriented edges counted by the shard: every edge of a light or idle pivot at the pivot, the edges a heavy pivot handed to its
// first members over inline at the pivot, every other edge of a heavy pivot where its entries live (tunits, written by the build);
// out[1] = id slots probed (per unit: 8 list, 14 byte-delta, 10 gap-12, 4 bitset words; 4 / 6 tail ids).  A shard = the pivots at the
// positions of `order` that shard_of() gives it.
__global__ __launch_bounds__(256) void k_tc_stats(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                  const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                  const int32_t *__restrict__ tunits, int32_t inline_limit, int inline_first, int64_t end, int nparts, int part,
                                                  unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long units = 0, probes = 0, bytes = 0;
    for (int64_t pos = wave0; pos < end; pos += nwaves) {
        if (nparts > 1 && shard_of(pos, nparts) != part) continue;
        const int32_t u = order[pos];
        const int du = dplus[u];
        if (du >= kHeavy) {  // forward + reverse edges whose entries live here (the heavy pivots are the first of `order`) …
            if (lane == 0) units += (unsigned long long)tunits[pos];
            // … and the edges to its first members that went inline (no entry anywhere)
            const int hl = min(int(hoff[u + 1] - hoff[u]), inline_first), tl = min(int(toff[u + 1] - toff[u]), inline_first - hl);
            int32_t v = -1;
            if (lane < hl) {
                const uint32_t x = hadj[hoff[u] + lane];
                if (x != 0xFFFFu) v = int32_t(x);
            } else if (lane - hl < tl) v = tadj[toff[u] + lane - hl];
            if (v >= 0 && lane > 0 && (v < inline_limit || dplus[v] >= kHeavy)) ++units;
            continue;
        }
        const int hl = int(hoff[u + 1] - hoff[u]);
        if (lane == 0) units += (unsigned long long)du;
        if (du < 2) continue;
        for (int64_t j = toff[u] + lane; j < toff[u + 1]; j += 64) {
            const int32_t v = tadj[j];
            if (v < inline_limit || dplus[v] >= kHeavy) continue;  // handed over: the ids are in v's inline rows, counted with its items
            // a light edge (k_tc_light): its 32-byte record, both parts of u's row (the tail part up to v) and of v's row
            const unsigned long long hv = (unsigned long long)(hoff[v + 1] - hoff[v]), tv = (unsigned long long)(toff[v + 1] - toff[v]);
            bytes += 32ull + 2ull * hl + 4ull * (unsigned long long)(j - toff[u]) + 2ull * hv + 4ull * tv;
            probes += hv + tv;  // v's ids, each compared with u's
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        units += __shfl_down(units, s);
        probes += __shfl_down(probes, s);
        bytes += __shfl_down(bytes, s);
    }
    if (lane == 0) {
        if (units) atomicAdd(&out[0], units);
        if (probes) atomicAdd(&out[1], probes);
        if (bytes) atomicAdd(&out[2], bytes);
    }
}
// one list of work items (hub or tail)
__global__ __launch_bounds__(256) void k_tc_item_stats(const int64_t *__restrict__ coff, int bytes_per_id, int tail, const TaskList task,
                                                       const gmsx_tc_item *__restrict__ items, int64_t n_items, int nparts, int part,
                                                       unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long probes = 0, bytes = 0;
    auto slots = [tail](unsigned long long d) -> unsigned long long {
        const unsigned long long n = d & 0x3fffffull, f = (d >> 22) & 3ull;
        return n * (tail ? (f == kFormDelta ? 6ull : 4ull) : (f == kFormList ? 8ull : f == kFormDelta ? 14ull : f == kFormGap12 ? 10ull : 4ull));
    };
    for (int64_t q = wave0; q < n_items; q += nwaves) {
        const gmsx_tc_item &it = items[q];
        if (nparts > 1 && shard_of(it.pos, nparts) != part) continue;
        const int ne = int(it.bc >> 40);
        const int64_t b = int64_t(it.bc & 0xffffffffffull);
        if (lane == 0) bytes += (unsigned long long)bytes_per_id * (unsigned long long)(coff[it.pivot + 1] - coff[it.pivot]);
        for (int i = lane; i < ne; i += 64) {
            const unsigned long long d = task.get(b + i);
            bytes += 6ull + 16ull * (d & 0x3fffffull);
            probes += slots(d);
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        probes += __shfl_down(probes, s);
        bytes += __shfl_down(bytes, s);
    }
    if (lane == 0) {
        if (probes) atomicAdd(&out[1], probes);
        if (bytes) atomicAdd(&out[2], bytes);
    }
}

// Diagnostics (gmsx_tc_stream_breakdown): the algorithmic stream bytes of one pass (= gmsx_stats.stream_bytes) by what is read.
//   out[0..2]  hub stream rows named by the work items' entries, by form (16-bit list, bitset, byte-delta)
//   out[3..4]  tail stream rows named by the entries (32-bit list, 16-bit delta)
//   out[5]     the entries themselves (6 bytes each)          out[6]  the pivots' own containers (hub part per hub item, tail part per tail item)
//   out[7]     of out[0] + out[3]: inline rows (ids handed over by light pivots; filled in by the host from the build's figures)
//   out[8..9]  light edges (k_tc_light): hub / tail parts of the far light members' rows
//   out[10]    light edges: the 32-byte records + the pivots' own hub / tail parts (once per edge)
//   out[11..14] counts: entries, inline entries, work items, light edges
//   out[15..20] reserved (0)
__global__ __launch_bounds__(256) void k_tc_breakdown(const int64_t *__restrict__ hoff, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                      const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                      const TaskList htask, const gmsx_tc_item *__restrict__ hitem, int64_t hitems,
                                                      const TaskList ttask, const gmsx_tc_item *__restrict__ titem, int64_t titems,
                                                      int32_t inline_limit, int64_t first_light, int64_t end_light,
                                                      unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long c[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t q = wave0; q < hitems + titems; q += nwaves) {
        const bool tail = q >= hitems;
        const gmsx_tc_item &it = tail ? titem[q - hitems] : hitem[q];
        const TaskList task = tail ? ttask : htask;
        const int ne = int(it.bc >> 40);
        const int64_t b0 = int64_t(it.bc & 0xffffffffffull);
        if (lane == 0) {
            c[6] += tail ? 4ull * (unsigned long long)(toff[it.pivot + 1] - toff[it.pivot]) : 2ull * (unsigned long long)(hoff[it.pivot + 1] - hoff[it.pivot]);
            c[11] += (unsigned long long)ne;
            c[13] += 1;
        }
        for (int i = lane; i < ne; i += 64) {
            const unsigned long long d = task.get(b0 + i);
            const unsigned long long db = 16ull * (d & 0x3fffffull);
            const int fd = int((d >> 22) & 3);
            if (tail) c[fd == kFormDelta ? 4 : 3] += db;
            else c[fd == kFormList ? 0 : fd == kFormBitset ? 1 : 2] += db;
            c[5] += 6;
        }
    }
    for (int64_t pos = first_light + wave0; pos < end_light; pos += nwaves) {
        const int32_t u = order[pos];
        const int64_t tb0 = toff[u], te = toff[u + 1];
        const unsigned long long hl = (unsigned long long)(hoff[u + 1] - hoff[u]);
        for (int64_t j = tb0 + lane; j < te; j += 64) {
            const int32_t v = tadj[j];
            if (v < inline_limit || dplus[v] >= kHeavy) continue;
            c[8] += 2ull * (unsigned long long)(hoff[v + 1] - hoff[v]);
            c[9] += 4ull * (unsigned long long)(toff[v + 1] - toff[v]);
            c[10] += 32ull + 2ull * hl + 4ull * (unsigned long long)(j - tb0);
            c[14] += 1;
        }
    }
    for (int k = 0; k < 15; ++k) {
        unsigned long long x = c[k];
        for (int s = 32; s > 0; s >>= 1) x += __shfl_down(x, s);
        if (lane == 0 && x) atomicAdd(&out[k], x);
    }
}

// Diagnostics behind gmsx_tc_row_histogram: how the stream rows the work items read are distributed over row lengths (a group of W
// lanes works on one row, so short rows keep few lanes busy).  out[(cls*24 + bin)*2 + {0,1}] = rows, units; cls: hub rows as list /
// bitset / byte-delta, tail rows as list / delta; bin: 1…16 units exactly, then 17-32, 33-64, … 1025+.  out[240…243]: entries, inline
// entries, work items, pivots' own container bytes; out[248…251]: Σ over the oriented edges (u,v) of heavy then light pivots u of the
// stream units of v's rows and of min(units of u's rows, units of v's rows) — what the smaller-endpoint rule is about.
__device__ __forceinline__ int hist_bin(unsigned long long units) {
    if (units <= 16) return int(units) - 1;
    int b = 16;
    for (unsigned long long lim = 32; units > lim && b < 23; lim <<= 1) ++b;
    return b;
}
// Diagnostics behind gmsx_tc_comembership: one key per entry of the hub items, (position of the item's pivot in `order`) / batch << 40 | (offset of
// the stream row the entry names) / 16, and the 16-byte units the entry streams of it — sorted and reduced by key on the host side of the call
__global__ __launch_bounds__(256) void k_tc_comember_keys(const TaskList htask, const gmsx_tc_item *__restrict__ hitem, int64_t hitems, int batch,
                                                         unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals, unsigned long long *__restrict__ sums) {
    unsigned long long ents = 0, units = 0;
    for (int64_t i = blockIdx.x; i < hitems; i += gridDim.x) {
        const gmsx_tc_item it = hitem[i];
        const unsigned long long first = it.bc & 0xffffffffffull;
        const int n = int(it.bc >> 40);
        const unsigned long long b = (unsigned long long)(it.pos / batch) << 40;
        for (int e = threadIdx.x; e < n; e += 256) {
            const unsigned long long d = htask.get(int64_t(first + e));
            keys[first + e] = b | (d >> 24);
            vals[first + e] = uint32_t(d & 0x3fffffull);
            ++ents;
            units += d & 0x3fffffull;
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        ents += __shfl_xor(ents, sft);
        units += __shfl_xor(units, sft);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], ents);
        atomicAdd(&sums[1], units);
    }
}
__global__ __launch_bounds__(256) void k_tc_sum_u32(const uint32_t *__restrict__ v, unsigned long long n, unsigned long long *__restrict__ out) {
    unsigned long long t = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) t += v[i];
    for (int sft = 32; sft > 0; sft >>= 1) t += __shfl_xor(t, sft);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t);
}
__global__ __launch_bounds__(256) void k_tc_row_hist(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                     const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                     const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                     const unsigned long long *__restrict__ srow, const unsigned long long *__restrict__ trow,
                                                     const TaskList htask, const gmsx_tc_item *__restrict__ hitem, int64_t hitems,
                                                     const TaskList ttask, const gmsx_tc_item *__restrict__ titem, int64_t titems,
                                                     int64_t end, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long h[256];
    for (int i = threadIdx.x; i < 256; i += 256) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    auto add = [&](int cls, unsigned long long units) {
        if (units == 0) return;
        const int b = (cls * 24 + hist_bin(units)) * 2;
        atomicAdd(&h[b], 1ull);
        atomicAdd(&h[b + 1], units);
    };
    for (int64_t q = wave0; q < hitems + titems; q += nwaves) {
        const bool tail = q >= hitems;
        const gmsx_tc_item &it = tail ? titem[q - hitems] : hitem[q];
        const TaskList task = tail ? ttask : htask;
        const int ne = int(it.bc >> 40);
        const int64_t b0 = int64_t(it.bc & 0xffffffffffull);
        if (lane == 0) {
            atomicAdd(&h[240], (unsigned long long)ne);
            atomicAdd(&h[242], 1ull);
            atomicAdd(&h[243], tail ? 4ull * (unsigned long long)(toff[it.pivot + 1] - toff[it.pivot]) : 2ull * (unsigned long long)(hoff[it.pivot + 1] - hoff[it.pivot]));
        }
        for (int i = lane; i < ne; i += 64) {
            const unsigned long long d = task.get(b0 + i);
            if (tail) add(((d >> 22) & 3) == kFormDelta ? 4 : 3, d & 0x3fffffull);
            else add(min(int((d >> 22) & 3), 2), d & 0x3fffffull);  // 12-bit-gap rows are counted with the byte-delta ones
        }
    }
    for (int64_t pos = wave0; pos < end; pos += nwaves) {
        if (!srow || !trow) break;  // the per-vertex descriptors are build-only unless GMSX_TC_KEEP_ROWS=1: no what-if estimates then
        const int32_t u = order[pos];
        const int du = dplus[u];
        if (du < 2) continue;
        const int64_t hb = hoff[u], tb = toff[u];
        const int hl = int(hoff[u + 1] - hb), tl = int(toff[u + 1] - tb);
        const unsigned long long su = (srow[u] & 0x3fffffull) + (trow[u] & 0x3fffffull);
        unsigned long long cur = 0, best = 0, cut = 0;
        for (int i = lane; i < hl + tl; i += 64) {
            int32_t v;
            if (i < hl) {
                const uint32_t x = hadj[hb + i];
                if (x == 0xFFFFu) continue;
                v = int32_t(x);
            } else v = tadj[tb + i - hl];
            const unsigned long long sv = (srow[v] & 0x3fffffull) + (trow[v] & 0x3fffffull);
            cur += sv;
            best += min(su, sv);
            // only the ids of u below v can be in N+(v): member i of u has i of them, about the first i/d+ of u's units
            cut += min(sv, dplus[v] >= kHeavy ? (su * (unsigned long long)i + du - 1) / (unsigned long long)du : ~0ull);
        }
        atomicAdd(&h[du >= kHeavy ? 248 : 250], cur);
        atomicAdd(&h[du >= kHeavy ? 249 : 251], best);
        if (du >= kHeavy) atomicAdd(&h[252], cut);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 256)
        if (h[i]) atomicAdd(&out[i], h[i]);
}

static int tc_one(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st);

static int tc_oriented(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st) {
    if (int rc = ensure_tc(g)) return rc;  // first call on a graph uploaded without GMSX_UPLOAD_FOR_TC: builds the task lists (untimed)
    if (g->tc_passes == 1) return tc_one(g, part, nparts, partial, st);
    // FALLBACK (device_graph.hpp, tc_passes): the containers of all pivots did not fit.  A whole-graph call walks the passes — shard p of
    // tc_passes resident at a time, rebuilt between the passes (untimed like every build; kernel_ms is the sum of the passes' kernels); a
    // sharded call builds exactly its shard.
    if (nparts > 1) {
        // A shard of a graph whose containers need tc_passes passes: the shard itself may not fit either (nparts < tc_passes).  It is cut
        // into J sub-shards of nparts * J — with J even (or 1) the snake rule nests: sub-shard a * nparts + (a even ? part : nparts - 1 - part),
        // a = 0 … J-1, lies inside shard `part` of nparts, and the J of them cover it — built and counted one after the other.  "Slower,
        // never refused": J doubles while a sub-shard does not fit.  On a failure the handle goes back to (pass 0 of tc_passes), not built.
        auto restore = [&](int rc) {
            gmsx_graph *m = const_cast<gmsx_graph *>(g);
            if (m->tc_ready && m->shard_part == 0 && m->shard_nparts == m->tc_passes) return rc;
            (void)ensure_tc_shard(g, 0, g->tc_passes);  // (leaves shard_part / shard_nparts at (0, tc_passes) whether or not the build succeeds)
            return rc;
        };
        int J = nparts >= g->tc_passes ? 1 : (g->tc_passes + nparts - 1) / nparts;
        if (J > 1 && (J & 1)) ++J;
        for (;; J *= 2) {
            uint64_t total = 0;
            gmsx_stats sum{};
            int rc = GMSX_OK;
            for (int a = 0; a < J && rc == GMSX_OK; ++a) {
                const int sub = a * nparts + ((a & 1) ? nparts - 1 - part : part);
                rc = ensure_tc_shard(g, sub, nparts * J);
                if (rc) break;
                uint64_t pp = 0;
                gmsx_stats sp{};
                rc = tc_one(g, sub, nparts * J, &pp, st ? &sp : nullptr);
                total += pp;
                sum.kernel_ms += sp.kernel_ms;
                sum.setup_ms += sp.setup_ms;
                sum.units += sp.units;
                sum.probes += sp.probes;
                sum.stream_bytes += sp.stream_bytes;
                sum.launches += sp.launches;
            }
            if (rc == GMSX_ERR_DEVICE_MEM && nparts * J <= (1 << 20)) continue;  // finer sub-shards
            if (rc) return restore(rc);
            *partial = total;
            if (st) *st = sum;
            return GMSX_OK;
        }
    }
    uint64_t total = 0;
    gmsx_stats sum{};
    for (int p = 0; p < g->tc_passes; ++p) {
        if (int rc = ensure_tc_shard(g, p, g->tc_passes)) return rc;
        uint64_t pp = 0;
        gmsx_stats sp{};
        if (int rc = tc_one(g, p, g->tc_passes, &pp, st ? &sp : nullptr)) return rc;
        total += pp;
        sum.kernel_ms += sp.kernel_ms;
        sum.setup_ms += sp.setup_ms;
        sum.units += sp.units;
        sum.probes += sp.probes;
        sum.stream_bytes += sp.stream_bytes;
        sum.launches += sp.launches;
    }
    *partial = total;
    if (st) {
        sum.alg_elements = g->alg_elements;
        *st = sum;
    }
    return GMSX_OK;
}

static int tc_one(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st) {
    // a sharded call on a FULL upload launches the work items of its shard only (compacted index lists, cached per (part, nparts)); a sharded
    // upload holds nothing but its own items
    const bool use_idx = nparts > 1 && g->shard_nparts == 1;
    if (use_idx)
        if (int rc = tc_shard_items(g, part, nparts)) return rc;
    const gmsx_tc_item *hitem = use_idx ? g->shard_hitem : g->hitem, *titem = use_idx ? g->shard_titem : g->titem;
    int64_t n_hitems = use_idx ? g->shard_hitems : g->hitems, n_titems = use_idx ? g->shard_titems : g->titems;
#ifdef GMSX_DEV_HOOKS  // profiling build only (WRONG counts): the hub items / the tail items / the light pivots alone
    const char *only = std::getenv("GMSX_TC_ONLY");
    if (only) {
        if (std::strcmp(only, "hub") != 0) n_hitems = 0;
        if (std::strcmp(only, "tail") != 0) n_titems = 0;
    }
#else
    const char *only = nullptr;
#endif
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    unsigned long long *acc = g->acc;  // persistent per-graph accumulators: no allocation on the call path
    static_assert(kAccSlots * kAccStride + 16 <= kAccWords, "gmsx_graph::acc too small");
    unsigned int *qhead = reinterpret_cast<unsigned int *>(acc + kAccSlots * kAccStride + 4);  // the two queue heads of k_tc_items (behind the 3 stats words)
    static_assert((kQueueStride + 1) * sizeof(unsigned int) <= 12 * sizeof(unsigned long long), "queue heads past gmsx_graph::acc");
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * kAccWords, s));
    GMSX_HIP(hipEventRecord(c.ev[1], s));

    int launches = 0;
    const int cus = c.compute_units > 0 ? c.compute_units : 256;
    const int64_t cap_blocks = int64_t(cus) * 16;
    const bool run_light = !only || std::strcmp(only, "light") == 0;
    // the light edges of this call: a full upload holds every edge (shard = shard_of(e, nparts)), a sharded one its own, densely
    const bool strided = nparts > 1 && g->shard_nparts == 1;
    const int64_t cnt_light = [&]() -> int64_t {
        if (!run_light) return 0;
        if (!strided) return g->n_ledge;
        const int64_t full = g->n_ledge / nparts, rem = g->n_ledge % nparts;  // whole stripes + the members of the last, partial one
        return full + (((full & 1) ? nparts - 1 - part : part) < rem ? 1 : 0);
    }();
    // LAUNCH PLAN (round 4).  k_tc_items — one persistent launch over the hub and the tail items — then k_tc_light, both on the launch stream.
    // GMSX_TC_OVERLAP=1 puts k_tc_light on a side stream behind the item kernel (its workgroups need no LDS and start wherever a persistent
    // workgroup has left): measured 72.0-72.4 ms against 71.3-71.6 one after the other at scale 26 — the items kernel is bound by memory and
    // the light edges are 10 GB of it — and beside it from the start 72-80.  GMSX_TC_PERSIST=0 (A/B): rounds 2-3's one-workgroup-per-item
    // kernels (with GMSX_TC_OVERLAP=1 the tail items on a side stream of their own).
    const int overlap = [] { const char *e = opt("TC_OVERLAP"); return e ? std::atoi(e) : 0; }();
    const int persist = [] { const char *e = opt("TC_PERSIST"); return e ? std::atoi(e) : 1; }();
    const bool sides = overlap && c.side[0] && c.side[1];
    struct Join {  // joins the side streams on every way out once they were forked (error returns included)
        Ctx &c;
        hipStream_t s;
        bool armed[2] = {false, false};
        ~Join() {
            for (int i = 0; i < 2; ++i)
                if (armed[i] && hipEventRecord(c.ev_join[i], c.side[i]) == hipSuccess) (void)hipStreamWaitEvent(s, c.ev_join[i], 0);
        }
    } join{c, s};
    const bool side_light = sides && cnt_light > 0 && n_hitems + n_titems > 0;
    const bool side_tail = sides && !persist && n_hitems > 0 && n_titems > 0;
    if (side_light || side_tail) GMSX_HIP(hipEventRecord(c.ev_fork, s));
    if (side_tail) {
        GMSX_HIP(hipStreamWaitEvent(c.side[0], c.ev_fork, 0));
        join.armed[0] = true;
    }
    if (side_light) {
        GMSX_HIP(hipStreamWaitEvent(c.side[1], c.ev_fork, 0));
        join.armed[1] = true;
    }
    if (persist) {
        const int wgs_env = [] { const char *e = opt("TC_ITEM_WGS"); return e ? std::atoi(e) : 0; }();  // workgroups per CU (0: what the kernel is compiled for)
        auto grid = [&](int64_t n, int wgs) {
            return unsigned(std::max<int64_t>(1, std::min<int64_t>((n + kGrab - 1) / kGrab, int64_t(cus) * (wgs_env > 0 ? wgs_env : wgs))));
        };
        if (n_hitems > 0) {
            hipLaunchKernelGGL(k_tc_items<false>, dim3(grid(n_hitems, GMSX_TC_HUB_MIN_WAVES)), dim3(256), 0, s, g->hadj, g->tadj, g->spool, g->htask, hitem, int(n_hitems), qhead, acc);
            ++launches;
        }
        if (n_titems > 0) {
            hipLaunchKernelGGL(k_tc_items<true>, dim3(grid(n_titems, GMSX_TC_TAIL_MIN_WAVES)), dim3(256), 0, s, g->hadj, g->tadj, g->tpool, g->ttask, titem, int(n_titems),
                               qhead + kQueueStride, acc);
            ++launches;
        }
    } else {
        if (n_hitems > 0) {
            hipLaunchKernelGGL(k_tc_block, dim3(unsigned(n_hitems)), dim3(256), 0, s, g->hoff, g->hadj, g->spool, g->htask, hitem, acc);
            ++launches;
        }
        if (n_titems > 0) {
            hipLaunchKernelGGL(k_tc_tail, dim3(unsigned(n_titems)), dim3(256), 0, side_tail ? c.side[0] : s, g->toff, g->tadj, g->tpool, g->ttask, titem, acc);
            ++launches;
        }
    }
    if (cnt_light > 0) {
        const int64_t blocks = std::min<int64_t>((cnt_light + 15) / 16, int64_t(cus) * 8);
        hipLaunchKernelGGL(k_tc_light, dim3(unsigned(blocks)), dim3(256), 0, side_light ? c.side[1] : s, g->hadj, g->tadj, g->ledge, strided ? nparts : 1,
                           strided ? part : 0, cnt_light, acc);
        ++launches;
    }
    for (int i = 0; i < 2; ++i)
        if (join.armed[i]) {
            join.armed[i] = false;
            GMSX_HIP(hipEventRecord(c.ev_join[i], c.side[i]));
            GMSX_HIP(hipStreamWaitEvent(s, c.ev_join[i], 0));
        }
    GMSX_HIP(hipEventRecord(c.ev[2], s));
    GMSX_HIP(hipGetLastError());

    const bool need_stats = st && !(g->stats_part == part && g->stats_nparts == nparts);
    if (need_stats) {  // untimed bookkeeping, once per shard (the graph is immutable)
        if (g->n > 0) {
            const int64_t blocks = std::min<int64_t>((g->n + 3) / 4, cap_blocks);
            hipLaunchKernelGGL(k_tc_stats, dim3(unsigned(blocks)), dim3(256), 0, s, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, g->order,
                               g->tunits, g->inline_limit, g->inline_first, g->n, nparts, part, acc + kAccSlots * kAccStride);
        }
        if (g->hitems > 0)
            hipLaunchKernelGGL(k_tc_item_stats, dim3(unsigned(std::min<int64_t>((g->hitems + 3) / 4, cap_blocks))), dim3(256), 0, s, g->hoff, 2, 0, g->htask, g->hitem,
                               g->hitems, nparts, part, acc + kAccSlots * kAccStride);
        if (g->titems > 0)
            hipLaunchKernelGGL(k_tc_item_stats, dim3(unsigned(std::min<int64_t>((g->titems + 3) / 4, cap_blocks))), dim3(256), 0, s, g->toff, 4, 1, g->ttask, g->titem,
                               g->titems, nparts, part, acc + kAccSlots * kAccStride);
    }
    unsigned long long host[kAccSlots * kAccStride + 3];
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
        if (need_stats) {
            g->stats_part = part;
            g->stats_nparts = nparts;
            g->stats_units = host[kAccSlots * kAccStride];
            g->stats_probes = host[kAccSlots * kAccStride + 1];
            g->stats_bytes = host[kAccSlots * kAccStride + 2];
        }
        st->units = g->stats_units;
        st->probes = g->stats_probes;
        st->alg_elements = nparts == 1 ? g->alg_elements : 0;
        st->launches = launches;
        st->reserved = 0;
        st->stream_bytes = g->stats_bytes;
    }
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_tc_divisor(int algo) { return algo == GMSX_TC_FULL ? 3 : 1; }

int gmsx_tc_stream_breakdown(const gmsx_graph *g, uint64_t *out21) {
    return gmsx::guard([&]() -> int {
        if (!g || !out21) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (int rc = ensure_tc(g)) return rc;
        hipStream_t s = ctx().stream;
        std::memset(out21, 0, 21 * sizeof(uint64_t));
        if (g->n == 0) return GMSX_OK;
        int64_t n_block = 0, n_work = 0;
        if (int rc = count_dplus_ge(g, kHeavy, &n_block)) return rc;
        if (int rc = count_dplus_ge(g, 2, &n_work)) return rc;
        unsigned long long *acc = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), 21 * 8));
        struct Guard { void *p; ~Guard() { (void)hipFree(p); } } g1{acc};
        GMSX_HIP(hipMemsetAsync(acc, 0, 21 * 8, s));
        const int cus = ctx().compute_units > 0 ? ctx().compute_units : 256;
        hipLaunchKernelGGL(k_tc_breakdown, dim3(unsigned(cus * 16)), dim3(256), 0, s, g->hoff, g->toff, g->tadj, g->dplus, g->order, g->htask,
                           g->hitem, g->hitems, g->ttask, g->titem, g->titems, g->inline_limit, n_block, n_work, acc);
        GMSX_HIP(hipMemcpyAsync(out21, acc, 21 * 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipGetLastError());
        out21[7] = uint64_t(g->inline_units) * 16ull;                         // the build's figures: the entries do not say what they name
        out21[12] = uint64_t(g->inline_hentries + g->inline_tentries);
        return GMSX_OK;
    });
}

int gmsx_tc_row_histogram(const gmsx_graph *g, uint64_t *out256) {
    return gmsx::guard([&]() -> int {
        uint64_t *out248 = out256;
        if (!g || !out248) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (int rc = ensure_tc(g)) return rc;
        hipStream_t s = ctx().stream;
        std::memset(out248, 0, 256 * sizeof(uint64_t));
        if (g->n == 0) return GMSX_OK;
        unsigned long long *acc = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), 256 * 8));
        struct Guard { void *p; ~Guard() { (void)hipFree(p); } } g1{acc};
        GMSX_HIP(hipMemsetAsync(acc, 0, 256 * 8, s));
        const int cus = ctx().compute_units > 0 ? ctx().compute_units : 256;
        hipLaunchKernelGGL(k_tc_row_hist, dim3(unsigned(cus * 8)), dim3(256), 0, s, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, g->order, g->srow, g->trow,
                           g->htask, g->hitem, g->hitems, g->ttask, g->titem, g->titems, g->n, acc);
        GMSX_HIP(hipGetLastError());
        GMSX_HIP(hipMemcpyAsync(out248, acc, 256 * 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipGetLastError());
        out248[241] = uint64_t(g->inline_hentries + g->inline_tentries);
        return GMSX_OK;
    });
}

int gmsx_tc_comembership(const gmsx_graph *g, int batch, uint64_t *out8) {
    return gmsx::guard([&]() -> int {
        if (!g || !out8 || batch < 1) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (int rc = ensure_tc(g)) return rc;
        hipStream_t s = ctx().stream;
        std::memset(out8, 0, 8 * sizeof(uint64_t));
        const int64_t ne = g->htask_entries;
        if (g->n == 0 || ne == 0 || g->hitems == 0) return GMSX_OK;
        struct Guard { void *p = nullptr; ~Guard() { (void)hipFree(p); } } gk0, gk1, gv0, gv1, gu, ga, gc, gt, gsum;
        unsigned long long *k0 = nullptr, *k1 = nullptr, *uk = nullptr, *cnt = nullptr, *sum = nullptr;
        uint32_t *v0 = nullptr, *v1 = nullptr, *agg = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&k0), size_t(ne) * 8)); gk0.p = k0;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&k1), size_t(ne) * 8)); gk1.p = k1;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&v0), size_t(ne) * 4)); gv0.p = v0;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&v1), size_t(ne) * 4)); gv1.p = v1;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&sum), 4 * 8)); gsum.p = sum;
        GMSX_HIP(hipMemsetAsync(k0, 0xff, size_t(ne) * 8, s));  // entries no item covers (none expected) sort to the end as one key of 0 units
        GMSX_HIP(hipMemsetAsync(v0, 0, size_t(ne) * 4, s));
        GMSX_HIP(hipMemsetAsync(sum, 0, 4 * 8, s));
        hipLaunchKernelGGL(k_tc_comember_keys, dim3(unsigned(std::min<int64_t>(g->hitems, 1 << 20))), dim3(256), 0, s, g->htask, g->hitem, g->hitems, batch, k0, v0, sum);
        size_t tb = 0;
        GMSX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, size_t(ne), 0, 64, s));
        void *tmp = nullptr;
        GMSX_HIP(hipMalloc(&tmp, tb ? tb : 8)); gt.p = tmp;
        GMSX_HIP(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, size_t(ne), 0, 64, s));
        // per distinct (batch, row): the longest cut of the row any pivot of the batch streams (k0 / v0 are free again: outputs)
        uk = k0;
        agg = v0;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&cnt), 8)); gc.p = cnt;
        size_t rb = 0;
        GMSX_HIP(rocprim::reduce_by_key(nullptr, rb, k1, v1, size_t(ne), uk, agg, cnt, rocprim::maximum<uint32_t>(), rocprim::equal_to<unsigned long long>(), s));
        void *tmp2 = nullptr;
        Guard g2;
        GMSX_HIP(hipMalloc(&tmp2, rb ? rb : 8)); g2.p = tmp2;
        GMSX_HIP(rocprim::reduce_by_key(tmp2, rb, k1, v1, size_t(ne), uk, agg, cnt, rocprim::maximum<uint32_t>(), rocprim::equal_to<unsigned long long>(), s));
        unsigned long long nuniq = 0;
        GMSX_HIP(hipMemcpyAsync(&nuniq, cnt, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_tc_sum_u32, dim3(1024), dim3(256), 0, s, agg, (unsigned long long)nuniq, sum + 2);
        unsigned long long h[4] = {0, 0, 0, 0};
        GMSX_HIP(hipMemcpyAsync(h, sum, sizeof(h), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipGetLastError());
        out8[0] = uint64_t(h[0]);   // entries the hub items cover
        out8[1] = uint64_t(h[1]);   // their 16-byte units: what the hub items stream per pass
        out8[2] = uint64_t(nuniq);  // distinct (batch, row) pairs
        out8[3] = uint64_t(h[2]);   // units if every row were streamed once per batch (at its longest cut)
        out8[4] = uint64_t(g->hitems);
        return GMSX_OK;
    });
}

int gmsx_tc_partial(const gmsx_graph *g, int algo, int part, int nparts, uint64_t *partial, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !partial || nparts < 1 || part < 0 || part >= nparts) return GMSX_ERR_INVALID;
        if (algo != GMSX_TC_AUTO && algo != GMSX_TC_ORIENTED && algo != GMSX_TC_FULL) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (algo == GMSX_TC_FULL) return tc_full_partial(g, part, nparts, partial, stats);
        // a sharded upload holds the task lists of ONE shard: that is the only one it can count
        if (g->tc_passes == 1 && g->shard_nparts > 1 && (nparts != g->shard_nparts || part != g->shard_part)) return GMSX_ERR_INVALID;
        return tc_oriented(g, part, nparts, partial, stats);
    });
}

int gmsx_tc_total(const gmsx_graph *g, int algo, uint64_t *triangles, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!triangles) return GMSX_ERR_INVALID;
        uint64_t partial = 0;
        if (int rc = gmsx_tc_partial(g, algo, 0, 1, &partial, stats)) return rc;
        const uint64_t div = uint64_t(gmsx_tc_divisor(algo));
        if (partial % div != 0) return GMSX_ERR_KERNEL;  // the reference asserts total % 3 == 0 (parallel/total.h:22)
        *triangles = partial / div;
        return GMSX_OK;
    });
}

}  // extern "C"
