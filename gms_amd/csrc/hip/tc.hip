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
//   4. k_tc_wave — light pivots with far light members: wave per pivot, private bitmap + bucket set, streams those members' rows.
//   The three kernels run side by side on three streams (hub items: bandwidth + VALU; tail items: short rows; light pivots: latency).
// No MFMA: integer/indexing work bounded by row streaming (HBM) and VALU issue of the decode + probe sequence.
#include "device_graph.hpp"

#include <algorithm>
#include <type_traits>
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

// The LIGHT-PIVOT kernel's row scan (k_tc_wave; the work-item kernels have their own typed loops below): streams up to 64 stream rows
// against the wave's LDS bitmap.  Lane l holds the packed descriptor of one row (srow[v]; 0 = no row).  A wave works as four 16-lane
// groups, each on its own row: one 16-byte unit per lane per step, two steps in flight.  Rows are handed out FORM BY FORM (a ballot per
// form, then the four lowest lanes of the ballot): the forms cost 12 / 32 / 81 VALU instructions per unit, and a hand-out that mixes them
// executes every branch with a quarter of the lanes.
struct RowHandout {
    const uint4 *row;
    int units;
};
__device__ __forceinline__ RowHandout take_rows(unsigned long long &todo, const uint32_t *__restrict__ pool, uint32_t dlo, uint32_t dhi, int grp) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo[k] = 0;
        hi[k] = 0;
        if (todo) {  // wave-uniform
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            lo[k] = __builtin_amdgcn_readlane(dlo, i);
            hi[k] = __builtin_amdgcn_readlane(dhi, i);
        }
    }
    const uint32_t l = grp == 0 ? lo[0] : grp == 1 ? lo[1] : grp == 2 ? lo[2] : lo[3];
    const uint32_t h = grp == 0 ? hi[0] : grp == 1 ? hi[1] : grp == 2 ? hi[2] : hi[3];
    return RowHandout{reinterpret_cast<const uint4 *>(pool) + ((uint64_t(h) << 8) | (l >> 24)), int(l & 0x3fffffu)};
}
__device__ __forceinline__ uint32_t scan_srows(const uint32_t *bm, const uint32_t *__restrict__ spool, unsigned long long desc, int lane) {
    const int grp = lane >> 4, sub = lane & 15;
    const uint32_t dlo = uint32_t(desc), dhi = uint32_t(desc >> 32);
    const bool any = (dlo & 0x3fffffu) != 0;
    const uint32_t form = (dlo >> 22) & 3u;
    uint32_t cnt = 0;
    for (unsigned long long todo = __ballot(any && form == kFormBitset); todo;) {  // AND the bitset with the pivot bitmap, 128 ids per unit
        const RowHandout r = take_rows(todo, spool, dlo, dhi, grp);
        for (int j = sub; j < r.units; j += 32) {
            const uint4 p = r.row[j];
            const uint4 q = *reinterpret_cast<const uint4 *>(bm + 4 * j);
            uint32_t c = uint32_t(__popc(p.x & q.x) + __popc(p.y & q.y) + __popc(p.z & q.z) + __popc(p.w & q.w));
            if (j + 16 < r.units) {
                const uint4 p2 = r.row[j + 16];
                const uint4 q2 = *reinterpret_cast<const uint4 *>(bm + 4 * (j + 16));
                c += uint32_t(__popc(p2.x & q2.x) + __popc(p2.y & q2.y) + __popc(p2.z & q2.z) + __popc(p2.w & q2.w));
            }
            cnt += c;
        }
    }
    for (unsigned long long todo = __ballot(any && form == kFormDelta); todo;) {
        const RowHandout r = take_rows(todo, spool, dlo, dhi, grp);
        int j = sub;
        for (; j + 16 < r.units; j += 32) {
            const uint4 p = r.row[j], q = r.row[j + 16];
            cnt += delta_unit_hits(bm, p);
            cnt += delta_unit_hits(bm, q);
        }
        if (j < r.units) cnt += delta_unit_hits(bm, r.row[j]);
    }
    for (unsigned long long todo = __ballot(any && form == kFormGap12); todo;) {
        const RowHandout r = take_rows(todo, spool, dlo, dhi, grp);
        int j = sub;
        for (; j + 16 < r.units; j += 32) {
            const uint4 p = r.row[j], q = r.row[j + 16];
            cnt += gap12_unit_hits(bm, p);
            cnt += gap12_unit_hits(bm, q);
        }
        if (j < r.units) cnt += gap12_unit_hits(bm, r.row[j]);
    }
    for (unsigned long long todo = __ballot(any && form == kFormList); todo;) {  // 16-bit list, 8 ids per unit, filler 0xFFFF
        const RowHandout r = take_rows(todo, spool, dlo, dhi, grp);
        int j = sub;
        for (; j + 16 < r.units; j += 32) {
            const uint4 p = r.row[j], q = r.row[j + 16];
            cnt += hub_hits8(bm, u4u{p.x, p.y, p.z, p.w});
            cnt += hub_hits8(bm, u4u{q.x, q.y, q.z, q.w});
        }
        if (j < r.units) {
            const uint4 p = r.row[j];
            cnt += hub_hits8(bm, u4u{p.x, p.y, p.z, p.w});
        }
    }
    return cnt;
}

// ---- tail side: open-addressing hash set in LDS (keys are rank ids >= kHub; -1 = empty) ------------------------
__device__ __forceinline__ uint32_t hash_slot(int32_t w, int shift) { return (uint32_t(w) * 0x9E3779B1u) >> shift; }

__device__ __forceinline__ void set_insert(int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift);
    while (atomicCAS(&tbl[h], -1, w) != -1) h = (h + 1) & mask;
}
__device__ __forceinline__ uint32_t set_contains(const int32_t *tbl, uint32_t mask, int shift, int32_t w) {
    uint32_t h = hash_slot(w, shift);
    while (true) {
        const int32_t x = tbl[h];
        if (x == w) return 1u;
        if (x == -1) return 0u;
        h = (h + 1) & mask;
    }
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
__device__ __forceinline__ uint32_t flt_bit(const uint32_t *flt, uint32_t, uint32_t id) { return __builtin_amdgcn_ubfe(flt[(id >> 5) & uint32_t(kFilterWords - 1)], id, 1u); }
__device__ __forceinline__ uint32_t tail_unit_hits(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p) {
    const uint32_t m = flt_bit(flt, mask, p.x) | (flt_bit(flt, mask, p.y) << 1) | (flt_bit(flt, mask, p.z) << 2) | (flt_bit(flt, mask, p.w) << 3);
    uint32_t c = 0;
    if (m) {  // rare: exact membership for the ids the filter let through (the filler -2 may pass the filter, it is never a key)
        if (m & 1u) c += set_contains(tbl, mask, shift, int32_t(p.x));
        if (m & 2u) c += set_contains(tbl, mask, shift, int32_t(p.y));
        if (m & 4u) c += set_contains(tbl, mask, shift, int32_t(p.z));
        if (m & 8u) c += set_contains(tbl, mask, shift, int32_t(p.w));
    }
    return c;
}
// one unit of the 16-bit delta form: 32-bit base, count (low half of word 1), five 16-bit gaps
__device__ __forceinline__ uint32_t tail_delta_unit_hits(const uint32_t *flt, const int32_t *tbl, uint32_t mask, int shift, uint4 p) {
    const uint32_t id0 = p.x, id1 = id0 + (p.y >> 16), id2 = id1 + (p.z & 0xffffu), id3 = id2 + (p.z >> 16), id4 = id3 + (p.w & 0xffffu),
                   id5 = id4 + (p.w >> 16);
    const uint32_t n = p.y & 0xffu;
    const uint32_t m = (flt_bit(flt, mask, id0) | (flt_bit(flt, mask, id1) << 1) | (flt_bit(flt, mask, id2) << 2) | (flt_bit(flt, mask, id3) << 3) |
                        (flt_bit(flt, mask, id4) << 4) | (flt_bit(flt, mask, id5) << 5)) & ((1u << n) - 1u);
    uint32_t c = 0;
    if (m) {
        if (m & 1u) c += set_contains(tbl, mask, shift, int32_t(id0));
        if (m & 2u) c += set_contains(tbl, mask, shift, int32_t(id1));
        if (m & 4u) c += set_contains(tbl, mask, shift, int32_t(id2));
        if (m & 8u) c += set_contains(tbl, mask, shift, int32_t(id3));
        if (m & 16u) c += set_contains(tbl, mask, shift, int32_t(id4));
        if (m & 32u) c += set_contains(tbl, mask, shift, int32_t(id5));
    }
    return c;
}
// Bucketed tail set for the light-pivot kernel (<= 63 keys): 64 buckets x 4 slots, 16-byte aligned, so a probe is ONE
// ds_read_b128 and four compares -- no probe loop, no divergence, and the four probes of a 16-byte load are
// independent.  A pivot whose keys overflow a bucket (five keys with the same hash) falls back to the open-addressing
// table above, built in the same 1 KB.
template <int NB>
__device__ __forceinline__ uint32_t bucket_of(int32_t w) {  // NB buckets (power of two)
    return NB == 64 ? (uint32_t(w) ^ (uint32_t(w) >> 6)) & 63u : (uint32_t(w) ^ (uint32_t(w) >> 9)) & uint32_t(NB - 1);
}
template <int NB>
__device__ __forceinline__ uint32_t bucket_contains(const int32_t *tbl, int32_t w) {
    const int4 b = *reinterpret_cast<const int4 *>(tbl + bucket_of<NB>(w) * 4);
    return uint32_t((b.x == w) | (b.y == w) | (b.z == w) | (b.w == w));
}
// Tail stream rows against an arbitrary exact membership probe (the light-pivot kernel: bucket set or open-addressing table, no filter).
template <class Probe>
__device__ __forceinline__ uint32_t tail_unit_probe(Probe probe, uint4 p, int form) {
    if (form == kFormDelta) {
        const uint32_t id0 = p.x, id1 = id0 + (p.y >> 16), id2 = id1 + (p.z & 0xffffu), id3 = id2 + (p.z >> 16), id4 = id3 + (p.w & 0xffffu),
                       id5 = id4 + (p.w >> 16);
        const uint32_t n = p.y & 0xffu;
        return probe(int32_t(id0)) + (n > 1 ? probe(int32_t(id1)) : 0u) + (n > 2 ? probe(int32_t(id2)) : 0u) + (n > 3 ? probe(int32_t(id3)) : 0u) +
               (n > 4 ? probe(int32_t(id4)) : 0u) + (n > 5 ? probe(int32_t(id5)) : 0u);
    }
    return probe(int32_t(p.x)) + probe(int32_t(p.y)) + probe(int32_t(p.z)) + probe(int32_t(p.w));  // the filler -2 is never a key
}
template <class Probe>
__device__ __forceinline__ uint32_t scan_trows_probe(Probe probe, const uint32_t *__restrict__ tpool, unsigned long long desc, int rows, int lane) {
    const int grp = lane >> 4, sub = lane & 15;
    uint32_t cnt = 0;
    for (int r0 = 0; r0 < rows; r0 += 4) {
        const int m0 = r0 & 63, m1 = (r0 + 1) & 63, m2 = (r0 + 2) & 63, m3 = (r0 + 3) & 63;
        const uint32_t lo0 = __builtin_amdgcn_readlane(uint32_t(desc), m0), lo1 = __builtin_amdgcn_readlane(uint32_t(desc), m1),
                       lo2 = __builtin_amdgcn_readlane(uint32_t(desc), m2), lo3 = __builtin_amdgcn_readlane(uint32_t(desc), m3);
        if (((lo0 | lo1 | lo2 | lo3) & 0x3fffffu) == 0) continue;  // wave-uniform
        const uint32_t hi0 = __builtin_amdgcn_readlane(uint32_t(desc >> 32), m0), hi1 = __builtin_amdgcn_readlane(uint32_t(desc >> 32), m1),
                       hi2 = __builtin_amdgcn_readlane(uint32_t(desc >> 32), m2), hi3 = __builtin_amdgcn_readlane(uint32_t(desc >> 32), m3);
        const uint32_t lo = grp == 0 ? lo0 : grp == 1 ? lo1 : grp == 2 ? lo2 : lo3;
        const uint32_t hi = grp == 0 ? hi0 : grp == 1 ? hi1 : grp == 2 ? hi2 : hi3;
        const int units = int(lo & 0x3fffffu), form = int((lo >> 22) & 3u);
        const uint4 *row = reinterpret_cast<const uint4 *>(tpool) + ((uint64_t(hi) << 8) | (lo >> 24));
        int j = sub;
        for (; j + 16 < units; j += 32) {
            const uint4 p = row[j], q = row[j + 16];
            cnt += tail_unit_probe(probe, p, form);
            cnt += tail_unit_probe(probe, q, form);
        }
        if (j < units) cnt += tail_unit_probe(probe, row[j], form);
    }
    return cnt;
}

// ---------------------------------------------------------------------------------------------
// Work items (device_graph.hpp): one workgroup per item = up to kTaskChunk consecutive entries of ONE pivot's hub-entry list (k_tc_block:
// the pivot's hub part staged as the 65536-bit bitmap, 8 KB) or tail-entry list (k_tc_tail: the pivot's tail part as filter + hash set,
// tiled if longer than half the table).  An entry is one 8-byte stream-row descriptor — a member's row, the cut row of an in-neighbour
// that handed its edge over, a 64-unit chunk of an inline row: the kernels cannot tell and need not.
// The lists are laid out CLASS BY CLASS at build time (class = form x ceil(log2 units)), so an item is a handful of runs of equally
// formed, similarly long rows.  The workgroup copies the item's descriptors to LDS, marks where each class begins and ends, and then runs
// one COMPILE-TIME-SHAPED loop per class: groups of W = 4 / 8 / 16 lanes (rows of <= 4 / <= 8 / more units), group g taking entries
// g, g + 256/W, … with the next descriptor already loaded while the current row is scanned.  No form ballots, no readlane hand-outs, no
// mixed rows in a wave: what was ~30 VALU instructions per four rows is one LDS read and an address computation per row.
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
#define GMSX_TC_TAIL_DEPTH 3  // 16-byte loads a lane of the TAIL scans keeps in flight
#endif
#ifndef GMSX_TC_HUB_DEPTH
#define GMSX_TC_HUB_DEPTH 2   // … of the hub scans (the decode of a byte-delta unit wants 30 registers of its own)
#endif
// Entries [lo, hi) of an item (descriptors in LDS), one form, rows of one width class, as a STREAM OF STEPS: a group of W lanes works on
// one row, a step = W consecutive units of it (one 16-byte load per lane), and a lane keeps D steps in flight across row boundaries.
// Every load is UNCONDITIONAL — a lane without a unit in the step re-reads unit 0 of its row, a group past its last entry unit 0 of the
// pool — because a load under a divergent branch cannot be counted: hipcc then waits with vmcnt(0) at every use, i.e. for the load it
// issued a moment ago, and a wave never has more than one load instruction in flight (rounds 2-3: that, times 32 waves per CU, was the
// 3.7 TB/s of the tail items — 0.9 KB per wave and memory latency).  With counted waits the D loads overlap.
// HIT(p, j) = hits of unit j of a row.
template <int W, int D, class Hit>
__device__ __forceinline__ uint32_t scan_run(const unsigned long long *sdesc, const uint32_t *__restrict__ pool, int lo, int hi, int tid, Hit hit) {
    constexpr int G = 256 / W;
    const int sub = tid % W;
    const uint4 *pool4 = reinterpret_cast<const uint4 *>(pool);
    uint32_t cnt = 0;
    // cursor of the step to ISSUE: entry e (descriptor d = first unit << 24 | form << 22 | units), unit j of this lane
    int e = lo + tid / W;
    unsigned long long d = e < hi ? sdesc[e] : 0ull;
    int j = sub;
    int pending = 0;  // steps of real entries in the ring
    uint4 p[D];
    int pj[D];  // unit index of the lane in that step, -1 = none
    auto issue = [&](int k) {
        const int units = int(uint32_t(d) & 0x3fffffu);
        const bool live = j < units;
        p[k] = pool4[(d >> 24) + (unsigned long long)(live ? j : 0)];
        pj[k] = live ? j : -1;
        pending += e < hi ? 1 : 0;
        j += W;
        if (j - sub >= units) {  // the row is through (uniform per group): next entry of the group
            e += G;
            d = e < hi ? sdesc[e] : 0ull;
            j = sub;
        }
    };
#pragma unroll
    for (int k = 0; k < D; ++k) issue(k);
    while (pending > 0) {  // the groups of a wave differ by the lengths of their rows
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const uint4 pc = p[k];
            const int jc = pj[k];
            pending -= 1;  // (slots of dead entries push it below zero: the loop ends at the first check after the last real step)
            issue(k);
            if (jc >= 0) cnt += hit(pc, jc);
        }
    }
    return cnt;
}
// One run per FORM (round 4): with the step streams a group moves from row to row on its own, so rows of unequal length side by side
// cost nothing, and a single width serves them all — kRowGroup = 8 lanes = one 128-byte line per row and step (rows average 40 units in
// the hub lists, 14 in the tail lists; 16 lanes left the short ones half empty).  Rounds 2-3 ran three loops per form (4 / 8 / 16 lanes):
// every loop has a prologue and a drain, and a typical item of ~110 rows paid for eight of them.
#ifndef GMSX_TC_HUB_GROUP
#define GMSX_TC_HUB_GROUP 16
#endif
#ifndef GMSX_TC_TAIL_GROUP
#define GMSX_TC_TAIL_GROUP 16
#endif
template <int W, int D, class Hit>
__device__ __forceinline__ uint32_t scan_form(const unsigned long long *sdesc, const uint32_t *__restrict__ pool, const unsigned short *rbeg, const unsigned short *rend,
                                              int form, int tid, Hit hit) {
    // the runs of a form are adjacent in the class-sorted list: [first non-empty begin, last non-empty end)
    const int r0 = form * 3;
    int lo = 0x7fffffff, hi = 0;
#pragma unroll
    for (int r = r0; r < r0 + 3; ++r)
        if (rend[r] > rbeg[r]) {
            lo = min(lo, int(rbeg[r]));
            hi = max(hi, int(rend[r]));
        }
    return hi > lo ? scan_run<W, D>(sdesc, pool, lo, hi, tid, hit) : 0u;
}
// copies the item's descriptors to LDS and records, per run type, where its run begins and ends (the list is class-sorted: one run each)
template <int NC, class ClassOf>
__device__ __forceinline__ void stage_item(const unsigned long long *__restrict__ ent, int ne, int tid, unsigned long long *sdesc, unsigned short *cbeg,
                                           unsigned short *cend, ClassOf class_of) {
    if (tid < NC) cbeg[tid] = cend[tid] = 0;
    for (int i = tid; i < ne; i += 256) sdesc[i] = ent[i];
    __syncthreads();
    for (int i = tid; i < ne; i += 256) {
        const int c = class_of(sdesc[i]);
        if (i == 0 || class_of(sdesc[i - 1]) != c) cbeg[c] = (unsigned short)i;
        if (i == ne - 1 || class_of(sdesc[i + 1]) != c) cend[c] = (unsigned short)(i + 1);
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
                                                  const uint32_t *__restrict__ spool, const unsigned long long *__restrict__ htask,
                                                  const gmsx_tc_item *__restrict__ items, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) uint32_t bm[kBitmapWords + 128];  // + slack: the delta probes of unused slots read up to 104 words past the bitmap
    __shared__ unsigned long long sdesc[kTaskChunk];
    __shared__ unsigned short cbeg[12], cend[12];  // run types
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gmsx_tc_item &it = items[blockIdx.x];
    const int32_t u = it.pivot;
    const int64_t hb = hoff[u];
    const int hl = int(hoff[u + 1] - hb);
    const int ne = int(it.bc >> 40);
    for (int i = tid; i < (kBitmapWords + 128) / 4; i += 256) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    stage_item<12>(htask + (it.bc & 0xffffffffffull), ne, tid, sdesc, cbeg, cend, [](unsigned long long d) { return run_type(d); });
    for (int i = tid; i < hl; i += 256) {
        const uint32_t id = hadj[hb + i];
        if (id != 0xFFFFu) atomicOr(&bm[id >> 5], 1u << (id & 31u));
    }
    __syncthreads();
    uint32_t cnt = 0;
    auto run = [&](auto form_tag) {
        constexpr int FORM = decltype(form_tag)::value;
        cnt += scan_form<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH>(sdesc, spool, cbeg, cend, FORM, tid, [](uint4 p, int j) { return hub_unit_hits<FORM>(bm, p, j); });
    };
#ifndef GMSX_TC_STAGING_ONLY  // (A/B build: what the per-item fixed cost alone takes)
    run(std::integral_constant<int, kFormList>{});
    run(std::integral_constant<int, kFormBitset>{});
    run(std::integral_constant<int, kFormDelta>{});
    run(std::integral_constant<int, kFormGap12>{});
#endif
    block_add(cnt, red, lane, wave, tid, acc);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tc_tail(const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                 const uint32_t *__restrict__ tpool, const unsigned long long *__restrict__ ttask,
                                                 const gmsx_tc_item *__restrict__ items, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) int32_t tbl[1 << kBlockLog];
    __shared__ __attribute__((aligned(16))) uint32_t flt[kFilterWords];
    __shared__ unsigned long long sdesc[kTaskChunk];
    __shared__ unsigned short cbeg[12], cend[12];  // run types (tail forms: list = 0, delta = 2)
    __shared__ unsigned long long red[4];
    constexpr int TILE = (1 << kBlockLog) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gmsx_tc_item &it = items[blockIdx.x];
    const int32_t u = it.pivot;
    const int64_t tb = toff[u];
    const int tl = int(toff[u + 1] - tb);
    const int ne = int(it.bc >> 40);
    stage_item<12>(ttask + (it.bc & 0xffffffffffull), ne, tid, sdesc, cbeg, cend, [](unsigned long long d) { return run_type(d); });
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
        auto run = [&](auto form_tag) {
            constexpr int FORM = decltype(form_tag)::value;
            cnt += scan_form<GMSX_TC_TAIL_GROUP, GMSX_TC_TAIL_DEPTH>(sdesc, tpool, cbeg, cend, FORM, tid, [mask, shift](uint4 p, int) { return tail_unit_hits_f<FORM>(flt, tbl, mask, shift, p); });
        };
#ifndef GMSX_TC_STAGING_ONLY
        run(std::integral_constant<int, kFormList>{});
        run(std::integral_constant<int, kFormDelta>{});
#endif
    }
    block_add(cnt, red, lane, wave, tid, acc);
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT work-item kernel (round 4): what k_tc_block + k_tc_tail did with one workgroup per item — 5.4 M workgroups per pass at scale
// 26, each opening with a chain of dependent round trips (item -> offsets -> descriptors / pivot ids -> LDS) before its first row load:
// a third of a tail item's slot time — as ONE launch of as many workgroups as the chip holds.  A workgroup walks items off two queues
// (hub items, tail items; tickets of kGrab consecutive items, fetched one chunk ahead) and keeps a three-deep pipeline over them:
//   item i+2   its 64-byte record is loading (one dword per lane; fields come back out with v_readlane: the record is wave-uniform)
//   item i+1   its descriptors and the pivot's ids are on their way into the OTHER half of the LDS staging buffers by LDS-DMA
//              (global_load_lds: no registers, no instructions at arrival)
//   item i     is scanned.
// Records are self-contained (device_graph.hpp: entries, the pivot's container part, run boundaries), so nothing in the chain depends on
// a second lookup.  The pivot bitmap is MAINTAINED instead of rebuilt: consecutive items of one pivot share it; otherwise the ids of the
// previous pivot (still in the other staging half) are XORed out and the new ones XORed in — both commute, so no barrier separates them
// and nothing clears 8 KB per item.  Tail items alias the bitmap's LDS as filter + hash set (sized per pivot) and leave it invalid.
// Roles: (blockIdx & 7) < tail_share starts on the tail queue, the rest on the hub queue — the latency-bound short tail rows run BESIDE
// the bandwidth-bound hub rows for the whole pass — and a workgroup whose queue runs dry moves to the other one.
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
    for (int base = wave * 64; base < ndw; base += 256)  // uniform per wave
        if (base + lane < ndw) glds_dword(src + base + lane, lds_byte_addr + 4u * uint32_t(base));
}
// run boundaries of the item whose record sits in LDS (dwords 6 … 12 of the record: thirteen 16-bit positions; uniform)
__device__ __forceinline__ int run_bound(const uint32_t *rec, int r) {
    const uint32_t w = uni32(rec[6 + (r >> 1)]);
    return int((r & 1) ? (w >> 16) : (w & 0xffffu));
}
template <int W, int D, class Hit>
__device__ __forceinline__ uint32_t scan_form_r(const unsigned long long *sdesc, const uint32_t *__restrict__ pool, const uint32_t *rec, int form, int tid, Hit hit) {
    uint32_t cnt = 0;
    const int r0 = form * 3;
    const int b0 = run_bound(rec, r0), b3 = run_bound(rec, r0 + 3);
    if (b3 > b0) cnt += scan_run<W, D>(sdesc, pool, b0, b3, tid, hit);
    return cnt;
}
// LDS of a k_tc_items workgroup (18.9 KB: eight per CU)
struct ItemLds {
    uint32_t bm[kBitmapWords + 128];  // hub items: the pivot bitmap (+ slack: the delta probes of unused slots read up to 104 words past it); tail items: table | filter
    unsigned long long sdesc[2][kTaskChunk];
    uint32_t idst[2][kIdStage];
    uint32_t rec[3][16];  // records of the items A (scanned), B (staging), C (arriving)
    unsigned long long red[4];
    int chunk[2];  // first item of the chunk behind the one the cursor walks (-1: the queue is dry)
};
// One queue (hub items or tail items) until it is dry.
template <bool TAIL>
__device__ __forceinline__ void item_loop(ItemLds &L, const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ pool,
                                          const unsigned long long *__restrict__ task, const gmsx_tc_item *__restrict__ items, int n_items,
                                          unsigned int *__restrict__ qhead, unsigned long long &total) {
    constexpr int TILE = (1 << kBlockLog) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *bm = L.bm;
    const uint32_t a_sdesc = lds_addr(&L.sdesc[0][0]), a_idst = lds_addr(&L.idst[0][0]), a_rec = lds_addr(&L.rec[0][0]);
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
        stage_dwords(reinterpret_cast<const uint32_t *>(task + first), a_sdesc + uint32_t(b) * uint32_t(sizeof(L.sdesc[0])), 2 * ne, lane, wave);
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
        const unsigned long long *sd = L.sdesc[buf];
        uint32_t c = 0;
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
#ifndef GMSX_TC_STAGING_ONLY
            c += scan_form_r<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH>(sd, pool, recA, kFormList, tid, [bm](uint4 p, int j) { return hub_unit_hits<kFormList>(bm, p, j); });
            c += scan_form_r<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH>(sd, pool, recA, kFormBitset, tid, [bm](uint4 p, int j) { return hub_unit_hits<kFormBitset>(bm, p, j); });
            c += scan_form_r<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH>(sd, pool, recA, kFormDelta, tid, [bm](uint4 p, int j) { return hub_unit_hits<kFormDelta>(bm, p, j); });
            c += scan_form_r<GMSX_TC_HUB_GROUP, GMSX_TC_HUB_DEPTH>(sd, pool, recA, kFormGap12, tid, [bm](uint4 p, int j) { return hub_unit_hits<kFormGap12>(bm, p, j); });
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
                for (int i = tid; i < tn; i += 256) {
                    const int32_t t = t0 + i < kIdStage ? int32_t(L.idst[buf][t0 + i]) : tadj[cb + t0 + i];
                    set_insert(tbl, mask, shift, t);
                    atomicOr(&flt[(uint32_t(t) >> 5) & uint32_t(kFilterWords - 1)], 1u << (uint32_t(t) & 31u));
                }
                __syncthreads();  // (2)
                if (!staged && vB) stage(L.rec[rb], buf ^ 1, lane, wave);
                staged = true;
#ifndef GMSX_TC_STAGING_ONLY
                c += scan_form_r<GMSX_TC_TAIL_GROUP, GMSX_TC_TAIL_DEPTH>(sd, pool, recA, kFormList, tid, [=](uint4 p, int) { return tail_unit_hits_f<kFormList>(flt, tbl, mask, shift, p); });
                c += scan_form_r<GMSX_TC_TAIL_GROUP, GMSX_TC_TAIL_DEPTH>(sd, pool, recA, kFormDelta, tid, [=](uint4 p, int) { return tail_unit_hits_f<kFormDelta>(flt, tbl, mask, shift, p); });
#endif
            }
            if (!staged) {  // (a pivot without tail ids: nothing can match)
                __syncthreads();
                if (vB) stage(L.rec[rb], buf ^ 1, lane, wave);
            }
        }
        total += c;
        if (fetch) L.chunk[1] = pend < unsigned(n_items) ? int(pend) : -1;  // the chunk behind the one the cursor just entered
        ra = rb;
        vA = vB;
        vB = vC;
        buf ^= 1;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tc_items(
    const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ spool, const uint32_t *__restrict__ tpool,
    const unsigned long long *__restrict__ htask, const unsigned long long *__restrict__ ttask, const gmsx_tc_item *__restrict__ hitem, int n_hitems,
    const gmsx_tc_item *__restrict__ titem, int n_titems, int tail_share, unsigned int *__restrict__ qhead, unsigned long long *__restrict__ acc) {
    __shared__ __attribute__((aligned(16))) ItemLds L;
    const int tid = threadIdx.x;
    const bool tail_first = (int(blockIdx.x) & 7) < tail_share;
    unsigned long long total = 0;
#pragma nounroll
    for (int phase = 0; phase < 2; ++phase) {  // a workgroup whose queue has run dry moves to the other one
        if ((phase == 0) == tail_first) {
            if (n_titems > 0) item_loop<true>(L, hadj, tadj, tpool, ttask, titem, n_titems, qhead + kQueueStride, total);
        } else {
            if (n_hitems > 0) item_loop<false>(L, hadj, tadj, spool, htask, hitem, n_hitems, qhead, total);
        }
    }
    block_add(total, L.red, tid & 63, tid >> 6, tid, acc);
}

// ---------------------------------------------------------------------------------------------
// Light pivots (2 <= d+ < 64): the rows of their FAR, LIGHT members (rank id >= inline_limit and d+ < 64).  Every other edge of a light
// pivot was handed over to the member at upload (inline rows, device_graph.hpp) and is counted by k_tc_block.  Each of the 4 waves of a
// workgroup owns a private bitmap (8 KB) and a 64 x 4 bucket set for the tail part and walks its pivots with a grid
// stride; pivots without tail members are skipped.  The bitmap is cleared once; each pivot sets its bits and clears
// exactly those words again.
// The dependent loads  order[] -> offsets -> member ids -> row extents  form a four-deep software pipeline over the
// pivots of the wave: while pivot D is scanned, the members and row extents of pivot C, the offsets of B and the id
// of A are in flight, so a pivot pays only the round trips of its own row scans.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tc_wave(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                 const uint32_t *__restrict__ spool, const uint32_t *__restrict__ tpool,
                                                 const unsigned long long *__restrict__ tdesc, const int64_t *__restrict__ toff,
                                                 const int32_t *__restrict__ tadj, const int32_t *__restrict__ tsplit,
                                                 const int32_t *__restrict__ order, int64_t first, int64_t end, int nparts, int part,
                                                 unsigned long long *__restrict__ acc) {
    constexpr int LOG = 8, SIZE = 1 << LOG, SHIFT = 32 - LOG;
    constexpr uint32_t MASK = SIZE - 1;
    __shared__ __attribute__((aligned(16))) uint32_t bm_all[4 * kBitmapWords + 128];  // + slack: the delta probes of unused slots read up to 104 words past a bitmap
    __shared__ __attribute__((aligned(16))) int32_t tbl_all[4 * SIZE];
    __shared__ uint32_t fill_all[4 * 64];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *bm = bm_all + wave * kBitmapWords;
    int32_t *tbl = tbl_all + wave * SIZE;
    uint32_t *fill = fill_all + wave * 64;
    for (int i = lane; i < kBitmapWords; i += 64) bm[i] = 0;
    const int64_t step = int64_t(gridDim.x) * 4 * nparts;
    const int64_t pos0 = first + (int64_t(blockIdx.x) * 4 + wave) * nparts + part;
    unsigned long long cnt = 0;

    // The dependent loads  order[] -> offsets -> (member ids, stream-row descriptors)  form a three-deep software pipeline over the
    // pivots of the wave.  The descriptors of a pivot's tail members sit next to the members (tdesc), so they no longer wait for the
    // member ids: stage C loads ids and descriptors together, stage D scans.
    int64_t hbC = 0, tbC = 0, hbB = 0, tbB = 0;
    int hlC = 0, tlC = 0, hlB = 0, tlB = 0;
    int tsC = 0, tsB = 0, tsD = 0;  // members below inline_limit come first in a tail row: handed over
    int32_t uA = -1;
    int tlD = 0;
    uint32_t hvD = 0xFFFFu;
    int32_t vD = -1;
    unsigned long long dsD = 0, dtD = 0;  // stream-row descriptors (hub part, tail part) of this lane's far member; 0 = nothing to stream
    auto load_members = [&](int64_t hb, int hl, int64_t tb, int tl, int ts, uint32_t &hv, int32_t &v, unsigned long long &ds, unsigned long long &dt) {
        hv = 0xFFFFu; v = -1; ds = 0; dt = 0;
        if (tl > 0) {
            if (lane < hl) hv = hadj[hb + lane];
            if (lane < tl) {
                v = tadj[tb + lane];
                if (lane >= ts) {
                    const ulonglong2 d = *reinterpret_cast<const ulonglong2 *>(tdesc + 2 * (tb + lane));
                    ds = d.x;
                    if (lane > 0) dt = d.y;  // the first tail member's tail ids are all below every tail id of the pivot: no match possible
                }
            }
        }
    };
    auto load_offsets = [&](int32_t u, int64_t &hb, int &hl, int64_t &tb, int &tl, int &ts) {
        hb = hoff[u];
        hl = int(hoff[u + 1] - hb);
        tb = toff[u];
        tl = int(toff[u + 1] - tb);
        ts = tsplit[u];
        if (tl <= ts) tl = 0;  // no member beyond inline_limit: nothing to stream for this pivot
    };
    {  // prologue
        int64_t hb0 = 0, tb0 = 0;
        int hl0 = 0;
        if (pos0 < end) load_offsets(order[pos0], hb0, hl0, tb0, tlD, tsD);
        if (pos0 + step < end) load_offsets(order[pos0 + step], hbC, hlC, tbC, tlC, tsC);
        if (pos0 + 2 * step < end) load_offsets(order[pos0 + 2 * step], hbB, hlB, tbB, tlB, tsB);
        if (pos0 + 3 * step < end) uA = order[pos0 + 3 * step];
        load_members(hb0, hl0, tb0, tlD, tsD, hvD, vD, dsD, dtD);
    }
    for (int64_t pos = pos0; pos < end; pos += step) {  // uniform per wave
        // loads of the later stages first; they complete while D is scanned
        int64_t hbN = 0, tbN = 0;
        int hlN = 0, tlN = 0, tsN = 0;
        if (uA >= 0) load_offsets(uA, hbN, hlN, tbN, tlN, tsN);
        const int32_t uN = (pos + 4 * step < end) ? order[pos + 4 * step] : -1;
        uint32_t hvC;
        int32_t vC;
        unsigned long long dsC, dtC;
        load_members(hbC, hlC, tbC, tlC, tsC, hvC, vC, dsC, dtC);
        if (tlD > 0) {
            uint32_t c = 0;
            __builtin_amdgcn_wave_barrier();
            // hub members: only their bits are needed here (their edges were handed over)
            if (hvD != 0xFFFFu) atomicOr(&bm[hvD >> 5], 1u << (hvD & 31u));
            for (int i = lane; i < SIZE; i += 64) tbl[i] = -1;
            fill[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            uint32_t slot = 0;
            if (vD >= 0) {
                slot = atomicAdd(&fill[bucket_of<64>(vD)], 1u);
                if (slot < 4) tbl[bucket_of<64>(vD) * 4 + slot] = vD;
            }
            const bool bucketed = __ballot(slot >= 4) == 0;
            if (!bucketed) {  // rare: some bucket took a fifth key; rebuild as an open-addressing table
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < SIZE; i += 64) tbl[i] = -1;
                __builtin_amdgcn_wave_barrier();
                if (vD >= 0) set_insert(tbl, MASK, SHIFT, vD);
            }
            __builtin_amdgcn_wave_barrier();
            c = scan_srows(bm, spool, dsD, lane);
            if (bucketed)
                c += scan_trows_probe([tbl](int32_t w) { return bucket_contains<64>(tbl, w); }, tpool, dtD, tlD, lane);
            else
                c += scan_trows_probe([tbl](int32_t w) { return set_contains(tbl, MASK, SHIFT, w); }, tpool, dtD, tlD, lane);
            cnt += c;
            __builtin_amdgcn_wave_barrier();
            if (hvD != 0xFFFFu) bm[hvD >> 5] = 0;  // every bit in this wave's bitmap belongs to this pivot
        }
        tlD = tlC; tsD = tsC; hvD = hvC; vD = vC; dsD = dsC; dtD = dtC;
        hbC = hbB; hlC = hlB; tbC = tbB; tlC = tlB; tsC = tsB;
        hbB = hbN; hlB = hlN; tbB = tbN; tlB = tlN; tsB = tsN;
        uA = uN;
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[(blockIdx.x & (kAccSlots - 1)) * kAccStride], t);
    }
}

// units / probes / algorithmic stream bytes of a shard (untimed bookkeeping for gmsx_stats).  out[2] follows what the count kernels
// read, byte for byte, assuming no on-chip reuse:
//   light pivot u (2 <= d+ < 64; k_tc_stats, wave per pivot position): its hub part if it has tail members; per far light member
//       (rank id >= inline_limit, d+ < 64) the member id, its two descriptors and the stream rows they describe;
//   work item (k_tc_item_stats, wave per item): the pivot's container (hub part for a hub item, tail part for a tail item) once; per
//       entry 8 bytes of descriptor and the stream row it describes (whole 16-byte units) — members' rows, cut rows and inline chunks alike.
// out[0] = oriented edges counted by the shard: every edge of a light or idle pivot at the pivot, the edges a heavy pivot handed to its
// first members over inline at the pivot, every other edge of a heavy pivot where its entries live (tunits, written by the build);
// out[1] = id slots probed (per unit: 8 list, 14 byte-delta, 10 gap-12, 4 bitset words; 4 / 6 tail ids).  A shard = the pivots at the
// positions of `order` that shard_of() gives it.
__global__ __launch_bounds__(256) void k_tc_stats(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                  const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                  const unsigned long long *__restrict__ srow, const unsigned long long *__restrict__ trow,
                                                  const int32_t *__restrict__ tunits, int32_t inline_limit, int64_t end, int nparts, int part,
                                                  unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long units = 0, probes = 0, bytes = 0;
    auto slots = [](unsigned long long d, bool tail) -> unsigned long long {
        const unsigned long long n = d & 0x3fffffull, f = (d >> 22) & 3ull;
        return n * (tail ? (f == kFormDelta ? 6ull : 4ull) : (f == kFormList ? 8ull : f == kFormDelta ? 14ull : f == kFormGap12 ? 10ull : 4ull));
    };
    for (int64_t pos = wave0; pos < end; pos += nwaves) {
        if (nparts > 1 && shard_of(pos, nparts) != part) continue;
        const int32_t u = order[pos];
        const int du = dplus[u];
        if (lane == 0) units += (unsigned long long)tunits[u];  // forward + reverse edges whose entries live here
        if (du >= kHeavy) {  // … and the edges to its first members that went inline (no entry anywhere)
            const int hl = min(int(hoff[u + 1] - hoff[u]), 64), tl = min(int(toff[u + 1] - toff[u]), 64 - hl);
            int32_t v = -1;
            if (lane < hl) {
                const uint32_t x = hadj[hoff[u] + lane];
                if (x != 0xFFFFu) v = int32_t(x);
            } else if (lane - hl < tl) v = tadj[toff[u] + lane - hl];
            if (v >= 0 && lane > 0 && (v < inline_limit || dplus[v] >= kHeavy)) ++units;
            continue;
        }
        const int hl = int(hoff[u + 1] - hoff[u]), tl = int(toff[u + 1] - toff[u]);
        if (lane == 0) {
            units += (unsigned long long)du;
            if (du >= 2 && tl > 0) bytes += 2ull * hl;  // k_tc_wave reads the hub part of every pivot that has tail members
        }
        if (du < 2) continue;
        for (int64_t j = toff[u] + lane; j < toff[u + 1]; j += 64) {
            const int32_t v = tadj[j];
            if (v < inline_limit || dplus[v] >= kHeavy) continue;  // handed over: the ids are in v's inline rows, counted with its items
            const unsigned long long d = srow[v], t = j > toff[u] ? trow[v] : 0ull;
            bytes += 16ull * ((d & 0x3fffffull) + (t & 0x3fffffull)) + 20ull;  // the rows + the member id and its descriptors
            probes += slots(d, false) + slots(t, true);
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
__global__ __launch_bounds__(256) void k_tc_item_stats(const int64_t *__restrict__ coff, int bytes_per_id, int tail, const unsigned long long *__restrict__ task,
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
            const unsigned long long d = task[b + i];
            bytes += 8ull + 16ull * (d & 0x3fffffull);
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
//   out[5]     the entries themselves (8 bytes each)          out[6]  the pivots' own containers (hub part per hub item, tail part per tail item)
//   out[7]     of out[0] + out[3]: inline rows (ids handed over by light pivots; filled in by the host from the build's figures)
//   out[8..9]  light pivots: hub / tail stream rows of their far light members (k_tc_wave)
//   out[10]    light pivots: their own hub parts + member ids + descriptors (k_tc_wave)
//   out[11..14] counts: entries, inline entries, work items, far light members streamed by k_tc_wave
//   out[15..20] reserved (0)
__global__ __launch_bounds__(256) void k_tc_breakdown(const int64_t *__restrict__ hoff, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                      const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                      const unsigned long long *__restrict__ srow, const unsigned long long *__restrict__ trow,
                                                      const unsigned long long *__restrict__ htask, const gmsx_tc_item *__restrict__ hitem, int64_t hitems,
                                                      const unsigned long long *__restrict__ ttask, const gmsx_tc_item *__restrict__ titem, int64_t titems,
                                                      int32_t inline_limit, int64_t first_light, int64_t end_light,
                                                      unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long c[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t q = wave0; q < hitems + titems; q += nwaves) {
        const bool tail = q >= hitems;
        const gmsx_tc_item &it = tail ? titem[q - hitems] : hitem[q];
        const unsigned long long *task = tail ? ttask : htask;
        const int ne = int(it.bc >> 40);
        const int64_t b0 = int64_t(it.bc & 0xffffffffffull);
        if (lane == 0) {
            c[6] += tail ? 4ull * (unsigned long long)(toff[it.pivot + 1] - toff[it.pivot]) : 2ull * (unsigned long long)(hoff[it.pivot + 1] - hoff[it.pivot]);
            c[11] += (unsigned long long)ne;
            c[13] += 1;
        }
        for (int i = lane; i < ne; i += 64) {
            const unsigned long long d = task[b0 + i];
            const unsigned long long db = 16ull * (d & 0x3fffffull);
            const int fd = int((d >> 22) & 3);
            if (tail) c[fd == kFormDelta ? 4 : 3] += db;
            else c[fd == kFormList ? 0 : fd == kFormBitset ? 1 : 2] += db;
            c[5] += 8;
        }
    }
    for (int64_t pos = first_light + wave0; pos < end_light; pos += nwaves) {
        const int32_t u = order[pos];
        const int64_t tb0 = toff[u], te = toff[u + 1];
        if (lane == 0 && te > tb0) c[10] += 2ull * (unsigned long long)(hoff[u + 1] - hoff[u]);
        for (int64_t j = tb0 + lane; j < te; j += 64) {
            const int32_t v = tadj[j];
            if (v < inline_limit || dplus[v] >= kHeavy) continue;
            c[8] += 16ull * (srow[v] & 0x3fffffull);
            if (j > tb0) c[9] += 16ull * (trow[v] & 0x3fffffull);
            c[10] += 20;
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
__global__ __launch_bounds__(256) void k_tc_row_hist(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                     const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                     const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                     const unsigned long long *__restrict__ srow, const unsigned long long *__restrict__ trow,
                                                     const unsigned long long *__restrict__ htask, const gmsx_tc_item *__restrict__ hitem, int64_t hitems,
                                                     const unsigned long long *__restrict__ ttask, const gmsx_tc_item *__restrict__ titem, int64_t titems,
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
        const unsigned long long *task = tail ? ttask : htask;
        const int ne = int(it.bc >> 40);
        const int64_t b0 = int64_t(it.bc & 0xffffffffffull);
        if (lane == 0) {
            atomicAdd(&h[240], (unsigned long long)ne);
            atomicAdd(&h[242], 1ull);
            atomicAdd(&h[243], tail ? 4ull * (unsigned long long)(toff[it.pivot + 1] - toff[it.pivot]) : 2ull * (unsigned long long)(hoff[it.pivot + 1] - hoff[it.pivot]));
        }
        for (int i = lane; i < ne; i += 64) {
            const unsigned long long d = task[b0 + i];
            if (tail) add(((d >> 22) & 3) == kFormDelta ? 4 : 3, d & 0x3fffffull);
            else add(min(int((d >> 22) & 3), 2), d & 0x3fffffull);  // 12-bit-gap rows are counted with the byte-delta ones
        }
    }
    for (int64_t pos = wave0; pos < end; pos += nwaves) {
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

static int64_t part_count(int64_t first, int64_t end, int nparts, int part) {
    const int64_t span = end - first - part;
    return span <= 0 ? 0 : (span + nparts - 1) / nparts;
}

static int tc_one(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st);

static int tc_oriented(const gmsx_graph *g, int part, int nparts, uint64_t *partial, gmsx_stats *st) {
    if (int rc = ensure_tc(g)) return rc;  // first call on a graph uploaded without GMSX_UPLOAD_FOR_TC: builds the task lists (untimed)
    if (g->tc_passes == 1) return tc_one(g, part, nparts, partial, st);
    // FALLBACK (device_graph.hpp, tc_passes): the containers of all pivots did not fit.  A whole-graph call walks the passes — shard p of
    // tc_passes resident at a time, rebuilt between the passes (untimed like every build; kernel_ms is the sum of the passes' kernels); a
    // sharded call builds exactly its shard.
    if (nparts > 1) {
        if (int rc = ensure_tc_shard(g, part, nparts)) return rc;
        return tc_one(g, part, nparts, partial, st);
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
    if (const char *only = std::getenv("GMSX_TC_ONLY")) {  // profiling (WRONG counts): the hub items / the tail items / the light pivots alone
        if (std::strcmp(only, "hub") != 0) n_hitems = 0;
        if (std::strcmp(only, "tail") != 0) n_titems = 0;
    }
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
    const int64_t cnt_light = [&] {
        const char *only = std::getenv("GMSX_TC_ONLY");
        return only && std::strcmp(only, "light") != 0 ? int64_t(0) : part_count(0, g->n_wave, nparts, part);
    }();
    // CO-SCHEDULING.  The hub-item kernel is bound by HBM bandwidth and VALU issue, the tail-item kernel streams short rows, the light-pivot
    // kernel (short rows behind dependent loads) is bound by memory latency: back to back each leaves what the others need idle and pays
    // its own drain.  So the light kernel goes to a side stream FIRST, as a persistent grid of a few workgroups per CU, the tail items to
    // a second side stream, and the hub items fill the remaining wave slots and LDS of every CU.  GMSX_TC_OVERLAP=0 restores the serial
    // order (full-width light grid), 2 forces co-scheduling of the light kernel on small graphs.
    const int overlap = [] { const char *e = std::getenv("GMSX_TC_OVERLAP"); return e ? std::atoi(e) : 1; }();
    const bool large = g->inline_limit > g->dense_limit;  // n >= 2^24: inline limit beyond the hub range
    const int wave_wgs = [large] { const char *e = std::getenv("GMSX_TC_WAVE_WGS"); return e ? std::max(1, std::atoi(e)) : (large ? 1 : 2); }();
    // measured (MI355X, round 2): scale 26 serial 86.4 ms, co-scheduled 84.6 (2 workgroups per CU) / 82.9 (1); scale 24 15.05 / 14.45 (2) /
    // erratic (1); scale 22 3.56 serial, 4.45 co-scheduled.  So the light kernel moves aside from 2^23 vertices on.
    const bool sides = overlap && c.side[0] && c.side[1];
    const bool co_wave = sides && n_hitems > 0 && cnt_light > 0 && (overlap > 1 || g->n >= (int64_t(1) << 23));
    const int tail_mode = [] { const char *e = std::getenv("GMSX_TC_TAIL"); return e ? std::atoi(e) : 1; }();  // A/B: 0 = tail items behind the hub items on the launch stream, 2 = before them
    const bool co_tail = sides && n_hitems > 0 && n_titems > 0 && tail_mode == 1;
    hipStream_t s_wave = co_wave ? c.side[1] : s, s_tail = co_tail ? c.side[0] : s;
    struct Join {  // joins the side streams on every way out once they were forked (error returns included)
        Ctx &c;
        hipStream_t s;
        bool armed[2] = {false, false};
        ~Join() {
            for (int i = 0; i < 2; ++i)
                if (armed[i] && hipEventRecord(c.ev_join[i], c.side[i]) == hipSuccess) (void)hipStreamWaitEvent(s, c.ev_join[i], 0);
        }
    } join{c, s};
    const int persist = [] { const char *e = std::getenv("GMSX_TC_PERSIST"); return e ? std::atoi(e) : 1; }();  // A/B: 0 = one workgroup per item (rounds 2-3)
    if (persist) {
        // ONE persistent launch walks the hub and the tail items (k_tc_items); the light-pivot kernel goes beside it as before
        const bool co = sides && cnt_light > 0 && n_hitems + n_titems > 0 && (overlap > 1 || g->n >= (int64_t(1) << 23));
        if (co) {
            GMSX_HIP(hipEventRecord(c.ev_fork, s));
            GMSX_HIP(hipStreamWaitEvent(c.side[1], c.ev_fork, 0));
            join.armed[1] = true;
        }
        auto launch_light = [&](hipStream_t sw, bool beside) {
            if (cnt_light <= 0) return;
            const int64_t b_wave = std::min<int64_t>((cnt_light + 3) / 4, beside ? int64_t(cus) * wave_wgs : cap_blocks);
            hipLaunchKernelGGL(k_tc_wave, dim3(unsigned(b_wave)), dim3(256), 0, sw, g->hoff, g->hadj, g->spool, g->tpool, g->tdesc, g->toff, g->tadj,
                               g->tsplit, g->worder, int64_t(0), g->n_wave, nparts, part, acc);
            ++launches;
        };
        if (co) launch_light(c.side[1], true);
        if (n_hitems + n_titems > 0) {
            static const int share_env = [] { const char *e = std::getenv("GMSX_TC_TAIL_SHARE"); return e ? std::atoi(e) : -1; }();
            static const int wgs_env = [] { const char *e = std::getenv("GMSX_TC_ITEM_WGS"); return e ? std::atoi(e) : 8; }();
            // workgroups that START on the tail queue, of every 8 (the queues drain into each other, so this only shapes the mix)
            const int tail_share = share_env >= 0 ? std::min(share_env, 8) : (n_titems == 0 ? 0 : n_hitems == 0 ? 8 : 2);
            const int64_t want = (n_hitems + n_titems + kGrab - 1) / kGrab;
            const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(want, int64_t(cus) * std::max(1, wgs_env)));
            hipLaunchKernelGGL(k_tc_items, dim3(unsigned(blocks)), dim3(256), 0, s, g->hadj, g->tadj, g->spool, g->tpool, g->htask, g->ttask, hitem, int(n_hitems),
                               titem, int(n_titems), tail_share, qhead, acc);
            ++launches;
        }
        if (!co) launch_light(s, false);
    } else if (tail_mode == 3 && sides && n_hitems > 0) {
        // A/B: the hub items ALONE first (they run at 7 TB/s by themselves), then the tail items on the launch stream with the light pivots
        // beside them (neither of the two is bandwidth-bound)
        hipLaunchKernelGGL(k_tc_block, dim3(unsigned(n_hitems)), dim3(256), 0, s, g->hoff, g->hadj, g->spool, g->htask, hitem, acc);
        ++launches;
        GMSX_HIP(hipEventRecord(c.ev_fork, s));
        GMSX_HIP(hipStreamWaitEvent(c.side[1], c.ev_fork, 0));
        join.armed[1] = true;
        if (cnt_light > 0) {
            const int64_t b_wave = std::min<int64_t>((cnt_light + 3) / 4, int64_t(cus) * 4);
            hipLaunchKernelGGL(k_tc_wave, dim3(unsigned(b_wave)), dim3(256), 0, c.side[1], g->hoff, g->hadj, g->spool, g->tpool, g->tdesc, g->toff, g->tadj,
                               g->tsplit, g->worder, int64_t(0), g->n_wave, nparts, part, acc);
            ++launches;
        }
        if (n_titems > 0) {
            hipLaunchKernelGGL(k_tc_tail, dim3(unsigned(n_titems)), dim3(256), 0, s, g->toff, g->tadj, g->tpool, g->ttask, titem, acc);
            ++launches;
        }
    } else {
    if (co_wave || co_tail) GMSX_HIP(hipEventRecord(c.ev_fork, s));
    if (co_tail) {
        GMSX_HIP(hipStreamWaitEvent(c.side[0], c.ev_fork, 0));
        join.armed[0] = true;
    }
    if (co_wave) {
        GMSX_HIP(hipStreamWaitEvent(c.side[1], c.ev_fork, 0));
        join.armed[1] = true;
    }
    auto launch_light = [&]() {
        if (cnt_light <= 0) return;
        const int64_t want = (cnt_light + 3) / 4;
        const int64_t b_wave = std::min<int64_t>(want, co_wave ? int64_t(cus) * wave_wgs : cap_blocks);
        hipLaunchKernelGGL(k_tc_wave, dim3(unsigned(b_wave)), dim3(256), 0, s_wave, g->hoff, g->hadj, g->spool, g->tpool, g->tdesc, g->toff, g->tadj,
                           g->tsplit, g->worder, int64_t(0), g->n_wave, nparts, part, acc);
        ++launches;
    };
    if (co_wave) launch_light();
    if (n_titems > 0 && tail_mode == 2) {
        hipLaunchKernelGGL(k_tc_tail, dim3(unsigned(n_titems)), dim3(256), 0, s, g->toff, g->tadj, g->tpool, g->ttask, titem, acc);
        ++launches;
    }
    if (n_hitems > 0) {
        hipLaunchKernelGGL(k_tc_block, dim3(unsigned(n_hitems)), dim3(256), 0, s, g->hoff, g->hadj, g->spool, g->htask, hitem, acc);
        ++launches;
    }
    if (n_titems > 0 && tail_mode != 2) {
        hipLaunchKernelGGL(k_tc_tail, dim3(unsigned(n_titems)), dim3(256), 0, s_tail, g->toff, g->tadj, g->tpool, g->ttask, titem, acc);
        ++launches;
    }
    if (!co_wave) launch_light();
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
            hipLaunchKernelGGL(k_tc_stats, dim3(unsigned(blocks)), dim3(256), 0, s, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, g->order, g->srow, g->trow,
                               g->tunits, g->inline_limit, g->n, nparts, part, acc + kAccSlots * kAccStride);
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
    hipLaunchKernelGGL(k_tc_breakdown, dim3(unsigned(cus * 16)), dim3(256), 0, s, g->hoff, g->toff, g->tadj, g->dplus, g->order, g->srow, g->trow, g->htask,
                       g->hitem, g->hitems, g->ttask, g->titem, g->titems, g->inline_limit, n_block, n_work, acc);
    GMSX_HIP(hipMemcpyAsync(out21, acc, 21 * 8, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    GMSX_HIP(hipGetLastError());
    out21[7] = uint64_t(g->inline_units) * 16ull;                         // the build's figures: the entries do not say what they name
    out21[12] = uint64_t(g->inline_hentries + g->inline_tentries);
    return GMSX_OK;
}

int gmsx_tc_row_histogram(const gmsx_graph *g, uint64_t *out256) {
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
}

int gmsx_tc_partial(const gmsx_graph *g, int algo, int part, int nparts, uint64_t *partial, gmsx_stats *stats) {
    if (!g || !partial || nparts < 1 || part < 0 || part >= nparts) return GMSX_ERR_INVALID;
    if (algo != GMSX_TC_AUTO && algo != GMSX_TC_ORIENTED && algo != GMSX_TC_FULL) return GMSX_ERR_INVALID;
    if (int rc = ensure_init()) return rc;
    if (algo == GMSX_TC_FULL) return tc_full_partial(g, part, nparts, partial, stats);
    // a sharded upload holds the task lists of ONE shard: that is the only one it can count
    if (g->tc_passes == 1 && g->shard_nparts > 1 && (nparts != g->shard_nparts || part != g->shard_part)) return GMSX_ERR_INVALID;
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
