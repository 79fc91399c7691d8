// k-clique counting on gfx950: the device replacement for
//   CliqueCount / RecursiveStepCliqueCount   (gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h:5-31)
// whose inner operator is Set::intersect (sorted_set.h:160-166 -> sorted_set_operations.h:36-42).
//
// The reference recurses over the SYMMETRIC graph and returns k! * C_k (every clique once per vertex order).  Here
// each clique is met once, on the degree-oriented DAG, and the result is multiplied by k! (mod 2^64, the reference's
// size_t arithmetic) at the ABI.  Per pivot u (one workgroup, or one wave for d+ <= 32):
//   1. BUILD ("LDS-staged intersect"): the pivot row N+(u) is staged into LDS — as the 65536-bit hub bitmap with
//      prefix popcounts (workgroup kernels: rank in the bitmap = local index) or as a 64 x 4 bucket set id -> local
//      index (wave kernel); the rows N+(v), v in N+(u), are streamed from HBM exactly like the triangle kernels do
//      (16-lane groups, 16-byte loads, bitset AND or list probes) and every hit sets one bit of the local adjacency
//      bit-matrix  rows[i] = N+(v_i) ∩ N+(u)  (d x d bits, strictly lower triangular; in LDS up to d = 1024, else in a
//      global slab built through an LDS row stage);
//   2. COUNT: recursive set intersection on bit rows: cand' = cand & rows[j] (bitmap AND), popcount at the last
//      level — the reference's `isect.intersect(N(vi))` recursion with sets as d-bit vectors.  k = 4 on wide matrices
//      runs wave-cooperatively with one neighbour j per lane (kc4_row), slab matrices band by band through LDS.
#include "device_graph.hpp"
#include "kc4_mfma.hpp"
#if defined(GMSX_KC_NO_TAIL_MEMBERS) && !defined(GMSX_DEV_HOOKS)  // (the A/B switches of device_graph.hpp's list + this round's: development builds only)
#error "A/B switches need -DGMSX_DEV_HOOKS (tools/ab_lib.sh sets it)"
#endif

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdlib>

namespace gmsx {

static constexpr int kAccSlots = 64;
static constexpr int kAccStride = 16;
static constexpr int kMaxK = 10;

// ---- id -> local index map of the wave kernel (open-addressing fallback of its bucket set; keys -1 = empty) ----
__device__ __forceinline__ uint32_t kc_hash(int32_t w, int shift) { return (uint32_t(w) * 0x9E3779B1u) >> shift; }

// ---- streamed build (workgroup kernels) ------------------------------------------------------------------------
// Same row streaming as the triangle kernels (tc.hip): a wave works as four 16-lane groups, each on the row of one
// pivot member, 16-byte loads, two in flight.  A streamed hub id is tested against the pivot bitmap `bm`; a hit becomes
// the local index  pre[word] + popcount(bits below)  (the hub list is sorted, so rank = position) and sets one bit of
// the member's adjacency row `orow` (LDS).  Hits are ~5-10 % of the stream: per load they are collected in a mask and
// resolved in a short loop, so the probe path stays branch-free.
struct __attribute__((packed, aligned(4))) kc_u4u { uint32_t x, y, z, w; };
// REVERSE ROWS (see ensure_kc_reverse below): what the pivots' kernels need of them.  rel == nullptr: every member is streamed forward.
struct KcRev {
    const uint32_t *rel;    // per hub-entry position of hadj: word offset of the member's row inside its pivot's span, ~0u = forward
    const int64_t *aoff;    // per rank id: the pivot's span in the arena
    const uint32_t *arena;  // rows written by k_kc_reverse / k_kc_reverse_tail earlier in the same call
    const uint32_t *relt;   // the same per tail-entry position of tadj (TAIL receivers, round 6b); nullptr: every tail member is streamed forward
};
static constexpr uint32_t kKcRelForward = 0xffffffffu;
// EXPORT (round 6; k = 4): the BUILD leaves the finished matrix of the pivot at position q of its launch in a pool (slot q, row stride kc4m_stride(d) words,
// d in dpool[q]) instead of counting it — the count of the launch's matrices is one k_kc4_mfma launch on the matrix cores (kc4_mfma.hpp).  pool == nullptr: off.
struct KcExport {
    uint32_t *pool;
    int32_t *dpool;
    unsigned long long slot_words;
};
// TRIANGULAR LDS matrix (round 6; k = 4, the bins of 512 < d+ <= 1472): row i is strictly below the diagonal, so it is stored with (i >> 5) + 1 words —
// every word that can hold a column below i, and word j >> 5 of row j exists for the pair loops — at word offset Σ_{t<i} ((t >> 5) + 1).  Half the
// LDS of the rectangular layout: two 1024-thread workgroups per CU up to d+ = 960, and a matrix in LDS (pair-list count) instead of a global slab
// (row-band count) up to d+ = 1472.
__device__ __host__ __forceinline__ uint32_t kc_tri_off(int i) {
    const int q = i >> 5, r = i & 31;
    return uint32_t((q + 1) * (16 * q + r));
}
// a finished row of `nwords` words from the arena into the (zeroed) matrix row, by the 16 lanes of a group
__device__ __forceinline__ void kc_copy_row(const uint32_t *__restrict__ src, int nwords, uint32_t *orow, int sub) {
    for (int t = sub; t < nwords; t += 16) orow[t] = src[t];
}

__device__ __forceinline__ void kc_set_hub_hit(const uint32_t *bm, const unsigned short *pre, uint32_t *orow, uint32_t id) {
    const uint32_t word = bm[id >> 5];
    const int idx = int(pre[id >> 5]) + __popc(word & ((1u << (id & 31u)) - 1u));
    atomicOr(&orow[idx >> 5], 1u << (idx & 31));
}

// the bit of the low / high 16-bit id of a packed pair: one shift + mask for the word address, v_bfe_u32 for the bit (it uses offset[4:0] only,
// so the low id needs no "& 31" and no "& 0xffff")
__device__ __forceinline__ uint32_t kc_bit_lo(const uint32_t *bm, uint32_t p) { return __builtin_amdgcn_ubfe(bm[(p >> 5) & 0x7ffu], p, 1u); }
__device__ __forceinline__ uint32_t kc_bit_hi(const uint32_t *bm, uint32_t p) {
    const uint32_t q = p >> 16;
    return __builtin_amdgcn_ubfe(bm[q >> 5], q, 1u);
}
__device__ __forceinline__ void kc_probe8(const uint32_t *bm, const unsigned short *pre, uint32_t *orow, kc_u4u p) {
#ifdef GMSX_KC_NO_PROBE  // A/B build (wrong counts): the row loads without their probes
    if ((p.x ^ p.y ^ p.z ^ p.w) == 0x12345678u) orow[0] = 1u;
    return;
#endif
    uint32_t mask = kc_bit_lo(bm, p.x) | (kc_bit_hi(bm, p.x) << 1) | (kc_bit_lo(bm, p.y) << 2) | (kc_bit_hi(bm, p.y) << 3) |
                    (kc_bit_lo(bm, p.z) << 4) | (kc_bit_hi(bm, p.z) << 5) | (kc_bit_lo(bm, p.w) << 6) | (kc_bit_hi(bm, p.w) << 7);
    const unsigned long long lo = (unsigned long long)p.x | ((unsigned long long)p.y << 32), hi = (unsigned long long)p.z | ((unsigned long long)p.w << 32);
#ifdef GMSX_KC_NO_HITS  // A/B build (wrong counts): the probes without the resolution of their hits
    if (mask == 0xdeadu) orow[0] = 1u;
    mask = 0;
#endif
    while (mask) {
        const int s = __ffs(mask) - 1;
        mask &= mask - 1;
        const uint32_t id = uint32_t(((s & 4) ? hi : lo) >> ((s & 3) * 16)) & 0xffffu;
        kc_set_hub_hit(bm, pre, orow, id);
    }
}

__device__ __forceinline__ kc_u4u kc_load8(const uint16_t *row, int j, int l) {
    kc_u4u p = *reinterpret_cast<const kc_u4u *>(row + j);
    const int valid = l - j;  // even, >= 2; ids beyond the row become 0xFFFF, which is never in the bitmap
    if (valid < 8) {
        if (valid < 6) p.z = 0xffffffffu;
        if (valid < 4) p.y = 0xffffffffu;
        p.w = 0xffffffffu;
    }
    return p;
}

// 16-bit list container [row, row + l) (l even, padded with 0xFFFF) of one member; sub = lane within the 16-lane group.
// A ring of kKcDepth loads per lane, refilled before the oldest one is probed; the loads are UNCONDITIONAL (an index past the row is
// clamped to its last pair: in bounds, probed never) so that the waits are counted — under a branch every wait is vmcnt(0) and the
// group has ONE load in flight: long member rows (the pivots of the big bins have members of d+ up to 3000) were a chain of round trips.
#ifndef GMSX_KC_STREAM_DEPTH
#define GMSX_KC_STREAM_DEPTH 2  // steps in flight per lane of the step-stream BUILD (mode 2)
#endif
#ifndef GMSX_KC_DEPTH
#define GMSX_KC_DEPTH 2  // (measured, k = 4 at scale 22 / 24: depth 2 22.6 / 143.4 ms, depth 3 24.9 / 154.5, depth 4 25.1 / 155.2 — the registers of a deeper ring cost a wave per SIMD, and the BUILD waits on its LDS probes, not on the row loads)
#endif
static constexpr int kKcDepth = GMSX_KC_DEPTH;
__device__ __forceinline__ kc_u4u kc_cut8(kc_u4u p, int valid) {  // ids behind the row become 0xFFFF, which is never in the bitmap (valid even, >= 2)
    if (valid < 8) {
        if (valid < 6) p.z = 0xffffffffu;
        if (valid < 4) p.y = 0xffffffffu;
        p.w = 0xffffffffu;
    }
    return p;
}
__device__ __forceinline__ void kc_stream_list_from(const uint16_t *__restrict__ row, int l, int j0, int sub, const uint32_t *bm, const unsigned short *pre,
                                                    uint32_t *orow) {
    (void)sub;
    if (l <= 0) return;  // uniform per group
    kc_u4u p[kKcDepth];
    const int jmax = l - 2;
#pragma unroll
    for (int k = 0; k < kKcDepth; ++k) p[k] = *reinterpret_cast<const kc_u4u *>(row + min(j0 + 128 * k, jmax));
    for (int j = j0; j < l; j += 128 * kKcDepth) {  // the lanes of a group differ by at most one step
#pragma unroll
        for (int k = 0; k < kKcDepth; ++k) {
            const kc_u4u cur = p[k];
            const int jc = j + 128 * k;
            p[k] = *reinterpret_cast<const kc_u4u *>(row + min(jc + 128 * kKcDepth, jmax));
            if (jc < l) kc_probe8(bm, pre, orow, kc_cut8(cur, l - jc));
        }
    }
}
__device__ __forceinline__ void kc_stream_list(const uint16_t *__restrict__ row, int l, int sub, const uint32_t *bm, const unsigned short *pre,
                                               uint32_t *orow) {
    kc_stream_list_from(row, l, sub * 8, sub, bm, pre, orow);
}

__device__ __forceinline__ void kc_and_word(uint32_t x, uint32_t pivot_word, int k, const unsigned short *pre, uint32_t *orow) {
#if defined(GMSX_KC_NO_HITS) || defined(GMSX_KC_NO_PROBE)
    if (x == 0xdeadbeefu) orow[0] = 1u;
    x = 0;
#endif
    while (x) {
        const int b = __ffs(x) - 1;
        x &= x - 1;
        const int idx = int(pre[k]) + __popc(pivot_word & ((1u << b) - 1u));
        atomicOr(&orow[idx >> 5], 1u << (idx & 31));
    }
}

// bitset container (nw words, multiple of 4) of one hub member: AND with the pivot bitmap, every surviving bit is a hit
__device__ __forceinline__ void kc_and4(uint4 p, uint4 q, int j, const unsigned short *pre, uint32_t *orow) {
    kc_and_word(p.x & q.x, q.x, j, pre, orow);
    kc_and_word(p.y & q.y, q.y, j + 1, pre, orow);
    kc_and_word(p.z & q.z, q.z, j + 2, pre, orow);
    kc_and_word(p.w & q.w, q.w, j + 3, pre, orow);
}
__device__ __forceinline__ void kc_stream_bitset_from(const uint32_t *__restrict__ brow, int nw, int j0, const uint32_t *bm, const unsigned short *pre,
                                                      uint32_t *orow) {
    if (nw <= 0) return;
    uint4 p[kKcDepth];
    const int jmax = nw - 4;  // nw is a multiple of 4
#pragma unroll
    for (int k = 0; k < kKcDepth; ++k) p[k] = *reinterpret_cast<const uint4 *>(brow + min(j0 + 64 * k, jmax));
    for (int j = j0; j < nw; j += 64 * kKcDepth) {
#pragma unroll
        for (int k = 0; k < kKcDepth; ++k) {
            const uint4 cur = p[k];
            const int jc = j + 64 * k;
            p[k] = *reinterpret_cast<const uint4 *>(brow + min(jc + 64 * kKcDepth, jmax));
            if (jc < nw) kc_and4(cur, *reinterpret_cast<const uint4 *>(bm + jc), jc, pre, orow);
        }
    }
}
__device__ __forceinline__ void kc_stream_bitset(const uint32_t *__restrict__ brow, int nw, int sub, const uint32_t *bm, const unsigned short *pre,
                                                 uint32_t *orow) {
    kc_stream_bitset_from(brow, nw, sub * 4, bm, pre, orow);
}

// position of w in the ascending list [lst, lst+len), or -1
__device__ __forceinline__ int sorted_find(const int32_t *__restrict__ lst, int len, int32_t w) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lst[mid] < w) lo = mid + 1; else hi = mid;
    }
    return (lo < len && lst[lo] == w) ? lo : -1;
}

// … behind a FILTER (round 5): 32 768 (8 192) bits in LDS, bit (id mod that) set for every tail member of the pivot.  The tail list itself is in global memory
// and the search a chain of ~log2(tc) dependent loads; almost every streamed tail id is a miss (as in the triangle kernels, tc.hip), and from scale 24
// on — where most vertices are tail vertices — those searches, not the rows, were what the BUILD waited for.
// (A filter of 8 192 bits in the bins of d+ <= 704 — a workgroup more per CU — measured: scale 24 116.2 instead of 112.6 ms, the 512-thread bin at scale 26
//  214 instead of < 149 ms: every false positive is a chain of global loads.  Hence also the SECOND bit per id, GMSX_KC_FILTER_BITS.)
#ifndef GMSX_KC_FILTER_BITS
#define GMSX_KC_FILTER_BITS 2
#endif
struct KcFilter {
    const uint32_t *w;
    uint32_t mask;  // words - 1
};
__device__ __forceinline__ uint32_t kc_filter_bit2(uint32_t w, uint32_t mask) { return (w * 0x9E3779B1u) >> 12 & (32u * mask + 31u); }  // bit index of the second hash
__device__ __forceinline__ int kc_tail_find(const KcFilter flt, const int32_t *__restrict__ lst, int len, int32_t w) {
    if (((flt.w[(uint32_t(w) >> 5) & flt.mask] >> (uint32_t(w) & 31u)) & 1u) == 0u) return -1;
    if (GMSX_KC_FILTER_BITS > 1) {
        const uint32_t b = kc_filter_bit2(uint32_t(w), flt.mask);
        if (((flt.w[b >> 5] >> (b & 31u)) & 1u) == 0u) return -1;
    }
    return sorted_find(lst, len, w);
}

// 32-bit tail container [row, row + l) of one tail member against the pivot's ascending tail list (local index hc + position), from id `base` on:
// lane `sub` of the group takes FOUR consecutive ids per step (one 16-byte load at a 4-byte-aligned address; tadj carries four ids of slack behind
// its last row) — round 5: one id per lane and step, 64 bytes per group and load, made the tail parts 114 of the 330 ms the LDS bins' BUILD took at
// scale 26 (44 % of the oriented edges are tail entries there).  Same ring of unconditional, counted loads as the hub lists.
__device__ __forceinline__ void kc_stream_tail_from(const int32_t *__restrict__ row, int l, int base, int sub, const int32_t *__restrict__ tail_list, int tc, int hc,
                                                    uint32_t *orow, const KcFilter flt) {
    if (l <= base) return;  // uniform per group
#ifdef GMSX_KC_NO_TAIL  // A/B build (wrong counts): the BUILD without the tail parts of its member rows
    return;
#endif
    kc_u4u p[kKcDepth];
    const int j0 = base + 4 * sub;
#pragma unroll
    for (int k = 0; k < kKcDepth; ++k) p[k] = *reinterpret_cast<const kc_u4u *>(row + (j0 + 64 * k < l ? j0 + 64 * k : 0));
    for (int j = j0; j < l; j += 64 * kKcDepth) {  // the lanes of a group differ by at most one step
#pragma unroll
        for (int k = 0; k < kKcDepth; ++k) {
            const kc_u4u cur = p[k];
            const int jc = j + 64 * k, jn = jc + 64 * kKcDepth;
            p[k] = *reinterpret_cast<const kc_u4u *>(row + (jn < l ? jn : 0));
            // the first filter bit of the unit's four ids TOGETHER (four LDS reads in flight, one wait), then only the ids that pass — well under 1 % —
            // go on to the second bit and the search.  (Rounds 4-5 ran kc_tail_find id by id: four exec-masked blocks per unit, each with its own
            // LDS round trip in front of a branch — the tail members' rows were the BUILD's most expensive bytes.)
            const uint32_t ids[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t fw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) fw[q] = flt.w[(ids[q] >> 5) & flt.mask];
            uint32_t pass = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) pass |= ((fw[q] >> (ids[q] & 31u)) & 1u) << q;
            const int nv = l - jc;  // ids of this unit that belong to the row (>= 1 here when jc < l)
            pass = jc < l ? (nv >= 4 ? pass : (pass & ((1u << nv) - 1u))) : 0u;
            while (pass) {
                const int q = __ffs(pass) - 1;
                pass &= pass - 1u;
                const uint32_t id = q == 0 ? ids[0] : q == 1 ? ids[1] : q == 2 ? ids[2] : ids[3];
                bool ok = true;
                if (GMSX_KC_FILTER_BITS > 1) {
                    const uint32_t b = kc_filter_bit2(id, flt.mask);
                    ok = ((flt.w[b >> 5] >> (b & 31u)) & 1u) != 0u;
                }
                if (ok) {
                    const int t = sorted_find(tail_list, tc, int32_t(id));
                    if (t >= 0) {
                        const int idx = hc + t;
                        atomicOr(&orow[idx >> 5], 1u << (idx & 31));
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void kc_stream_tail(const int32_t *__restrict__ row, int l, int sub, const int32_t *__restrict__ tail_list, int tc, int hc,
                                               uint32_t *orow, const KcFilter flt) {
    kc_stream_tail_from(row, l, 0, sub, tail_list, tc, hc, orow, flt);
}

// ---- the BUILD as a pipeline over the members of a lane group (round 4) ---------------------------------------------------------------
// A member's row sits behind a chain of dependent loads: member id -> extents (hoff / toff / bmoff) -> first units of the row.  Walked
// member by member (rounds 1-3) the group paid the three round trips for every member — k_kc_block spent 62 % of its wave cycles waiting.
// Here the chain is three stages deep ACROSS the members of the group: while member t is probed, the first units of member t+1, the
// extents of member t+2 and the id of member t+3 are in flight.  Every load of the pipeline is unconditional (clamped to a valid
// address, its result ignored) so that the waits are counted, not vmcnt(0).
struct KcExt {      // what stage B fetches for a member
    int64_t hb, tb, bo;
    int hl, tl;
};
struct KcFirst {    // … and stage C: the first 16-byte unit of its hub container (list or bitset) and the first id of its tail container
    uint4 h;
    int32_t t;
};
__device__ __forceinline__ KcExt kc_load_ext(const int64_t *__restrict__ hoff, const int64_t *__restrict__ toff, const int64_t *__restrict__ bmoff,
                                             int32_t dense_limit, int32_t v) {
    KcExt e;
    e.hb = hoff[v];
    e.hl = int(hoff[v + 1] - e.hb);
    e.tb = toff[v];
    e.tl = int(toff[v + 1] - e.tb);
    e.bo = bmoff[max(min(v, dense_limit - 1), 0)];
    return e;
}
// does member v (a hub member of the pivot, below dense_limit) stream its bitset container rather than its list?
__device__ __forceinline__ bool kc_use_bitset(int32_t v, bool is_hub, int32_t dense_limit, int hl) {
    return is_hub && v < dense_limit && int(bitset_words(v)) * 4 + 32 < hl * 2;
}
__device__ __forceinline__ KcFirst kc_load_first(const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ bmpool,
                                                 const KcExt &e, bool bitset, int nw, int sub) {
    KcFirst f;
    // (an empty or too short container: unit 0 of the array — loaded, never probed)
    const uint16_t *lrow = hadj + (sub * 8 < e.hl ? e.hb + sub * 8 : 0);
    const uint32_t *brow = bmpool + (sub * 4 < nw ? e.bo + sub * 4 : 0);
    const kc_u4u x = *reinterpret_cast<const kc_u4u *>(bitset ? reinterpret_cast<const uint16_t *>(brow) : lrow);
    f.h = make_uint4(x.x, x.y, x.z, x.w);
    f.t = tadj[sub < e.tl ? e.tb + sub : 0];
    return f;
}
// member v's row into orow, its first units already here
__device__ __forceinline__ void kc_build_member_first(const uint16_t *__restrict__ hadj, const int32_t *__restrict__ tadj, const uint32_t *__restrict__ bmpool,
                                                      const KcExt &e, const KcFirst &f, bool bitset, int nw, bool is_hub, int hc,
                                                      const int32_t *__restrict__ tail_list, int tc, const uint32_t *bm, const unsigned short *pre,
                                                      uint32_t *orow, int sub, const KcFilter flt) {
    if (bitset) {
        const int j = sub * 4;
        if (j < nw) kc_and4(f.h, *reinterpret_cast<const uint4 *>(bm + j), j, pre, orow);
        kc_stream_bitset_from(bmpool + e.bo, nw, j + 64, bm, pre, orow);
        return;
    }
    if (hc > 0) {
        const int j = sub * 8;
        if (j < e.hl) kc_probe8(bm, pre, orow, kc_cut8(kc_u4u{f.h.x, f.h.y, f.h.z, f.h.w}, e.hl - j));
        kc_stream_list_from(hadj + e.hb, e.hl, j + 128, sub, bm, pre, orow);
    }
    if (!is_hub && tc > 0) {
        if (sub < e.tl) {
            const int t = kc_tail_find(flt, tail_list, tc, f.t);
            if (t >= 0) atomicOr(&orow[(hc + t) >> 5], 1u << ((hc + t) & 31));
        }
        kc_stream_tail_from(tadj + e.tb, e.tl, 16, sub, tail_list, tc, hc, orow, flt);
    }
}

// Adjacency row of pivot member i (rank id v) into orow: the smaller of v's bitset / list hub containers, plus the tail
// container when v is a tail vertex and the pivot has tail members.
__device__ __forceinline__ void kc_build_member(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                const int64_t *__restrict__ bmoff, const uint32_t *__restrict__ bmpool, int32_t dense_limit,
                                                int32_t v, bool is_hub, int hc, const int32_t *__restrict__ tail_list, int tc,
                                                const uint32_t *bm, const unsigned short *pre, uint32_t *orow, int sub, const KcFilter flt) {
    const int64_t hb = hoff[v];
    const int hl = int(hoff[v + 1] - hb);
    if (is_hub && v < dense_limit) {
        const int nw = int(bitset_words(v));
        if (nw * 4 + 32 < hl * 2) {
            kc_stream_bitset(bmpool + bmoff[v], nw, sub, bm, pre, orow);
            return;
        }
    }
    if (hc > 0) kc_stream_list(hadj + hb, hl, sub, bm, pre, orow);
    if (!is_hub && tc > 0) {
        const int64_t tb = toff[v];
        kc_stream_tail(tadj + tb, int(toff[v + 1] - tb), sub, tail_list, tc, hc, orow, flt);
    }
}

// ---- the BUILD as ONE STREAM OF STEPS per 16-lane group (round 5; mode 2 of k_kc_block) ---------------------------------------------------
// Member by member (mode 0) a group pays, per member, the chain member id -> extents -> first units of the row before it streams it; from scale 24 on
// the oriented containers no longer sit in the Infinity Cache and every link is an HBM round trip: the BUILD of k = 4 at scale 26 moved its 1.6 TB of
// member rows at 2.1 TB/s.  Here, as in the triangle kernels' StepStream (tc.hip): the extents of up to 256 members are fetched by
// 256 threads AT ONCE into LDS descriptors, and a group then walks its members' rows as one stream of steps (16 consecutive 16-byte units, one per
// lane) with D steps in flight per lane ACROSS member boundaries — every load unconditional (a lane without a unit in the step re-reads the row's last
// unit, a group past its last member the fallback address) so that the waits are counted.  A member's row = its hub part (bitset words or 16-bit list)
// followed by its tail part (32-bit ids); a step carries its member's index, the probe reads what else it needs from the descriptor.
struct __attribute__((aligned(16))) KcDesc {
    unsigned long long ph, pt;  // byte addresses of the hub part (list or bitset container) and of the tail part
    uint32_t hu, tu;            // 16-byte units of the two parts; bit 31 of hu: the hub part is a bitset container
    uint32_t hl, tl;            // ids of the list / tail part (the last unit of a part is cut there)
};
template <int D>
struct KcStream {
    static constexpr int S = D == 1 ? 2 : D;
    const KcDesc *desc;
    const char *fallback;
    int hi, sub, G;
    int e, units, hu, last, lim, j;
    const char *ph, *pt;
    int pending;
    kc_u4u p[S];
    int pj[S], pe[S];
    __device__ __forceinline__ void open_row() {
        if (e < hi) {
            const KcDesc d = desc[e];
            ph = reinterpret_cast<const char *>(d.ph);
            pt = reinterpret_cast<const char *>(d.pt);
            hu = int(d.hu & 0x7fffffffu);
            units = hu + int(d.tu);
        } else {
            ph = pt = fallback;
            hu = units = 0;
        }
        last = max(units - 1, 0);
        lim = units + sub;
        j = sub;
    }
    __device__ __forceinline__ void issue(int k) {
        const int jj = min(j, last);
        const char *a = jj < hu ? ph + 16 * size_t(jj) : pt + 16 * size_t(jj - hu);
        p[k] = *reinterpret_cast<const kc_u4u *>(units > 0 ? a : fallback);
        pj[k] = j < units ? j : -1;
        pe[k] = e;
        pending += e < hi ? 1 : 0;
        j += 16;
        if (j >= lim) {  // the member's row is through (uniform per group): the group's next member
            e += G;
            open_row();
        }
    }
    __device__ __forceinline__ void start(const KcDesc *d, const void *fb, int n, int tid, int nthreads) {
        desc = d;
        fallback = static_cast<const char *>(fb);
        hi = n;
        sub = tid & 15;
        G = nthreads >> 4;
        e = tid >> 4;
        pending = 0;
        open_row();
#pragma unroll
        for (int k = 0; k < S - (D == 1 ? 1 : 0); ++k) issue(k);
    }
    template <class Probe>
    __device__ __forceinline__ void run(Probe probe) {
        if constexpr (D == 1) {
            while (pending > 0) {
                pending -= 1;
                issue(1);
                if (pj[0] >= 0) probe(p[0], pj[0], pe[0]);
                if (!(pending > 0)) break;
                pending -= 1;
                issue(0);
                if (pj[1] >= 0) probe(p[1], pj[1], pe[1]);
            }
        } else {
            while (pending > 0) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const kc_u4u pc = p[k];
                    const int jc = pj[k], ec = pe[k];
                    pending -= 1;
                    issue(k);
                    if (jc >= 0) probe(pc, jc, ec);
                }
            }
        }
    }
};

// ---- counting ------------------------------------------------------------------------------------------------
// Per-lane recursion for local graphs of <= 32 vertices (one 32-bit word per set).  Number of LV-cliques inside cand.
template <int LV>
__device__ __forceinline__ unsigned long long lane_cliques(const uint32_t *rows, uint32_t cand) {
    if constexpr (LV == 1) {
        return (unsigned long long)__popc(cand);
    } else {
        unsigned long long s = 0;
        uint32_t it = cand;
        while (it) {
            const int j = __ffs(it) - 1;
            it &= it - 1;
            s += lane_cliques<LV - 1>(rows, cand & rows[j]);
        }
        return s;
    }
}

// Wave-cooperative recursion: a set of up to 64*32*WPL local vertices is WPL words per lane.  Returns this lane's
// share of the number of LV-cliques inside cand (sum over lanes = the count).  `rows` may be LDS or global.
template <int LV, int WPL>
__device__ __forceinline__ unsigned long long wave_cliques(const uint32_t *rows, int W, const uint32_t (&cand)[WPL], int lane);

// expands every member of word-slice K of cand
template <int LV, int WPL, int K>
__device__ __forceinline__ unsigned long long expand_slice(const uint32_t *rows, int W, const uint32_t (&cand)[WPL], int lane) {
    unsigned long long s = 0;
    uint32_t it = cand[K];
    while (true) {
        const unsigned long long nz = __ballot(it != 0);
        if (!nz) break;
        const int L = __ffsll((long long)nz) - 1;                 // wave-uniform
        const uint32_t word = __builtin_amdgcn_readlane(it, L);    // wave-uniform
        const int bit = __ffs(word) - 1;
        if (lane == L) it &= it - 1;
        const int j = ((K * 64 + L) << 5) + bit;
        const uint32_t *rj = rows + size_t(j) * W;
        uint32_t nc[WPL];
        nc[0] = lane < W ? (cand[0] & rj[lane]) : 0u;
        if constexpr (WPL > 1) nc[1] = 64 + lane < W ? (cand[1] & rj[64 + lane]) : 0u;
        s += wave_cliques<LV - 1, WPL>(rows, W, nc, lane);
    }
    return s;
}

template <int LV, int WPL>
__device__ __forceinline__ unsigned long long wave_cliques(const uint32_t *rows, int W, const uint32_t (&cand)[WPL], int lane) {
    static_assert(WPL == 1 || WPL == 2, "bit rows are one or two words per lane");
    if constexpr (LV == 1) {
        unsigned long long s = (unsigned long long)__popc(cand[0]);
        if constexpr (WPL > 1) s += (unsigned long long)__popc(cand[1]);
        return s;
    } else {
        unsigned long long s = expand_slice<LV, WPL, 0>(rows, W, cand, lane);
        if constexpr (WPL > 1) s += expand_slice<LV, WPL, 1>(rows, W, cand, lane);
        return s;
    }
}

// Per-LANE enumeration for bit-matrices of any width: the path so far is a list of row pointers, the candidate set of a
// level is never materialised — its words are recomputed as the AND of the path's rows, so a level costs two registers
// (word index, remaining bits).  R = vertices still to choose.  Returns the number of R-cliques inside ∧ path rows.
struct KcPath {
    const uint32_t *row[10];
};
template <int R, int P>
__device__ __forceinline__ unsigned long long lane_enum(const uint32_t *rows, int WS, int W, KcPath &path) {
    unsigned long long s = 0;
    for (int t = 0; t < W; ++t) {
        uint32_t c = path.row[0][t];
#pragma unroll
        for (int q = 1; q < P; ++q) c &= path.row[q][t];
        if constexpr (R == 1) {
            s += (unsigned long long)__popc(c);
        } else {
            while (c) {
                const int l = (t << 5) + __ffs(c) - 1;
                c &= c - 1;
                path.row[P] = rows + size_t(l) * WS;
                s += lane_enum<R - 1, P + 1>(rows, WS, t + 1, path);  // the matrix is strictly lower triangular
            }
        }
    }
    return s;
}

// popc(x) + acc in ONE instruction: v_bcnt_u32_b32 has an accumulate operand, but the compiler re-associates a sum of popcounts into bcnt(x, 0) plus
// v_add3 — a quarter more VALU instructions in the k = 4 count loops, which are bound by exactly those (round 6: and + bcnt per word, two chains)
__device__ __forceinline__ uint32_t kc_popc_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// Σ_{q < nq} popc(word q of the row held one-word-per-lane in `wl`  &  rj[q]); nq is wave-uniform, four LDS reads in flight
__device__ __forceinline__ uint32_t kc4_and_popc(uint32_t wl, const uint32_t *rj, int nq) {
    uint32_t acc = 0, acc1 = 0;
    int q = 0;
    for (; q + 8 <= nq; q += 8) {
        const uint32_t b0 = rj[q], b1 = rj[q + 1], b2 = rj[q + 2], b3 = rj[q + 3], b4 = rj[q + 4], b5 = rj[q + 5], b6 = rj[q + 6], b7 = rj[q + 7];
        acc = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q)) & b0, acc);
        acc1 = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 1)) & b1, acc1);
        acc = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 2)) & b2, acc);
        acc1 = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 3)) & b3, acc1);
        acc = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 4)) & b4, acc);
        acc1 = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 5)) & b5, acc1);
        acc = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 6)) & b6, acc);
        acc1 = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q + 7)) & b7, acc1);
    }
    for (; q + 4 <= nq; q += 4) {
        const uint32_t b0 = rj[q], b1 = rj[q + 1], b2 = rj[q + 2], b3 = rj[q + 3];
        const uint32_t a0 = uint32_t(__builtin_amdgcn_readlane(int(wl), q)), a1 = uint32_t(__builtin_amdgcn_readlane(int(wl), q + 1)),
                       a2 = uint32_t(__builtin_amdgcn_readlane(int(wl), q + 2)), a3 = uint32_t(__builtin_amdgcn_readlane(int(wl), q + 3));
        acc = kc_popc_acc(a0 & b0, acc);
        acc1 = kc_popc_acc(a1 & b1, acc1);
        acc = kc_popc_acc(a2 & b2, acc);
        acc1 = kc_popc_acc(a3 & b3, acc1);
    }
    for (; q < nq; ++q) acc = kc_popc_acc(uint32_t(__builtin_amdgcn_readlane(int(wl), q)) & rj[q], acc);
    return acc + acc1;
}

// k = 4 count of one matrix row i, wave-cooperative:  Σ_{j ∈ rows[i], j in the band} popc(rows[i] & rows[j]).
// The lanes take the SET BITS j of row i (64 neighbours per chunk, found through a popcount prefix over the row's
// words: lane t holds word t), so every lane walks words 0..j>>5 of its own row j — neighbouring lanes have
// neighbouring j, i.e. the same trip count, and rows[i][q] is a wave-uniform v_readlane (no LDS read).  `band` holds
// the rows of the vertices j0, j0+1, … (stride BS words, zero above the diagonal) in LDS; only the words [jw0, jw1)
// of row i (the band's columns) are enumerated.
template <int WPL>
__device__ __forceinline__ unsigned long long kc4_row(const uint32_t (&wt)[WPL], const uint32_t *band, int BS, int j0, int jw0, int jw1, int lane) {
    unsigned long long total = 0;
#pragma unroll
    for (int h = 0; h < WPL; ++h) {
        const int t_me = lane + 64 * h;
        const uint32_t w = (t_me >= jw0 && t_me < jw1) ? wt[h] : 0u;
        const int pc = __popc(w);
        int P = pc;  // inclusive prefix over the lanes
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(P, sft);
            if (lane >= sft) P += o;
        }
        const int nb = __builtin_amdgcn_readlane(P, 63);
        const int ex_me = P - pc;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const bool act = c0 + lane < nb;
            const int rr = act ? c0 + lane : nb - 1;
            int L = 0;  // number of lanes whose prefix is <= rr = the lane holding my neighbour's word
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1)
                if (__shfl(P, L + sft - 1) <= rr) L += sft;
            uint32_t word = uint32_t(__shfl(int(w), L));
            int k = rr - __shfl(ex_me, L);  // k-th set bit of word (0-based)
            int pos = 0;
#pragma unroll
            for (int sft = 16; sft > 0; sft >>= 1) {
                const int c = __popc((word >> pos) & ((1u << sft) - 1u));
                if (k >= c) {
                    k -= c;
                    pos += sft;
                }
            }
            const int tj = L + 64 * h;
            const int j = (tj << 5) + pos;
            const int tmax = __builtin_amdgcn_readlane(tj, min(63, nb - c0 - 1));  // ranks ascend with the lane: the last active lane has the largest word index
            const uint32_t *rj = band + size_t(j - j0) * BS;
            uint32_t acc = 0;
#ifdef GMSX_KC_NO_INNER  // A/B build (wrong counts): the k = 4 count without its AND + popcount loops (what finding the neighbours alone takes)
            acc = uint32_t(rj[0] & 1u) + uint32_t(tmax);
#else
#pragma unroll
            for (int g = 0; g < WPL; ++g)
                if (tmax >= 64 * g) acc += kc4_and_popc(wt[g], rj + 64 * g, min(tmax - 64 * g, 63) + 1);
#endif
            if (act) total += acc;
        }
    }
    return total;
}

// … the same sum with the neighbours of the row found WITHOUT the 64-lane shuffle scan, the six-step search and the five-step select (round 5, the
// slab matrices: 0.16 s of the 0.39 s their bins took at scale 26 were those, 0.12 s the AND + popcount loops): only the band's column words
// [jw0, jw1) of row i can hold neighbours, so the lanes that hold them write their set bits j - j0 into the wave's LDS list behind a DPP prefix sum,
// and lane r of a chunk reads entry r.  (Tried and dropped: the lanes on the WORDS of the two rows, the neighbours walked by scalar code — no scan,
// no list, one ds_read_b32 + v_and + v_bcnt per neighbour: 2.5 times SLOWER, 0.92 s for the slab bins: one LDS round trip per neighbour and wave
// instead of 64 neighbours in flight.)
static constexpr int kKcRowList = 256;  // entries of a wave's neighbour list (2 bytes each)
template <int WPL>
__device__ __forceinline__ unsigned long long kc4_row_list(const uint32_t (&wt)[WPL], const uint32_t *band, int BS, int j0, int jw0, int jw1, int lane,
                                                           unsigned short *wbuf, int cap) {
    unsigned long long total = 0;
#pragma unroll
    for (int h = 0; h < WPL; ++h) {
        if (jw1 <= 64 * h || jw0 >= 64 * (h + 1)) continue;  // (uniform: the band's columns lie in one half, rarely in two)
        const int t_me = lane + 64 * h;
        uint32_t bits = (t_me >= jw0 && t_me < jw1) ? wt[h] : 0u;
        while (__ballot(bits != 0u) != 0) {  // (one trip unless the row has more neighbours in the band than the list holds)
            const int pc = __popc(bits);
            int incl = pc;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
            {
                const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
                incl += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
            }
            const bool fits = incl <= cap;  // a run of lanes from lane 0 on (a word has at most 32 bits <= cap)
            const int nfit = __popcll(__ballot(fits));
            const int nb = __builtin_amdgcn_readlane(incl, uni32(nfit - 1));
            if (fits) {
                int at = incl - pc;
                const int jb = (t_me << 5) - j0;
                while (bits) {
                    wbuf[at++] = (unsigned short)(jb + __ffs(bits) - 1);
                    bits &= bits - 1u;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int c0 = 0; c0 < nb; c0 += 64) {
                const bool act = c0 + lane < nb;
                const int jr = wbuf[act ? c0 + lane : nb - 1];  // j - j0, ascending with the lane
                const int tmax = (__builtin_amdgcn_readlane(jr, min(63, nb - c0 - 1)) + j0) >> 5;  // the last active lane has the largest j
                const uint32_t *rj = band + size_t(jr) * BS;
                uint32_t acc = 0;
#ifdef GMSX_KC_NO_INNER  // A/B build (wrong counts): the count without its AND + popcount loops
                acc = uint32_t(rj[0] & 1u) + uint32_t(tmax);
#else
#pragma unroll
                for (int g = 0; g < WPL; ++g)
                    if (tmax >= 64 * g) acc += kc4_and_popc(wt[g], rj + 64 * g, min(tmax - 64 * g, 63) + 1);
#endif
                if (act) total += acc;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    return total;
}

// k = 4 count of a whole LDS bit-matrix, by PAIR LISTS (round 5):  Σ_i Σ_{j ∈ rows[i]} popc(rows[i] & rows[j]).
// Measured on round 4's kernels (k = 4, scale 22): with the AND + popcount loops compiled out of kc4_row the call still took 20.9 of 22.6 ms — finding
// the neighbours j of a row (a 64-lane prefix scan, a six-step binary search and a five-step select per row, for 33 neighbours on average) cost five
// times what the intersections did; the narrower matrices' lane-per-word loop idled behind its longest lane.  Here a wave takes 64 matrix cells of ONE
// COLUMN WORD w — rows 64 b … 64 b + 63 — writes their set bits (i, j) into its own LDS pair buffer behind one wave prefix sum, and then every lane
// intersects one pair: all pairs of the task have j in word w, so every lane walks exactly the words 0 … w of its two rows (row j has no bit at or
// above j) — full lanes, one trip count, no search.  A task with more pairs than the buffer holds is taken in runs of lanes.
template <bool TRI = false>
__device__ __forceinline__ unsigned long long kc4_count_pairs(const uint32_t *rows, int WS, int d, int W, uint32_t *wbuf, int cap, int *next_task, int lane) {
    auto rowp = [&](int i) -> const uint32_t * { return TRI ? rows + kc_tri_off(i) : rows + size_t(i) * WS; };
    unsigned long long total = 0;
    d = uni32(d);
    // tasks = (column word w, block of 64 rows b) with rows beyond column 32 w, i.e. b >= w / 2 (the matrix is strictly lower triangular: the blocks
    // above the diagonal are empty and get no ticket): the pairs (k, b), k <= b < nblk, of a triangular enumeration, two column words 2k, 2k + 1 each
    const int nblk = (d + 63) >> 6, ntri = nblk * (nblk + 1) / 2, ntask = 2 * ntri;
    while (true) {
        // tasks from a workgroup-wide ticket counter in LDS (zeroed before the barrier in front of the count), the long ones first: the waves of a
        // workgroup end together.  The ticket is pinned to a scalar register: the task, its column word w and the trip count of the intersections below
        // are then scalar — a loop the compiler unrolls and waits for once, not an exec-masked one
        int task = 0;
        if (lane == 0) task = atomicAdd(next_task, 1);
        task = uni32(task);
        if (task >= ntask) break;
        const int tt = ntri - 1 - (task >> 1);  // the far blocks of the high column words first: the long tasks
        int blk = int((__builtin_sqrtf(8.0f * float(tt) + 1.0f) - 1.0f) * 0.5f);
        while (blk * (blk + 1) / 2 > tt) --blk;
        while ((blk + 1) * (blk + 2) / 2 <= tt) ++blk;
        const int w = 2 * (tt - blk * (blk + 1) / 2) + (task & 1);
        if (w >= W) continue;  // (an odd number of column words)
        const int i = blk * 64 + lane;
        uint32_t bits = (i < d && (w << 5) < i) ? rowp(i)[w] : 0u;
#ifdef GMSX_KC_CELLS_ONLY  // A/B build (wrong counts): the count phase reads the matrix cells and nothing else
        total += __popc(bits);
        bits = 0u;
#endif
        while (__ballot(bits != 0u) != 0) {  // wave-uniform
            const int pc = __popc(bits);
            // inclusive prefix over the 64 lanes without an LDS round trip (six dependent ds_bpermutes per task were a third of the phase): DPP row_shr
            // scans inside the four rows of 16, then the totals of the rows in front through three v_readlanes
            int incl = pc;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
            {
                const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
                incl += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
            }
            const bool fits = incl <= cap;  // a run of lanes from lane 0 on (a cell has at most 32 pairs <= cap: at least one lane)
            const int nfit = __popcll(__ballot(fits));
            const int npairs = __builtin_amdgcn_readlane(incl, uni32(nfit - 1));
            if (fits) {
                int at = incl - pc;
                while (bits) {
                    wbuf[at++] = (uint32_t(i) << 16) | uint32_t((w << 5) + __ffs(bits) - 1);
                    bits &= bits - 1u;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int c0 = 0; c0 < npairs; c0 += 64) {
                const int r = c0 + lane;
                const bool act = r < npairs;
                const uint32_t pr = wbuf[act ? r : 0];
                const uint32_t *ri = rowp(int(pr >> 16)), *rj = rowp(int(pr & 0xffffu));
                uint32_t acc = 0;
                int q = 0;
#ifdef GMSX_KC_NO_INNER
                acc = (ri[0] ^ rj[0]) & 1u;
                q = w + 1;
#endif
                uint32_t acc1 = 0;  // two chains of kc_popc_acc (and + bcnt per word: no separate adds)
                for (; q + 8 <= w + 1; q += 8) {  // eight words per trip: the two address increments and the loop control once per 16 LDS words
                    const uint32_t a0 = ri[q], a1 = ri[q + 1], a2 = ri[q + 2], a3 = ri[q + 3], a4 = ri[q + 4], a5 = ri[q + 5], a6 = ri[q + 6], a7 = ri[q + 7];
                    const uint32_t b0 = rj[q], b1 = rj[q + 1], b2 = rj[q + 2], b3 = rj[q + 3], b4 = rj[q + 4], b5 = rj[q + 5], b6 = rj[q + 6], b7 = rj[q + 7];
                    acc = kc_popc_acc(a0 & b0, acc);
                    acc1 = kc_popc_acc(a1 & b1, acc1);
                    acc = kc_popc_acc(a2 & b2, acc);
                    acc1 = kc_popc_acc(a3 & b3, acc1);
                    acc = kc_popc_acc(a4 & b4, acc);
                    acc1 = kc_popc_acc(a5 & b5, acc1);
                    acc = kc_popc_acc(a6 & b6, acc);
                    acc1 = kc_popc_acc(a7 & b7, acc1);
                }
                for (; q + 4 <= w + 1; q += 4) {
                    const uint32_t a0 = ri[q], a1 = ri[q + 1], a2 = ri[q + 2], a3 = ri[q + 3];
                    const uint32_t b0 = rj[q], b1 = rj[q + 1], b2 = rj[q + 2], b3 = rj[q + 3];
                    acc = kc_popc_acc(a0 & b0, acc);
                    acc1 = kc_popc_acc(a1 & b1, acc1);
                    acc = kc_popc_acc(a2 & b2, acc);
                    acc1 = kc_popc_acc(a3 & b3, acc1);
                }
                for (; q <= w; ++q) acc = kc_popc_acc(ri[q] & rj[q], acc);
                if (act) total += acc + acc1;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// S: wave per pivot, 1 <= d+ <= 32.  The 32 x 32 adjacency matrix is one word per member:
//   * rows of HUB members come from inverted gathers into their bitset containers (as k_tc_wave_hub): lane j asks
//     "is member w_j in N+(v_i)", two rows per gather (one per half-wave), eight gathers in flight, and the ballot IS
//     the pair of row words — no atomics, no LDS;
//   * rows of TAIL members are streamed (four 16-lane groups, 16-byte loads) and every id — hub or tail — is probed in
//     one 64 x 4 bucket set of the pivot's members (id -> local index; one ds_read_b128 + four compares).  A pivot
//     with five members in one bucket uses the open-addressing map instead.
// No bitmap: 1.4 KB of LDS per wave, full occupancy.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t readlane64(int64_t x, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(uint64_t(x)), l);
    const uint32_t hi = __builtin_amdgcn_readlane(uint32_t(uint64_t(x) >> 32), l);
    return int64_t((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ uint32_t kcs_bucket(int32_t w) { return (uint32_t(w) ^ (uint32_t(w) >> 6)) & 63u; }

template <bool BUCKET>
__device__ __forceinline__ int kcs_find(const int32_t *keys, const unsigned char *vals, int32_t w) {
    if (BUCKET) {
        const uint32_t b = kcs_bucket(w) * 4;
        const int4 k4 = *reinterpret_cast<const int4 *>(keys + b);
        const uint32_t m = uint32_t(k4.x == w) | (uint32_t(k4.y == w) << 1) | (uint32_t(k4.z == w) << 2) | (uint32_t(k4.w == w) << 3);
        if (!m) return -1;
        return int(vals[b + (__ffs(m) - 1)]);
    } else {
        uint32_t h = kc_hash(w, 24);
        while (true) {
            const int32_t x = keys[h];
            if (x == w) return int(vals[h]);
            if (x == -1) return -1;
            h = (h + 1) & 255u;
        }
    }
}

// both containers of the rows of the tail members i0 .. i0+3 (one per 16-lane group); hits set bits of rows[i]
template <bool BUCKET>
__device__ __forceinline__ void kcs_tail_rows(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff,
                                              const int32_t *__restrict__ tadj, const int32_t *keys, const unsigned char *vals, uint32_t *rows,
                                              int32_t my, int hc, int d, int lane, uint32_t fwd_mask /* members whose row is not in the arena */) {
    const int grp = lane >> 4, sub = lane & 15;
    // the extents of the NEXT trip's rows are on their way while this trip's rows are streamed (round 5: extents, then rows — two dependent round trips
    // per trip of four members — were most of what a pivot of a dozen tail members waited for)
    int32_t vn = __shfl(my, min(hc + grp, d - 1));
    int64_t n_hb = hoff[vn], n_he = hoff[vn + 1], n_tb = toff[vn], n_te = toff[vn + 1];
    for (int i0 = hc; i0 < d; i0 += 4) {
        const int i = i0 + grp;
        const int64_t hb = n_hb, tb = n_tb;
        const int hl = int(n_he - n_hb), tl = int(n_te - n_tb);
        const int32_t m_max = __shfl(my, max(i - 1, 0));  // member i - 1
        vn = __shfl(my, min(i + 4, d - 1));
        n_hb = hoff[vn];
        n_he = hoff[vn + 1];
        n_tb = toff[vn];
        n_te = toff[vn + 1];
        if (i >= d || !((fwd_mask >> i) & 1u)) continue;  // per group
        uint32_t bits = 0;
        // (round 5) both parts of the row ascend, and only the members BELOW v can be in it: the hub part is of use up to the pivot's largest hub
        // member — not at all without hub members —, the tail part up to the tail member in front of v — not at all for the first tail member
        if (hc > 0) {
            const uint32_t h_max = uint32_t(__builtin_amdgcn_readlane(my, (hc - 1) & 63));
            const uint16_t *row = hadj + hb;
            for (int j = sub * 8; j < hl; j += 128) {
                const kc_u4u p = kc_load8(row, j, hl);
                if ((p.x & 0xffffu) > h_max) break;  // (the pad 0xFFFF included)
                const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t lo = pw[q] & 0xffffu, hi = pw[q] >> 16;
                    const int a = lo != 0xffffu ? kcs_find<BUCKET>(keys, vals, int32_t(lo)) : -1;  // 0xFFFF = pad, also a possible tail id
                    const int c = hi != 0xffffu ? kcs_find<BUCKET>(keys, vals, int32_t(hi)) : -1;
                    if (a >= 0) bits |= 1u << a;
                    if (c >= 0) bits |= 1u << c;
                }
            }
        }
        if (i > hc) {
            const int32_t *row = tadj + tb;
            for (int j = sub; j < tl; j += 16) {
                const int32_t x = row[j];
                if (x > m_max) break;
                const int a = kcs_find<BUCKET>(keys, vals, x);
                if (a >= 0) bits |= 1u << a;
            }
        }
        if (bits) atomicOr(&rows[i], bits);
    }
}

template <int LV, bool VTX>
__global__ __launch_bounds__(256) void k_kc_small(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                  const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                  const int64_t *__restrict__ bmoff, const uint32_t *__restrict__ bmpool,
                                                  const int32_t *__restrict__ order, int64_t first, int64_t end, int nparts,
                                                  int part, unsigned long long *__restrict__ acc, const int32_t *__restrict__ oldid,
                                                  unsigned long long *__restrict__ vcounts, KcRev rv) {
    constexpr int SIZE = 256;
    __shared__ __attribute__((aligned(16))) int32_t keys_all[4 * SIZE];
    __shared__ __attribute__((aligned(4))) unsigned char vals_all[4 * SIZE];
    __shared__ uint32_t rows_all[4 * 32];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *keys = keys_all + wave * SIZE;
    unsigned char *vals = vals_all + wave * SIZE;
    uint32_t *rows = rows_all + wave * 32;
    const int64_t nwaves = int64_t(gridDim.x) * 4;
    const int half = lane >> 5, jl = lane & 31;
    unsigned long long cnt = 0;
    // (Software-pipelining the pivots of a wave — the extents of pivot k + 1 and the id of pivot k + 2 loaded under pivot k — was measured SLOWER,
    // 73 against 59 ms at scale 26: the grid gives a wave one or two pivots, and the staged prologue is two more round trips for each.)
    for (int64_t q = int64_t(blockIdx.x) * 4 + wave;; q += nwaves) {
        const int64_t pos = first + q * nparts + part;
        if (pos >= end) break;  // uniform per wave
        const int32_t u = order[pos];
        const int64_t hb = hoff[u], tb = toff[u];
        int hc = int(hoff[u + 1] - hb);
        const int tc = int(toff[u + 1] - tb);
        if (hc > 0 && hadj[hb + hc - 1] == 0xFFFFu) --hc;  // drop the pad
        const int d = hc + tc;                               // <= 32
        for (int i = lane; i < SIZE; i += 64) keys[i] = -1;
        if (lane < 32) rows[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        int32_t my = -1;
        int64_t rbm = 0;
        if (lane < hc) {
            my = int32_t(hadj[hb + lane]);
            rbm = bmoff[my];
        } else if (lane < d) {
            my = tadj[tb + (lane - hc)];
        }
        // tail members whose receiver took their edge (round 6b: k_kc_reverse_tail serves the narrow pivots too): the finished row is ONE word of the arena;
        // only the others are streamed forward — and only for them is the member set below built at all
        uint32_t fwd_mask = 0u;
        {
            uint32_t r = kKcRelForward;
            if (rv.relt && lane >= hc && lane < d && lane > 0) r = rv.relt[tb + (lane - hc)];
            if (r != kKcRelForward) rows[lane] = rv.arena[rv.aoff[u] + r];
            fwd_mask = uint32_t(__ballot(lane >= hc && lane < d && lane > 0 && r == kKcRelForward));
        }
        bool bucketed = true;
        if (fwd_mask) {  // the member set is only probed by the streamed rows of tail members
            bool ovf = false;
            if (lane < d) {
                int32_t *b = keys + kcs_bucket(my) * 4;
                int s = 0;
                for (; s < 4; ++s)
                    if (atomicCAS(b + s, -1, my) == -1) break;
                if (s < 4) vals[(b - keys) + s] = (unsigned char)lane;
                else ovf = true;
            }
            bucketed = __ballot(ovf) == 0;
            if (!bucketed) {  // rare: rebuild as an open-addressing map in the same table
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < SIZE; i += 64) keys[i] = -1;
                __builtin_amdgcn_wave_barrier();
                if (lane < d) {
                    uint32_t h = kc_hash(my, 24);
                    while (atomicCAS(&keys[h], -1, my) != -1) h = (h + 1) & 255u;
                    vals[h] = (unsigned char)lane;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // rows of the hub members 1 .. hc-1 (member 0 has no smaller member): two rows per gather, eight gathers in flight
        if (hc > 1) {
            const int32_t wj = __shfl(my, jl);
            const bool jv = jl < hc;
            for (int p0 = 0; p0 * 2 < hc; p0 += 8) {
                uint32_t wd[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ia = 2 * (p0 + k), ib = ia + 1;  // <= 31 + 16: masked below
                    const int32_t va = ia < hc ? __builtin_amdgcn_readlane(my, ia & 63) : 0, vb = ib < hc ? __builtin_amdgcn_readlane(my, ib & 63) : 0;
                    const int64_t ba = readlane64(rbm, ia & 63), bb = readlane64(rbm, ib & 63);
                    const int32_t vi = half ? vb : va;
                    const int64_t rb = half ? bb : ba;
                    wd[k] = (jv && wj < vi) ? bmpool[rb + (uint32_t(wj) >> 5)] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ia = 2 * (p0 + k), ib = ia + 1;
                    const unsigned long long m = __ballot((wd[k] >> (uint32_t(wj) & 31u)) & 1u);
                    if (lane == 0 && ia < hc) rows[ia] = uint32_t(m);
                    if (lane == 32 && ib < hc) rows[ib] = uint32_t(m >> 32);
                }
            }
        }
        if (fwd_mask) {
            if (bucketed)
                kcs_tail_rows<true>(hoff, hadj, toff, tadj, keys, vals, rows, my, hc, d, lane, fwd_mask);
            else
                kcs_tail_rows<false>(hoff, hadj, toff, tadj, keys, vals, rows, my, hc, d, lane, fwd_mask);
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (VTX) {
            // per-vertex triangle counts (vertex_count2 = 2 x triangles at the vertex): a set bit (i, j) is the triangle
            // (u, v_i, v_j); member t collects its row popcount and its column sum, the pivot the number of bits
            uint32_t rp = 0, cs = 0;
            if (lane < d) rp = uint32_t(__popc(rows[lane]));
            for (int i = 0; i < d; ++i) cs += (rows[i] >> lane) & 1u;  // rows[i] is a broadcast read; lanes >= d see zeros
            if (lane < d && rp + cs) atomicAdd(&vcounts[oldid[my]], 2ull * (rp + cs));
            uint32_t ps = rp;
            for (int sft = 32; sft > 0; sft >>= 1) ps += __shfl_xor(ps, sft);
            if (lane == 0 && ps) atomicAdd(&vcounts[oldid[u]], 2ull * ps);
            cnt += rp;
        } else {
            if (lane < d) cnt += lane_cliques<LV>(rows, rows[lane]);
        }
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

// ---------------------------------------------------------------------------------------------
// M / L: workgroup per pivot.  W = words per bit row, WS = row stride.  GLOBAL_ROWS=false: bit-matrix in dynamic LDS
// (d <= 1024); true: in a per-workgroup global slab (d up to 64*32*WPL), workgroups walk their pivots with a grid stride.
// LDS layout: LDS variants — static [bm: 2048 u32][pre: 2048 u16][filter: 1024 u32] at LDS address 0, dynamic [rows][col][fwd]; slab variants — static filter,
// dynamic [bm: 2048 u32] [pre: 2048 u16] [row stage: nwaves*4*W u32]
// ---------------------------------------------------------------------------------------------
template <int LV, int WPL, bool GLOBAL_ROWS, bool VTX, int PIPE = GLOBAL_ROWS ? 1 : 0 /* the BUILD: 0 member by member, 1 three-stage member pipeline, 2 step stream */,
          bool TRI = false /* LDS matrix stored triangularly (kc_tri_off; k = 4 only) */,
          bool EXPORT = false /* BUILD only: the finished matrix leaves for the pool (KcExport), k_kc4_mfma counts it */>
__global__ __launch_bounds__(1024) void k_kc_block(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                   const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                   const int64_t *__restrict__ bmoff, const uint32_t *__restrict__ bmpool,
                                                   int32_t dense_limit, const int32_t *__restrict__ order, int64_t first, int64_t end,
                                                   int nparts, int part, int dmax, int W, int WS, int WT, uint32_t *__restrict__ slabs,
                                                   unsigned long long *__restrict__ acc, const int32_t *__restrict__ oldid,
                                                   unsigned long long *__restrict__ vcounts, KcRev rv, KcExport ex) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ unsigned long long red[16];
    __shared__ int wave_tot[16];
    __shared__ int s_nfwd;  // members of the current pivot that are streamed forward (the list `fwd` below; LDS matrices only)
    constexpr int kKcFilterWords = 1024;
    // LDS-matrix variants (round 6): bitmap, prefix counts and tail filter as ONE STATIC array — the only 16-byte aligned static of the kernel, so it is laid out
    // at LDS address 0 and every probe's word address IS its ds_read address (the triangle kernels have always had this); as part of the dynamic LDS the
    // bitmap's base had to be added to each of the eight word addresses of a streamed unit.  The slab variants keep the bitmap in the dynamic part: their
    // k = 4 count reuses everything from the first dynamic word on as its row band, and 16 KB of static LDS beside that would not fit the CU.
    constexpr int kStaticWords = GLOBAL_ROWS ? kKcFilterWords : kBitmapWords + kBitmapWords / 2 + kKcFilterWords;
    __shared__ __attribute__((aligned(16))) uint32_t kc_fixed[kStaticWords];
    uint32_t *fltw = GLOBAL_ROWS ? kc_fixed : kc_fixed + kBitmapWords + kBitmapWords / 2;  // the pivot's tail members, one bit per (id mod 32 x words): kc_tail_find
    const KcFilter flt{fltw, uint32_t(kKcFilterWords - 1)};
#ifdef GMSX_KC_NO_PAIRS  // A/B build: round 4's k = 4 counts (kc4_row for wide matrices, one lane per matrix word for the others)
    constexpr bool kc4_pairs_enabled = false;
#else
    constexpr bool kc4_pairs_enabled = !GLOBAL_ROWS;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = int(blockDim.x), nwaves = nthreads >> 6;  // 256 … 1024 threads: big LDS bit-matrices leave one workgroup per CU
    // WS = row stride in words: W for the wave-cooperative recursion (lane = word index), W + 1 for the lane-per-pair
    // k = 4 count (random rows per lane: an odd stride spreads them over the LDS banks)
    static_assert(!TRI || (LV == 2 && !GLOBAL_ROWS && !VTX && PIPE != 2), "the triangular layout serves the k = 4 count on LDS matrices");
    const size_t mat_words = TRI ? size_t(kc_tri_off(dmax)) : size_t(dmax) * WS;  // words of the LDS matrix of this bin
    static_assert(!EXPORT || (LV == 2 && !VTX && (GLOBAL_ROWS || TRI) && PIPE == 1), "the variants whose k = 4 count can run on the matrix cores");
    constexpr bool kCanExport = EXPORT, exporting = EXPORT;  // (compile-time: the BUILD-only kernels carry none of the count phases' registers)
    uint32_t *rows = GLOBAL_ROWS ? slabs + size_t(blockIdx.x) * size_t(dmax) * size_t(WS) : smem;
    uint32_t *bm = GLOBAL_ROWS ? smem : kc_fixed;
    // dynamic LDS of the LDS-matrix variants: [rows: mat_words][column counters (VTX)][forward list / step-stream descriptors]
    auto rowp = [&](int i) -> uint32_t * { return TRI ? rows + kc_tri_off(i) : rows + size_t(i) * WS; };
    unsigned short *pre = reinterpret_cast<unsigned short *>(bm + kBitmapWords);

    unsigned long long cnt = 0;
    bool flt_dirty = true;
    for (int64_t q = blockIdx.x;; q += gridDim.x) {
        const int64_t pos = first + q * nparts + part;
        if (pos >= end) break;  // uniform per block
        const int32_t u = order[pos];
        const int64_t hb = hoff[u], tb = toff[u];
        int hc = int(hoff[u + 1] - hb);
        const int tc = int(toff[u + 1] - tb);
        if (hc > 0 && hadj[hb + hc - 1] == 0xFFFFu) --hc;
        const int d = hc + tc;
        const uint16_t *hub_list = hadj + hb;
        const int32_t *tail_list = tadj + tb;
        // reverse rows of this pivot's hub members (nullptr: none; ~0u per member: streamed forward)
        const uint32_t *rel_u = rv.rel ? rv.rel + hb : nullptr;
        const uint32_t *relt_u = (rv.rel && rv.relt) ? rv.relt + tb : nullptr;
        const uint32_t *arow = rv.rel ? rv.arena + rv.aoff[u] : nullptr;
        // where member i's finished row lies in the pivot's arena span (~0u: nowhere, it is streamed forward)
        auto relat = [&](int i) -> uint32_t { return !rel_u ? kKcRelForward : i < hc ? rel_u[i] : relt_u ? relt_u[i - hc] : kKcRelForward; };
        // slab rows: stride WSr, Wr words written per row.  EXPORT: the slab is slot q of the pool in the layout k_kc4_mfma reads (stride kc4m_stride(d) <= W)
        int WSr = WS, Wr = W;
        if constexpr (kCanExport && GLOBAL_ROWS)
            if (exporting) {
                WSr = Wr = kc4m_stride(d);
                rows = ex.pool + size_t(q) * size_t(ex.slot_words);
            }
        __syncthreads();  // previous pivot's counting is done
        // (16-byte stores: the clears are a quarter of the LDS instructions of a pivot of a few dozen members)
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid; i < kBitmapWords / 4; i += nthreads) reinterpret_cast<uint4 *>(bm)[i] = zero4;
        if (flt_dirty)  // (uniform: the previous pivot of this workgroup had tail members, or this is its first)
            for (int i = tid; i < kKcFilterWords / 4; i += nthreads) reinterpret_cast<uint4 *>(fltw)[i] = zero4;
        flt_dirty = tc > 0;
        if (tid == 0) s_nfwd = 0;
        if (!GLOBAL_ROWS)  // the slab variant writes every row word at its flush.  (Rounded up to 16 bytes: at most into the first words of the bitmap behind the rows — zero as well.)
            for (int i = tid; i < (int(TRI ? kc_tri_off(d) : uint32_t(d * WS)) + 3) / 4; i += nthreads) reinterpret_cast<uint4 *>(rows)[i] = zero4;
        __syncthreads();
        // the bitmap, and for every word that holds a member the local index of its first one — the exclusive prefix popcount a hit adds its rank
        // inside the word to.  The hub list is ascending, so that index is the position of the first member whose id falls into the word: written by
        // that member (round 5; rounds 1-4 counted the bits of all 2 048 words and scanned them per pivot: a barrier and 24 LDS operations per thread
        // that the pivots of a few dozen members paid as dearly as the wide ones).  Words without a member keep stale values: no hit reads them.
        for (int i = tid; i < hc; i += nthreads) {
            const uint32_t w = hub_list[i];
            atomicOr(&bm[w >> 5], 1u << (w & 31u));
            if (i == 0 || (uint32_t(hub_list[i - 1]) >> 5) != (w >> 5)) pre[w >> 5] = (unsigned short)i;
        }
        for (int i = tid; i < tc; i += nthreads) {
            const uint32_t w = uint32_t(tail_list[i]);
            atomicOr(&fltw[(w >> 5) & uint32_t(kKcFilterWords - 1)], 1u << (w & 31u));
            if (GMSX_KC_FILTER_BITS > 1) {
                const uint32_t b2 = kc_filter_bit2(w, uint32_t(kKcFilterWords - 1));
                atomicOr(&fltw[b2 >> 5], 1u << (b2 & 31u));
            }
        }
        __syncthreads();
        // LDS matrices (round 6): FIRST the rows that need no stream — one THREAD per member copies its finished row from the reverse-row arena (a
        // contiguous ⌈i / 32⌉ words; neighbouring members' rows are neighbours in the arena), all members of the pivot at once: one dependent chain
        // rel -> row per pivot instead of one per trip of four members — and the members that ARE streamed are compacted into `fwd` (ballot + one
        // LDS atomic per wave), so that the group loops below walk forward members only: a trip no longer idles three groups behind one stream, and a
        // pivot whose hub edges were all handed over has no trips at all.  (The slab variant keeps deciding per trip: its rows pass through the stage.)
        // EXPORT (the matrix leaves for the pool): rows that lie in the arena never pass through LDS or the stage — they are copied arena -> pool in bulk below
        // (slab variant: here; LDS variant: by the copy-out) — and BOTH variants walk the forward list only.  (With every hub edge handed to its receiver the
        // slab BUILD spent 40 ms at scale 26 copying finished rows four per wave and trip through its stage.)
        constexpr bool kUseFwd = (!GLOBAL_ROWS && PIPE != 2) || EXPORT;
        unsigned short *fwd = GLOBAL_ROWS ? reinterpret_cast<unsigned short *>(reinterpret_cast<uint32_t *>(pre + kBitmapWords) + size_t(nwaves) * 4 * W)  // (slab: behind the stage)
                                          : reinterpret_cast<unsigned short *>(smem + mat_words + (VTX ? dmax : 0));
        int nfwd = d;
        if constexpr (kUseFwd) {
            for (int i0 = 0; i0 < d; i0 += nthreads) {
                const int i = i0 + tid;
                bool fw = false;
                if (i < d && i > 0) {  // (row 0 is empty: nobody is below the first member)
                    const uint32_t r = relat(i);
                    if (r != kKcRelForward) {
                        if constexpr (!EXPORT) {
                            const uint32_t *src = arow + r;
                            uint32_t *dst = rowp(i);
                            for (int t = 0; t < ((i + 31) >> 5); ++t) dst[t] = src[t];
                        }
                    } else {
                        fw = true;
#ifdef GMSX_KC_NO_TAIL_MEMBERS  // A/B build (wrong counts): what the forward rows of the TAIL members cost (they cannot be receivers: no bitset container)
                        fw = i < hc;
#endif
                    }
                }
                const unsigned long long m = __ballot(fw);
                int base = 0;
                if (lane == 0 && m) base = atomicAdd(&s_nfwd, __popcll(m));
                base = __shfl(base, 0);
                if (fw) fwd[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            }
            if constexpr (EXPORT && GLOBAL_ROWS) {
                // slab: the arena rows (and the empty row 0) go straight to the pool slot, 16 bytes per thread and step; the forward rows are written by the flushes
                const int q4 = WSr >> 2;
                uint4 *dst = reinterpret_cast<uint4 *>(rows);
                for (int x = tid; x < d * q4; x += nthreads) {
                    const int i = x / q4, c = (x - i * q4) << 2;
                    const uint32_t r = i > 0 ? relat(i) : kKcRelForward;
                    if (r == kKcRelForward && i > 0) continue;
                    const uint32_t *src = arow + r;
                    const int nw = i > 0 ? (i + 31) >> 5 : 0;
                    uint4 v;
                    v.x = c < nw ? src[c] : 0u;
                    v.y = c + 1 < nw ? src[c + 1] : 0u;
                    v.z = c + 2 < nw ? src[c + 2] : 0u;
                    v.w = c + 3 < nw ? src[c + 3] : 0u;
                    dst[x] = v;
                }
            }
            __syncthreads();
            nfwd = s_nfwd;
        }
        // four rows per wave and trip, one per 16-lane group; the global-slab variant builds them in an LDS stage and
        // writes finished rows out with coalesced stores (no global atomics)
        {
            const int grp = lane >> 4, sub = lane & 15;
            uint32_t *stage = GLOBAL_ROWS ? reinterpret_cast<uint32_t *>(pre + kBitmapWords) + size_t(wave) * 4 * W : nullptr;
            if constexpr (PIPE == 2) {
                // ---- mode 2: descriptors of up to 256 members at a time, then one step stream per group (KcStream above) ----
                KcDesc *desc = reinterpret_cast<KcDesc *>(smem + size_t(dmax) * WS + (VTX ? dmax : 0));
                for (int base = 0; base < d; base += 256) {
                    const int nb = min(256, d - base);
                    if (tid < nb) {
                        const int i = base + tid;
                        const bool is_hub = i < hc;
                        const int32_t v = is_hub ? int32_t(hub_list[i]) : tail_list[i - hc];
                        const uint32_t r = relat(i);
                        const bool skip = r != kKcRelForward || i == 0;  // the row comes from the arena (copied here, by this one thread: the step stream is not the default BUILD) or is empty
                        if (r != kKcRelForward)
                            for (int t = 0; t < ((i + 31) >> 5); ++t) rows[size_t(i) * WS + t] = arow[r + t];
                        const KcExt e = kc_load_ext(hoff, toff, bmoff, dense_limit, skip ? 0 : v);
                        const bool bs = kc_use_bitset(skip ? 0 : v, is_hub, dense_limit, e.hl);
                        KcDesc dd;
                        dd.ph = bs ? reinterpret_cast<unsigned long long>(bmpool + e.bo) : reinterpret_cast<unsigned long long>(hadj + e.hb);
                        dd.pt = reinterpret_cast<unsigned long long>(tadj + e.tb);
                        dd.hu = bs ? (uint32_t(bitset_words(v) / 4) | 0x80000000u) : (hc > 0 ? uint32_t((e.hl + 7) / 8) : 0u);
                        dd.tu = (!is_hub && tc > 0) ? uint32_t((e.tl + 3) / 4) : 0u;
                        if (skip) dd.hu = dd.tu = 0u;
                        dd.hl = uint32_t(e.hl);
                        dd.tl = uint32_t(e.tl);
                        desc[tid] = dd;
                    }
                    __syncthreads();
                    KcStream<GMSX_KC_STREAM_DEPTH> st;
                    st.start(desc, hadj, nb, tid, nthreads);
                    st.run([&](kc_u4u pu, int ju, int eu) {
                        const KcDesc dd = desc[eu];
                        uint32_t *orow = rows + size_t(base + eu) * WS;
                        const int hu = int(dd.hu & 0x7fffffffu);
                        if (ju < hu) {
                            if (dd.hu & 0x80000000u) kc_and4(make_uint4(pu.x, pu.y, pu.z, pu.w), *reinterpret_cast<const uint4 *>(bm + 4 * ju), 4 * ju, pre, orow);
                            else kc_probe8(bm, pre, orow, kc_cut8(pu, int(dd.hl) - 8 * ju));
                        } else {
                            const int t0 = 4 * (ju - hu), nt = int(dd.tl) - t0;  // 1 … 4 ids of this unit are the row's
                            const int32_t ids[4] = {int32_t(pu.x), int32_t(pu.y), int32_t(pu.z), int32_t(pu.w)};
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4)
                                if (q4 < nt) {
                                    const int t = kc_tail_find(flt, tail_list, tc, ids[q4]);
                                    if (t >= 0) atomicOr(&orow[(hc + t) >> 5], 1u << ((hc + t) & 31));
                                }
                        }
                    });
                    __syncthreads();  // the descriptors are rewritten by the next batch
                }
            } else if constexpr (PIPE == 1) {
                const int istep = nwaves * 4;
                // member i of the pivot: its rank id (an index past the row: vertex 0 — loaded, never used)
                auto member = [&](int i) -> int32_t { return i < hc ? int32_t(hub_list[i]) : tail_list[i < d ? i - hc : 0]; };
                // SLAB matrices walk ALL members, four consecutive ones per trip (their rows leave through the stage together), and decide per member where the
                // row comes from — a member whose row lies in the arena (or member 0: no row) walks the pipeline as rank id 0, the top hub, d+ = 0: its extents
                // are two cached loads and its "row" is empty, so every stage is a no-op for it.  LDS matrices walk the forward list only (pass 1 above).
                const int K = kUseFwd ? nfwd : d;  // positions to walk
                auto at = [&](int k) -> int { return !kUseFwd ? min(k, d - 1) : (nfwd > 0 ? int(fwd[min(k, nfwd - 1)]) : 0); };  // position -> member index (clamped)
                auto relof = [&](int i) -> uint32_t { return !kUseFwd ? relat(i) : kKcRelForward; };
                auto piped = [&](int i, uint32_t r) -> int32_t { return (r != kKcRelForward || i == 0) ? 0 : member(i); };
                // the pipeline (kc_load_ext / kc_load_first above): ids three members ahead, extents two, first units one
                const int kg = wave * 4 + grp;
                int i_0 = at(kg), i_1 = at(kg + istep), i_2 = at(kg + 2 * istep);
                uint32_t r0 = relof(i_0), r1 = relof(i_1), r2 = relof(i_2);
                int32_t v0 = piped(i_0, r0), v1 = piped(i_1, r1), v2 = piped(i_2, r2);
                KcExt e0 = kc_load_ext(hoff, toff, bmoff, dense_limit, v0), e1 = kc_load_ext(hoff, toff, bmoff, dense_limit, v1);
                bool b0 = kc_use_bitset(v0, i_0 < hc, dense_limit, e0.hl);
                KcFirst f0 = kc_load_first(hadj, tadj, bmpool, e0, b0, b0 ? int(bitset_words(v0)) : 0, sub);
                for (int k0 = wave * 4; k0 < K; k0 += istep) {
                    const int kf = k0 + grp;
                    const int i = i_0;
                    // later stages first: they complete while member i is probed
                    const int i_3 = at(kf + 3 * istep);
                    const uint32_t r3 = relof(i_3);
                    const int32_t v3 = piped(i_3, r3);
                    const KcExt e2 = kc_load_ext(hoff, toff, bmoff, dense_limit, v2);
                    const bool b1 = kc_use_bitset(v1, i_1 < hc, dense_limit, e1.hl);
                    const KcFirst f1 = kc_load_first(hadj, tadj, bmpool, e1, b1, b1 ? int(bitset_words(v1)) : 0, sub);
                    if (GLOBAL_ROWS) {
                        for (int t = lane; t < 4 * W; t += 64) stage[t] = 0;
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (kf < K) {
                        const bool is_hub = i < hc;
                        uint32_t *orow = GLOBAL_ROWS ? stage + grp * W : rowp(i);
#ifdef GMSX_KC_NO_ROWS  // A/B build (wrong counts): the BUILD phase without its row streams
                        if (v0 == -7) orow[0] = 1;
#else
                        if (r0 != kKcRelForward) kc_copy_row(arow + r0, (i + 31) >> 5, orow, sub);
                        else if (i > 0) kc_build_member_first(hadj, tadj, bmpool, e0, f0, b0, b0 ? int(bitset_words(v0)) : 0, is_hub, hc, tail_list, tc, bm, pre, orow, sub, flt);
#endif
                    }
                    if (GLOBAL_ROWS) {
                        __builtin_amdgcn_wave_barrier();
                        const int nr = min(4, K - k0);
                        if constexpr (kUseFwd) {  // (EXPORT: the trip's rows are four members of the forward list)
                            for (int t = lane; t < nr * Wr; t += 64) rows[size_t(fwd[k0 + t / Wr]) * WSr + t % Wr] = stage[(t / Wr) * W + t % Wr];
                        } else {
                            for (int t = lane; t < nr * Wr; t += 64) rows[size_t(k0 + t / Wr) * WSr + t % Wr] = stage[(t / Wr) * W + t % Wr];
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    v0 = v1; v1 = v2; v2 = v3;
                    r0 = r1; r1 = r2; r2 = r3;
                    i_0 = i_1; i_1 = i_2; i_2 = i_3;
                    e0 = e1; e1 = e2;
                    b0 = b1;
                    f0 = f1;
                }
            } else {
                // small matrices, several workgroups per CU: the 30 registers of the pipeline would cost a third of the waves (80 -> 110
                // VGPRs: 6 -> 4 per SIMD; measured 22.6 -> 25.9 ms at scale 22 with the pipeline everywhere) — member by member
                for (int k0 = wave * 4; k0 < nfwd; k0 += nwaves * 4) {
                    const int kf = k0 + grp;
                    if (kf < nfwd) {
                        const int i = fwd[kf];
                        const bool is_hub = i < hc;
                        const int32_t v = is_hub ? int32_t(hub_list[i]) : tail_list[i - hc];
#ifdef GMSX_KC_NO_ROWS
                        if (v == -7) rows[size_t(i) * WS] = 1;
#else
                        kc_build_member(hoff, hadj, toff, tadj, bmoff, bmpool, dense_limit, v, is_hub, hc, tail_list, tc, bm, pre, rowp(i), sub, flt);
#endif
                    }
                }
            }
        }
        if constexpr (GLOBAL_ROWS && EXPORT) {
            // nobody reads the pool slot inside this kernel (the count is another launch): no fence — the agent-scope release / acquire pair below made every
            // workgroup wait for the slot's 0.4 MB of stores at every pivot
        } else if (GLOBAL_ROWS) {
            // the slab's readers are this workgroup's own waves: its stores only have to be done (the L1 writes through) — a workgroup-scope release; an agent-scope
            // one (__threadfence) also waits for the L2 to write its dirty lines back, every workgroup's, at every pivot
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop L1 lines of the slab cached for an earlier pivot
        } else {
            if (tid == 0) wave_tot[0] = 0;  // the ticket counter of kc4_count_pairs (the prefix phase is long done with wave_tot)
            __syncthreads();
        }
        if constexpr (kCanExport)
            if (exporting) {
                // the count of this matrix is k_kc4_mfma's: an LDS matrix leaves for its pool slot (triangular -> rectangular rows of kc4m_stride(d) words,
                // zero-filled, 16-byte stores); a slab matrix was built there
                if constexpr (!GLOBAL_ROWS) {
                    const int q4 = kc4m_stride(d) >> 2;  // 16-byte units per row
                    uint4 *dst = reinterpret_cast<uint4 *>(ex.pool + size_t(q) * size_t(ex.slot_words));
                    for (int x = tid; x < d * q4; x += nthreads) {
                        const int i = x / q4, c = (x - i * q4) << 2;
                        const uint32_t r = i > 0 ? relat(i) : kKcRelForward;  // (a row of the arena never went through LDS)
                        const uint32_t *src = r != kKcRelForward ? arow + r : rows + kc_tri_off(i);
                        const int nw = r != kKcRelForward ? (i + 31) >> 5 : (i >> 5) + 1;
                        uint4 v;
                        v.x = c < nw ? src[c] : 0u;
                        v.y = c + 1 < nw ? src[c + 1] : 0u;
                        v.z = c + 2 < nw ? src[c + 2] : 0u;
                        v.w = c + 3 < nw ? src[c + 3] : 0u;
                        dst[x] = v;
                    }
                }
                if (tid == 0) ex.dpool[q] = d;
                continue;
            }
#ifdef GMSX_KC_BUILD_ONLY  // A/B build: what the BUILD phase alone takes (k = 4, scale 22: 12.4 of 22.6 ms)
        if (true) {
        } else
#endif
        if constexpr (VTX) {
            // per-vertex triangle counts (vertex_count2 = 2 x triangles at the vertex): a set bit (i, j) is the triangle
            // (u, v_i, v_j).  Column sums with byte-sliced counters: lane = word (32 columns), eight registers of four 8-bit
            // counters each, the rows split over the waves, flushed to LDS counters every 255 rows; then member t adds
            // its row popcount and its column sum to counts[v_t] (one global atomic), the pivot gets the number of bits.
            uint32_t *col = GLOBAL_ROWS ? reinterpret_cast<uint32_t *>(pre + kBitmapWords) + size_t(nwaves) * 4 * W : smem + mat_words;
            __shared__ uint32_t piv_sum;
            for (int t = tid; t < d; t += nthreads) col[t] = 0;
            if (tid == 0) piv_sum = 0;
            __syncthreads();
            const int Wd = (d + 31) >> 5;
            for (int w0 = 0; w0 < Wd; w0 += 64) {
                const int w = w0 + lane;
                uint32_t r[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                int since = 0;
                for (int i = wave; i < d; i += nwaves) {
                    const uint32_t x = w < Wd ? rows[size_t(i) * WS + w] : 0u;
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] += (x >> q) & 0x01010101u;
                    if (++since == 255 || i + nwaves >= d) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const uint32_t c = (r[q] >> (8 * b)) & 0xffu;
                                if (c) atomicAdd(&col[(w << 5) + q + 8 * b], c);
                            }
                            r[q] = 0;
                        }
                        since = 0;
                    }
                }
            }
            __syncthreads();
            for (int t = tid; t < d; t += nthreads) {
                uint32_t rp = 0;
                const int nw = (t + 31) >> 5;  // row t has no bits at or above t
                for (int w = 0; w < nw; ++w) rp += uint32_t(__popc(rows[size_t(t) * WS + w]));
                const int32_t v = t < hc ? int32_t(hub_list[t]) : tail_list[t - hc];
                const unsigned long long tot = (unsigned long long)rp + col[t];
                if (tot) atomicAdd(&vcounts[oldid[v]], 2ull * tot);
                if (rp) atomicAdd(&piv_sum, rp);
                cnt += rp;
            }
            __syncthreads();
            if (tid == 0 && piv_sum) atomicAdd(&vcounts[oldid[u]], 2ull * piv_sum);
        } else if constexpr (LV == 2 && GLOBAL_ROWS) {
            // k = 4 on a slab matrix, by ROW BANDS: the rows j0 .. j0+JB-1 are staged in LDS (over the dead bitmap / stage
            // area), then every later row i (one per wave, its words in registers straight from the slab) meets its
            // neighbours inside the band.  JB = WT here (host-chosen so that JB x (W+1) words fit the LDS).
            uint32_t *band = smem;
            const int JB = WT, BS = W + 1;
            unsigned short *wlist = reinterpret_cast<unsigned short *>(band + size_t(JB) * BS) + size_t(wave) * kKcRowList;  // (behind the band: the host adds nwaves x kKcRowList x 2 bytes)
            for (int j0 = 0; j0 + 1 < d; j0 += JB) {
                const int j1 = min(j0 + JB, d);
                __syncthreads();
                const int bw = (j1 + 31) >> 5;  // rows below j1 have no bits at or above j1
                for (int x = tid; x < (j1 - j0) * bw; x += nthreads) {
                    const int r = x / bw, cw = x - r * bw;
                    band[r * BS + cw] = rows[size_t(j0 + r) * WS + cw];
                }
                __syncthreads();
                const int jw0 = j0 >> 5, jw1 = (j1 + 31) >> 5;  // the band's columns
                // the rows come from the slab in global memory, one per wave and trip — with the NEXT row of the wave on its way while this one meets
                // the band (round 5: a short probe load, a ballot, then the row's load, then the count — two dependent round trips per row with four
                // waves per SIMD — were 0.16 of the 0.39 s the slab bins took at scale 26)
                constexpr int R = 4;  // rows of the wave in flight (one ahead: the trip took as long as the load, 1.2 us; the count of a row is shorter)
                uint32_t ring[R][WPL];
                const int i_first = j0 + 1 + wave;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int h = 0; h < WPL; ++h) ring[r][h] = 64 * h + lane < W ? rows[size_t(min(i_first + r * nwaves, d - 1)) * WS + 64 * h + lane] : 0u;
                for (int i = i_first; i < d; i += R * nwaves) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int ii = i + r * nwaves;
                        uint32_t wt[WPL];
#pragma unroll
                        for (int h = 0; h < WPL; ++h) {
                            wt[h] = ring[r][h];
                            ring[r][h] = 64 * h + lane < W ? rows[size_t(min(ii + R * nwaves, d - 1)) * WS + 64 * h + lane] : 0u;
                        }
                        if (ii >= d) continue;  // (uniform)
                        bool any = false;  // a row without a neighbour inside the band is skipped (sparse matrices of large d)
#pragma unroll
                        for (int h = 0; h < WPL; ++h) any |= 64 * h + lane >= jw0 && 64 * h + lane < jw1 && wt[h] != 0u;
                        if (__ballot(any) == 0) continue;
#ifdef GMSX_KC_NO_SLAB_CALL  // A/B build (wrong counts): the slab count's bands and row loads without the per-row work
                        cnt += any;
                        continue;
#endif
#ifdef GMSX_KC_SLAB_ROW_SCAN  // A/B build: rounds 1-4's neighbour search (shuffle scan + binary search + select)
                        cnt += kc4_row<WPL>(wt, band, BS, j0, jw0, jw1, lane);
#else
                        cnt += kc4_row_list<WPL>(wt, band, BS, j0, jw0, jw1, lane, wlist, kKcRowList);
#endif
                    }
                }
            }
        } else if constexpr (LV == 1) {
            // k = 3: Σ_i popc(rows[i]) — one lane per matrix word
            for (int cell = tid; cell < d * W; cell += nthreads) cnt += (unsigned long long)__popc(rows[size_t(cell / W) * WS + cell % W]);
        } else if constexpr (LV >= 3 && WPL > 2) {
            // not instantiated by the host (k >= 5 stops at d+ = 4096)
        } else if (LV >= 3 && W > 8) {
            // deep recursion on wide rows: wave-cooperative (one word per lane, the candidate set of every level is kept),
            // which reads each row word once per visited node instead of re-ANDing the whole path
            for (int i = wave; i < d; i += nwaves) {
                uint32_t cand[WPL];
                cand[0] = lane < W ? rows[size_t(i) * WS + lane] : 0u;
                if constexpr (WPL > 1) cand[1] = 64 + lane < W ? rows[size_t(i) * WS + 64 + lane] : 0u;
                cnt += wave_cliques<LV, WPL>(rows, WS, cand, lane);
            }
        } else if (LV == 2 && kc4_pairs_enabled) {
            // k = 4 on an LDS matrix: pair lists in the dead bitmap / prefix area (12 KB), a share per wave (kc4_count_pairs)
            uint32_t *wbuf = bm + size_t(wave) * size_t((kBitmapWords + kBitmapWords / 2) / nwaves);
            cnt += kc4_count_pairs<TRI>(rows, WS, d, W, wbuf, (kBitmapWords + kBitmapWords / 2) / nwaves, &wave_tot[0], lane);
        } else if (LV == 2 && W >= 22) {
            // k = 4, wide matrix in LDS: a single band (the whole matrix), one row per wave and trip.  (Below ~700 vertices
            // the per-row prefix / select overhead of kc4_row exceeds what its divergence-free inner loop saves.)
            for (int i = 1 + wave; i < d; i += nwaves) {
                uint32_t wt[1];
                wt[0] = lane < W ? rows[size_t(i) * WS + lane] : 0u;
                cnt += kc4_row<1>(wt, rows, WS, 0, 0, W, lane);
            }
        } else {
            // k = 4 on narrower matrices, and narrow rows at k >= 5: one LANE per matrix word (i, w): it walks the set bits j of its word and enumerates the cliques below
            // the path (i, j) by ANDing the path's rows word by word — the reference's isect.intersect(N(vi)) recursion
            // (k = 4: Σ_i Σ_{j ∈ rows[i]} popc(rows[i] & rows[j])).  Odd row stride: no LDS bank conflicts.
            const int words = d * W;
            for (int cell = tid; cell < words; cell += nthreads) {
                const int i = cell / W, w = cell - i * W;
                uint32_t bits = rows[size_t(i) * WS + w];
                KcPath path;
                path.row[0] = rows + size_t(i) * WS;
                while (bits) {
                    const int j = (w << 5) + __ffs(bits) - 1;
                    bits &= bits - 1;
                    path.row[1] = rows + size_t(j) * WS;
                    cnt += lane_enum<LV - 1, 2>(rows, WS, w + 1, path);  // row j has no bits at or above j: words 0..w only
                }
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < nwaves; ++w) t += red[w];
        if (t) atomicAdd(&acc[(blockIdx.x & (kAccSlots - 1)) * kAccStride], t);
    }
}

// ---------------------------------------------------------------------------------------------
// Generic path — no limit on d+ or k (k <= kMaxGenericK): the reference's recursion (k_clique_count_set_based.h:5-17) with
// sets as sorted id lists in a per-wave global slab instead of bit rows.  One wave per (pivot u, first member v_i) task;
// below that an iterative depth-first search: level sets  S' = { w in S[0..t) : w in N+(S[t]) }  are filtered 64 candidates
// at a time (ballot compaction), membership = one bit of a bitset container (rank id < bitset_limit) or a binary search
// in the member's sorted 16-bit / 32-bit list.  Slower than the bit-matrix kernels by an order of magnitude; it serves the
// pivots they cannot hold (d+ > 8192 for k <= 4, > 4096 beyond) and k > kMaxK, so that no request fails for its size.
// ---------------------------------------------------------------------------------------------
static constexpr int kMaxGenericK = 64;

__device__ __forceinline__ bool kc_has_edge(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff,
                                            const int32_t *__restrict__ tadj, const int64_t *__restrict__ bmoff, const uint32_t *__restrict__ bmpool,
                                            int32_t bitset_limit, int32_t hub_limit, int32_t v, int32_t w) {  // w < v: is w in N+(v)?
    if (v < bitset_limit) return (bmpool[bmoff[v] + (uint32_t(w) >> 5)] >> (uint32_t(w) & 31u)) & 1u;
    if (w < hub_limit) {
        int64_t lo = hoff[v], hi = hoff[v + 1];
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (int32_t(hadj[mid]) < w) lo = mid + 1; else hi = mid;  // the 0xFFFF pad sorts last
        }
        return lo < hoff[v + 1] && int32_t(hadj[lo]) == w;
    }
    int64_t lo = toff[v], hi = toff[v + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (tadj[mid] < w) lo = mid + 1; else hi = mid;
    }
    return lo < toff[v + 1] && tadj[lo] == w;
}

__global__ __launch_bounds__(256) void k_kc_generic(const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                    const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                    const int64_t *__restrict__ bmoff, const uint32_t *__restrict__ bmpool, int32_t bitset_limit,
                                                    int32_t hub_limit, const int32_t *__restrict__ dplus, const int32_t *__restrict__ order,
                                                    int64_t first, int64_t end, int nparts, int part, int k, int32_t *__restrict__ slab,
                                                    int64_t level_stride, unsigned long long *__restrict__ acc,
                                                    const int32_t *__restrict__ oldid, unsigned long long *__restrict__ vcounts) {
    __shared__ int s_idx[4][kMaxGenericK], s_len[4][kMaxGenericK];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pos = first + int64_t(blockIdx.y) * nparts + part;
    unsigned long long cnt = 0;
    if (pos < end) {
        const int32_t u = order[pos];
        const int d = dplus[u];
        const int64_t hb = hoff[u], tb = toff[u];
        const int tl = int(toff[u + 1] - tb), hreal = d - tl;  // members: hreal hub ids (ascending), then tl tail ids (ascending)
        auto member = [&](int p) -> int32_t { return p < hreal ? int32_t(hadj[hb + p]) : tadj[tb + (p - hreal)]; };
        const int64_t wave_id = int64_t(blockIdx.x) * 4 + wave, nwaves = int64_t(gridDim.x) * 4;
        int32_t *base = slab + (int64_t(blockIdx.y) * nwaves + wave_id) * level_stride * int64_t(k > 3 ? k - 3 : 1);
        int *idx = s_idx[wave], *len = s_len[wave];
        for (int64_t i = wave_id + (k - 2); i < d; i += nwaves) {  // v_i needs k-2 members below it
            const int32_t v = member(int(i));
            // level 0: S = { w = member(p), p < i : w in N+(v) } — counted for k = 3, materialised otherwise
            int n0 = 0;
            for (int p0 = 0; p0 < int(i); p0 += 64) {
                const int p = p0 + lane;
                int32_t w = 0;
                bool hit = false;
                if (p < int(i)) {
                    w = member(p);
                    hit = kc_has_edge(hoff, hadj, toff, tadj, bmoff, bmpool, bitset_limit, hub_limit, v, w);
                }
                const unsigned long long m = __ballot(hit);
                if (k > 3 && hit) base[n0 + __popcll(m & ((1ull << lane) - 1ull))] = w;
                // per-vertex counts (gmsx_tc_vertex_count2 on pivots wider than the bit-matrix kernels hold; k = 3 only): the triangle {u, v, w}
                // adds 2 to each of its corners (counts[x] = Σ_{y∈N(x)} |N(x)∩N(y)| meets every triangle at x twice) — w here, v and u below
                if (vcounts && hit) atomicAdd(&vcounts[oldid[w]], 2ull);
                n0 += __popcll(m);
            }
            if (vcounts && lane == 0 && n0 > 0) {
                atomicAdd(&vcounts[oldid[v]], 2ull * (unsigned long long)n0);
                atomicAdd(&vcounts[oldid[u]], 2ull * (unsigned long long)n0);
            }
            if (k == 3) {
                cnt += (lane == 0) ? (unsigned long long)n0 : 0ull;
                continue;
            }
            if (n0 < k - 2) continue;  // the clique still needs k-2 vertices from this set
            // depth-first over the levels: level L holds a set of len[L] ids from which `need` = k-2-L pairwise adjacent vertices
            // are still to be chosen — R(need, S) of the reference recursion; R(1, S) = |S|
            __builtin_amdgcn_wave_barrier();
            int L = 0;
            if (lane == 0) { len[0] = n0; idx[0] = n0; }
            __builtin_amdgcn_wave_barrier();
            while (L >= 0) {
                const int need = k - 2 - L;  // vertices to pick from level L (>= 1)
                const int n = len[L];
                if (need == 1) {  // R(1, S) = |S|
                    cnt += (lane == 0) ? (unsigned long long)n : 0ull;
                    --L;
                    continue;
                }
                int t = idx[L] - 1;  // next member of this level to branch on (descending)
                if (t < need - 1) { --L; continue; }  // fewer than need-1 candidates below it
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) idx[L] = t;
                const int32_t *cur = base + int64_t(L) * level_stride;
                int32_t *nxt = base + int64_t(L + 1) * level_stride;
                const int32_t x = cur[t];
                int nn = 0;
                for (int p0 = 0; p0 < t; p0 += 64) {
                    const int p = p0 + lane;
                    int32_t w = 0;
                    bool hit = false;
                    if (p < t) {
                        w = cur[p];
                        hit = kc_has_edge(hoff, hadj, toff, tadj, bmoff, bmpool, bitset_limit, hub_limit, x, w);
                    }
                    const unsigned long long m = __ballot(hit);
                    if (need > 2 && hit) nxt[nn + __popcll(m & ((1ull << lane) - 1ull))] = w;
                    nn += __popcll(m);
                }
                if (need == 2) {  // the next level would only be counted
                    cnt += (lane == 0) ? (unsigned long long)nn : 0ull;
                } else if (nn >= need - 1) {
                    ++L;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { len[L] = nn; idx[L] = nn; }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_down(cnt, s);
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&acc[((blockIdx.x + blockIdx.y) & (kAccSlots - 1)) * kAccStride], t);
    }
}

static int64_t part_count(int64_t first, int64_t end, int nparts, int part) {
    const int64_t span = end - first - part;
    return span <= 0 ? 0 : (span + nparts - 1) / nparts;
}

// generic list recursion for the pivots at positions [first, end) of the d+ order — the slow path.  The pivots run in chunks (grid.y) sized
// from a SLAB BUDGET: a wave needs (k-3) level lists of round64(d+) ids, and `order` is d+-descending, so the chunk's own widest pivot is
// its first one — pivots per chunk = budget / (4 waves * levels * stride of that pivot), never more than 60000 (grid.y < 65536) and never
// fewer than one (one workgroup: k <= 64 levels of the widest row, megabytes).  One slab, grown only when a chunk needs more than the
// last, serves every chunk: no request fails for its size and there is no multi-GB malloc / free per chunk.
static int launch_generic(const gmsx_graph *g, int k, int64_t first, int64_t end, int part, int nparts, unsigned long long *acc, int *launches,
                          unsigned long long *vcounts = nullptr) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    const int cu = c.compute_units > 0 ? c.compute_units : 256;
    const int64_t levels = k > 3 ? k - 3 : 1;
    int64_t budget_ints = (int64_t(2) << 30) / 4;  // 2 GB of level lists in flight
    if (const char *e = opt("KC_SLAB_MB")) {  // test hook: a tiny budget forces many chunks on small graphs
        const long long v = std::atoll(e);
        if (v >= 1) budget_ints = int64_t(v) * (1 << 20) / 4;
    }
    int32_t *slab = nullptr;
    int64_t slab_ints = 0;
    struct Guard { int32_t *&p; ~Guard() { (void)hipFree(p); } } guard{slab};
    for (int64_t lo = first; lo < end;) {
        const int64_t left = part_count(lo, end, nparts, part);
        if (left <= 0) break;
        int32_t u0 = 0, d0 = 0;  // this shard's first pivot of the chunk is its widest
        GMSX_HIP(hipMemcpyAsync(&u0, g->order + lo + part, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        GMSX_HIP(hipMemcpyAsync(&d0, g->dplus + u0, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        const int64_t stride = (std::max<int64_t>(d0, 1) + 63) & ~int64_t(63);
        const int64_t per_block = 4 * levels * stride;  // ints one workgroup (4 waves) needs
        const int64_t pivots = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(60000, budget_ints / per_block), left));
        const int64_t hi = std::min<int64_t>(end, lo + pivots * nparts);
        // waves per pivot: enough to fill the chip a few times over, inside the budget
        int64_t blocks_x = std::max<int64_t>(1, std::min<int64_t>((int64_t(d0) + 3) / 4, std::max<int64_t>(1, int64_t(cu) * 8 / pivots)));
        while (blocks_x > 1 && pivots * blocks_x * per_block > budget_ints) blocks_x /= 2;
        const int64_t need = pivots * blocks_x * per_block;
        if (need > slab_ints) {
            GMSX_HIP(hipStreamSynchronize(s));  // earlier chunks still read the old slab
            (void)hipFree(slab);
            slab = nullptr;
            slab_ints = 0;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&slab), size_t(need) * 4));
            slab_ints = need;
        }
        hipLaunchKernelGGL(k_kc_generic, dim3(unsigned(blocks_x), unsigned(pivots)), dim3(256), 0, s, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff,
                           g->bmpool, g->bitset_limit, g->dense_limit, g->dplus, g->order, lo, hi, nparts, part, k, slab, stride, acc, g->oldid, vcounts);
        ++*launches;
        GMSX_HIP(hipGetLastError());
        lo = hi;
    }
    GMSX_HIP(hipStreamSynchronize(s));  // the slab is freed on return
    return GMSX_OK;
}

// =====================================================================================================================================================
// REVERSE ROWS (round 6).  rows[i] = N+(v_i) ∩ N+(u) can be had from either end of the oriented edge (u, v_i).  FORWARD (rounds 1-5, everything above):
// stream N+(v_i) — the HIGHER-degree endpoint's row — against u's bitmap.  REVERSE: nothing of N+(v_i) can hit but the members below v_i, and those are
// a PREFIX of u's own ascending hub list: i ids, 2 i bytes, probed against N+(v_i).  For a hub member w = v_i that set already exists as w's bitset
// container in bmpool.  So the edges whose reverse side is at least twice cheaper are handed to their receiver w (the triangle kernels' per-edge
// choice, §5.1 of DESIGN.md): k_kc_reverse stages w's bitset in LDS once per 512 of its in-neighbours, streams their prefixes through it with the
// stream position as the local index — no prefix popcount, no atomics — and writes every finished row to an arena; the pivots' BUILD copies those
// rows (⌈i / 32⌉ words) instead of streaming the member.  Which edges go which way is decided ONCE per graph (ensure_kc_reverse: like the triangle-count
// task lists an immutable container of the graph, built at the first k-clique call and timed in gmsx_stats.setup_ms); every call still performs every probe.
// v1 limits: hub receivers only (a tail receiver has no bitset container), pivots of 32 < d+ (k_kc_small keeps its inverted gathers).
// =====================================================================================================================================================
static constexpr int kRevMinEdges = 64;      // a receiver takes its edges only if it gets at least this many (its bitset is staged once per work item)
static constexpr int kRevItem = 512;         // records per work item
static constexpr int kRevMinD = 33;          // narrower pivots run on k_kc_small
static constexpr uint32_t kRelForward = 0xffffffffu;
__device__ __forceinline__ int rev_row_words(int i) { return (i + 31) >> 5; }
// bytes the forward BUILD streams for hub member w (kc_use_bitset's choice) — and what handing the edge over costs: the prefix, the record, the row written and read back
__device__ __forceinline__ bool rev_is_cheaper(const int64_t *__restrict__ hoff, int32_t dense_limit, int32_t w, int i, int factor10) {
    const int hl = int(hoff[w + 1] - hoff[w]);
    const int bw = int(bitset_words(w)) * 4;
    const int fwd = (w < dense_limit && bw + 32 < hl * 2) ? bw : hl * 2;
    const int rev = 2 * i + 16 + 8 * rev_row_words(i) + 32;
    return factor10 * rev < 10 * fwd;  // (factor10 = 20: handed over when at least twice cheaper; 1, the default: unless ten times dearer)
}
// the TAIL receivers' side of the three passes (tl.rcnt == nullptr: none): counters by rank id - H
struct KcrTail {
    const int64_t *toff;
    const int32_t *tadj;
    int32_t H;
    int min_edges;
    uint32_t *relt;         // [toff[n]]
    uint32_t *rcnt, *rcur;  // [n - H]
    const int64_t *roff;    // [n - H + 1] (pass 3)
    ulonglong2 *rec;        // two halves per record (pass 3)
    int64_t n_all;          // positions of the d+ order the passes walk: [0, n_piv) all members, [n_piv, n_all) tail members only
};
static constexpr int kRevTailMaxI = 2048;  // a tail record's row is assembled in 64 words of LDS per group (wider pivots have next to no tail members)
// pass 1 (MODE 0): mark the candidate edges (kc_rel = 0 / ~0) and count them per receiver.  pass 2 (MODE 1): the arena words of every pivot (the edges whose
// receiver takes them).  pass 3 (MODE 2): relative offsets into kc_rel, records to the receivers.  One 16-lane group per pivot position of the d+ order.
template <int MODE>
__global__ __launch_bounds__(256) void k_kcr_edges(int64_t n_piv, const int32_t *__restrict__ order, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                   int32_t dense_limit, int min_edges, int factor10, uint32_t *__restrict__ rel, uint32_t *__restrict__ rcnt,
                                                   int64_t *__restrict__ words /* by rank id */, const int64_t *__restrict__ aoff, const int64_t *__restrict__ roff,
                                                   uint32_t *__restrict__ rcur, ulonglong2 *__restrict__ rec, KcrTail tl) {
    const int sub = threadIdx.x & 15;
    const int64_t g0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 4, ng = (int64_t(gridDim.x) * blockDim.x) >> 4;
    for (int64_t pos = g0; pos < tl.n_all; pos += ng) {
        const int32_t u = order[pos];
        const int64_t hb = hoff[u];
        int hc = int(hoff[u + 1] - hb);
        if (hc > 0 && hadj[hb + hc - 1] == 0xFFFFu) --hc;
        int64_t run = 0;  // arena words of the accepted members in front (uniform per group)
        // (positions [n_piv, n_all): the pivots of d+ <= 32 — the wave kernel builds their hub members' rows by gathers, only their tail members are handed over)
        for (int i0 = 0; i0 < (pos < n_piv ? hc : 0); i0 += 16) {
            const int i = i0 + sub;
            bool take = false;
            int32_t w = 0;
            if (i < hc) {
                w = int32_t(hadj[hb + i]);
                if (MODE == 0) {
                    take = i >= 1 && rev_is_cheaper(hoff, dense_limit, w, i, factor10);
                    rel[hb + i] = take ? 0u : kRelForward;
                    if (take) atomicAdd(&rcnt[w], 1u);
                } else {
                    take = rel[hb + i] != kRelForward && rcnt[w] >= uint32_t(min_edges);
                }
            }
            if (MODE >= 1) {
                int wi = take ? rev_row_words(i) : 0, pre = wi;  // inclusive prefix over the 16 lanes of the group
#pragma unroll
                for (int sft = 1; sft < 16; sft <<= 1) {
                    const int t = __shfl_up(pre, sft, 16);
                    if (sub >= sft) pre += t;
                }
                const int tot = __shfl(pre, 15, 16);
                if (MODE == 2 && i < hc) {
                    const int64_t r = run + pre - wi;
                    rel[hb + i] = take ? uint32_t(r) : kRelForward;
                    if (take) {
                        const uint32_t slot = atomicAdd(&rcur[w], 1u);
                        ulonglong2 e;
                        e.x = (unsigned long long)hb | ((unsigned long long)i << 40);
                        e.y = (unsigned long long)(aoff[u] + r) | ((unsigned long long)pos << 36);
                        rec[roff[w] + slot] = e;
                    }
                }
                run += tot;
            }
        }
        // … and the tail members: member i = hc + k is handed to its receiver whenever that one qualifies (its row would be streamed forward through the
        // pivot's bitmap AND tail set; the receiver streams hc + k ids).  Same arena span, behind the hub members' rows.
        if (tl.rcnt) {
            const int64_t tb = tl.toff[u];
            const int tc = int(tl.toff[u + 1] - tb);
            for (int k0 = 0; k0 < tc; k0 += 16) {
                const int k = k0 + sub, i = hc + k;
                bool take = false;
                int32_t w = 0;
                if (k < tc) {
                    w = tl.tadj[tb + k];
                    if (MODE == 0) {
                        take = i >= 1 && i < kRevTailMaxI;
                        tl.relt[tb + k] = take ? 0u : kRelForward;
                        if (take) atomicAdd(&tl.rcnt[w - tl.H], 1u);
                    } else {
                        take = tl.relt[tb + k] != kRelForward && tl.rcnt[w - tl.H] >= uint32_t(tl.min_edges);
                    }
                }
                if (MODE >= 1) {
                    int wi = take ? rev_row_words(i) : 0, pre = wi;
#pragma unroll
                    for (int sft = 1; sft < 16; sft <<= 1) {
                        const int t = __shfl_up(pre, sft, 16);
                        if (sub >= sft) pre += t;
                    }
                    const int tot = __shfl(pre, 15, 16);
                    if (MODE == 2 && k < tc) {
                        const int64_t r = run + pre - wi;
                        tl.relt[tb + k] = take ? uint32_t(r) : kRelForward;
                        if (take) {
                            const uint32_t slot = atomicAdd(&tl.rcur[w - tl.H], 1u);
                            ulonglong2 e0, e1;
                            e0.x = (unsigned long long)hb | ((unsigned long long)i << 40);
                            e0.y = (unsigned long long)(aoff[u] + r) | ((unsigned long long)pos << 36);
                            e1.x = (unsigned long long)tb | ((unsigned long long)hc << 40);
                            e1.y = 0ull;
                            const int64_t at = 2 * (tl.roff[w - tl.H] + slot);
                            tl.rec[at] = e0;
                            tl.rec[at + 1] = e1;
                        }
                    }
                    run += tot;
                }
            }
        }
        if (MODE == 1 && sub == 0) words[u] = run;
    }
}
// tail receivers: accepted records and work items per receiver (device-side: there can be millions of them)
__global__ void k_kcr_tail_sizes(int64_t nt, const uint32_t *__restrict__ rcnt, int min_edges, int64_t *__restrict__ recs, int64_t *__restrict__ items) {
    const int64_t x = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (x > nt) return;
    const int64_t a = (x < nt && rcnt[x] >= uint32_t(min_edges)) ? int64_t(rcnt[x]) : 0;
    recs[x] = a;
    items[x] = (a + kRevItem - 1) / kRevItem;
}
__global__ void k_kcr_items(int64_t H, const int64_t *__restrict__ roff, const int64_t *__restrict__ ioff, uint4 *__restrict__ items, int32_t base) {
    const int64_t w = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // receiver base + w
    if (w >= H) return;
    const int64_t first = roff[w], cnt = roff[w + 1] - first, i0 = ioff[w];
    for (int64_t t = 0; t * kRevItem < cnt; ++t) {
        const int64_t f = first + t * kRevItem;
        items[i0 + t] = make_uint4(uint32_t(base + w), uint32_t(min<int64_t>(kRevItem, cnt - t * kRevItem)), uint32_t(f), uint32_t(f >> 32));
    }
}
// Which shard a pivot belongs to: the pivots' kernels stride over the positions of THEIR BIN — position lo + q * nparts + part of the d+ order, lo = where the
// bin starts (launch_all) — so the rule needs the bins' starts: lo[] ascending, 0 for "not yet reached".
struct KcBins {
    static constexpr int kN = 16;
    int64_t lo[kN];  // unused slots: 0
    __device__ __forceinline__ int part_of(int64_t pos, int nparts) const {
        int64_t l = 0;
#pragma unroll
        for (int b = 0; b < kN; ++b)
            if (lo[b] <= pos) l = max(l, lo[b]);
        return int((pos - l) % nparts);
    }
};
// the receivers' kernel: persistent workgroups take work items (<= 512 records of one receiver w) from a queue; w's bitset container over [0, w) goes to LDS
// with 16-byte copies; a 16-lane group per record streams the pivot's prefix (16-byte loads, 8 ids per lane and step), tests every id's bit, and each lane
// stores its 8 hit bits as ONE BYTE of the row — byte b of a row = local indices 8 b … 8 b + 7, i.e. exactly the little-endian words the BUILD copies.
// SHARDED: the call is one rank's part of a multi-GPU count — records of other ranks' pivots are skipped.  A template parameter, not a run-time test: the
// compiler evaluated KcBins::part_of (sixteen 64-bit compares and selects) for EVERY record ahead of `nparts <= 1 ||` — half the vector instructions of the
// single-GPU kernel.  (Its run time did not move: 72.7 against 75 ms at scale 26 — the pass waits on memory, 77 % of its wave cycles, and what it pulls
// through the fabric, ≈ 200 M KB of FETCH_SIZE per call, is the bandwidth of the box: every added load made it slower, every removed instruction left it as it was.)
template <int GW /* lanes per record: 16, or 8 = twice the records in flight per wave (most prefixes fit one step of 64 ids) */, bool SHARDED>
__global__ __launch_bounds__(256) void k_kc_reverse(const uint4 *__restrict__ items, int64_t n_items, const ulonglong2 *__restrict__ rec, const int64_t *__restrict__ bmoff,
                                                    const uint32_t *__restrict__ bmpool, const uint16_t *__restrict__ hadj, uint32_t *__restrict__ arena, int nparts,
                                                    int part, KcBins bins, unsigned int *__restrict__ queue) {
    __shared__ __attribute__((aligned(16))) uint32_t bm[kBitmapWords];
    __shared__ unsigned int s_item;
    const int tid = threadIdx.x, grp = tid / GW, sub = tid % GW;
    constexpr int NG = 256 / GW;  // records per round
    while (true) {
        __syncthreads();  // the previous item's probes are done with bm
        if (tid == 0) s_item = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t it = s_item;
        if (it >= n_items) break;
        const uint4 item = items[it];
        const int32_t w = int32_t(item.x);
        const int cnt = int(item.y);
        const int64_t first = int64_t(item.z) | (int64_t(item.w) << 32);
        const int nw4 = int(bitset_words(w)) >> 2;  // 16-byte units of the container (a multiple of 4 words)
        const uint4 *src = reinterpret_cast<const uint4 *>(bmpool + bmoff[w]);
        for (int t = tid; t < nw4; t += 256) reinterpret_cast<uint4 *>(bm)[t] = src[t];
        __syncthreads();
        // three records in flight per group: the record of entry e + 32, the first unit of the row of entry e + 16, the probes of entry e — every load
        // unconditional (an entry past the item re-reads its last record: in bounds, skipped below), so the waits are counted and a group does not sit out the
        // record -> row chain of every entry (first version, one entry at a time: 32.8 ms at scale 26 for 288 M records)
        const int e_last = cnt - 1;
        auto unit_of = [&](const ulonglong2 &r) -> kc_u4u {
            const int i = int(r.x >> 40);
            return *reinterpret_cast<const kc_u4u *>(hadj + (r.x & ((1ull << 40) - 1ull)) + min(sub * 8, (i - 1) & ~7));
        };
        ulonglong2 r0 = rec[first + min(grp, e_last)], r1 = rec[first + min(grp + NG, e_last)];
        kc_u4u u0 = unit_of(r0);
        for (int e = grp; e < cnt; e += NG) {
            const ulonglong2 r2 = rec[first + min(e + 2 * NG, e_last)];
            const kc_u4u u1 = unit_of(r1);
            const ulonglong2 r = r0;
            bool mine = true;
            if constexpr (SHARDED) mine = bins.part_of(int64_t(r.y >> 36), nparts) == part;  // (another rank's pivot: skipped)
            if (mine) {
                const int i = int(r.x >> 40);
                const uint16_t *row = hadj + (r.x & ((1ull << 40) - 1ull));
                unsigned char *out = reinterpret_cast<unsigned char *>(arena + (r.y & ((1ull << 36) - 1ull)));
                const int nbytes = rev_row_words(i) * 4, last = (i - 1) & ~7;  // i >= 1
                // (round 6b, measured and dropped: two more units of a long prefix in flight behind the one being probed — 81.6 / 88.2 against 72.8 ms at scale 26,
                //  always or only for prefixes beyond 128 ids: the kernel waits on its LDS probes, not on these loads)
                for (int b = sub; b < nbytes; b += GW) {
                    const int p0 = b * 8;
                    const kc_u4u p = b == sub ? u0 : *reinterpret_cast<const kc_u4u *>(row + min(p0, last));  // (clamped: in bounds, its bits masked below)
                    uint32_t m = kc_bit_lo(bm, p.x) | (kc_bit_hi(bm, p.x) << 1) | (kc_bit_lo(bm, p.y) << 2) | (kc_bit_hi(bm, p.y) << 3) | (kc_bit_lo(bm, p.z) << 4) |
                                 (kc_bit_hi(bm, p.z) << 5) | (kc_bit_lo(bm, p.w) << 6) | (kc_bit_hi(bm, p.w) << 7);
                    const int valid = i - p0;  // ids of this unit that belong to the prefix (<= 0: a byte of the row's last word behind the prefix)
                    m = valid >= 8 ? m : valid > 0 ? (m & ((1u << valid) - 1u)) : 0u;
                    out[b] = (unsigned char)m;
                }
            }
            r0 = r1;
            r1 = r2;
            u0 = u1;
        }
    }
}

// … the same for TAIL receivers (rank id >= 65 535: no bitset container).  Per work item the receiver's own two lists become its LDS images — the hub part a
// 65 536-bit bitmap (atomic ORs), the tail part a 16 384-bit hash filter in front of a sorted copy (binary search only behind a filter hit; a list too long for
// its 4 KB is searched where it lies) — and a 16-lane group per record assembles the row in its 256 bytes of LDS: the pivot's WHOLE hub list streamed through the
// bitmap (every hub member lies below a tail member), one byte per lane and step as above; then the pivot's tail members below the receiver, one id per lane
// and step, a set bit per find; then the row leaves for the arena.  Three records in flight per group, as above: the record of entry e + 32, the first hub
// unit and the first tail id of entry e + 16, the probes of entry e (first version, every load behind its record: 230 ms at scale 26 for 363 M records).
static constexpr int kRevTailList = 1024, kRevTailFilterWords = 512;
__device__ __forceinline__ uint32_t kcr_tail_hash(int32_t id) { return (uint32_t(id) * 0x9E3779B1u) >> 18; }  // 14 bits
template <int GW, bool SHARDED>
__global__ __launch_bounds__(256) void k_kc_reverse_tail(const uint4 *__restrict__ items, int64_t n_items, const ulonglong2 *__restrict__ rec, const int64_t *__restrict__ hoff,
                                                         const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                         uint32_t *__restrict__ arena, int nparts, int part, KcBins bins, unsigned int *__restrict__ queue) {
    __shared__ __attribute__((aligned(16))) uint32_t bm[kBitmapWords];
    __shared__ __attribute__((aligned(16))) uint32_t flt[kRevTailFilterWords];
    __shared__ int32_t tl[kRevTailList];
    constexpr int NG = 256 / GW;  // records per round
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[NG][kRevTailMaxI / 32];
    __shared__ unsigned int s_item;
    const int tid = threadIdx.x, grp = tid / GW, sub = tid % GW;
    // (Measured and dropped: the next item's header, extents and first 256 ids of each list loaded while this item's records are probed — 440 against 427 ms at
    //  scale 26.  What the per-item start costs is paid instead by the threshold: a receiver takes its edges from 256 records on (KC_REV_TAIL_MIN; 16: 434 ms,
    //  64: 426, 256: 423 at scale 26; 60.6 / 58.1 / 59.3 at scale 24).)
    while (true) {
        __syncthreads();  // the previous item's probes are done with bm / flt / tl
        if (tid == 0) s_item = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t it = s_item;
        if (it >= n_items) break;
        const uint4 item = items[it];
        const int32_t w = int32_t(item.x);
        const int cnt = int(item.y);
        const int64_t first = int64_t(item.z) | (int64_t(item.w) << 32);
        const int64_t whb = hoff[w], wtb = toff[w];
        const int whl = int(hoff[w + 1] - whb), wtl = int(toff[w + 1] - wtb);
        for (int t = tid; t < kBitmapWords / 4; t += 256) reinterpret_cast<uint4 *>(bm)[t] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < kRevTailFilterWords / 4) reinterpret_cast<uint4 *>(flt)[tid] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int t = tid; t < whl; t += 256) {
            const uint32_t id = hadj[whb + t];  // (a padding id 0xFFFF sets a bit no pivot's hub list asks for)
            atomicOr(&bm[id >> 5], 1u << (id & 31u));
        }
        const bool tl_lds = wtl <= kRevTailList;
        for (int t = tid; t < wtl; t += 256) {
            const int32_t id = tadj[wtb + t];
            if (tl_lds) tl[t] = id;
            const uint32_t hsh = kcr_tail_hash(id);
            atomicOr(&flt[hsh >> 5], 1u << (hsh & 31u));
        }
        __syncthreads();
        const int32_t *tlist = tadj + wtb;
        uint32_t *rb = rowbuf[grp];
        unsigned char *rbb = reinterpret_cast<unsigned char *>(rb);
        const int e_last = cnt - 1;
        // stage 2 of an entry: its first hub unit and its first tail id (clamped: in bounds, masked / skipped below)
        auto first_unit = [&](const ulonglong2 &a, const ulonglong2 &b) -> kc_u4u {
            const int hc = int(b.x >> 40);
            return *reinterpret_cast<const kc_u4u *>(hadj + (hc > 0 ? (a.x & ((1ull << 40) - 1ull)) : 0ull) + min(sub * 8, max(hc - 1, 0) & ~7));  // (no hub member: any unit)
        };
        auto first_tail = [&](const ulonglong2 &a, const ulonglong2 &b) -> int32_t {
            const int k = int(a.x >> 40) - int(b.x >> 40);
            return tadj[(b.x & ((1ull << 40) - 1ull)) + min(sub, max(k - 1, 0))];
        };
        int64_t e0i = first + min(grp, e_last), e1i = first + min(grp + NG, e_last);
        ulonglong2 a0 = rec[2 * e0i], b0 = rec[2 * e0i + 1], a1 = rec[2 * e1i], b1 = rec[2 * e1i + 1];
        kc_u4u u0 = first_unit(a0, b0);
        int32_t t0 = first_tail(a0, b0);
        for (int e = grp; e < cnt; e += NG) {
            const int64_t e2i = first + min(e + 2 * NG, e_last);
            const ulonglong2 a2 = rec[2 * e2i], b2 = rec[2 * e2i + 1];
            const kc_u4u u1 = first_unit(a1, b1);
            const int32_t t1 = first_tail(a1, b1);
            bool mine = true;
            if constexpr (SHARDED) mine = bins.part_of(int64_t(a0.y >> 36), nparts) == part;  // (another rank's pivot: skipped)
            if (mine) {
                const int i = int(a0.x >> 40), hc = int(b0.x >> 40), k = i - hc;  // member i = hc + k of its pivot
                const uint16_t *hrow = hadj + (a0.x & ((1ull << 40) - 1ull));
                const int32_t *trow = tadj + (b0.x & ((1ull << 40) - 1ull));
                const int nwords = rev_row_words(i);
                for (int t = sub; t < nwords; t += GW) rb[t] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // hub columns [0, hc): 8 ids per lane and step -> one byte of the row
                const int hbytes = (hc + 7) >> 3, last = (hc - 1) & ~7;
                for (int b = sub; b < hbytes; b += GW) {
                    const int p0 = b * 8;
                    const kc_u4u p = b == sub ? u0 : *reinterpret_cast<const kc_u4u *>(hrow + min(p0, last));
                    uint32_t m = kc_bit_lo(bm, p.x) | (kc_bit_hi(bm, p.x) << 1) | (kc_bit_lo(bm, p.y) << 2) | (kc_bit_hi(bm, p.y) << 3) | (kc_bit_lo(bm, p.z) << 4) |
                                 (kc_bit_hi(bm, p.z) << 5) | (kc_bit_lo(bm, p.w) << 6) | (kc_bit_hi(bm, p.w) << 7);
                    const int valid = hc - p0;  // (>= 1)
                    m = valid >= 8 ? m : (m & ((1u << valid) - 1u));
                    rbb[b] = (unsigned char)m;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // tail columns [hc, i): the pivot's tail members below the receiver: filter bit, then the receiver's sorted tail list
                for (int q = sub; q < k; q += GW) {
                    const int32_t id = q == sub ? t0 : trow[q];
                    const uint32_t hsh = kcr_tail_hash(id);
                    if ((flt[hsh >> 5] >> (hsh & 31u)) & 1u) {
                        int lo = 0, hi = wtl;
                        if (tl_lds) {
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (tl[mid] < id) lo = mid + 1; else hi = mid;
                            }
                            if (lo < wtl && tl[lo] == id) atomicOr(&rb[(hc + q) >> 5], 1u << ((hc + q) & 31));
                        } else {
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (tlist[mid] < id) lo = mid + 1; else hi = mid;
                            }
                            if (lo < wtl && tlist[lo] == id) atomicOr(&rb[(hc + q) >> 5], 1u << ((hc + q) & 31));
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t *out = arena + (a0.y & ((1ull << 36) - 1ull));
                for (int t = sub; t < nwords; t += GW) out[t] = rb[t];
            }
            a0 = a1; b0 = b1;
            a1 = a2; b1 = b2;
            u0 = u1;
            t0 = t1;
        }
    }
}

// builds the lists above for the graph (once); leaves kc_rel == nullptr when no receiver qualifies or the option KC_REVERSE = 0 says no
static int ensure_kc_reverse(const gmsx_graph *g, int max_d) {
    if (g->kc_rev_tried) return GMSX_OK;
    g->kc_rev_tried = true;
    if (const char *e = opt("KC_REVERSE"); e && std::atoi(e) == 0) return GMSX_OK;
    if (!g->rows_sorted || g->dense_limit <= 0 || g->n >= (int64_t(1) << 28)) return GMSX_OK;  // (28 bits of a record hold the pivot's position)
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    const auto t_begin = std::chrono::steady_clock::now();
    int64_t n_piv = 0, hub_total = 0;
    if (int rc = count_dplus_ge(g, kRevMinD, &n_piv)) return rc;
    int64_t n_over = 0;
    if (int rc = count_dplus_ge(g, max_d + 1, &n_over)) return rc;  // (pivots beyond the bit-matrix kernels take part too: harmless, their lists are never read)
    (void)n_over;
    if (n_piv <= 0) return GMSX_OK;
    GMSX_HIP(hipMemcpyAsync(&hub_total, g->hoff + g->n, 8, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    if (hub_total <= 0) return GMSX_OK;
    const int32_t H = g->dense_limit;
    struct Dev { void *p = nullptr; ~Dev() { (void)hipFree(p); } };
    Dev d_rel, d_rcnt, d_words, d_aoff, d_roff, d_ioff, d_rec, d_item;
    Dev d_relt, d_rcntt, d_sizes, d_rect, d_itemt;  // tail receivers (option KC_REV_TAIL = 0: none)
    auto fail = [&](int rc) { return rc; };
    if (hipMalloc(&d_rel.p, size_t(hub_total) * 4) != hipSuccess || hipMalloc(&d_rcnt.p, size_t(H + 1) * 4 * 2) != hipSuccess ||
        hipMalloc(&d_words.p, size_t(g->n + 1) * 8) != hipSuccess || hipMalloc(&d_aoff.p, size_t(g->n + 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        return fail(GMSX_OK);  // no room for the lists: the forward BUILD needs none
    }
    uint32_t *rel = static_cast<uint32_t *>(d_rel.p), *rcnt = static_cast<uint32_t *>(d_rcnt.p), *rcur = rcnt + (H + 1);
    int64_t *words = static_cast<int64_t *>(d_words.p), *aoff = static_cast<int64_t *>(d_aoff.p);
    GMSX_HIP(hipMemsetAsync(rel, 0xff, size_t(hub_total) * 4, s));
    GMSX_HIP(hipMemsetAsync(rcnt, 0, size_t(H + 1) * 8, s));
    GMSX_HIP(hipMemsetAsync(words, 0, size_t(g->n + 1) * 8, s));
    const int cu = c.compute_units > 0 ? c.compute_units : 256;
    const int min_edges = int(std::max<long long>(1, opt_int("KC_REV_MIN", kRevMinEdges)));  // (option: a lower threshold lets small test graphs hand edges over)
    // tail receivers: rank ids [H, n)
    const int64_t nt = g->n - int64_t(H);
    int64_t tail_total = 0;
    GMSX_HIP(hipMemcpyAsync(&tail_total, g->toff + g->n, 8, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    KcrTail tl{g->toff, g->tadj, H, int(std::max<long long>(1, opt_int("KC_REV_TAIL_MIN", 256))), nullptr, nullptr, nullptr, nullptr, nullptr, n_piv};
    const bool want_tail = nt > 0 && tail_total > 0 && !(opt("KC_REV_TAIL") && std::atoi(opt("KC_REV_TAIL")) == 0);
    if (want_tail) {
        if (hipMalloc(&d_relt.p, size_t(tail_total) * 4) == hipSuccess && hipMalloc(&d_rcntt.p, size_t(nt) * 4 * 2) == hipSuccess) {
            tl.relt = static_cast<uint32_t *>(d_relt.p);
            tl.rcnt = static_cast<uint32_t *>(d_rcntt.p);
            tl.rcur = tl.rcnt + nt;
            GMSX_HIP(hipMemsetAsync(tl.relt, 0xff, size_t(tail_total) * 4, s));
            GMSX_HIP(hipMemsetAsync(tl.rcnt, 0, size_t(nt) * 8, s));
            if (int rc = count_dplus_ge(g, 2, &tl.n_all)) return rc;  // … the tail members of the narrow pivots too (k_kc_small)
            tl.n_all = std::max(tl.n_all, n_piv);
        } else {
            (void)hipGetLastError();  // no room: the tail members stay forward
        }
    }
    // (option: 10 x how much cheaper in BYTES the reverse side must be.  Default 1 — handed over unless it moves ten times the bytes — since the matrix-core count:
    //  a forward row's hits are resolved bit by bit (prefix popcount + LDS atomic, divergent per lane), a reverse row costs one LDS probe per prefix id, and
    //  with the count off the clock the BUILD is what a call waits for.  20 -> 1 at scales 22 / 24 / 26: 13.6 -> 11.3, 70.0 -> 59.3, 496 -> 470 ms, for lists + arena
    //  of 31.3 instead of 13.4 GB at scale 26 and 69 instead of 34 ms of one-off list build)
    const int factor10 = int(std::max<long long>(0, opt_int("KC_REV_FACTOR", 0)));  // (0: every hub edge whose receiver qualifies)
    const unsigned blocks = unsigned(std::min<int64_t>((tl.n_all + 15) / 16, int64_t(cu) * 32));
    hipLaunchKernelGGL(k_kcr_edges<0>, dim3(blocks), dim3(256), 0, s, n_piv, g->order, g->hoff, g->hadj, g->dense_limit, min_edges, factor10, rel, rcnt, words, nullptr, nullptr, rcur, nullptr, tl);
    hipLaunchKernelGGL(k_kcr_edges<1>, dim3(blocks), dim3(256), 0, s, n_piv, g->order, g->hoff, g->hadj, g->dense_limit, min_edges, factor10, rel, rcnt, words, nullptr, nullptr, rcur, nullptr, tl);
    GMSX_HIP(hipGetLastError());
    if (int rc = exclusive_scan_i64(words, aoff, g->n + 1, s)) return rc;
    // the receivers' side on the host: at most 65 535 counters
    std::vector<uint32_t> h_cnt(size_t(H) + 1);
    int64_t arena_words = 0;
    GMSX_HIP(hipMemcpyAsync(h_cnt.data(), rcnt, size_t(H + 1) * 4, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipMemcpyAsync(&arena_words, aoff + g->n, 8, hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> h_roff(size_t(H) + 1), h_ioff(size_t(H) + 1);
    int64_t recs = 0, items = 0;
    for (int32_t w = 0; w < H; ++w) {
        h_roff[size_t(w)] = recs;
        h_ioff[size_t(w)] = items;
        const int64_t a = h_cnt[size_t(w)] >= uint32_t(min_edges) ? int64_t(h_cnt[size_t(w)]) : 0;
        recs += a;
        items += (a + kRevItem - 1) / kRevItem;
    }
    h_roff[size_t(H)] = recs;
    h_ioff[size_t(H)] = items;
    // the tail receivers' side on the device: sizes -> two scans (records, work items)
    int64_t recs_t = 0, items_t = 0;
    int64_t *rofft = nullptr, *iofft = nullptr;
    if (tl.rcnt) {
        if (hipMalloc(&d_sizes.p, size_t(nt + 1) * 8 * 4) != hipSuccess) {
            (void)hipGetLastError();
            return GMSX_OK;
        }
        int64_t *sz_r = static_cast<int64_t *>(d_sizes.p), *sz_i = sz_r + (nt + 1);
        rofft = sz_i + (nt + 1);
        iofft = rofft + (nt + 1);
        hipLaunchKernelGGL(k_kcr_tail_sizes, dim3(unsigned((nt + 1 + 255) / 256)), dim3(256), 0, s, nt, tl.rcnt, tl.min_edges, sz_r, sz_i);
        GMSX_HIP(hipGetLastError());
        if (int rc = exclusive_scan_i64(sz_r, rofft, nt + 1, s)) return rc;
        if (int rc = exclusive_scan_i64(sz_i, iofft, nt + 1, s)) return rc;
        GMSX_HIP(hipMemcpyAsync(&recs_t, rofft + nt, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipMemcpyAsync(&items_t, iofft + nt, 8, hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        tl.roff = rofft;
    }
    if (recs + recs_t == 0 || arena_words <= 0 || arena_words >= (int64_t(1) << 36)) return GMSX_OK;  // nothing worth handing over (small or flat graphs)
    if (hipMalloc(&d_roff.p, size_t(H + 1) * 8) != hipSuccess || hipMalloc(&d_ioff.p, size_t(H + 1) * 8) != hipSuccess ||
        hipMalloc(&d_rec.p, size_t(std::max<int64_t>(recs, 1)) * sizeof(ulonglong2)) != hipSuccess ||
        hipMalloc(&d_item.p, size_t(std::max<int64_t>(items, 1)) * sizeof(uint4)) != hipSuccess ||
        hipMalloc(&d_rect.p, size_t(std::max<int64_t>(recs_t, 1)) * 2 * sizeof(ulonglong2)) != hipSuccess ||
        hipMalloc(&d_itemt.p, size_t(std::max<int64_t>(items_t, 1)) * sizeof(uint4)) != hipSuccess) {
        (void)hipGetLastError();
        return GMSX_OK;
    }
    tl.rec = static_cast<ulonglong2 *>(d_rect.p);
    uint32_t *arena = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&arena), size_t(arena_words) * 4 + 64) != hipSuccess) {
        (void)hipGetLastError();
        return GMSX_OK;
    }
    Dev d_arena;
    d_arena.p = arena;
    int64_t *roff = static_cast<int64_t *>(d_roff.p), *ioff = static_cast<int64_t *>(d_ioff.p);
    GMSX_HIP(hipMemcpyAsync(roff, h_roff.data(), size_t(H + 1) * 8, hipMemcpyHostToDevice, s));
    GMSX_HIP(hipMemcpyAsync(ioff, h_ioff.data(), size_t(H + 1) * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_kcr_edges<2>, dim3(blocks), dim3(256), 0, s, n_piv, g->order, g->hoff, g->hadj, g->dense_limit, min_edges, factor10, rel, rcnt, words, aoff, roff, rcur,
                       static_cast<ulonglong2 *>(d_rec.p), tl);
    hipLaunchKernelGGL(k_kcr_items, dim3(unsigned((H + 255) / 256)), dim3(256), 0, s, int64_t(H), roff, ioff, static_cast<uint4 *>(d_item.p), int32_t(0));
    if (tl.rcnt && items_t > 0)
        hipLaunchKernelGGL(k_kcr_items, dim3(unsigned((nt + 255) / 256)), dim3(256), 0, s, nt, rofft, iofft, static_cast<uint4 *>(d_itemt.p), int32_t(H));
    GMSX_HIP(hipGetLastError());
    GMSX_HIP(hipStreamSynchronize(s));
    if (tl.rcnt && recs_t > 0) {
        g->kc_relt = tl.relt; d_relt.p = nullptr;
        g->kc_rect = tl.rec; d_rect.p = nullptr;
        g->kc_itemt = static_cast<uint4 *>(d_itemt.p); d_itemt.p = nullptr;
        g->kc_recst = recs_t;
        g->kc_itemst = items_t;
    }
    g->kc_rel = rel; d_rel.p = nullptr;
    g->kc_aoff = aoff; d_aoff.p = nullptr;
    g->kc_arena = arena; d_arena.p = nullptr;
    g->kc_rec = static_cast<ulonglong2 *>(d_rec.p); d_rec.p = nullptr;
    g->kc_item = static_cast<uint4 *>(d_item.p); d_item.p = nullptr;
    g->kc_recs = recs;
    g->kc_items = items;
    g->kc_arena_words = arena_words;
    g->kc_rev_bytes = hub_total * 4 + (g->n + 1) * 8 + arena_words * 4 + 64 + recs * int64_t(sizeof(ulonglong2)) + items * int64_t(sizeof(uint4));
    if (g->kc_relt) g->kc_rev_bytes += tail_total * 4 + recs_t * 2 * int64_t(sizeof(ulonglong2)) + items_t * int64_t(sizeof(uint4));
    const_cast<gmsx_graph *>(g)->device_bytes += g->kc_rev_bytes;
    g->kc_rev_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (opt_on("TIMING"))
        std::fprintf(stderr, "[gmsx kclique] reverse rows: %lld of the hub edges of %lld pivots handed to %lld work items of hub receivers, %lld tail edges to %lld work items of tail receivers, arena %.3f GB, lists %.3f GB, built in %.1f ms\n",
                     (long long)recs, (long long)n_piv, (long long)items, (long long)recs_t, (long long)items_t, double(arena_words) * 4e-9, double(g->kc_rev_bytes - arena_words * 4) * 1e-9, g->kc_rev_build_ms);
    return GMSX_OK;
}
// the receivers' pass of one call (inside the timed region, ahead of the pivots' kernels)
static int launch_kc_reverse(const gmsx_graph *g, int part, int nparts, const KcBins &bins, unsigned long long *acc, int *launches, hipStream_t s) {
    if (!g->kc_rel || (g->kc_items <= 0 && g->kc_itemst <= 0)) return GMSX_OK;
    Ctx &c = ctx();
    unsigned int *queue = reinterpret_cast<unsigned int *>(acc + (kAccSlots - 1) * kAccStride + 8);  // a spare word of the accumulator array (zeroed by the caller)
    const int cu = c.compute_units > 0 ? c.compute_units : 256;
    const unsigned blocks = unsigned(std::min<int64_t>(g->kc_items, int64_t(cu) * 8));
    if (g->kc_items > 0) {
        const bool gw8 = opt("KC_REV_GW") && std::atoi(opt("KC_REV_GW")) == 8;  // (hub receivers: 16 lanes per record — 8: 84 against 71 ms at scale 26; option for A/B)
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, g->kc_item, g->kc_items, g->kc_rec, g->bmoff, g->bmpool, g->hadj, g->kc_arena, nparts, part, bins, queue); };
        if (nparts > 1) { if (gw8) go(k_kc_reverse<8, true>); else go(k_kc_reverse<16, true>); }
        else { if (gw8) go(k_kc_reverse<8, false>); else go(k_kc_reverse<16, false>); }
        ++*launches;
    }
    if (g->kc_relt && g->kc_itemst > 0) {
        const unsigned blocks_t = unsigned(std::min<int64_t>(g->kc_itemst, int64_t(cu) * 8));
        const bool gw8 = opt("KC_REV_GW") && std::atoi(opt("KC_REV_GW")) == 8;  // (8 lanes per record: 94 against ~105 ms alone at scale 26, but 433 against 425 ms for the call — the passes overlap)
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(blocks_t), dim3(256), 0, s, g->kc_itemt, g->kc_itemst, g->kc_rect, g->hoff, g->hadj, g->toff, g->tadj, g->kc_arena, nparts, part, bins, queue + 1);
        };
        if (nparts > 1) { if (gw8) go(k_kc_reverse_tail<8, true>); else go(k_kc_reverse_tail<16, true>); }
        else { if (gw8) go(k_kc_reverse_tail<8, false>); else go(k_kc_reverse_tail<16, false>); }
        ++*launches;
    }
    return GMSX_OK;
}

// widths of the two triangular LDS bins (k = 4): the matrix of d+ <= 1472 and its bitmap / prefix / forward list fill one CU's LDS (153.6 of the 155 KB a
// launch may ask for); two matrices of d+ <= 960 fit it together (2 x 78.0 KB with the static 4.3 KB each)
// dynamic LDS the LDS-matrix variants may ask for: the CU's 160 KB minus their 16.2 KB of static LDS (bitmap + prefix counts + tail filter + reduction slots)
static constexpr int kKcLdsDynMax = 143 * 1024;
static constexpr int kKcTriTop = 1472, kKcTriTwo = 960;  // (multiples of 32: W = dmax / 32 words)
static bool kc_tri_enabled() {
    const char *e = opt("KC_TRI");  // option: 0 = rectangular LDS matrices up to 1024 and the slab beyond, as in rounds 1-5 (A/B)
    return !(e && std::atoi(e) == 0);
}
static bool stream_build_default(const gmsx_graph *g) {
    (void)g;
    return false;
}
static bool pipe_all_default(const gmsx_graph *g) {
    (void)g;
    return false;
}

// The pool the exporting BUILD kernels leave their matrices in (KcExport) — allocated at the first k = 4 call and kept (option KC_POOL_MB, default 6 144: a
// region per stream; a region is reused by the next chunk of the same stream, so stream order alone protects it), with the d's of a chunk's slots behind it.
struct KcPool {
    uint32_t *base = nullptr;
    size_t bytes = 0;
    int32_t *dpool = nullptr;
};
static constexpr int kKcPoolRegions = 3, kKcChunkMax = 65536;
// words between the slots of a chunk: 2 048 x 64 and 4 096 x 128 words are powers of two — the workgroups of a BUILD (and the teams of a count) would walk their
// slots in step on the same memory channels
static constexpr size_t kKcSlotPad = 2080;
static KcPool &kc_pool() {
    static KcPool p;
    return p;
}
// `needed`: what this call's matrices would take in one piece; the pool is min(needed, default) — or exactly KC_POOL_MB — and only ever grows
static int ensure_kc_pool(size_t needed) {
    KcPool &p = kc_pool();
    size_t want = std::min(size_t(6144) << 20, std::max(needed, size_t(96) << 20));
    bool exact = false;
    if (const char *e = opt("KC_POOL_MB")) {
        const long long v = std::atoll(e);
        if (v >= 0) {
            want = size_t(v) << 20;
            exact = true;
        }
    }
    if (p.base && (exact ? p.bytes == want : p.bytes >= want)) return GMSX_OK;
    if (p.base) {
        GMSX_HIP(hipDeviceSynchronize());
        (void)hipFree(p.base);
        p.base = nullptr;
        p.bytes = 0;
    }
    if (!p.dpool) {
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&p.dpool), sizeof(int32_t) * kKcPoolRegions * kKcChunkMax));
    }
    for (size_t b = want; b >= (size_t(96) << 20); b >>= 1) {  // (a smaller pool = more chunks; below 96 MB the bins keep their in-kernel counts)
        void *q = nullptr;
        if (hipMalloc(&q, b) == hipSuccess) {
            p.base = static_cast<uint32_t *>(q);
            p.bytes = b;
            break;
        }
        (void)hipGetLastError();
    }
    return GMSX_OK;
}
static bool kc_mfma_enabled() {
    const char *e = opt("KC_MFMA");  // option: 0 = the k = 4 count of every bin by AND + popcount inside the BUILD kernels, as before the matrix-core count (A/B)
    return !(e && std::atoi(e) == 0);
}

template <int LV, bool VTX = false>
static int launch_all(const gmsx_graph *g, int part, int nparts, unsigned long long *acc, int *launches, uint32_t **slab_out,
                      unsigned long long *vcounts = nullptr) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    const int k = LV + 2;
    int64_t over = 0, n_min = 0;
    constexpr int kMaxD = (LV <= 2) ? 8192 : 4096;  // widest bit rows: four words per lane for k = 3, 4 and the per-vertex counts, two for k >= 5
    int max_d = kMaxD;
    if (const char *e = opt("KC_MAXD")) {  // test hook: a lower limit sends more pivots through the generic path
        const int v = std::atoi(e);
        if (v >= 1 && v < kMaxD) max_d = v;
    }
    if (int rc = count_dplus_ge(g, max_d + 1, &over)) return rc;
    if (!g->rows_sorted) return GMSX_ERR_UNSUPPORTED;  // > 2^32 container entries: rows were not sorted at upload
    if (over > 0) {
        // pivots wider than the bit-matrix kernels hold: the generic list recursion takes positions [0, over) of the d+ order — with the per-vertex
        // counts too (round 6: gmsx_tc_vertex_count2 used to leave the whole graph to one full-row intersect per CSR entry as soon as ONE pivot was
        // wider than 8192; now only the wide pivots themselves take the slow path)
        if (int rc = launch_generic(g, k, 0, over, part, nparts, acc, launches, VTX ? vcounts : nullptr)) return rc;
    }
    if (int rc = count_dplus_ge(g, std::max(k - 1, 1), &n_min)) return rc;
    const int cu = c.compute_units > 0 ? c.compute_units : 256;
    // positions [lo, hi) in the d+-sorted order of the pivots with a < d+ <= b, cut at the smallest useful d+
    auto range = [&](int a, int b, int64_t *lo, int64_t *hi) -> int {
        if (int rc = count_dplus_ge(g, b + 1, lo)) return rc;
        if (int rc = count_dplus_ge(g, a + 1, hi)) return rc;
        *hi = std::min(*hi, n_min);
        *lo = std::max(*lo, over);  // positions [0, over) went through the generic path
        *hi = std::max(*hi, *lo);
        return GMSX_OK;
    };

    // Two streams: the wide bins (slab, 1024, 704: one or two workgroups per CU, LDS-bound occupancy) on the caller's
    // stream, the narrow bins and the wave kernel on a side stream — their small workgroups fill the wave slots and the
    // LDS the wide ones leave free.  All kernels add into the same accumulators; the side stream is joined below.
    // FOUR streams (round 3; two before): the bins differ in workgroup size and LDS footprint (one workgroup per CU for the 1024-wide
    // matrices … sixteen for the narrow ones), each leaves wave slots and LDS the others can use, and a bin's last workgroups no longer hold
    // up the next bin of the same stream.  GMSX_KC_STREAMS=1 … 4 (A/B).
    constexpr int kSides = 4;  // three for the bins (round robin with the launch stream), the last for the wave kernel of d+ <= 32
    static hipStream_t sides[kSides] = {nullptr, nullptr, nullptr, nullptr};
    static hipEvent_t ev_fork = nullptr, ev_joins[kSides] = {nullptr, nullptr, nullptr, nullptr};
    if (!sides[0]) {
        for (int i = 0; i < kSides; ++i) {
            GMSX_HIP(hipStreamCreateWithFlags(&sides[i], hipStreamNonBlocking));
            GMSX_HIP(hipEventCreateWithFlags(&ev_joins[i], hipEventDisableTiming));
        }
        GMSX_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    }
    const int n_streams = [] { const char *e = opt("KC_STREAMS"); const int v = e ? std::atoi(e) : 4; return v < 1 ? 1 : v > 4 ? 4 : v; }();
    GMSX_HIP(hipEventRecord(ev_fork, s));
    for (int i = 0; i < kSides; ++i) GMSX_HIP(hipStreamWaitEvent(sides[i], ev_fork, 0));
    hipStream_t side = n_streams > 1 ? sides[0] : s;
    // The receivers' pass (the rows the pivots copy instead of streaming their member) and the wave kernel of d+ <= 32 — which reads none of those rows —
    // start TOGETHER, each on a side stream of its own: both are latency-bound (60 - 70 % of their wave cycles waiting, 9 / 6 KB of LDS per workgroup),
    // so they share the CUs instead of queueing (28.5 + 58 ms one after the other at scale 26).  Every other bin waits for the rows (ev_rev).
    static hipEvent_t ev_rev = nullptr;
    if (!ev_rev) GMSX_HIP(hipEventCreateWithFlags(&ev_rev, hipEventDisableTiming));
    hipStream_t rev_stream = n_streams > 1 ? sides[kSides - 2] : s, small_stream = n_streams > 1 ? sides[kSides - 1] : s;
    // the bins of this call: slab widths (each from half its width on), then the LDS-matrix widths, each from the next one on — and, for the receivers'
    // pass, where each starts in the d+ order (`range` below): the shards of a bin are strides of ITS positions
    // k = 4 (count only): the LDS matrices of 512 < d+ <= 1472 are stored TRIANGULARLY (kc_tri_off) — two 1024-thread workgroups per CU up to 960, and
    // the pivots of 1024 < d+ <= 1472 (most of what used to be the first slab bin) count by pair lists in LDS instead of row bands over a global slab
    const bool tri = LV == 2 && !VTX && kc_tri_enabled();
    const int l_dmax[3] = {4096, 2048, 8192}, l_from[3] = {2048, tri ? kKcTriTop : 1024, 4096};  // slab bins: (from, dmax]
    const int m_rect[] = {1024, 704, 640, 512, 384, 256, 192, 128, 96, 64, 32}, m_tri[] = {kKcTriTop, kKcTriTwo, 512, 384, 256, 192, 128, 96, 64, 32};
    const int *m_dmax = tri ? m_tri : m_rect;  // LDS bins: (next entry, entry]; the last entry = lower end of the last bin
    const int n_m = tri ? int(sizeof(m_tri) / sizeof(int)) : int(sizeof(m_rect) / sizeof(int));
    KcBins bins;
    {
        int nb = 0;
        for (int64_t &x : bins.lo) x = 0;
        auto add = [&](int width) -> int {
            int64_t lo = 0;
            if (int rc = count_dplus_ge(g, width + 1, &lo)) return rc;
            if (nb < KcBins::kN) bins.lo[nb++] = std::max(lo, over);
            return GMSX_OK;
        };
        for (int w : l_dmax)
            if (int rc = add(w)) return rc;
        for (int b = 0; b < n_m; ++b)
            if (int rc = add(m_dmax[b])) return rc;
    }
    // (Measured and dropped: the tail receivers' pass on a stream of its own beside the hub receivers' — 425 … 442 against 421 ms at scale 26.)
    if (int rc = launch_kc_reverse(g, part, nparts, bins, acc, launches, rev_stream)) return rc;
    const KcRev rv{g->kc_rel, g->kc_aoff, g->kc_arena, g->kc_relt};
    if (g->kc_rel && (g->kc_items > 0 || g->kc_itemst > 0) && n_streams > 1) {
        GMSX_HIP(hipEventRecord(ev_rev, rev_stream));
        GMSX_HIP(hipStreamWaitEvent(s, ev_rev, 0));
        for (int i = 0; i + 2 < kSides; ++i) GMSX_HIP(hipStreamWaitEvent(sides[i], ev_rev, 0));
        if (g->kc_relt) GMSX_HIP(hipStreamWaitEvent(small_stream, ev_rev, 0));  // (the wave kernel reads its tail members' rows from the arena too)
    }
    int next_stream = 0;
    auto pick = [&]() -> hipStream_t {  // round robin over the launch stream and the side streams in use
        const int i = next_stream++ % n_streams;
        return i == 0 ? s : sides[i - 1];
    };
    // every way out of this function joins the side streams again — error returns included, so that no later call on the launch
    // stream can overtake kernels still running beside it
    struct Join {
        hipStream_t main;
        hipStream_t *sides;
        hipEvent_t *evs;
        bool armed = true;
        ~Join() {
            if (!armed) return;
            for (int i = 0; i < kSides; ++i)
                if (hipEventRecord(evs[i], sides[i]) == hipSuccess) (void)hipStreamWaitEvent(main, evs[i], 0);
        }
    } join{s, sides, ev_joins};
    // S: k-1 <= d+ <= 32 — launched first, beside the receivers' pass (see above)
    {
        int64_t lo = 0, hi = 0;
        if (int rc = range(0, 32, &lo, &hi)) return rc;
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (cnt > 0) {
            const int64_t blocks = std::min<int64_t>((cnt + 3) / 4, int64_t(cu) * 32);
            hipLaunchKernelGGL((k_kc_small<LV, VTX>), dim3(unsigned(blocks)), dim3(256), 0, small_stream, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff, g->bmpool,
                               g->order, lo, hi, nparts, part, acc, g->oldid, vcounts, rv);
            ++*launches;
        }
    }
    const bool timing = opt("TIMING") != nullptr;  // the bins' pivot counts on stderr
    // k = 4 on the MATRIX CORES (round 6): the bins of d+ > 512 BUILD their matrices into a pool — chunk by chunk, a pool region and a stream per chunk in
    // turn — and each chunk's count is one k_kc4_mfma launch behind its BUILD on the same stream (kc4_mfma.hpp): the BUILD of the next chunk (rows streamed
    // from HBM: latency and bandwidth) runs beside the count of this one (matrix cores and VALU, operands from the L2).
    bool exported_slab[3] = {false, false, false}, exported_tri = false;
    if constexpr (LV == 2 && !VTX) {
        // (a graph without a pivot wider than 512 allocates nothing)
        size_t needed = 0;
        if (kc_mfma_enabled()) {
            const int froms[5] = {4096, 2048, tri ? kKcTriTop : 1024, kKcTriTwo, 512}, tos[5] = {8192, 4096, 2048, kKcTriTop, kKcTriTwo};
            for (int b = 0; b < (tri ? 5 : 3); ++b) {
                int64_t lo = 0, hi = 0;
                if (int rc = range(froms[b], tos[b], &lo, &hi)) return rc;
                needed += size_t(part_count(lo, hi, nparts, part)) * (size_t(tos[b]) * size_t(kc4m_stride(tos[b])) + kKcSlotPad) * 4;
            }
        }
        if (needed > 0) {
            if (int rc = ensure_kc_pool(needed + (size_t(1) << 20))) return rc;
            KcPool &pool = kc_pool();
            const int n_regions = std::min(n_streams, kKcPoolRegions);
            const size_t region_words = pool.bytes / 4 / size_t(n_regions) & ~size_t(63);
            static bool x_attr = false;
            if (!x_attr) {
                GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<2, 1, true, false, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<2, 2, true, false, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<2, 4, true, false, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<2, 1, false, false, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kKcLdsDynMax));
                x_attr = true;
            }
            int chunk_seq = 0;
            // one bin (from, dmax]: slab variant WPL = 1 / 2 / 4 (dmax 2048 / 4096 / 8192) or the triangular LDS variant (WPL = 0)
            auto export_bin = [&](int from, int dmax, int wpl, bool *done) -> int {
                *done = false;
                if (!pool.base) return GMSX_OK;
                const size_t slot_words = size_t(dmax) * size_t(kc4m_stride(dmax)) + kKcSlotPad;
                const int64_t cap = std::min<int64_t>(int64_t(region_words / slot_words), kKcChunkMax);
                if (cap < 1) return GMSX_OK;  // the pool cannot hold one matrix of this bin: it keeps its in-kernel count
                int64_t lo = 0, hi = 0;
                if (int rc = range(from, dmax, &lo, &hi)) return rc;
                const int64_t cnt = part_count(lo, hi, nparts, part);
                *done = true;
                if (timing && cnt > 0) std::fprintf(stderr, "[gmsx kclique] matrix-core bin d+ <= %d: %lld pivots, %lld per chunk\n", dmax, (long long)cnt, (long long)cap);
                const int W = dmax / 32;
                for (int64_t q0 = 0; q0 < cnt; q0 += cap) {
                    const int64_t nq = std::min(cap, cnt - q0);
                    const int r = chunk_seq++ % n_regions;
                    hipStream_t st = r == 0 ? s : sides[r - 1];
                    const KcExport ex{pool.base + size_t(r) * region_words, pool.dpool + size_t(r) * kKcChunkMax, (unsigned long long)slot_words};
                    const int64_t first = lo + q0 * nparts, end = std::min(hi, lo + (q0 + nq) * nparts);
                    if (wpl > 0) {
                        constexpr int slab_threads = 1024;
                        const size_t lds = size_t(kBitmapWords) * 4 + size_t(kBitmapWords) * 2 + size_t(slab_threads / 64) * 4 * W * 4 +  // bitmap + prefix + row stage
                                           ((size_t(dmax) * 2 + 15) & ~size_t(15));                                                            // + the forward list
                        const unsigned blocks = unsigned(std::min<int64_t>(nq, int64_t(cu)));  // (128 registers x 1024 threads: one workgroup per CU)
                        if (wpl == 1)
                            hipLaunchKernelGGL((k_kc_block<2, 1, true, false, 1, false, true>), dim3(blocks), dim3(slab_threads), lds, st, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff, g->bmpool,
                                               g->dense_limit, g->order, first, end, nparts, part, dmax, W, W, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, ex);
                        else if (wpl == 2)
                            hipLaunchKernelGGL((k_kc_block<2, 2, true, false, 1, false, true>), dim3(blocks), dim3(slab_threads), lds, st, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff, g->bmpool,
                                               g->dense_limit, g->order, first, end, nparts, part, dmax, W, W, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, ex);
                        else
                            hipLaunchKernelGGL((k_kc_block<2, 4, true, false, 1, false, true>), dim3(blocks), dim3(slab_threads), lds, st, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff, g->bmpool,
                                               g->dense_limit, g->order, first, end, nparts, part, dmax, W, W, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, ex);
                    } else {
                        const size_t lds = size_t(kc_tri_off(dmax)) * 4 + ((size_t(dmax) * 2 + 15) & ~size_t(15));
                        const unsigned blocks = unsigned(std::min<int64_t>(nq, int64_t(cu) * 64));
                        constexpr int tri_threads = 1024;
                        hipLaunchKernelGGL((k_kc_block<2, 1, false, false, 1, true, true>), dim3(blocks), dim3(tri_threads), lds, st, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff,
                                           g->bmpool, g->dense_limit, g->order, first, end, nparts, part, dmax, W, W | 1, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid,
                                           vcounts, rv, ex);
                    }
                    // teams of G workgroups of one XCD per matrix (kc4_mfma.hpp): the wider the matrices, the more of an XCD's CUs share one — measured on 1 024
                    // matrices of d+ = 600 / 1 200 / 1 800 / 3 000: G = 1 / 2 / 4 / 8 are the fastest (0.092 / 0.220 / 0.717 / 1.46 ms; G = 1: 0.092 / 0.244 / 0.822 / 1.87)
                    const int G = dmax <= kKcTriTwo ? 1 : dmax <= kKcTriTop ? 2 : dmax <= 2048 ? 4 : 8;
                    const unsigned cgrid = unsigned(std::max(cu / (8 * G), 1) * 8 * G);  // (a multiple of 8 G; teams beyond the chunk's matrices leave at once)
                    // (Measured at scale 26 and dropped: 256- / 512-thread count workgroups, two or four per CU, with 512- / 768-thread BUILD workgroups beside them, so
                    //  that a count wave per SIMD fits next to the BUILD's — 504 … 545 against 489 ms: the two kernels slow each other more than the overlap gains.)
                    hipLaunchKernelGGL((k_kc4_mfma<2, 1024>), dim3(cgrid), dim3(1024), 0, st, ex.pool, slot_words, ex.dpool, int(nq), G, acc, kAccSlots, kAccStride);
                    *launches += 2;
                }
                return GMSX_OK;
            };
            if (int rc = export_bin(4096, 8192, 4, &exported_slab[2])) return rc;
            if (int rc = export_bin(2048, 4096, 2, &exported_slab[0])) return rc;
            if (int rc = export_bin(tri ? kKcTriTop : 1024, 2048, 1, &exported_slab[1])) return rc;
            if (tri) {
                bool a = false, b = false;
                if (int rc = export_bin(kKcTriTwo, kKcTriTop, 0, &a)) return rc;
                if (int rc = export_bin(512, kKcTriTwo, 0, &b)) return rc;
                exported_tri = a && b;
                if (a != b) return GMSX_ERR_KERNEL;  // (both bins have slots of at most 283 KB: a pool that holds one holds the other)
            }
        }
    }
    // L: 1024 < d+ <= 4096 (8192 for k <= 4), bit-matrix in a global slab per workgroup; one launch per row width (one / two / four words per lane)
    constexpr int NL = (LV <= 2) ? 3 : 2;
    size_t slab_bytes[3] = {0, 0, 0};
    int64_t l_lo[3], l_hi[3], l_cnt[3] = {0, 0, 0}, l_blocks[3];
    for (int b = 0; b < NL; ++b) {
        if (exported_slab[b]) continue;
        if (int rc = range(l_from[b], l_dmax[b], &l_lo[b], &l_hi[b])) return rc;
        l_cnt[b] = part_count(l_lo[b], l_hi[b], nparts, part);
        l_blocks[b] = std::min<int64_t>(l_cnt[b], cu);  // one workgroup per CU: the LDS tile / stage fills it
        if (timing && l_cnt[b] > 0) std::fprintf(stderr, "[gmsx kclique] slab bin d+ <= %d: %lld pivots\n", l_dmax[b], (long long)l_cnt[b]);
        if (l_cnt[b] > 0) slab_bytes[b] = size_t(l_blocks[b]) * l_dmax[b] * (l_dmax[b] / 32 + 1) * sizeof(uint32_t);
    }
    if (slab_bytes[0] + slab_bytes[1] + slab_bytes[2] > 0) {
        uint32_t *slabs = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&slabs), std::max({slab_bytes[0], slab_bytes[1], slab_bytes[2]})));  // the launches run back to back
        *slab_out = slabs;
        static bool l_attr[kMaxK + 1] = {false};
        if (!l_attr[VTX ? kMaxK : LV]) {
            GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 1, true, VTX>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
            GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 2, true, VTX>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
            if constexpr (LV <= 2)
                GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 4, true, VTX>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
            l_attr[VTX ? kMaxK : LV] = true;
        }
        for (int b = 0; b < NL; ++b) {
            if (l_cnt[b] <= 0) continue;
            const int dmax = l_dmax[b], W = dmax / 32, WS = W + 1;
            const int threads = LV == 2 ? 1024 : 512;
            // (Round 6 tried the other cut of a slab matrix — a CHUNK of 16 / 8 column words of every row in LDS and one lane per pair (i, j) as on the LDS
            //  matrices, pairs enumerated from the slab into per-wave lists: bit-exact, and SLOWER — 270.7 against 249.5 ms for the d+ <= 2048 bin, 171.6 against
            //  98.7 for d+ <= 4096 at scale 26 (profiles/r06/kc26_slab_chunks.txt).  The slab matrices are dense enough that the count is the AND + popcount
            //  words themselves, and a lane-per-pair loop reads BOTH rows from LDS for every word where the band loop below holds row i in registers.)
            const int WT = dmax == 8192 ? 128 : dmax == 4096 ? 288 : 576;  // k = 4 row band: WT rows x (W+1) words of LDS (multiple of 32)
            size_t lds = size_t(kBitmapWords) * 4 + size_t(kBitmapWords) * 2 + size_t(threads / 64) * 4 * W * 4;  // bitmap + prefix + row stage
            if (LV == 2) lds = std::max(lds, size_t(WT) * (W + 1) * 4 + size_t(threads / 64) * kKcRowList * sizeof(unsigned short));  // + the waves' neighbour lists (kc4_row_list)
            if (VTX) lds += size_t(dmax) * 4;  // column counters
            if (b == 2) {
                if constexpr (LV <= 2)
                    hipLaunchKernelGGL((k_kc_block<LV, 4, true, VTX>), dim3(unsigned(l_blocks[b])), dim3(threads), lds, s, g->hoff, g->hadj, g->toff, g->tadj,
                                       g->bmoff, g->bmpool, g->dense_limit, g->order, l_lo[b], l_hi[b], nparts, part, dmax, W, WS, WT, slabs, acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            } else if (b == 0)
                hipLaunchKernelGGL((k_kc_block<LV, 2, true, VTX>), dim3(unsigned(l_blocks[b])), dim3(threads), lds, s, g->hoff, g->hadj, g->toff, g->tadj,
                                   g->bmoff, g->bmpool, g->dense_limit, g->order, l_lo[b], l_hi[b], nparts, part, dmax, W, WS, WT, slabs, acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            else
                hipLaunchKernelGGL((k_kc_block<LV, 1, true, VTX>), dim3(unsigned(l_blocks[b])), dim3(threads), lds, s, g->hoff, g->hadj, g->toff, g->tadj,
                                   g->bmoff, g->bmpool, g->dense_limit, g->order, l_lo[b], l_hi[b], nparts, part, dmax, W, WS, WT, slabs, acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            ++*launches;
        }
    }
    // M: 32 < d+ <= 1024, bit-matrix in LDS; one launch per bin — the bins are cut where another workgroup fits a CU
    static bool attr_set[kMaxK + 1] = {false};
    if (!attr_set[VTX ? kMaxK : LV]) {
        GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 1, false, VTX, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kKcLdsDynMax));
        if constexpr (LV == 2 && !VTX)
            GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 1, false, VTX, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kKcLdsDynMax));
        GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 1, false, VTX, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kKcLdsDynMax));
        GMSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_block<LV, 1, false, VTX, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kKcLdsDynMax));
        attr_set[VTX ? kMaxK : LV] = true;
    }
    // the pipelined BUILD (member id -> extents -> first units, three members deep) in the narrower bins too: GMSX_KC_PIPE_ALL = 1 always, 0 never, unset =
    // when the oriented containers outgrow the 256 MB Infinity Cache (a member's row then costs HBM round trips, not cache hits — see DESIGN.md §5.2)
    const bool pipe_all = [&] {
        if (const char *e = opt("KC_PIPE_ALL")) return std::atoi(e) != 0;
        return pipe_all_default(g);
    }();
    // the step-stream BUILD (mode 2): GMSX_KC_STREAM_BUILD = 1 always, 0 never, unset = by graph size (stream_build_default)
    const bool stream_build = [&] {
        if (const char *e = opt("KC_STREAM_BUILD")) return std::atoi(e) != 0;
        return stream_build_default(g);
    }();
    for (int b = 0; b + 1 < n_m; ++b) {
        const int dmax = m_dmax[b], W = dmax / 32, WS = W | 1;  // odd stride: rows of one column spread over the LDS banks
        const bool tri_bin = tri && dmax > 512;
        if (tri_bin && exported_tri) continue;
        int64_t lo = 0, hi = 0;
        if (int rc = range(m_dmax[b + 1], dmax, &lo, &hi)) return rc;
        const int64_t cnt = part_count(lo, hi, nparts, part);
        if (timing && cnt > 0) std::fprintf(stderr, "[gmsx kclique] LDS bin d+ <= %d: %lld pivots\n", dmax, (long long)cnt);
        if (cnt > 0) {
            // (the bitmap, the prefix counts and the tail filter — 16 KB — are static LDS of the LDS-matrix variants)
            const size_t lds = (tri_bin ? size_t(kc_tri_off(dmax)) : size_t(dmax) * WS) * 4 + (VTX ? size_t(dmax) * 4 : 0) +
                               ((size_t(dmax) * 2 + 15) & ~size_t(15));  // + the list of the members streamed forward (2 bytes each; no bin loses a workgroup per CU to it)
            const int64_t blocks = std::min<int64_t>(cnt, int64_t(cu) * 64);
            const int threads = dmax >= 640 ? 1024 : dmax >= 384 ? 512 : 256;  // (d+ <= 640: TWO 1024-thread workgroups fit a CU — 53.8 KB of matrix each)
            // (one 1024-thread workgroup per CU from d+ = 513 on: four waves per SIMD whatever the registers — the pipelined BUILD pays there)
            // (the step-stream BUILD adds 256 descriptors to the dynamic LDS: where they no longer fit beside the matrix and the 4 KB static filter —
            // the d+ <= 1024 bin with per-vertex counts — the bin keeps its default BUILD instead of failing its launch: ADVICE r5)
            if (tri_bin) {
                if constexpr (LV == 2 && !VTX)
                    hipLaunchKernelGGL((k_kc_block<LV, 1, false, VTX, 1, true>), dim3(unsigned(blocks)), dim3(1024), lds, n_streams > 2 ? pick() : s, g->hoff, g->hadj,
                                       g->toff, g->tadj, g->bmoff, g->bmpool, g->dense_limit, g->order, lo, hi, nparts, part, dmax, W, WS, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            } else if (stream_build && lds + 256 * sizeof(KcDesc) <= size_t(kKcLdsDynMax))
                hipLaunchKernelGGL((k_kc_block<LV, 1, false, VTX, 2>), dim3(unsigned(blocks)), dim3(threads), lds + 256 * sizeof(KcDesc), n_streams > 2 ? pick() : (dmax >= 704 ? s : side), g->hoff, g->hadj,
                                   g->toff, g->tadj, g->bmoff, g->bmpool, g->dense_limit, g->order, lo, hi, nparts, part, dmax, W, WS, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            else if (threads == 1024 || pipe_all)
                hipLaunchKernelGGL((k_kc_block<LV, 1, false, VTX, 1>), dim3(unsigned(blocks)), dim3(threads), lds, n_streams > 2 ? pick() : (dmax >= 704 ? s : side), g->hoff, g->hadj,
                                   g->toff, g->tadj, g->bmoff, g->bmpool, g->dense_limit, g->order, lo, hi, nparts, part, dmax, W, WS, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            else
                hipLaunchKernelGGL((k_kc_block<LV, 1, false, VTX, 0>), dim3(unsigned(blocks)), dim3(threads), lds, n_streams > 2 ? pick() : (dmax >= 704 ? s : side), g->hoff, g->hadj,
                                   g->toff, g->tadj, g->bmoff, g->bmpool, g->dense_limit, g->order, lo, hi, nparts, part, dmax, W, WS, 0, static_cast<uint32_t *>(nullptr), acc, g->oldid, vcounts, rv, KcExport{nullptr, nullptr, 0ull});
            ++*launches;
        }
    }
    join.armed = false;
    for (int i = 0; i < kSides; ++i) {
        GMSX_HIP(hipEventRecord(ev_joins[i], sides[i]));
        GMSX_HIP(hipStreamWaitEvent(s, ev_joins[i], 0));
    }
    return GMSX_OK;
}

// gmsx_stats.stream_bytes of a k-clique call: the ALGORITHMIC bytes of this formulation, no cache assumed — per pivot u of the shard (d+ >= k - 1; position in
// the d+ order ≡ part mod nparts) its own containers once and, per member v, the containers of N+(v) the BUILD reads: the hub part as bitset words or
// 16-bit list, whichever kc_use_bitset picks (nothing when the pivot has no hub member), the tail part (4 bytes per id) when v is a tail member and the pivot
// has tail members; pivots of d+ <= 32 read a hub member's row as one 4-byte gather per lower hub member instead; a pivot wider than `slab_from` (1024; k = 4: 1472,
// or 512 when the count runs on the matrix cores) writes its d x d bit-matrix to global memory and reads it back once.  One wave per pivot.
__global__ __launch_bounds__(256) void k_stat_kc_bytes(int64_t n_min, int nparts, int part, const int32_t *__restrict__ order, const int32_t *__restrict__ dplus,
                                                     const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff,
                                                     const int32_t *__restrict__ tadj, int32_t dense_limit, const uint32_t *__restrict__ rel, const uint32_t *__restrict__ relt,
                                                     int slab_from, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long b = 0;
    for (int64_t qi = wave0;; qi += nwaves) {
        const int64_t pos = qi * nparts + part;
        if (pos >= n_min) break;
        const int32_t u = order[pos];
        const int64_t hb = hoff[u], tb = toff[u];
        int hc = int(hoff[u + 1] - hb);
        if (hc > 0 && hadj[hb + hc - 1] == 0xFFFFu) --hc;  // row padding
        const int tc = int(toff[u + 1] - tb), d = hc + tc;
        if (lane == 0) {
            b += 2ull * (unsigned long long)hc + 4ull * (unsigned long long)tc;
            if (d > slab_from) b += 2ull * (unsigned long long)d * (unsigned long long)((d + 31) / 32) * 4ull;
        }
        for (int i = lane; i < d; i += 64) {
            const bool is_hub = i < hc;
            const int32_t v = is_hub ? int32_t(hadj[hb + i]) : tadj[tb + (i - hc)];
            const int hl = int(hoff[v + 1] - hoff[v]), tl = int(toff[v + 1] - toff[v]);
            if (d <= 32 && is_hub) b += 4ull * (unsigned long long)i;  // inverted gathers into v's bitset container, one per lower hub member
            else if (d > 32 && i == 0) b += 0ull;  // the first member has nobody below it: not streamed
            else if (is_hub && rel && rel[hb + i] != 0xffffffffu)  // a reverse row: the prefix streamed at the receiver, its record, the row written and read back
                b += 2ull * (unsigned long long)i + 16ull + 8ull * (unsigned long long)((i + 31) >> 5);
            else if (!is_hub && relt && relt[tb + (i - hc)] != 0xffffffffu)  // … of a tail receiver: the whole hub list + the tail members below it, a 32-byte record, the row
                b += 2ull * (unsigned long long)hc + 4ull * (unsigned long long)(i - hc) + 32ull + 8ull * (unsigned long long)((i + 31) >> 5);
            else if (is_hub && v < dense_limit && int(bitset_words(v)) * 4 + 32 < hl * 2) b += 4ull * (unsigned long long)bitset_words(v);
            else {
                if (hc > 0) b += 2ull * (unsigned long long)hl;
                if (!is_hub && tc > 0) b += 4ull * (unsigned long long)tl;
            }
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) b += __shfl_xor(b, sft);
    if (lane == 0 && b) atomicAdd(out, b);
}

static int kclique_partial(const gmsx_graph *g, int k, int part, int nparts, uint64_t *out, gmsx_stats *st) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    if (k == 2) {  // edges: Σ d+ over the shard
        *out = 0;
        if (part == 0) *out = uint64_t(g->m);  // every part but the first reports 0; the sum over parts is m
        if (st) *st = gmsx_stats{0.0, 0.0, uint64_t(g->n), 0, 0, 0, 0};
        return GMSX_OK;
    }
    // the reverse-row lists of the graph: built by the first k-clique call on it, outside the timed region and reported in setup_ms (an immutable
    // container like the triangle-count task lists; the reference's harness likewise times its SetGraph build apart, k_clique_count_set_based.h:22)
    const bool rev_was_there = g->kc_rev_tried;
    if (k <= kMaxK && g->rows_sorted)
        if (int rc0 = ensure_kc_reverse(g, 8192)) return rc0;
    const double setup_ms = rev_was_there ? 0.0 : g->kc_rev_build_ms;
    unsigned long long *acc = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), sizeof(unsigned long long) * kAccSlots * kAccStride));
    struct Guard { void *p; ~Guard() { (void)hipFree(p); } } guard{acc};
    uint32_t *slabs = nullptr;
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * kAccSlots * kAccStride, s));
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    int launches = 0, rc = GMSX_OK;
    switch (k) {
        case 3: rc = launch_all<1>(g, part, nparts, acc, &launches, &slabs); break;
        case 4: rc = launch_all<2>(g, part, nparts, acc, &launches, &slabs); break;
        case 5: rc = launch_all<3>(g, part, nparts, acc, &launches, &slabs); break;
        case 6: rc = launch_all<4>(g, part, nparts, acc, &launches, &slabs); break;
        case 7: rc = launch_all<5>(g, part, nparts, acc, &launches, &slabs); break;
        case 8: rc = launch_all<6>(g, part, nparts, acc, &launches, &slabs); break;
        case 9: rc = launch_all<7>(g, part, nparts, acc, &launches, &slabs); break;
        case 10: rc = launch_all<8>(g, part, nparts, acc, &launches, &slabs); break;
        default: {  // k > kMaxK: every pivot that can head a k-clique (d+ >= k-1) through the generic list recursion
            int64_t n_min = 0;
            rc = g->rows_sorted ? count_dplus_ge(g, k - 1, &n_min) : GMSX_ERR_UNSUPPORTED;
            if (!rc) rc = launch_generic(g, k, 0, n_min, part, nparts, acc, &launches);
        }
    }
    Guard slab_guard{slabs};
    if (rc) return rc;
    GMSX_HIP(hipEventRecord(c.ev[1], s));
    GMSX_HIP(hipGetLastError());
    unsigned long long host[kAccSlots * kAccStride];
    GMSX_HIP(hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    unsigned long long total = 0;
    for (int i = 0; i < kAccSlots; ++i) total += host[i * kAccStride];
    *out = total;
    if (st) {
        float ms = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        unsigned long long alg = 0;  // outside the timed region
        int64_t n_min = 0;
        if (g->rows_sorted && count_dplus_ge(g, std::max(k - 1, 1), &n_min) == GMSX_OK && n_min > 0) {
            GMSX_HIP(hipMemsetAsync(acc, 0, 8, s));
            const int cu = c.compute_units > 0 ? c.compute_units : 256;
            hipLaunchKernelGGL(k_stat_kc_bytes, dim3(unsigned(cu * 8)), dim3(256), 0, s, n_min, nparts, part, g->order, g->dplus, g->hoff, g->hadj, g->toff, g->tadj,
                               g->dense_limit, g->kc_rel, g->kc_relt, (k == 4 && kc_tri_enabled()) ? ((kc_mfma_enabled() && kc_pool().base) ? 512 : kKcTriTop) : 1024, acc);
            GMSX_HIP(hipMemcpyAsync(&alg, acc, 8, hipMemcpyDeviceToHost, s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        *st = gmsx_stats{double(ms), setup_ms, uint64_t(part_count(0, g->n, nparts, part)), 0, 0, launches, 0, uint64_t(alg)};
    }
    return GMSX_OK;
}

// per-vertex counts through the k = 3 machinery (used by gmsx_tc_vertex_count2): vcounts is indexed by ORIGINAL vertex id
int kclique_vertex_counts(const gmsx_graph *g, unsigned long long *d_counts, gmsx_stats *st) {
    Ctx &c = ctx();
    hipStream_t s = c.stream;
    unsigned long long *acc = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&acc), sizeof(unsigned long long) * kAccSlots * kAccStride));
    struct Guard { void *p; ~Guard() { (void)hipFree(p); } } guard{acc};
    uint32_t *slabs = nullptr;
    if (g->rows_sorted)
        if (int rc0 = ensure_kc_reverse(g, 8192)) return rc0;
    GMSX_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * kAccSlots * kAccStride, s));
    GMSX_HIP(hipEventRecord(c.ev[0], s));
    int launches = 0;
    const int rc = launch_all<1, true>(g, 0, 1, acc, &launches, &slabs, d_counts);
    Guard slab_guard{slabs};
    if (rc) return rc;
    GMSX_HIP(hipEventRecord(c.ev[1], s));
    GMSX_HIP(hipGetLastError());
    unsigned long long host[kAccSlots * kAccStride];
    GMSX_HIP(hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    if (st) {
        unsigned long long total = 0;
        for (int i = 0; i < kAccSlots; ++i) total += host[i * kAccStride];
        float ms = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        *st = gmsx_stats{double(ms), 0.0, uint64_t(g->n), total, 0, launches, 0};  // probes = triangles met
    }
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_kclique_partial(const gmsx_graph *g, int k, int part, int nparts, uint64_t *cliques_partial, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !cliques_partial || nparts < 1 || part < 0 || part >= nparts || k < 2) return GMSX_ERR_INVALID;
        if (k > kMaxGenericK) return GMSX_ERR_UNSUPPORTED;  // the generic recursion keeps its per-level cursors in LDS: k <= 64
        if (int rc = ensure_init()) return rc;
        return kclique_partial(g, k, part, nparts, cliques_partial, stats);
    });
}

int gmsx_kclique_count(const gmsx_graph *g, int k, uint64_t *ordered_count, uint64_t *cliques, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!ordered_count) return GMSX_ERR_INVALID;
        uint64_t c = 0;
        if (int rc = gmsx_kclique_partial(g, k, 0, 1, &c, stats)) return rc;
        uint64_t fact = 1;
        for (int i = 2; i <= k; ++i) fact *= uint64_t(i);  // mod 2^64, like the reference's size_t sum of ordered cliques
        *ordered_count = c * fact;
        if (cliques) *cliques = c;
        return GMSX_OK;
    });
}

int gmsx_kclique_star_count(const gmsx_graph *g, int k, uint64_t *stars, uint64_t *star_members, gmsx_stats *stats) {
    return gmsx::guard([&]() -> int {
        if (!g || !stars || k < 1) return GMSX_ERR_INVALID;
        if (k + 1 > kMaxGenericK) return GMSX_ERR_UNSUPPORTED;
        if (int rc = ensure_init()) return rc;
        gmsx_stats st_k{}, st_k1{};
        uint64_t ck = uint64_t(g->n), ck1 = 0;  // C_1 = n: an isolated vertex is a 1-clique with an empty star
        if (k >= 2)
            if (int rc = kclique_partial(g, k, 0, 1, &ck, &st_k)) return rc;
        if (star_members) {
            if (int rc = kclique_partial(g, k + 1, 0, 1, &ck1, &st_k1)) return rc;
            *star_members = uint64_t(k + 1) * ck1;
        }
        *stars = ck;
        if (stats) {
            *stats = st_k1;
            stats->kernel_ms += st_k.kernel_ms;
            stats->setup_ms += st_k.setup_ms;
            stats->launches += st_k.launches;
            stats->units = uint64_t(g->n);
        }
        return GMSX_OK;
    });
}

}  // extern "C"
