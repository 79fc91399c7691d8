// libgmsx device layer: process/device binding, graph upload and the device-side construction of the
// set representations (the analogue of SetGraph::FromCGraph, representations/graphs/set_graph.h:86-89,152-181).
// gfx950 only.
#include "device_graph.hpp"

#include <algorithm>
#include <mutex>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>  // before rocprim: its texture iterator calls the host memset
#include <memory>
#include <new>
#include <random>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "../host/gmsx_internal.hpp"

namespace gmsx {

Ctx &ctx() {
    static Ctx c;
    return c;
}

int ensure_init() { return ctx().device >= 0 ? GMSX_OK : gmsx_init(-1); }

// ---- preprocessing kernels ------------------------------------------------------------------

// One wave per row: checks the canonical-row invariant and accumulates Σ_{u<v}(d_u+d_v) and max degree.
// flags[0] |= 1 unsorted/duplicate, 2 id out of range, 4 self loop; acc[3..6] = the arc-set hashes of the symmetry check.
// SYMMETRY without a search per arc: the rows are strictly ascending (checked here), so the arcs form a SET, and the graph is symmetric iff
// that set equals its transpose.  Both are hashed as multisets — Σ mix(u, v) and Σ mix(v, u) mod 2^64, twice with independent keyed mixes
// (the keys are drawn per process) — and compared on the host: equal sets give equal sums, different sets collide with probability 2^-128.
// A binary search of u in row v per arc (2.1 G random probes at scale 26) was 0.39 s of a 1.3 s upload; this streams the CSR once.
// Σ_{u<v}(d_u+d_v) = Σ_u d_u² on a symmetric graph (every vertex is counted once per incident edge): no degree gathers either.
__device__ __forceinline__ unsigned long long arc_mix(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
__global__ __launch_bounds__(256) void k_validate(int64_t n, const int64_t *__restrict__ off,
                                                  const int32_t *__restrict__ adj, int check_symmetry, unsigned long long key0,
                                                  unsigned long long key1,
                                                  unsigned long long *__restrict__ acc /* [0]=flags [1]=elements [2]=maxdeg [3..6]=hashes */) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned flags = 0;
    unsigned long long elems = 0;
    unsigned long long maxdeg = 0;
    unsigned long long f0 = 0, r0 = 0, f1 = 0, r1 = 0;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const unsigned long long du = (unsigned long long)(e - b);
        if (du > maxdeg) maxdeg = du;
        if (lane == 0) elems += du * du;
        for (int64_t j = b + lane; j < e; j += 64) {
            const int32_t v = adj[j];
            if (v < 0 || v >= n) { flags |= 2; continue; }
            if (v == u) flags |= 4;
            if (j + 1 < e && adj[j + 1] <= v) flags |= 1;
            if (check_symmetry) {
                const unsigned long long uv = ((unsigned long long)uint32_t(u) << 32) | uint32_t(v), vu = ((unsigned long long)uint32_t(v) << 32) | uint32_t(u);
                f0 += arc_mix(uv ^ key0);
                r0 += arc_mix(vu ^ key0);
                f1 += arc_mix(uv * 0x9e3779b97f4a7c15ull + key1);
                r1 += arc_mix(vu * 0x9e3779b97f4a7c15ull + key1);
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        elems += __shfl_down(elems, s);
        flags |= __shfl_down(flags, s);
        const unsigned long long o = __shfl_down(maxdeg, s);
        maxdeg = o > maxdeg ? o : maxdeg;
        f0 += __shfl_down(f0, s);
        r0 += __shfl_down(r0, s);
        f1 += __shfl_down(f1, s);
        r1 += __shfl_down(r1, s);
    }
    if (lane == 0) {
        if (flags) atomicOr(&acc[0], (unsigned long long)flags);
        if (elems) atomicAdd(&acc[1], elems);
        atomicMax(&acc[2], maxdeg);
        if (check_symmetry) {
            atomicAdd(&acc[3], f0);
            atomicAdd(&acc[4], r0);
            atomicAdd(&acc[5], f1);
            atomicAdd(&acc[6], r1);
        }
    }
}

// ---- rank ids: vertices by decreasing (degree, id) -----------------------------------------------
__global__ void k_rank_keys(int64_t n, const int64_t *__restrict__ off, unsigned long long *__restrict__ keys) {
    const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u < n) keys[u] = ((unsigned long long)(off[u + 1] - off[u]) << 32) | (unsigned long long)u;
}
__global__ void k_assign_ids(int64_t n, const unsigned long long *__restrict__ sorted_keys, int32_t *__restrict__ oldid,
                             int32_t *__restrict__ newid) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t o = int32_t(sorted_keys[i] & 0xffffffffull);
        oldid[i] = o;
        newid[o] = int32_t(i);
    }
}

// sizes of the two oriented containers of every vertex, written at its rank id (hub size padded to even)
__global__ __launch_bounds__(256) void k_count_parts(int64_t n, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                     const int32_t *__restrict__ newid, int hub_limit, int64_t *__restrict__ hcnt,
                                                     int64_t *__restrict__ tcnt, int32_t *__restrict__ dplus) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const int32_t nu = newid[u];
        int ch = 0, ct = 0;
        for (int64_t j = b + lane; j < e; j += 64) {
            const int32_t nv = newid[adj[j]];
            if (nv < nu) {
                if (nv < hub_limit) ch++; else ct++;
            }
        }
        for (int s = 32; s > 0; s >>= 1) {
            ch += __shfl_down(ch, s);
            ct += __shfl_down(ct, s);
        }
        if (lane == 0) {
            hcnt[nu] = (ch + 1) & ~1;
            tcnt[nu] = ct;
            dplus[nu] = ch + ct;
        }
    }
}

// order-preserving compaction of the oriented neighbours (as rank ids) into the hub / tail containers
__global__ __launch_bounds__(256) void k_fill_parts(int64_t n, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                    const int32_t *__restrict__ newid, int hub_limit, const int64_t *__restrict__ hoff,
                                                    const int64_t *__restrict__ toff, uint16_t *__restrict__ hadj,
                                                    int32_t *__restrict__ tadj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t u = wave0; u < n; u += nwaves) {
        const int64_t b = off[u], e = off[u + 1];
        const int32_t nu = newid[u];
        int64_t ho = hoff[nu], to = toff[nu];
        const int64_t hend = hoff[nu + 1];
        for (int64_t base = b; base < e; base += 64) {
            const int64_t j = base + lane;
            bool kh = false, kt = false;
            int32_t nv = 0;
            if (j < e) {
                nv = newid[adj[j]];
                kh = nv < nu && nv < hub_limit;
                kt = nv < nu && nv >= hub_limit;
            }
            const unsigned long long mh = __ballot(kh), mt = __ballot(kt);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (kh) hadj[ho + __popcll(mh & below)] = uint16_t(nv);
            if (kt) tadj[to + __popcll(mt & below)] = nv;
            ho += __popcll(mh);
            to += __popcll(mt);
        }
        if (lane == 0 && ho < hend) hadj[ho] = 0xFFFF;  // pad to an even count
    }
}

// ---- stream rows: per row the cheapest of list / bitset / byte-delta, in whole 16-byte units ------------------------------------
// units of the byte-delta form of one sorted 16-bit list: unit = 16-bit base id, one count byte, 13 gap bytes; a gap above 255
// ends the unit (the next unit's base is free), unused gap bytes stay 0 and are cut off by the count.  `emit` != nullptr writes
// the units.  One THREAD per row: the encoding is sequential.
__device__ inline uint32_t delta_encode(const uint16_t *__restrict__ row, int len, uint32_t *__restrict__ emit) {
    uint32_t units = 0;
    int i = 0;
    while (i < len && row[i] != 0xFFFFu) {
        uint32_t w[4] = {row[i], 0u, 0u, 0u};  // base id in the low half of word 0
        uint32_t cur = row[i];
        int slot = 3;  // byte position of the next gap (3 … 15); byte 2 holds the number of ids of the unit
        ++i;
        while (i < len && row[i] != 0xFFFFu && slot < 16) {
            const uint32_t gap = uint32_t(row[i]) - cur;  // >= 1: rows are strictly ascending
            if (gap > 255u) break;                         // does not fit a byte: the next unit starts at this id (its base is free)
            w[slot >> 2] |= gap << ((slot & 3) * 8);
            ++slot;
            cur = row[i];
            ++i;
        }
        w[0] |= uint32_t(slot - 2) << 16;  // ids in this unit: 1 … 14
        if (emit) {
            emit[units * 4 + 0] = w[0]; emit[units * 4 + 1] = w[1]; emit[units * 4 + 2] = w[2]; emit[units * 4 + 3] = w[3];
        }
        ++units;
    }
    return units;
}
// units of the 12-bit-gap form (form 3): unit = 16-bit base id, 4-bit count (1 … 10), nine 12-bit gaps at bit 20 + 12(k-1); a gap above
// 4095 ends the unit, unused gaps stay 0 (the running id then repeats the last one; the count cuts them off).
__device__ inline uint32_t gap12_encode(const uint16_t *__restrict__ row, int len, uint32_t *__restrict__ emit) {
    uint32_t units = 0;
    int i = 0;
    while (i < len && row[i] != 0xFFFFu) {
        uint32_t w[5] = {row[i], 0u, 0u, 0u, 0u};
        uint32_t cur = row[i];
        int k = 1;  // ids in the unit so far
        ++i;
        while (i < len && row[i] != 0xFFFFu && k < 10) {
            const uint32_t gap = uint32_t(row[i]) - cur;
            if (gap > 4095u) break;
            const int pos = 20 + 12 * (k - 1), word = pos >> 5, sh = pos & 31;
            w[word] |= gap << sh;
            if (sh > 20) w[word + 1] |= gap >> (32 - sh);
            ++k;
            cur = row[i];
            ++i;
        }
        w[0] |= uint32_t(k) << 16;
        if (emit) {
            emit[units * 4 + 0] = w[0]; emit[units * 4 + 1] = w[1]; emit[units * 4 + 2] = w[2]; emit[units * 4 + 3] = w[3];
        }
        ++units;
    }
    return units;
}
// delta_mode: 0 = never, 1 = when it is at least 15 % smaller than the list, 2 = whenever possible (test hook)
// Every row has TWO slots (2v, 2v+1).  Slot 1 is used by the HYBRID form only (round 3): the ids of a heavy row below B — 512 … 4096, the
// rank ids of the biggest hubs, where the neighbours of every vertex of a power-law graph crowd — as a PREFIX BITMAP over [0, B) in slot 0
// (B/128 units, probed by AND + popcount like a bitset row) and the ids from B on as a list / byte-delta row of their own in slot 1.  It
// is Roaring's per-chunk choice of container (bitset where dense, array where sparse) applied at the one boundary that matters; on the
// forward streams of RMAT scale 21 it takes 10 % off the hub stream units (numpy estimate), and a bitmap unit costs 14 VALU instructions
// against 32 (list) / 81 (delta).  ksplit[v] = ids below B (0 = not hybrid).
static constexpr int kHybridB[4] = {512, 1024, 2048, 4096};
__device__ inline int lower_bound_u16(const uint16_t *__restrict__ row, int len, uint32_t key) {  // first index with row[i] >= key (the 0xFFFF pad sorts last)
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (uint32_t(row[mid]) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ void k_srow_sizes(int64_t n, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int32_t *__restrict__ dplus,
                             int32_t dense_limit, int delta_mode, int delta_pct, int gap12_mode, int hybrid_mode, int64_t *__restrict__ units_out,
                             int64_t *__restrict__ small_out, uint32_t *__restrict__ real_out, unsigned char *__restrict__ form_out,
                             int32_t *__restrict__ ksplit) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v > n) return;
    if (v == n) { units_out[2 * n] = 0; small_out[2 * n] = 0; return; }
    const int64_t b = hoff[v];
    const int len = int(hoff[v + 1] - b);
    uint32_t best = uint32_t((len + 7) / 8);  // list: 8 ids per unit
    int form = kFormList;
    if (len > 0 && v < dense_limit) {
        const uint32_t bits = uint32_t(bitset_words(int32_t(v)) / 4);
        if (bits * 16u + 32u < uint32_t(len) * 2u) { best = bits; form = kFormBitset; }  // same rule as the kernels always used
    }
    if (len > 0 && delta_mode > 0 && (delta_mode == 2 || len >= 24)) {  // also against the bitset: between 1/16 and 1/9 density delta is smaller
        const uint32_t d = delta_encode(hadj + b, len, nullptr);
        if ((delta_mode == 2 && form == kFormList) || d * 100u <= best * uint32_t(delta_pct)) { best = d; form = kFormDelta; }
    }
    if (len > 0 && gap12_mode > 0 && form == kFormList && (gap12_mode == 2 || len >= 24)) {  // rows too sparse for byte gaps: 10 ids per unit instead of 8
        const uint32_t d = gap12_encode(hadj + b, len, nullptr);
        if (gap12_mode == 2 || d * 100u <= best * 90u) { best = d; form = kFormGap12; }
    }
    // hybrid: prefix bitmap over [0, B) + the rest as list / delta.  Heavy rows only (the light-pivot kernel reads ONE descriptor per
    // member), and only when it is at least 10 % smaller in all
    uint32_t best2 = 0;
    int form2 = kFormList, ks = 0;
    if (hybrid_mode > 0 && len >= 32 && dplus[v] >= kHeavy && form != kFormBitset) {
        uint32_t tot = best;
        for (int t = 0; t < 4; ++t) {
            const int B = kHybridB[t];
            if (v <= B) break;
            const int k = lower_bound_u16(hadj + b, len, uint32_t(B));
            if (k * 2 <= B / 8 + 16) continue;  // the prefix as a list would be no bigger than its bitmap
            const int rl = len - k;
            uint32_t r = uint32_t((rl + 7) / 8);
            int rf = kFormList;
            if (rl >= 24 && delta_mode > 0) {
                const uint32_t d = delta_encode(hadj + b + k, rl, nullptr);
                if (d * 100u <= r * uint32_t(delta_pct)) { r = d; rf = kFormDelta; }
            }
            const uint32_t cand = uint32_t(B / 128) + r;
            if (cand < tot && (hybrid_mode == 2 || cand * 100u <= best * 90u)) {
                tot = cand;
                ks = k;
                best2 = r;
                form2 = rf;
            }
        }
        if (ks > 0) {
            best = tot - best2;  // B / 128 units of prefix bitmap
            form = kFormBitset;
        }
    }
    ksplit[v] = ks;
    // rows of 8 units (128 bytes) or more start on 128-byte boundaries of their own region of the pool: a row fetch then touches
    // ceil(L/128) lines instead of L/128 + 1 (≈5 % of the heavy-pivot traffic); the small rows are packed behind them
    const uint32_t u2[2] = {best, best2};
    const int f2[2] = {form, form2};
    for (int sl = 0; sl < 2; ++sl) {
        const bool big = u2[sl] >= 8u;
        units_out[2 * v + sl] = big ? int64_t((u2[sl] + 7u) & ~7u) : 0;
        small_out[2 * v + sl] = big ? 0 : int64_t(u2[sl]);
        real_out[2 * v + sl] = u2[sl];
        form_out[2 * v + sl] = (unsigned char)f2[sl];
    }
}
__global__ void k_srow_fill(int64_t n, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj, const int64_t *__restrict__ bmoff,
                            const uint32_t *__restrict__ bmpool, const int64_t *__restrict__ uoff, const int64_t *__restrict__ soff,
                            const uint32_t *__restrict__ real, const unsigned char *__restrict__ form, const int32_t *__restrict__ ksplit,
                            unsigned long long *__restrict__ srow, unsigned long long *__restrict__ srow2, uint32_t *__restrict__ spool) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int64_t b = hoff[v];
    const int len = int(hoff[v + 1] - b);
    const int ks = ksplit[v];  // > 0: slot 0 = prefix bitmap of the first ks ids, slot 1 = the ids behind them
    for (int sl = 0; sl < 2; ++sl) {
        const int64_t units = int64_t(real[2 * v + sl]);
        const int64_t u0 = units >= 8 ? uoff[2 * v + sl] : uoff[2 * n] + soff[2 * v + sl];  // big rows: 128-byte aligned region first; small rows behind it
        const int fm = form[2 * v + sl];
        const unsigned long long d = units ? ((unsigned long long)u0 << 24) | ((unsigned long long)fm << 22) | (unsigned long long)units : 0ull;
        if (sl == 0) srow[v] = units ? d : ((unsigned long long)u0 << 24) | ((unsigned long long)fm << 22);
        else srow2[v] = d;
        if (units == 0) continue;
        uint32_t *dst = spool + u0 * 4;
        const uint16_t *row = hadj + b + (sl ? ks : 0);
        const int rl = sl ? len - ks : len;
        if (sl == 0 && ks > 0) {  // prefix bitmap
            for (int64_t w = 0; w < units * 4; ++w) dst[w] = 0;
            for (int i = 0; i < ks; ++i) dst[row[i] >> 5] |= 1u << (row[i] & 31u);
        } else if (fm == kFormDelta) {
            delta_encode(row, rl, dst);
        } else if (fm == kFormGap12) {
            gap12_encode(row, rl, dst);
        } else if (fm == kFormBitset) {
            const uint32_t *src = bmpool + bmoff[v];  // bitset_words(v) is a multiple of 4 words = whole units
            for (int64_t w = 0; w < units * 4; ++w) dst[w] = src[w];
        } else {
            for (int64_t w = 0; w < units * 4; ++w) {
                const int i = int(w) * 2;
                const uint32_t lo = i < rl ? row[i] : 0xFFFFu, hi = i + 1 < rl ? row[i + 1] : 0xFFFFu;
                dst[w] = lo | (hi << 16);
            }
        }
    }
}

// ---- stream rows of the tail parts: 32-bit list or 16-bit delta, whole 16-byte units ------------------------------------------
// 16-bit delta form of a tail row: unit = 32-bit base id, count (1 … 6) in the low half of word 1, five 16-bit gaps; a gap above
// 65535 ends the unit early.  6 ids per unit: 2.67 B/id against 4 B/id.
__device__ inline uint32_t tail_delta_encode(const int32_t *__restrict__ row, int len, uint32_t *__restrict__ emit) {
    uint32_t units = 0;
    int i = 0;
    while (i < len) {
        uint32_t w[4] = {uint32_t(row[i]), 0u, 0u, 0u};
        uint32_t cur = uint32_t(row[i]);
        int slot = 1;  // half-word index in words 1..3 (half-word 0 = the count): gap k lives in half-word k
        ++i;
        while (i < len && slot < 6) {
            const uint32_t gap = uint32_t(row[i]) - cur;
            if (gap > 65535u) break;
            w[1 + (slot >> 1)] |= gap << ((slot & 1) * 16);
            ++slot;
            cur = uint32_t(row[i]);
            ++i;
        }
        w[1] |= uint32_t(slot);  // ids in this unit
        if (emit) {
            emit[units * 4 + 0] = w[0]; emit[units * 4 + 1] = w[1]; emit[units * 4 + 2] = w[2]; emit[units * 4 + 3] = w[3];
        }
        ++units;
    }
    return units;
}
__global__ void k_trow_sizes(int64_t n, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, int delta_mode,
                             int64_t *__restrict__ units_out, int64_t *__restrict__ small_out, uint32_t *__restrict__ real_out, unsigned char *__restrict__ form_out) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v > n) return;
    if (v == n) { units_out[n] = 0; small_out[n] = 0; return; }
    const int64_t b = toff[v];
    const int len = int(toff[v + 1] - b);
    uint32_t best = uint32_t((len + 3) / 4);  // list: 4 ids per unit
    int form = kFormList;
    if (len >= 8 && delta_mode > 0) {
        const uint32_t d = tail_delta_encode(tadj + b, len, nullptr);
        if (delta_mode == 2 || d * 100u <= best * 85u) { best = d; form = kFormDelta; }
    }
    // like the hub rows: rows of 8 units (128 bytes) or more start on 128-byte boundaries of their own region, the small rows are packed
    // behind them — a row fetch touches ceil(L/128) lines instead of L/128 + 1 (the tail rows average 14 units: 2.8 lines unaligned)
    const bool big = best >= 8u;
    units_out[v] = big ? int64_t((best + 7u) & ~7u) : 0;
    small_out[v] = big ? 0 : int64_t(best);
    real_out[v] = best;
    form_out[v] = (unsigned char)form;
}
__global__ void k_trow_fill(int64_t n, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, const int64_t *__restrict__ uoff,
                            const int64_t *__restrict__ soff, const uint32_t *__restrict__ real, const unsigned char *__restrict__ form,
                            unsigned long long *__restrict__ trow, uint32_t *__restrict__ tpool) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int64_t units = int64_t(real[v]);
    const int64_t u0 = units >= 8 ? uoff[v] : uoff[n] + soff[v];
    trow[v] = ((unsigned long long)u0 << 24) | ((unsigned long long)form[v] << 22) | (unsigned long long)units;
    if (units == 0) return;
    uint32_t *dst = tpool + u0 * 4;
    const int64_t b = toff[v];
    const int64_t len = toff[v + 1] - b;
    if (form[v] == kFormDelta) tail_delta_encode(tadj + b, int(len), dst);
    else
        for (int64_t w = 0; w < units * 4; ++w) dst[w] = w < len ? uint32_t(tadj[b + w]) : 0xfffffffeu;
}

__global__ void k_dense_sizes(int32_t limit, const int32_t *__restrict__ dplus, int64_t *__restrict__ sizes) {
    const int32_t v = int32_t(blockIdx.x * blockDim.x + threadIdx.x);
    // every hub vertex with out-neighbours gets a bitset (<= 268 MB in total); the kernels pick bitset or list per use
    if (v < limit) sizes[v] = bitset_words(v);  // also for d+ = 0: an all-zero bitset answers every gather with "no"
    if (v == limit) sizes[v] = 0;
}
__global__ __launch_bounds__(256) void k_dense_fill(int32_t limit, const int64_t *__restrict__ hoff, const uint16_t *__restrict__ hadj,
                                                    const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                    const int64_t *__restrict__ bmoff, uint32_t *__restrict__ pool) {
    const int lane = threadIdx.x & 63;
    const int32_t wave0 = int32_t((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    const int32_t nwaves = int32_t((int64_t(gridDim.x) * blockDim.x) >> 6);
    for (int32_t v = wave0; v < limit; v += nwaves) {
        const int64_t b = bmoff[v];
        if (bmoff[v + 1] == b) continue;
        for (int64_t j = hoff[v] + lane; j < hoff[v + 1]; j += 64) {
            const uint32_t id = hadj[j];
            if (id != 0xFFFFu) atomicOr(&pool[b + (id >> 5)], 1u << (id & 31u));
        }
        for (int64_t j = toff[v] + lane; j < toff[v + 1]; j += 64) {  // (rows beyond the hub range would contribute their tail targets, all < v, too; none has a bitset today)
            const uint32_t id = uint32_t(tadj[j]);
            atomicOr(&pool[b + (id >> 5)], 1u << (id & 31u));
        }
    }
}
// tsplit[u] = number of tail targets of u below `limit` (tail rows are ascending when rows_sorted; otherwise a plain count,
// which the kernels then do not use as a position)
__global__ __launch_bounds__(256) void k_tail_split(int64_t n, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj, int32_t limit,
                                                    int32_t *__restrict__ tsplit) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t u = wave0; u < n; u += nwaves) {
        int c = 0;
        for (int64_t j = toff[u] + lane; j < toff[u + 1]; j += 64) c += tadj[j] < limit;
        for (int s = 32; s > 0; s >>= 1) c += __shfl_down(c, s);
        if (lane == 0) tsplit[u] = c;
    }
}

__global__ void k_order_keys(int64_t n, const int32_t *__restrict__ dplus, int32_t *__restrict__ keys, int32_t *__restrict__ vals) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = dplus[i];
        vals[i] = int32_t(i);
    }
}

__global__ void k_warm(unsigned long long *p) {  // gmsx_init: loads the code object
    if (p) p[0] = 0;
}
// d+ descends along `order`; out[0] = number of vertices with d+ >= thr
__global__ void k_count_ge(int64_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ dplus, int32_t thr, int64_t *__restrict__ out) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (dplus[order[mid]] >= thr) lo = mid + 1; else hi = mid;
    }
    out[0] = lo;
}
// the task entries' fields (device_graph.hpp, TaskList): out[0] |= 1 when a stream row is longer than its list can say
__global__ void k_task_limits(int64_t n, const unsigned long long *__restrict__ srow, const unsigned long long *__restrict__ srow2,
                              const unsigned long long *__restrict__ trow, uint32_t hub_max, uint32_t tail_max, unsigned long long *__restrict__ out) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if ((srow[v] & 0x3fffffull) > hub_max || (srow2[v] & 0x3fffffull) > hub_max || (trow[v] & 0x3fffffull) > tail_max) atomicOr(out, 1ull);
}

// ---- inline rows (build step 4c / 5b) --------------------------------------------------------------------------------------------
static constexpr int kInlineChunk = 64;  // units of an inline row per task entry
// A heavy pivot hands the edges to its first gmsx_graph::inline_first (<= 64) members over inline too: their cut rows are 1-8 units, 2 bytes
// per id inline against 6 bytes of entry + a line behind a pointer.  The copies grow with the SQUARE of that number (member i receives i ids:
// 2016 ids per heavy pivot at 64): scale 26, device bytes / ms per pass at 64, 48, 40, 32 = 48.6 GB / 67.2, 44.3 / 67.6, 42.7 / 67.7,
// 41.4 / 68.0 (GMSX_TC_INLINE_FIRST; round 4 took 48: -4.2 GB for +0.6 %).  Round 5, with 6-byte task entries: 48 / 36 / 32 / 28 =
// 40.9 GB / 66.4 ms, 38.5 / 66.9, 37.9 / 67.1, 37.3 / 67.1 — member 28 receives 56 bytes of ids inline, and its cut row would be 6 bytes
// of entry + four 16-byte units: from there on the copy is no smaller than what it replaces.  Round 5 takes 28.
static constexpr int kDefaultInlineFirst = 28;
static constexpr int kInlineFirstMax = 64;
__device__ __forceinline__ bool takes_inline(int32_t v, int32_t inline_limit, const int32_t *__restrict__ dplus) {
    return v < inline_limit || dplus[v] >= kHeavy;
}
// Wave per pivot u (positions [first, end) of `order`: every pivot with d+ >= 2), one lane per member (hub part — padded at its end — in
// the low lanes, tail part behind it, both ascending; a light pivot has hl + tl <= 64, a heavy one takes part with its first inline_first).  COUNT: ids handed over per receiving member; FILL: copies them (cnt_* are
// the cursors then).  (A far member that is HEAVY takes the edge over inline like a near one; k_tc_light's edge list leaves it out.)
template <bool FILL>
__global__ __launch_bounds__(256) void k_inline_rows(int64_t first, int64_t end, const int32_t *__restrict__ order, const int64_t *__restrict__ hoff,
                                                     const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                     const int32_t *__restrict__ dplus, int32_t inline_limit, int inline_first, const int32_t *__restrict__ opos, int nparts, int part,
                                                     unsigned long long *__restrict__ cnt_h,
                                                     unsigned long long *__restrict__ cnt_t, const int64_t *__restrict__ ihoff,
                                                     const int64_t *__restrict__ itoff, int64_t base_h, uint16_t *__restrict__ pool_h, int64_t base_t,
                                                     int32_t *__restrict__ pool_t) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t pos = first + wave0; pos < end; pos += nwaves) {
        const int32_t u = order[pos];
        const int64_t hb = hoff[u], tb = toff[u];
        // a heavy pivot takes part with its first inline_first members only (the low lanes of its combined list); a light one with all
        const int first_u = dplus[u] >= kHeavy ? inline_first : kInlineFirstMax;
        const int hl = min(int(hoff[u + 1] - hb), first_u), tl = min(int(toff[u + 1] - tb), first_u - hl);
        int32_t mv = 0x7fffffff;
        if (lane < hl) {
            const uint32_t x = hadj[hb + lane];
            if (x != 0xFFFFu) mv = int32_t(x);
        } else if (lane - hl < tl) {
            mv = tadj[tb + (lane - hl)];
        }
        const bool valid = mv != 0x7fffffff, is_tail = valid && lane >= hl;
        const unsigned long long vmask = __ballot(valid);
        const int nhub = __popcll(__ballot(valid && lane < hl));  // the valid hub members are the lanes [0, nhub)
        const int below = __popcll(vmask & ((1ull << lane) - 1ull));
        const bool give = valid && below > 0 && takes_inline(mv, inline_limit, dplus);
        // a sharded upload (gmsx_graph_upload_shard) keeps the inline rows of the receivers this rank owns only; the hand-over itself (the
        // blanked descriptors below) is the same on every rank
        const bool mine = give && (nparts <= 1 || shard_of(opos[mv], nparts) == part);
        const int nh = mine ? (is_tail ? nhub : below) : 0;
        const int nt = mine && is_tail ? lane - hl : 0;
        if (!FILL) {
            if (nh) atomicAdd(&cnt_h[mv], (unsigned long long)nh);
            if (nt) atomicAdd(&cnt_t[mv], (unsigned long long)nt);
        } else {
            int64_t at_h = 0, at_t = 0;
            if (nh) at_h = (base_h + ihoff[mv]) * 8 + int64_t(atomicAdd(&cnt_h[mv], (unsigned long long)nh));
            if (nt) at_t = (base_t + itoff[mv]) * 4 + int64_t(atomicAdd(&cnt_t[mv], (unsigned long long)nt));
            for (int k = 0; k < nhub; ++k) {  // wave-uniform trip count
                const int32_t x = __shfl(mv, k);
                if (k < nh) pool_h[at_h + k] = uint16_t(x);
            }
            for (int k = 0; k < tl; ++k) {
                const int32_t x = __shfl(mv, hl + k);
                if (k < nt) pool_t[at_t + k] = x;
            }
        }
    }
}
__global__ void k_inline_units(int64_t n, const unsigned long long *__restrict__ ids_h, const unsigned long long *__restrict__ ids_t,
                               int64_t *__restrict__ units_h, int64_t *__restrict__ units_t) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v > n) return;
    units_h[v] = v < n ? int64_t((ids_h[v] + 7) / 8) : 0;
    units_t[v] = v < n ? int64_t((ids_t[v] + 3) / 4) : 0;
}
static constexpr int kDefaultHotWindows = 0, kDefaultHotKB = 2048, kDefaultHotMin = 16;
// ---- receivers -----------------------------------------------------------------------------------------------------------------------------
// A RECEIVER is a vertex that owns task lists: every heavy pivot (d+ >= kHeavy; receiver index = its position in `order`, which lists the
// heavy pivots first) and every light vertex of rank id < inline_limit (it can only receive inline rows; index n_heavy + its rank among
// those).  The build keeps kClasses counters per receiver: entries per class, then (after k_list_sizes) the class offsets inside the list.
__device__ __forceinline__ int64_t receiver_of(int32_t w, const int32_t *__restrict__ dplus, const int32_t *__restrict__ opos, const int64_t *__restrict__ lidx,
                                               int64_t n_heavy) {
    return dplus[w] >= kHeavy ? int64_t(opos[w]) : n_heavy + lidx[w];
}
__global__ void k_opos(int64_t n, const int32_t *__restrict__ order, int32_t *__restrict__ opos) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p < n) opos[order[p]] = int32_t(p);
}
__global__ void k_light_flags(int64_t limit, const int32_t *__restrict__ dplus, int64_t *__restrict__ flags) {
    const int64_t w = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w <= limit) flags[w] = (w < limit && dplus[w] < kHeavy) ? 1 : 0;
}
__global__ void k_recv_vertices(int64_t n_heavy, int64_t limit, const int32_t *__restrict__ order, const int32_t *__restrict__ dplus, const int64_t *__restrict__ lidx,
                                int32_t *__restrict__ recv_v) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_heavy) recv_v[i] = order[i];
    if (i < limit && dplus[i] < kHeavy) recv_v[n_heavy + lidx[i]] = int32_t(i);
}
// the inline entries of every receiver: its hub inline row and its tail inline row, each in chunks of kInlineChunk units (list form).
// MODE 0 counts them into the class counters, MODE 1 writes them (cursor = cur).
template <bool FILL>
__global__ void k_inline_entries(int64_t n, const int32_t *__restrict__ dplus, const int32_t *__restrict__ opos, const int64_t *__restrict__ lidx, int64_t n_heavy,
                                 int32_t inline_limit, const int64_t *__restrict__ ihoff, const int64_t *__restrict__ itoff, int64_t base_h, int64_t base_t,
                                 uint32_t *__restrict__ cnt, uint32_t *__restrict__ cur, const int64_t *__restrict__ hbeg, const int64_t *__restrict__ tbeg,
                                 TaskList htask, TaskList ttask, unsigned long long *__restrict__ totals, TcClasses cc) {
    const int kClasses = cc.count();
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int64_t uh = ihoff[v + 1] - ihoff[v], ut = itoff[v + 1] - itoff[v];
    if (uh == 0 && ut == 0) return;
    if (dplus[v] < kHeavy && v >= inline_limit) return;  // cannot happen: only heavy vertices and rank ids < inline_limit receive
    const int64_t p = receiver_of(int32_t(v), dplus, opos, lidx, n_heavy);
    uint32_t *c = cnt + p * kClasses;
    for (int64_t o = 0; o < uh; o += kInlineChunk) {
        const unsigned long long units = (unsigned long long)min(int64_t(kInlineChunk), uh - o);
        const unsigned long long d = ((unsigned long long)(base_h + ihoff[v] + o) << 24) | ((unsigned long long)kFormList << 22) | units;
        const int cls = hub_class(cc, d);
        if (!FILL) atomicAdd(&c[cls], 1u);
        else htask.put(hbeg[p] + c[cls] + atomicAdd(&cur[p * kClasses + cls], 1u), d);
    }
    for (int64_t o = 0; o < ut; o += kInlineChunk) {
        const unsigned long long units = (unsigned long long)min(int64_t(kInlineChunk), ut - o);
        const unsigned long long d = ((unsigned long long)(base_t + itoff[v] + o) << 24) | ((unsigned long long)kFormList << 22) | units;
        const int cls = tail_class(cc, d);
        if (!FILL) atomicAdd(&c[cls], 1u);
        else ttask.put(tbeg[p] + c[cls] + atomicAdd(&cur[p * kClasses + cls], 1u), d);
    }
    if (!FILL) {
        atomicAdd(&totals[0], (unsigned long long)((uh + kInlineChunk - 1) / kInlineChunk));
        atomicAdd(&totals[1], (unsigned long long)((ut + kInlineChunk - 1) / kInlineChunk));
    }
}
// per receiver: the class counters become the class offsets inside its hub list / tail list; the list lengths go to hcnt / tcnt
__global__ void k_list_sizes(int64_t n_recv, uint32_t *__restrict__ cnt, int64_t *__restrict__ hcnt, int64_t *__restrict__ tcnt, TcClasses cc) {
    const int kClasses = cc.count(), hub_end = cc.tail_base();
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p > n_recv) return;
    if (p == n_recv) { hcnt[p] = 0; tcnt[p] = 0; return; }
    uint32_t *c = cnt + p * kClasses;
    uint32_t run = 0;
    for (int k = 0; k < hub_end; ++k) { const uint32_t x = c[k]; c[k] = run; run += x; }
    hcnt[p] = run;
    run = 0;
    for (int k = hub_end; k < kClasses; ++k) { const uint32_t x = c[k]; c[k] = run; run += x; }
    tcnt[p] = run;
}

// ---- light edges (device_graph.hpp): thread per light pivot (positions [first, end) of `order`) ----------------------------------------
__global__ void k_ledge_count(int64_t first, int64_t end, const int32_t *__restrict__ order, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                              const int32_t *__restrict__ tsplit, const int32_t *__restrict__ dplus, int64_t *__restrict__ cnt) {
    const int64_t pos = first + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (pos > end) return;
    int64_t c = 0;
    if (pos < end) {
        const int32_t u = order[pos];
        const int64_t tb = toff[u];
        const int tl = int(toff[u + 1] - tb);
        for (int i = tsplit[u]; i < tl; ++i) c += dplus[tadj[tb + i]] < kHeavy ? 1 : 0;
    }
    cnt[pos - first] = c;
}
__global__ void k_ledge_fill(int64_t first, int64_t end, const int32_t *__restrict__ order, const int64_t *__restrict__ hoff, const int64_t *__restrict__ toff,
                             const int32_t *__restrict__ tadj, const int32_t *__restrict__ tsplit, const int32_t *__restrict__ dplus,
                             const int64_t *__restrict__ ebeg, int nparts, int part, uint4 *__restrict__ ledge) {
    const int64_t pos = first + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (pos >= end) return;
    const int32_t u = order[pos];
    const int64_t hb = hoff[u], tb = toff[u];
    const int tl = int(toff[u + 1] - tb);
    const unsigned long long hu = (unsigned long long)hb | ((unsigned long long)(hoff[u + 1] - hb) << 40);
    int64_t e = ebeg[pos - first];
    for (int i = tsplit[u]; i < tl; ++i) {
        const int32_t v = tadj[tb + i];
        if (dplus[v] >= kHeavy) continue;
        if (nparts <= 1 || shard_of(e, nparts) == part) {  // (every stripe of nparts consecutive edges has one member per shard: slot e / nparts)
            const unsigned long long tu = (unsigned long long)tb | ((unsigned long long)i << 40);  // the tail ids of u in front of v
            const unsigned long long hv = (unsigned long long)hoff[v] | ((unsigned long long)(hoff[v + 1] - hoff[v]) << 40);
            const unsigned long long tv = (unsigned long long)toff[v] | ((unsigned long long)(toff[v + 1] - toff[v]) << 40);
            const int64_t slot = nparts <= 1 ? e : e / nparts;
            ledge[2 * slot] = make_uint4(uint32_t(hu), uint32_t(hu >> 32), uint32_t(tu), uint32_t(tu >> 32));
            ledge[2 * slot + 1] = make_uint4(uint32_t(hv), uint32_t(hv >> 32), uint32_t(tv), uint32_t(tv >> 32));
        }
        ++e;
    }
}

// ---- task lists of the heavy pivots (device_graph.hpp) -------------------------------------------------------------------------
// The rule, evaluated once per oriented edge (u,v), u heavy: the edge is handed to v ("reverse") iff v is heavy too and the part of u's
// rows that v has to stream (cut at v's id) is strictly fewer 16-byte units than what u would stream of v's; otherwise u keeps it.
// A handed-over row is needed only up to the receiving pivot's id: N+(v) lies below v.  The units of u's hub stream row that can hold
// ids below hub member v (which has `below` hub members of u in front of it), and of u's tail stream row for tail member v:
__device__ __forceinline__ uint32_t cut_hub_units(const uint32_t *__restrict__ spool, unsigned long long d, int below, int32_t v) {
    const uint32_t units = uint32_t(d) & 0x3fffffu, form = (uint32_t(d) >> 22) & 3u;
    if (form == kFormList) return min(units, uint32_t(below + 7) / 8u);
    if (form == kFormBitset) return min(units, uint32_t(v + 127) / 128u);
    const uint4 *row = reinterpret_cast<const uint4 *>(spool) + (d >> 24);  // byte-delta / 12-bit gaps: units ascend by their 16-bit base id
    uint32_t lo = 0, hi = units;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (int32_t(row[mid].x & 0xffffu) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t cut_tail_units(const uint32_t *__restrict__ tpool, unsigned long long d, int below, int32_t v) {
    const uint32_t units = uint32_t(d) & 0x3fffffu, form = (uint32_t(d) >> 22) & 3u;
    if (form != kFormDelta) return min(units, uint32_t(below + 3) / 4u);
    const uint4 *row = reinterpret_cast<const uint4 *>(tpool) + (d >> 24);  // 16-bit delta: units ascend by their 32-bit base id
    uint32_t lo = 0, hi = units;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (int32_t(row[mid].x) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// wave-aggregated claim of `one` slot per active lane in the counter of class `cls` of ONE receiver (all lanes the same receiver, a few
// distinct classes): one atomic per distinct class; returns this lane's offset behind the counter's old value (FILL) / nothing (COUNT)
__device__ __forceinline__ uint32_t claim_by_class(uint32_t *ctr, int cls, bool active, int lane) {
    uint32_t mine = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __builtin_amdgcn_readlane(cls, leader);
        const unsigned long long same = __ballot(active && cls == c);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&ctr[c], uint32_t(__popcll(same)));
        base = uint32_t(__builtin_amdgcn_readlane(int(base), leader));
        if (active && cls == c) mine = base + uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
        todo &= ~same;
    }
    return mine;
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_task_lists(int64_t n_heavy, const int32_t *__restrict__ order, const int64_t *__restrict__ hoff,
                                                    const uint16_t *__restrict__ hadj, const int64_t *__restrict__ toff, const int32_t *__restrict__ tadj,
                                                    const int32_t *__restrict__ dplus, const unsigned long long *__restrict__ srow,
                                                    const unsigned long long *__restrict__ srow2, const int32_t *__restrict__ ksplit,
                                                    const unsigned long long *__restrict__ trow, int two_sided, const int32_t *__restrict__ opos,
                                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ cur, const int64_t *__restrict__ hbeg,
                                                    const int64_t *__restrict__ tbeg, TaskList htask, TaskList ttask,
                                                    int32_t *__restrict__ tunits, unsigned long long *__restrict__ reversed, const uint32_t *__restrict__ spool,
                                                    const uint32_t *__restrict__ tpool, int32_t inline_limit, int inline_first, int nparts, int part, TcClasses cc) {
    const int kClasses = cc.count();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    unsigned long long rev = 0;
    for (int64_t pos = wave0; pos < n_heavy; pos += nwaves) {  // the heavy pivots are the first n_heavy of `order`: receiver index = pos
        const int32_t u = order[pos];
        const int64_t hb = hoff[u], tb = toff[u];
        const int hl = int(hoff[u + 1] - hb), tl = int(toff[u + 1] - tb);
        const unsigned long long du_s = srow[u], du_s2 = srow2[u], du_t = trow[u];  // du_s2 != 0: hybrid row — du_s is the prefix bitmap, du_s2 the ids behind it
        const int ks_u = ksplit[u];
        uint32_t *cu = cnt + pos * kClasses, *ru = cur + pos * kClasses;
        const bool own_u = nparts <= 1 || shard_of(pos, nparts) == part;  // a sharded upload writes the lists of the receivers this rank owns only
        int kept = 0;
        for (int base = 0; base < hl + tl; base += 64) {
            const int i = base + lane;
            int32_t v = -1;
            if (i < hl) {
                const uint32_t x = hadj[hb + i];
                if (x != 0xFFFFu) v = int32_t(x);
            } else if (i < hl + tl) v = tadj[tb + i - hl];
            if (v >= 0 && i > 0 && i < inline_first && takes_inline(v, inline_limit, dplus)) v = -1;  // handed over inline (k_inline_rows): no entry
            bool reverse = false;
            uint32_t ch = 0, ch2 = 0, ct = 0;  // units of u's rows that v would have to stream (ch2: the second slot of a hybrid row)
            if (v >= 0 && i > 0 && two_sided && dplus[v] >= kHeavy) {
                if (i < hl) {
                    ch = cut_hub_units(spool, du_s, i, v);  // (a prefix bitmap is cut like a bitset row: ceil(v / 128) units)
                    if (du_s2 && i > ks_u) ch2 = cut_hub_units(spool, du_s2, i - ks_u, v);  // members behind the prefix: i - ks_u ids of the second slot lie below v
                } else {  // a tail member sees the whole hub part
                    ch = uint32_t(du_s) & 0x3fffffu;
                    ch2 = uint32_t(du_s2) & 0x3fffffu;
                }
                ct = (i < hl || toff[v + 1] == toff[v]) ? 0u : cut_tail_units(tpool, du_t, i - hl, v);  // a pivot without tail part has nothing to match
                const uint32_t keep = uint32_t(srow[v] & 0x3fffffull) + uint32_t(srow2[v] & 0x3fffffull) + (i > hl ? uint32_t(trow[v] & 0x3fffffull) : 0u);  // what u would stream
                reverse = ch + ch2 + ct < keep;
            }
            const bool fwd = v >= 0 && !reverse && own_u;
            if (reverse && nparts > 1 && shard_of(opos[v], nparts) != part) v = -1;
            // forward: v's rows against u — the first member has no member below it (the edge closes no triangle); the tail ids of the first
            // tail member lie below every tail id of the pivot; a hub member has no tail part
            const unsigned long long fh = (fwd && i > 0) ? srow[v] : 0ull, fh2 = (fwd && i > 0) ? srow2[v] : 0ull, ft = (fwd && i > hl) ? trow[v] : 0ull;
            const bool fh_on = (fh & 0x3fffffull) != 0, fh2_on = (fh2 & 0x3fffffull) != 0, ft_on = (ft & 0x3fffffull) != 0;
            const uint32_t sh = claim_by_class(FILL ? ru : cu, hub_class(cc, fh), fh_on, lane);
            const uint32_t sh2 = claim_by_class(FILL ? ru : cu, hub_class(cc, fh2), fh2_on, lane);
            const uint32_t st = claim_by_class(FILL ? ru : cu, tail_class(cc, ft), ft_on, lane);
            if (FILL) {
                if (fh_on) htask.put(hbeg[pos] + cu[hub_class(cc, fh)] + sh, fh);
                if (fh2_on) htask.put(hbeg[pos] + cu[hub_class(cc, fh2)] + sh2, fh2);
                if (ft_on) ttask.put(tbeg[pos] + cu[tail_class(cc, ft)] + st, ft);
            }
            kept += __popcll(__ballot(fwd));
            if (v >= 0 && reverse) {  // u's rows, cut at v, against v
                const int64_t pv = opos[v];
                const unsigned long long rh = ch ? (du_s & ~0x3fffffull) | ch : 0ull, rh2 = ch2 ? (du_s2 & ~0x3fffffull) | ch2 : 0ull,
                                         rt = ct ? (du_t & ~0x3fffffull) | ct : 0ull;
                uint32_t *cv = cnt + pv * kClasses;
                if (!FILL) {
                    if (rh) atomicAdd(&cv[hub_class(cc, rh)], 1u);
                    if (rh2) atomicAdd(&cv[hub_class(cc, rh2)], 1u);
                    if (rt) atomicAdd(&cv[tail_class(cc, rt)], 1u);
                    atomicAdd(&tunits[pv], 1);
                    ++rev;
                } else {
                    uint32_t *rv = cur + pv * kClasses;
                    if (rh) htask.put(hbeg[pv] + cv[hub_class(cc, rh)] + atomicAdd(&rv[hub_class(cc, rh)], 1u), rh);
                    if (rh2) htask.put(hbeg[pv] + cv[hub_class(cc, rh2)] + atomicAdd(&rv[hub_class(cc, rh2)], 1u), rh2);
                    if (rt) ttask.put(tbeg[pv] + cv[tail_class(cc, rt)] + atomicAdd(&rv[tail_class(cc, rt)], 1u), rt);
                }
            }
        }
        if (!FILL && lane == 0 && kept) atomicAdd(&tunits[pos], kept);
    }
    if (!FILL) {
        for (int s = 32; s > 0; s >>= 1) rev += __shfl_down(rev, s);
        if (lane == 0 && rev) atomicAdd(reversed, rev);
    }
}
// work items: every receiver's list in chunks of kTaskChunk entries (run once for the hub lists, once for the tail lists)
// (hub lists: phase by phase — TcClasses — with the items of phase 0 of ALL receivers first: slot ph * n_recv + p; the tail lists have one phase)
__global__ void k_item_counts(int64_t n_recv, const int64_t *__restrict__ lbeg, const uint32_t *__restrict__ cnt, TcClasses cc, int phases, int64_t *__restrict__ items) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p > n_recv) return;
    if (p == n_recv) { items[int64_t(phases) * n_recv] = 0; return; }
    for (int ph = 0; ph < phases; ++ph) {
        int64_t b = 0, e = lbeg[p + 1] - lbeg[p];
        if (phases > 1) hub_phase_range(cc, cnt + p * cc.count(), e, ph, &b, &e);
        items[int64_t(ph) * n_recv + p] = (e - b + kTaskChunk - 1) / kTaskChunk;
    }
}
// one thread per receiver: its list in chunks of kTaskChunk entries, each a self-contained record (device_graph.hpp); the pivot's
// container part from coff (hoff / toff)
__global__ void k_item_fill(int64_t n_recv, const int32_t *__restrict__ recv_v, const int32_t *__restrict__ opos, const int64_t *__restrict__ lbeg,
                            const int64_t *__restrict__ ioff, TaskList task, const int64_t *__restrict__ coff, int kind,
                            const uint32_t *__restrict__ cnt, TcClasses cc, int phases, gmsx_tc_item *__restrict__ items) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n_recv) return;
    const int32_t w = recv_v[p];
    const uint64_t cont = uint64_t(coff[w]) | (uint64_t(coff[w + 1] - coff[w]) << 40);
    for (int ph = 0; ph < phases; ++ph) {
    int64_t b = 0, e = lbeg[p + 1] - lbeg[p];
    if (phases > 1) hub_phase_range(cc, cnt + p * cc.count(), e, ph, &b, &e);
    b += lbeg[p];
    e += lbeg[p];
    int64_t k = ioff[int64_t(ph) * n_recv + p];
    for (int64_t x = b; x < e; x += kTaskChunk, ++k) {
        const int ne = int(min(int64_t(kTaskChunk), e - x));
        gmsx_tc_item it{};
        it.bc = uint64_t(x) | (uint64_t(ne) << 40);
        it.cont = cont;
        it.pivot = w;
        it.pos = opos[w];
        it.kind = uint32_t(kind);
        for (int f = 0; f < 4; ++f) {  // first entry of form >= f (binary search over the form-sorted chunk)
            int lo = 0, hi = ne;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (int((uint32_t(task.get(x + mid)) >> 22) & 3u) < f) lo = mid + 1; else hi = mid;
            }
            it.fbeg[f] = uint16_t(lo);
        }
        items[k] = it;
    }
    }
}

static int grid_for_waves(int64_t rows) {
    // wave-per-row grid-stride kernels: enough 256-thread blocks to fill the chip a few times over
    const int64_t want = (rows + 3) / 4;
    const int64_t cap = int64_t(ctx().compute_units > 0 ? ctx().compute_units : 256) * 32;
    return int(want < 1 ? 1 : (want > cap ? cap : want));
}

template <class T>
static int dmalloc(T **p, int64_t count, gmsx_graph *g, int line = __builtin_LINE()) {
    const size_t bytes = size_t(count > 0 ? count : 1) * sizeof(T);
    const bool trace = opt("MEM_TRACE") != nullptr;  // where the device bytes go: one line per allocation that stays with the graph
    if (trace && g && bytes >= (1u << 20)) std::fprintf(stderr, "gmsx mem: device_graph.hip:%d %.3f GB (elements of %zu B)\n", line, double(bytes) * 1e-9, sizeof(T));
    if (g && g->tc_building && g->tc_limit_bytes > 0 && g->device_bytes + int64_t(bytes) - g->tc_base_bytes > g->tc_limit_bytes) {
        *p = nullptr;  // test hook: pretend the device is this small
        return GMSX_ERR_DEVICE_MEM;
    }
    if (hipMalloc(reinterpret_cast<void **>(p), bytes) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return GMSX_ERR_DEVICE_MEM;
    }
    if (g) g->device_bytes += int64_t(bytes);
    return GMSX_OK;
}

struct DevGuard {
    void *p;
    ~DevGuard() { (void)hipFree(p); }
};

// the triangle-count containers only (a failed or abandoned build_tc_sets leaves the base layout usable)
static void free_tc(gmsx_graph *g) {
    auto drop = [](auto *&p) {
        (void)hipFree(p);
        p = nullptr;
    };
    drop(g->tsplit); drop(g->srow); drop(g->srow2); drop(g->ksplit); drop(g->spool); drop(g->trow); drop(g->htask.lo); drop(g->htask.hi); drop(g->ttask.lo); drop(g->ttask.hi); drop(g->hitem); drop(g->titem); drop(g->tunits);
    drop(g->ledge); drop(g->tpool); drop(g->shard_hitem); drop(g->shard_titem);
    g->shard_idx_part = g->shard_idx_nparts = -1;
    g->device_bytes -= g->tc_bytes;
    g->tc_bytes = 0;
    g->tc_ready = false;
    g->htask_entries = g->ttask_entries = g->hitems = g->titems = g->inline_hentries = g->inline_tentries = 0;
    g->n_ledge = g->ledge_total = 0;
    g->task_reverse = g->inline_units = g->spool_units = g->tpool_units = 0;
    g->stats_part = g->stats_nparts = -1;
}

static void free_graph(gmsx_graph *g) {
    if (!g) return;
    (void)hipFree(g->off);
    (void)hipFree(g->adj);
    (void)hipFree(g->newid);
    (void)hipFree(g->oldid);
    (void)hipFree(g->hoff);
    (void)hipFree(g->hadj);
    (void)hipFree(g->toff);
    (void)hipFree(g->tadj);
    (void)hipFree(g->bmoff);
    (void)hipFree(g->bmpool);
    (void)hipFree(g->tsplit);
    (void)hipFree(g->srow);
    (void)hipFree(g->srow2);
    (void)hipFree(g->ksplit);
    (void)hipFree(g->spool);
    (void)hipFree(g->trow);
    (void)hipFree(g->htask.lo);
    (void)hipFree(g->htask.hi);
    (void)hipFree(g->ttask.lo);
    (void)hipFree(g->ttask.hi);
    (void)hipFree(g->hitem);
    (void)hipFree(g->titem);
    (void)hipFree(g->tunits);
    (void)hipFree(g->shard_hitem);
    (void)hipFree(g->shard_titem);
    (void)hipFree(g->ledge);
    (void)hipFree(g->tpool);
    (void)hipFree(g->dplus);
    (void)hipFree(g->order);
    (void)hipFree(g->kc_rel);
    (void)hipFree(g->kc_aoff);
    (void)hipFree(g->kc_arena);
    (void)hipFree(g->kc_rec);
    (void)hipFree(g->kc_item);
    (void)hipFree(g->kc_relt);
    (void)hipFree(g->kc_rect);
    (void)hipFree(g->kc_itemt);
    (void)hipFree(g->scratch);
    (void)hipFree(g->acc);
    delete g;
}

int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t count, hipStream_t s) {
    size_t tmp_bytes = 0;
    GMSX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, int64_t(0), size_t(count), rocprim::plus<int64_t>(), s));
    void *tmp = nullptr;
    GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    DevGuard g_tmp{tmp};
    GMSX_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, int64_t(0), size_t(count), rocprim::plus<int64_t>(), s));
    GMSX_HIP(hipStreamSynchronize(s));
    return GMSX_OK;
}

// ---- rows ascending: segmented radix sort in vertex ranges of < 2^31 entries ---------------------------------------------------------
struct OffsetMinus {
    int64_t base;
    __host__ __device__ int64_t operator()(int64_t x) const { return x - base; }
};
__global__ void k_lower_bound_i64(int64_t count, const int64_t *__restrict__ a, int64_t target, int64_t *__restrict__ out) {
    int64_t lo = 0, hi = count;  // first index with a[i] >= target (a ascending)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < target) lo = mid + 1; else hi = mid;
    }
    out[0] = lo;
}
template <class K>
static int sort_rows(K *keys, int64_t entries, int64_t n, const int64_t *d_off, int end_bit, hipStream_t s) {
    if (n <= 0 || entries <= 0) return GMSX_OK;
    int64_t chunk = int64_t(1) << 31;
    if (const char *e = opt("SORT_CHUNK")) {  // test hook: force several ranges on a small graph
        const long long v = std::atoll(e);
        if (v > 0) chunk = v;
    }
    int64_t *d_v = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_v), sizeof(int64_t)));
    DevGuard g_v{d_v};
    GMSX_HIP(hipStreamSynchronize(s));  // the offsets are read with blocking copies below: whatever produced them must be done
    int64_t v0 = 0, base = 0;
    while (v0 < n) {
        // the range [v0, v1): as many whole rows as fit below base + chunk, at least one
        int64_t v1 = n;
        if (entries - base > chunk) {
            hipLaunchKernelGGL(k_lower_bound_i64, dim3(1), dim3(1), 0, s, n + 1, d_off, base + chunk + 1, d_v);
            GMSX_HIP(hipStreamSynchronize(s));
            GMSX_HIP(hipMemcpy(&v1, d_v, sizeof(int64_t), hipMemcpyDeviceToHost));
            v1 = std::max(v0 + 1, std::min(n, v1 - 1));  // off[v1] <= base + chunk
        }
        int64_t end = 0;
        GMSX_HIP(hipMemcpy(&end, d_off + v1, sizeof(int64_t), hipMemcpyDeviceToHost));
        const int64_t cnt = end - base;
        if (cnt >= (int64_t(1) << 32)) return GMSX_ERR_UNSUPPORTED;  // one row of 2^32 entries: not a graph this library can hold
        if (cnt > 0) {
            K *sorted = nullptr;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&sorted), size_t(cnt) * sizeof(K)));
            DevGuard g_sorted{sorted};
            auto begin = rocprim::make_transform_iterator(d_off + v0, OffsetMinus{base});
            auto endit = rocprim::make_transform_iterator(d_off + v0 + 1, OffsetMinus{base});
            size_t tmp_bytes = 0;
            GMSX_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, keys + base, sorted, unsigned(cnt), unsigned(v1 - v0), begin, endit, 0, end_bit, s));
            void *tmp = nullptr;
            GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
            DevGuard g_tmp{tmp};
            GMSX_HIP(rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, keys + base, sorted, unsigned(cnt), unsigned(v1 - v0), begin, endit, 0, end_bit, s));
            GMSX_HIP(hipMemcpyAsync(keys + base, sorted, size_t(cnt) * sizeof(K), hipMemcpyDeviceToDevice, s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        v0 = v1;
        base = end;
    }
    return GMSX_OK;
}

// the same for (key, value) pairs: values reordered in place by their keys, segment by segment
template <class K, class V>
static int sort_segment_pairs(K *keys, V *vals, int64_t entries, int64_t n, const int64_t *d_off, int end_bit, hipStream_t s) {
    if (n <= 0 || entries <= 0) return GMSX_OK;
    int64_t chunk = int64_t(1) << 31;
    if (const char *e = opt("SORT_CHUNK")) {
        const long long v = std::atoll(e);
        if (v > 0) chunk = v;
    }
    int64_t *d_v = nullptr;
    GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&d_v), sizeof(int64_t)));
    DevGuard g_v{d_v};
    GMSX_HIP(hipStreamSynchronize(s));  // the offsets are read with blocking copies below: whatever produced them must be done
    int64_t v0 = 0, base = 0;
    while (v0 < n) {
        int64_t v1 = n;
        if (entries - base > chunk) {
            hipLaunchKernelGGL(k_lower_bound_i64, dim3(1), dim3(1), 0, s, n + 1, d_off, base + chunk + 1, d_v);
            GMSX_HIP(hipStreamSynchronize(s));
            GMSX_HIP(hipMemcpy(&v1, d_v, sizeof(int64_t), hipMemcpyDeviceToHost));
            v1 = std::max(v0 + 1, std::min(n, v1 - 1));
        }
        int64_t end = 0;
        GMSX_HIP(hipMemcpy(&end, d_off + v1, sizeof(int64_t), hipMemcpyDeviceToHost));
        const int64_t cnt = end - base;
        if (cnt >= (int64_t(1) << 32)) return GMSX_ERR_UNSUPPORTED;
        if (cnt > 0) {
            K *k_out = nullptr;
            V *v_out = nullptr;
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&k_out), size_t(cnt) * sizeof(K)));
            DevGuard g_ko{k_out};
            GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&v_out), size_t(cnt) * sizeof(V)));
            DevGuard g_vo{v_out};
            auto begin = rocprim::make_transform_iterator(d_off + v0, OffsetMinus{base});
            auto endit = rocprim::make_transform_iterator(d_off + v0 + 1, OffsetMinus{base});
            size_t tmp_bytes = 0;
            GMSX_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, keys + base, k_out, vals + base, v_out, unsigned(cnt), unsigned(v1 - v0), begin, endit, 0,
                                                         end_bit, s));
            void *tmp = nullptr;
            GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
            DevGuard g_tmp{tmp};
            GMSX_HIP(rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, keys + base, k_out, vals + base, v_out, unsigned(cnt), unsigned(v1 - v0), begin, endit, 0,
                                                         end_bit, s));
            GMSX_HIP(hipMemcpyAsync(vals + base, v_out, size_t(cnt) * sizeof(V), hipMemcpyDeviceToDevice, s));
            GMSX_HIP(hipStreamSynchronize(s));
        }
        v0 = v1;
        base = end;
    }
    return GMSX_OK;
}

// GMSX_TIMING=1: phase times of the device-side builds on stderr (each mark synchronises the stream)
struct PhaseTimer {
    hipStream_t s;
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char *what;
    PhaseTimer(hipStream_t st, const char *w) : s(st), on(opt("TIMING") != nullptr), t0(std::chrono::steady_clock::now()), what(w) {}
    void mark(const char *phase) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gmsx %s] %-28s %8.1f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Host -> device copy of a caller-owned, PAGEABLE array.  One hipMemcpyAsync of 8 GiB that the runtime touches for the first time moves at 26 GB/s on the
// MI355X hosts (its staging pipeline runs on one thread; tools/probes/h2d_probe.hip: 0.333 s); hipHostRegister pins at the same 26 GB/s before the copy starts
// (0.308 + 0.149 s).  Here: two pinned 64 MB staging buffers (kept for the life of the process), filled by the host substrate's threads while the other
// one is on the wire — 0.173 s for the same 8 GiB (50 GB/s; a warm buffer copies at 56).  Small copies go the plain way.
static constexpr size_t kH2dChunk = size_t(64) << 20;
struct H2dStage {
    char *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = false;
    H2dStage() {
        ok = hipHostMalloc(reinterpret_cast<void **>(&pin[0]), kH2dChunk, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&pin[1]), kH2dChunk, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        else {
            std::memset(pin[0], 0, kH2dChunk);  // first touch here, not under the first upload's clock
            std::memset(pin[1], 0, kH2dChunk);
        }
    }
};
static bool h2d_staged_off() {
    const bool off = [] { const char *e = opt("UPLOAD_STAGED"); return e && std::atoi(e) == 0; }();  // A/B: 0 = one hipMemcpyAsync (round 4)
    return off;
}
static H2dStage &h2d_stage() {
    static H2dStage st;  // (one upload at a time per process: the library's calls are not re-entrant on one stream anyway)
    return st;
}
static int staged_h2d(void *dst, const void *src, size_t bytes, hipStream_t s) {
    constexpr size_t kChunk = kH2dChunk;
    if (bytes < 2 * kChunk || h2d_staged_off()) {
        GMSX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return GMSX_OK;
    }
    H2dStage &st = h2d_stage();
    if (!st.ok) {
        GMSX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return GMSX_OK;
    }
    static std::mutex stage_mutex;          // the two pinned buffers, their events and `used` are process-wide: two host threads uploading different graphs
    std::lock_guard<std::mutex> lock(stage_mutex);  // take turns here instead of overwriting each other's chunks
    static bool used[2] = {false, false};  // across calls: the last chunks of the previous copy may still be on the wire when the next one starts filling
    int b = 0;
    for (size_t at = 0; at < bytes; at += kChunk, b ^= 1) {
        const size_t n = std::min(kChunk, bytes - at);
        if (used[b]) GMSX_HIP(hipEventSynchronize(st.ev[b]));
        char *to = st.pin[b];
        parallel_memcpy(to, static_cast<const char *>(src) + at, n);
        GMSX_HIP(hipMemcpyAsync(static_cast<char *>(dst) + at, to, n, hipMemcpyHostToDevice, s));
        GMSX_HIP(hipEventRecord(st.ev[b], s));
        used[b] = true;
    }
    return GMSX_OK;
}

static int build_device_sets(gmsx_graph *g, uint32_t flags) {
    hipStream_t s = ctx().stream;
    PhaseTimer pt(s, "base build");
    const int64_t n = g->n;
    if (int rc = dmalloc(&g->scratch, 16, g)) return rc;
    if (int rc = dmalloc(&g->acc, kAccWords, g)) return rc;
    GMSX_HIP(hipMemsetAsync(g->scratch, 0, 16 * sizeof(unsigned long long), s));

    // 1. invariant check + Σ(d_u+d_v) + max degree
    const int check_sym = (flags & GMSX_UPLOAD_TRUSTED) ? 0 : 1;
    static const std::array<unsigned long long, 2> sym_keys = []() noexcept {  // per-process keys of the symmetry hashes: no fixed input collides in every process
        try {
            std::random_device rd;  // (may throw where no entropy source is reachable)
            return std::array<unsigned long long, 2>{((unsigned long long)rd() << 32) | rd(), ((unsigned long long)rd() << 32) | rd()};
        } catch (...) {  // fallback: clock + address-space layout — still different from process to process
            const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
            const unsigned long long a = (unsigned long long)reinterpret_cast<uintptr_t>(&t);
            return std::array<unsigned long long, 2>{t * 0x9E3779B97F4A7C15ull ^ a, (a * 0xC2B2AE3D27D4EB4Full) ^ (t << 17)};
        }
    }();
    if (n > 0)
        hipLaunchKernelGGL(k_validate, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, check_sym, sym_keys[0], sym_keys[1], g->scratch);
    unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};
    GMSX_HIP(hipMemcpyAsync(acc, g->scratch, sizeof(acc), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    if (check_sym && (acc[3] != acc[4] || acc[5] != acc[6])) acc[0] |= 8;  // the arc set differs from its transpose
    if (!(flags & GMSX_UPLOAD_TRUSTED) && acc[0] != 0) return GMSX_ERR_NOT_CANONICAL;
    if ((flags & GMSX_UPLOAD_TRUSTED) && (acc[0] & 2)) return GMSX_ERR_NOT_CANONICAL;  // out-of-range ids are never tolerated
    if (!(flags & GMSX_UPLOAD_TRUSTED) && (g->nnz & 1)) return GMSX_ERR_NOT_CANONICAL;
    g->alg_elements = acc[1];
    g->max_deg = int32_t(acc[2]);
    g->m = g->nnz / 2;

    pt.mark("validate");
    const unsigned tb = unsigned((n + 255) / 256);
    // 2. rank ids: sort (degree, id) descending
    if (int rc = dmalloc(&g->newid, n, g)) return rc;
    if (int rc = dmalloc(&g->oldid, n, g)) return rc;
    if (n > 0) {
        unsigned long long *k_in = nullptr, *k_out = nullptr;
        if (int rc = dmalloc(&k_in, n, nullptr)) return rc;
        DevGuard g1{k_in};
        if (int rc = dmalloc(&k_out, n, nullptr)) return rc;
        DevGuard g2{k_out};
        hipLaunchKernelGGL(k_rank_keys, dim3(tb), dim3(256), 0, s, n, g->off, k_in);
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::radix_sort_keys_desc(nullptr, tmp_bytes, k_in, k_out, size_t(n), 0, 64, s));
        void *tmp = nullptr;
        GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
        DevGuard g3{tmp};
        GMSX_HIP(rocprim::radix_sort_keys_desc(tmp, tmp_bytes, k_in, k_out, size_t(n), 0, 64, s));
        hipLaunchKernelGGL(k_assign_ids, dim3(tb), dim3(256), 0, s, n, k_out, g->oldid, g->newid);
        GMSX_HIP(hipStreamSynchronize(s));
    }

    pt.mark("rank ids");
    // 3. container sizes -> offsets.  Test hook: bits 8..23 of `flags` shrink the hub id range so that small graphs
    //    exercise the tail containers (0 = the production value kHub).
    int hub_limit = int((flags >> 8) & 0xffffu);
    if (hub_limit == 0 || hub_limit > kHub) hub_limit = kHub;
    int64_t *hcnt = nullptr, *tcnt = nullptr;
    if (int rc = dmalloc(&hcnt, n + 1, nullptr)) return rc;
    DevGuard g_h{hcnt};
    if (int rc = dmalloc(&tcnt, n + 1, nullptr)) return rc;
    DevGuard g_t{tcnt};
    GMSX_HIP(hipMemsetAsync(hcnt, 0, size_t(n + 1) * sizeof(int64_t), s));
    GMSX_HIP(hipMemsetAsync(tcnt, 0, size_t(n + 1) * sizeof(int64_t), s));
    if (int rc = dmalloc(&g->dplus, n, g)) return rc;
    if (n > 0)
        hipLaunchKernelGGL(k_count_parts, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, g->newid, hub_limit, hcnt, tcnt, g->dplus);
    if (int rc = dmalloc(&g->hoff, n + 1, g)) return rc;
    if (int rc = dmalloc(&g->toff, n + 1, g)) return rc;
    if (int rc = exclusive_scan_i64(hcnt, g->hoff, n + 1, s)) return rc;
    if (int rc = exclusive_scan_i64(tcnt, g->toff, n + 1, s)) return rc;
    GMSX_HIP(hipMemcpy(&g->hub_entries, g->hoff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    GMSX_HIP(hipMemcpy(&g->tail_entries, g->toff + n, sizeof(int64_t), hipMemcpyDeviceToHost));

    // 4. containers (+16 bytes of slack each: the 16-byte row loads of the count kernels may overrun the last row)
    if (int rc = dmalloc(&g->hadj, g->hub_entries + 8, g)) return rc;
    if (int rc = dmalloc(&g->tadj, g->tail_entries + 4, g)) return rc;
    GMSX_HIP(hipMemsetAsync(g->hadj + g->hub_entries, 0xff, 8 * sizeof(uint16_t), s));
    GMSX_HIP(hipMemsetAsync(g->tadj + g->tail_entries, 0xff, 4 * sizeof(int32_t), s));
    if (n > 0)
        hipLaunchKernelGGL(k_fill_parts, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->off, g->adj, g->newid, hub_limit, g->hoff, g->toff,
                           g->hadj, g->tadj);

    pt.mark("containers count+fill");
    // 4a. both containers of every row ascending by rank id (the 0xFFFF pad sorts last): "the members below v" are then the entries in
    //     front of v — what the inline rows, the cut of handed-over rows, the delta forms and the k-clique binary searches rely on.
    //     rocPRIM's segmented sort counts items in 32 bits, so the rows are sorted in vertex ranges of < 2^31 entries each.
    if (int rc = sort_rows(g->hadj, g->hub_entries, n, g->hoff, 16, s)) return rc;
    if (int rc = sort_rows(g->tadj, g->tail_entries, n, g->toff, 32, s)) return rc;
    g->rows_sorted = true;

    pt.mark("row sorts");
    // 4b. bitset containers: every hub row (rank id < hub_limit) as a bitmap over [0, v) — the dense streaming form of the triangle
    //     kernel and the edge test of the k-clique recursion.  <= 268 MB.
    g->dense_limit = int32_t(std::min<int64_t>(n, hub_limit));
    g->bitset_limit = g->dense_limit;
    {
        const int32_t K = g->bitset_limit;
        int64_t *sizes = nullptr;
        if (int rc = dmalloc(&sizes, int64_t(K) + 1, nullptr)) return rc;
        DevGuard g_sz{sizes};
        if (int rc = dmalloc(&g->bmoff, int64_t(K) + 1, g)) return rc;
        hipLaunchKernelGGL(k_dense_sizes, dim3(unsigned(K / 256 + 1)), dim3(256), 0, s, K, g->dplus, sizes);
        if (int rc = exclusive_scan_i64(sizes, g->bmoff, int64_t(K) + 1, s)) return rc;
        GMSX_HIP(hipMemcpy(&g->bmpool_words, g->bmoff + K, sizeof(int64_t), hipMemcpyDeviceToHost));
        if (int rc = dmalloc(&g->bmpool, g->bmpool_words + 4, g)) return rc;
        GMSX_HIP(hipMemsetAsync(g->bmpool, 0, size_t(g->bmpool_words + 4) * sizeof(uint32_t), s));
        if (K > 0 && g->bmpool_words > 0)
            hipLaunchKernelGGL(k_dense_fill, dim3(grid_for_waves(K)), dim3(256), 0, s, K, g->hoff, g->hadj, g->toff, g->tadj, g->bmoff, g->bmpool);
    }
    pt.mark("bitsets");
    // 5. work-sorted launch order: rank ids by decreasing d+
    if (int rc = dmalloc(&g->order, n, g)) return rc;
    if (n > 0) {
        int32_t *keys_in = nullptr, *vals_in = nullptr, *keys_out = nullptr;  // (the sorted keys are dplus[order[i]]: not kept)
        if (int rc = dmalloc(&keys_out, n, nullptr)) return rc;
        DevGuard g_ko{keys_out};
        if (int rc = dmalloc(&keys_in, n, nullptr)) return rc;
        DevGuard g_ki{keys_in};
        if (int rc = dmalloc(&vals_in, n, nullptr)) return rc;
        DevGuard g_vi{vals_in};
        hipLaunchKernelGGL(k_order_keys, dim3(tb), dim3(256), 0, s, n, g->dplus, keys_in, vals_in);
        size_t tmp_bytes = 0;
        GMSX_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, keys_in, keys_out, vals_in, g->order, size_t(n), 0, 32, s));
        void *tmp = nullptr;
        GMSX_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
        DevGuard g_tmp{tmp};
        GMSX_HIP(rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys_in, keys_out, vals_in, g->order, size_t(n), 0, 32, s));
        int32_t top = 0;
        GMSX_HIP(hipMemcpyAsync(&top, keys_out, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GMSX_HIP(hipStreamSynchronize(s));
        g->max_dplus = top;
    }

    pt.mark("d+ order");
    GMSX_HIP(hipStreamSynchronize(s));
    GMSX_HIP(hipGetLastError());
    g->hub_limit = hub_limit;
    g->upload_flags = flags;
    return GMSX_OK;
}

// The triangle-count containers on top of the base layout: inline rows, stream rows, descriptors, task lists, work items (device_graph.hpp).
// Built by gmsx_graph_upload(…GMSX_UPLOAD_FOR_TC), gmsx_graph_prepare(g, GMSX_PREPARE_TC) or the first oriented triangle-count call —
// a k-clique or Bron–Kerbosch user never pays for them (the reference's set-based k-clique harness builds its SGraph INSIDE the timed
// region, k_clique_count_set_based.h:22).
static int build_tc_sets(gmsx_graph *g) {
    hipStream_t s = ctx().stream;
    PhaseTimer pt(s, "tc build");
    const int64_t n = g->n;
    const uint32_t flags = g->upload_flags;
    const int hub_limit = g->hub_limit;
    const unsigned tb = unsigned((n + 255) / 256);
    int32_t *opos = nullptr;  // position of every vertex in `order`: the receiver index of a heavy pivot, and what shard_of() is evaluated on
    if (int rc = dmalloc(&opos, n + 1, nullptr)) return rc;
    DevGuard g_opos{opos};
    if (n > 0) hipLaunchKernelGGL(k_opos, dim3(tb), dim3(256), 0, s, n, g->order, opos);
    // 4c. INLINE LIMIT.  A light pivot u (2 <= d+ < 64) hands the edge (u,v) over to v whenever v is a pivot of the workgroup kernel
    //     anyway (d+ >= 64) or a popular target (rank id < inline_limit): the members of u below v — the only ids of N+(u) that can
    //     be in N+(v) — are copied into v's INLINE ROWS, two more stream rows of v (16-bit hub ids / 32-bit tail ids, list form) that
    //     v's work items scan against v's own row like any other entry.  A 20-byte row behind a pointer would cost a 128-byte line
    //     per fetch; inline it is streamed.  Only the far, light members stay with the light-pivot kernel (k_tc_wave).
    {
        int64_t want = std::min<int64_t>(524288, n / 256);
        bool forced = false;
        if (const char *e = opt("INLINE_LIMIT")) {  // tuning / test knob
            const long long v = std::atoll(e);
            if (v >= 0 && v <= (1ll << 31) - 1) { want = v; forced = true; }
        }
        if (!forced && ((flags >> 8) & 0xffffu)) want = int64_t(4) * hub_limit;  // hub-limit test hook: near AND far tail on small graphs
        g->inline_limit = int32_t(std::min<int64_t>(n, std::max<int64_t>(want, g->dense_limit)));
        g->inline_first = kDefaultInlineFirst;
        if (const char *e = opt("TC_INLINE_FIRST")) g->inline_first = std::max(1, std::min(kInlineFirstMax, std::atoi(e)));  // A/B knob
        if (int rc = dmalloc(&g->tsplit, n, g)) return rc;
        if (n > 0) hipLaunchKernelGGL(k_tail_split, dim3(grid_for_waves(n)), dim3(256), 0, s, n, g->toff, g->tadj, g->inline_limit, g->tsplit);
    }

    pt.mark("tail split");
    // 5b. sizes of the inline rows (4c): ids handed over per receiving vertex
    int64_t n_heavy = 0, n_work = 0;
    unsigned long long *inl_h = nullptr, *inl_t = nullptr;  // [n + 1] ids per receiver, later the fill cursors
    int64_t *ihoff = nullptr, *itoff = nullptr;             // [n + 1] first 16-byte unit of the receiver's inline rows (relative to the inline region)
    if (int rc = dmalloc(&inl_h, n + 1, nullptr)) return rc;
    DevGuard g_inl_h{inl_h};
    if (int rc = dmalloc(&inl_t, n + 1, nullptr)) return rc;
    DevGuard g_inl_t{inl_t};
    if (int rc = dmalloc(&ihoff, n + 1, nullptr)) return rc;
    DevGuard g_ihoff{ihoff};
    if (int rc = dmalloc(&itoff, n + 1, nullptr)) return rc;
    DevGuard g_itoff{itoff};
    int64_t inline_h_units = 0, inline_t_units = 0;
    {
        if (n > 0) {
            int64_t *d_cnt = nullptr;
            if (int rc = dmalloc(&d_cnt, 2, nullptr)) return rc;
            DevGuard g_c{d_cnt};
            hipLaunchKernelGGL(k_count_ge, dim3(1), dim3(1), 0, s, n, g->order, g->dplus, int32_t(kHeavy), d_cnt);
            hipLaunchKernelGGL(k_count_ge, dim3(1), dim3(1), 0, s, n, g->order, g->dplus, int32_t(2), d_cnt + 1);
            GMSX_HIP(hipStreamSynchronize(s));
            int64_t h[2] = {0, 0};
            GMSX_HIP(hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost));
            n_heavy = h[0];
            n_work = h[1];
        }
        GMSX_HIP(hipMemsetAsync(inl_h, 0, size_t(n + 1) * sizeof(unsigned long long), s));
        GMSX_HIP(hipMemsetAsync(inl_t, 0, size_t(n + 1) * sizeof(unsigned long long), s));
        if (n_work > 0)
            hipLaunchKernelGGL(k_inline_rows<false>, dim3(grid_for_waves(n_work)), dim3(256), 0, s, int64_t(0), n_work, g->order, g->hoff, g->hadj,
                               g->toff, g->tadj, g->dplus, g->inline_limit, g->inline_first, opos, g->shard_nparts, g->shard_part, inl_h, inl_t, ihoff, itoff, int64_t(0), static_cast<uint16_t *>(nullptr),
                               int64_t(0), static_cast<int32_t *>(nullptr));
        int64_t *uh = nullptr, *ut = nullptr;
        if (int rc = dmalloc(&uh, n + 1, nullptr)) return rc;
        DevGuard g_uh{uh};
        if (int rc = dmalloc(&ut, n + 1, nullptr)) return rc;
        DevGuard g_ut{ut};
        hipLaunchKernelGGL(k_inline_units, dim3(unsigned(n / 256 + 1)), dim3(256), 0, s, n, inl_h, inl_t, uh, ut);
        if (int rc = exclusive_scan_i64(uh, ihoff, n + 1, s)) return rc;
        if (int rc = exclusive_scan_i64(ut, itoff, n + 1, s)) return rc;
        GMSX_HIP(hipMemcpy(&inline_h_units, ihoff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(&inline_t_units, itoff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
        g->inline_units = inline_h_units + inline_t_units;
    }

    pt.mark("inline row sizes");
    // 4d. stream rows of the heavy-pivot triangle kernel (needs the hub rows sorted: the delta form encodes ascending ids)
    int64_t inline_h_base = 0, inline_t_base = 0;  // first unit of the inline regions of spool / tpool
    {
        int delta_mode = g->rows_sorted ? 1 : 0;
        if (const char *e = opt("TC_DELTA")) {  // 0 = lists and bitsets only, 2 = delta wherever possible (test hook)
            const int v = std::atoi(e);
            if (v >= 0 && v <= 2 && g->rows_sorted) delta_mode = v;
        }
        int delta_pct = 85;  // take the delta form when it is at most this percentage of the list form (its decode costs ~30 % more VALU per id)
        if (const char *e = opt("TC_DELTA_PCT")) {
            const int v = std::atoi(e);
            if (v >= 10 && v <= 100) delta_pct = v;
        }
        // measured (MI355X, scale 26): the 12-bit-gap form takes 37 GB (9 %) off the algorithmic stream bytes but costs 6.3 VALU instructions
        // per id against 4 for a list: 87.5 -> 90.7 ms (scale 24: 16.3 -> 17.3).  The pass is on both roofs; the form stays OFF by default.
        int gap12_mode = 0;
        if (const char *e = opt("TC_GAP12")) {  // 1 = when at least 10 % smaller than the list, 2 = every row that would be a list (test hook)
            const int v = std::atoi(e);
            if (v >= 0 && v <= 2 && g->rows_sorted) gap12_mode = v;
        }
        // measured (MI355X, scale 26): mode 1 takes 0.3 % off the algorithmic stream bytes, mode 2 3.4 % — the 10 % of the forward-stream
        // estimate does not survive the two-sided design (the rows that end up streamed are the SMALLER ones of every edge and the cuts
        // already drop their low-id prefixes for the receivers that matter) — and the pass time does not move (80.5–80.9 ms all three).  OFF.
        int hybrid_mode = 0;
        if (const char *e = opt("TC_HYBRID")) {  // 1 = when at least 10 % smaller than the row's best single form, 2 = wherever smaller at all
            const int v = std::atoi(e);
            if (v >= 0 && v <= 2 && g->rows_sorted) hybrid_mode = v;
        }
        int64_t *units = nullptr, *uoff = nullptr, *small = nullptr, *soff = nullptr;
        uint32_t *real = nullptr;
        unsigned char *form = nullptr;
        const int64_t n2 = 2 * n;  // two slots per row
        if (int rc = dmalloc(&units, n2 + 1, nullptr)) return rc;
        DevGuard g_u{units};
        if (int rc = dmalloc(&uoff, n2 + 1, nullptr)) return rc;
        DevGuard g_o{uoff};
        if (int rc = dmalloc(&small, n2 + 1, nullptr)) return rc;
        DevGuard g_s{small};
        if (int rc = dmalloc(&soff, n2 + 1, nullptr)) return rc;
        DevGuard g_so{soff};
        if (int rc = dmalloc(&real, n2 + 1, nullptr)) return rc;
        DevGuard g_r{real};
        if (int rc = dmalloc(&form, n2 + 1, nullptr)) return rc;
        DevGuard g_f{form};
        if (int rc = dmalloc(&g->ksplit, n + 1, g)) return rc;
        hipLaunchKernelGGL(k_srow_sizes, dim3(unsigned(n / 256 + 1)), dim3(256), 0, s, n, g->hoff, g->hadj, g->dplus, g->dense_limit, delta_mode, delta_pct, gap12_mode,
                           hybrid_mode, units, small, real, form, g->ksplit);
        if (int rc = exclusive_scan_i64(units, uoff, n2 + 1, s)) return rc;
        if (int rc = exclusive_scan_i64(small, soff, n2 + 1, s)) return rc;
        int64_t big_units = 0, small_units = 0;
        GMSX_HIP(hipMemcpy(&big_units, uoff + n2, sizeof(int64_t), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(&small_units, soff + n2, sizeof(int64_t), hipMemcpyDeviceToHost));
        g->spool_units = big_units + small_units + inline_h_units;
        inline_h_base = big_units + small_units;
        if (g->spool_units >= (int64_t(1) << 40)) return GMSX_ERR_DEVICE_MEM;  // 40 offset bits in a srow entry (16 TB)
        if (int rc = dmalloc(&g->srow, n, g)) return rc;
        if (int rc = dmalloc(&g->srow2, n, g)) return rc;
        if (int rc = dmalloc(&g->spool, (g->spool_units + kPoolSlack) * 4, g)) return rc;  // + slack: the scans load whole lane groups past a row's end
        GMSX_HIP(hipMemsetAsync(g->spool, 0, size_t((g->spool_units + kPoolSlack) * 4) * sizeof(uint32_t), s));  // the alignment gaps are never read, but keep them defined
        if (inline_h_units > 0) GMSX_HIP(hipMemsetAsync(g->spool + inline_h_base * 4, 0xff, size_t(inline_h_units) * 16, s));  // list filler 0xFFFF
        if (n > 0)
            hipLaunchKernelGGL(k_srow_fill, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, g->hoff, g->hadj, g->bmoff, g->bmpool, uoff, soff, real, form,
                               g->ksplit, g->srow, g->srow2, g->spool);
        GMSX_HIP(hipStreamSynchronize(s));
    }
    // 4e. … and of the tail parts
    {
        int delta_mode = g->rows_sorted ? 1 : 0;
        if (const char *e = opt("TC_TAIL_DELTA")) {  // 0 = 32-bit lists only, 2 = delta wherever possible (test hook)
            const int v = std::atoi(e);
            if (v >= 0 && v <= 2 && g->rows_sorted) delta_mode = v;
        }
        int64_t *units = nullptr, *uoff = nullptr, *small = nullptr, *soff = nullptr;
        uint32_t *real = nullptr;
        unsigned char *form = nullptr;
        if (int rc = dmalloc(&units, n + 1, nullptr)) return rc;
        DevGuard g_u{units};
        if (int rc = dmalloc(&uoff, n + 1, nullptr)) return rc;
        DevGuard g_o{uoff};
        if (int rc = dmalloc(&small, n + 1, nullptr)) return rc;
        DevGuard g_s{small};
        if (int rc = dmalloc(&soff, n + 1, nullptr)) return rc;
        DevGuard g_so{soff};
        if (int rc = dmalloc(&real, n + 1, nullptr)) return rc;
        DevGuard g_r{real};
        if (int rc = dmalloc(&form, n + 1, nullptr)) return rc;
        DevGuard g_f{form};
        hipLaunchKernelGGL(k_trow_sizes, dim3(unsigned(n / 256 + 1)), dim3(256), 0, s, n, g->toff, g->tadj, delta_mode, units, small, real, form);
        if (int rc = exclusive_scan_i64(units, uoff, n + 1, s)) return rc;
        if (int rc = exclusive_scan_i64(small, soff, n + 1, s)) return rc;
        int64_t big_units = 0, small_units = 0;
        GMSX_HIP(hipMemcpy(&big_units, uoff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(&small_units, soff + n, sizeof(int64_t), hipMemcpyDeviceToHost));
        inline_t_base = big_units + small_units;
        g->tpool_units = inline_t_base + inline_t_units;
        if (g->tpool_units >= (int64_t(1) << 40)) return GMSX_ERR_DEVICE_MEM;
        if (int rc = dmalloc(&g->trow, n, g)) return rc;
        if (int rc = dmalloc(&g->tpool, (g->tpool_units + kPoolSlack) * 4, g)) return rc;
        GMSX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(g->tpool), int(0xFFFFFFFEu), size_t(g->tpool_units + kPoolSlack) * 4, s));  // filler -2 (alignment gaps, inline rows, the slack the scans may load)
        if (n > 0)
            hipLaunchKernelGGL(k_trow_fill, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, g->toff, g->tadj, uoff, soff, real, form, g->trow, g->tpool);
        GMSX_HIP(hipStreamSynchronize(s));
    }

    pt.mark("tail stream rows");
    // 5c. inline rows: copy the handed-over ids (pools exist now), blank the light pivots' descriptors of handed-over far members
    if (n_work > 0) {
        GMSX_HIP(hipMemsetAsync(inl_h, 0, size_t(n + 1) * sizeof(unsigned long long), s));  // now the fill cursors
        GMSX_HIP(hipMemsetAsync(inl_t, 0, size_t(n + 1) * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_inline_rows<true>, dim3(grid_for_waves(n_work)), dim3(256), 0, s, int64_t(0), n_work, g->order, g->hoff, g->hadj, g->toff,
                           g->tadj, g->dplus, g->inline_limit, g->inline_first, opos, g->shard_nparts, g->shard_part, inl_h, inl_t, ihoff, itoff, inline_h_base, reinterpret_cast<uint16_t *>(g->spool),
                           inline_t_base, reinterpret_cast<int32_t *>(g->tpool));
    }
    pt.mark("inline rows fill");
    // (the ids arrive in the inline rows through atomic cursors, in any order — and stay so: a receiver's inline rows are scanned whole by
    //  whichever rank owns the receiver, and the count does not depend on the order.  Round 2 sorted them, 0.13–0.8 s at scale 26, because
    //  its shards cut the lists by position.)
    // 5e. light edges (k_tc_light): count per light pivot -> scan -> fill (the same list on every rank: built with a scan)
    {
        const int64_t nl = n_work - n_heavy;
        g->n_ledge = g->ledge_total = 0;
        if (nl > 0) {
            int64_t *ecnt = nullptr, *ebeg = nullptr;
            if (int rc = dmalloc(&ecnt, nl + 1, nullptr)) return rc;
            DevGuard g_ec{ecnt};
            if (int rc = dmalloc(&ebeg, nl + 1, nullptr)) return rc;
            DevGuard g_eb{ebeg};
            hipLaunchKernelGGL(k_ledge_count, dim3(unsigned(nl / 256 + 1)), dim3(256), 0, s, n_heavy, n_work, g->order, g->toff, g->tadj, g->tsplit, g->dplus, ecnt);
            if (int rc = exclusive_scan_i64(ecnt, ebeg, nl + 1, s)) return rc;
            GMSX_HIP(hipMemcpy(&g->ledge_total, ebeg + nl, sizeof(int64_t), hipMemcpyDeviceToHost));
            const int np = g->shard_nparts > 1 ? g->shard_nparts : 1, pp = g->shard_nparts > 1 ? g->shard_part : 0;
            {
                const int64_t full = g->ledge_total / np, rem = g->ledge_total % np;  // whole stripes of np edges + this shard's share of the partial one
                g->n_ledge = np <= 1 ? g->ledge_total : full + (((full & 1) ? np - 1 - pp : pp) < rem ? 1 : 0);
            }
            if (int rc = dmalloc(&g->ledge, 2 * g->n_ledge + 2, g)) return rc;
            hipLaunchKernelGGL(k_ledge_fill, dim3(unsigned(nl / 256 + 1)), dim3(256), 0, s, n_heavy, n_work, g->order, g->hoff, g->toff, g->tadj, g->tsplit, g->dplus, ebeg,
                               np, pp, g->ledge);
            GMSX_HIP(hipStreamSynchronize(s));
        }
    }
    pt.mark("light pivot list");
    // 6. task lists: per receiver a hub-entry list and a tail-entry list, each laid out class by class.  COUNT (inline chunks, then every
    //    oriented edge of a heavy pivot at the endpoint whose row is the bigger one) -> class offsets + list offsets (scans) -> FILL.  No sort:
    //    the order inside a class is the arrival order of atomic cursors, and nothing depends on it (a multi-GPU shard is a set of pivots).
    {
        int two_sided = 1;
        if (const char *e = opt("TC_TWO_SIDED")) two_sided = std::atoi(e) != 0;  // 0 = every heavy pivot keeps all its edges (A/B knob)
        const unsigned vb = unsigned(n / 256 + 1);
        // light receivers: rank ids below inline_limit that are not heavy
        const int64_t L = g->inline_limit;
        int64_t *lflag = nullptr, *lidx = nullptr;
        if (int rc = dmalloc(&lflag, L + 1, nullptr)) return rc;
        DevGuard g_lf{lflag};
        if (int rc = dmalloc(&lidx, L + 1, nullptr)) return rc;
        DevGuard g_li{lidx};
        hipLaunchKernelGGL(k_light_flags, dim3(unsigned(L / 256 + 1)), dim3(256), 0, s, L, g->dplus, lflag);
        if (int rc = exclusive_scan_i64(lflag, lidx, L + 1, s)) return rc;
        int64_t n_light_recv = 0;
        GMSX_HIP(hipMemcpy(&n_light_recv, lidx + L, sizeof(int64_t), hipMemcpyDeviceToHost));
        const int64_t n_recv = n_heavy + n_light_recv;
        int32_t *recv_v = nullptr;
        if (int rc = dmalloc(&recv_v, n_recv + 1, nullptr)) return rc;
        DevGuard g_rv{recv_v};
        if (std::max<int64_t>(n_heavy, L) > 0)
            hipLaunchKernelGGL(k_recv_vertices, dim3(unsigned(std::max<int64_t>(n_heavy, L) / 256 + 1)), dim3(256), 0, s, n_heavy, L, g->order, g->dplus, lidx, recv_v);
        // hot windows of the hub lists (device_graph.hpp): GMSX_TC_HOT_WINDOWS x GMSX_TC_HOT_KB of the pool's front, GMSX_TC_HOT_MIN entries
        {
            auto env_int = [](const char *name, int dflt) { return int(opt_int(name, dflt)); };
            g->tc_hot_windows = std::max(0, std::min(kMaxHotWindows, env_int("TC_HOT_WINDOWS", kDefaultHotWindows)));
            g->tc_window_units = uint32_t(std::max(1, env_int("TC_HOT_KB", kDefaultHotKB))) * 64u;
            g->tc_hot_min = std::max(1, env_int("TC_HOT_MIN", kDefaultHotMin));
        }
        const TcClasses cc{g->tc_hot_windows, g->tc_window_units, g->tc_hot_min};
        const int kClasses = cc.count();
        uint32_t *cnt = nullptr, *cur = nullptr;
        if (int rc = dmalloc(&cnt, n_recv * kClasses + 1, nullptr)) return rc;
        DevGuard g_cnt{cnt};
        if (int rc = dmalloc(&cur, n_recv * kClasses + 1, nullptr)) return rc;
        DevGuard g_cur{cur};
        GMSX_HIP(hipMemsetAsync(cnt, 0, size_t(n_recv * kClasses + 1) * sizeof(uint32_t), s));
        GMSX_HIP(hipMemsetAsync(cur, 0, size_t(n_recv * kClasses + 1) * sizeof(uint32_t), s));
        if (int rc = dmalloc(&g->tunits, n_heavy + 1, g)) return rc;  // only heavy pivots keep or receive entries of edges: by position in `order`
        GMSX_HIP(hipMemsetAsync(g->tunits, 0, size_t(n_heavy + 1) * sizeof(int32_t), s));
        unsigned long long *totals = nullptr;  // [0] hub inline entries [1] tail inline entries [2] reversed edges [3] rows too long for a task entry
        if (int rc = dmalloc(&totals, 4, nullptr)) return rc;
        DevGuard g_tot{totals};
        GMSX_HIP(hipMemsetAsync(totals, 0, 4 * sizeof(unsigned long long), s));
        int64_t *hcnt = nullptr, *tcnt = nullptr, *hbeg = nullptr, *tbeg = nullptr;
        if (int rc = dmalloc(&hcnt, n_recv + 1, nullptr)) return rc;
        DevGuard g_hc{hcnt};
        if (int rc = dmalloc(&tcnt, n_recv + 1, nullptr)) return rc;
        DevGuard g_tc{tcnt};
        if (int rc = dmalloc(&hbeg, n_recv + 1, nullptr)) return rc;
        DevGuard g_hb{hbeg};
        if (int rc = dmalloc(&tbeg, n_recv + 1, nullptr)) return rc;
        DevGuard g_tb{tbeg};
        const int grid = grid_for_waves(n_heavy);
        // a task entry has 32 bits for the first unit of its row and 14 / 15 for the row's units (TaskList)
        if (g->spool_units + kPoolSlack >= (int64_t(1) << 32) || g->tpool_units + kPoolSlack >= (int64_t(1) << 32)) return GMSX_ERR_DEVICE_MEM;
        {
            uint32_t hub_max = kTaskHubUnitsMax, tail_max = kTaskTailUnitsMax;
            if (const char *e = opt("TC_TEST_MAX_UNITS")) hub_max = tail_max = uint32_t(std::max(1, std::atoi(e)));  // test hook: pretend the fields are this narrow
            if (n > 0) hipLaunchKernelGGL(k_task_limits, dim3(vb), dim3(256), 0, s, n, g->srow, g->srow2, g->trow, hub_max, tail_max, totals + 3);
        }
        // COUNT
        if (n > 0)
            hipLaunchKernelGGL(k_inline_entries<false>, dim3(vb), dim3(256), 0, s, n, g->dplus, opos, lidx, n_heavy, g->inline_limit, ihoff, itoff, inline_h_base,
                               inline_t_base, cnt, cur, hbeg, tbeg, TaskList{}, TaskList{}, totals, cc);
        if (n_heavy > 0)
            hipLaunchKernelGGL(k_task_lists<false>, dim3(grid), dim3(256), 0, s, n_heavy, g->order, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, g->srow, g->srow2, g->ksplit, g->trow,
                               two_sided, opos, cnt, cur, hbeg, tbeg, TaskList{}, TaskList{}, g->tunits,
                               totals + 2, g->spool, g->tpool, g->inline_limit, g->inline_first, g->shard_nparts, g->shard_part, cc);
        pt.mark("task lists count");
        // class offsets, list offsets
        hipLaunchKernelGGL(k_list_sizes, dim3(unsigned(n_recv / 256 + 1)), dim3(256), 0, s, n_recv, cnt, hcnt, tcnt, cc);
        if (int rc = exclusive_scan_i64(hcnt, hbeg, n_recv + 1, s)) return rc;
        if (int rc = exclusive_scan_i64(tcnt, tbeg, n_recv + 1, s)) return rc;
        unsigned long long tot[4] = {0, 0, 0, 0};
        GMSX_HIP(hipMemcpy(&g->htask_entries, hbeg + n_recv, sizeof(int64_t), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(&g->ttask_entries, tbeg + n_recv, sizeof(int64_t), hipMemcpyDeviceToHost));
        GMSX_HIP(hipMemcpy(tot, totals, sizeof(tot), hipMemcpyDeviceToHost));
        if (tot[3]) return GMSX_ERR_UNSUPPORTED;  // a row of > 16 383 hub / 32 767 tail units (no graph that fits the device has one): not a matter of memory — building a share of the pivots at a time would not help
        g->inline_hentries = int64_t(tot[0]);
        g->inline_tentries = int64_t(tot[1]);
        g->task_reverse = int64_t(tot[2]);
        if (g->htask_entries >= (int64_t(1) << 40) || g->ttask_entries >= (int64_t(1) << 40)) return GMSX_ERR_DEVICE_MEM;  // 40 position bits in a work item
        if (int rc = dmalloc(&g->htask.lo, g->htask_entries + 2, g)) return rc;
        if (int rc = dmalloc(&g->htask.hi, g->htask_entries + 2, g)) return rc;
        if (int rc = dmalloc(&g->ttask.lo, g->ttask_entries + 2, g)) return rc;
        if (int rc = dmalloc(&g->ttask.hi, g->ttask_entries + 2, g)) return rc;
        // FILL
        if (n > 0)
            hipLaunchKernelGGL(k_inline_entries<true>, dim3(vb), dim3(256), 0, s, n, g->dplus, opos, lidx, n_heavy, g->inline_limit, ihoff, itoff, inline_h_base,
                               inline_t_base, cnt, cur, hbeg, tbeg, g->htask, g->ttask, totals, cc);
        if (n_heavy > 0)
            hipLaunchKernelGGL(k_task_lists<true>, dim3(grid), dim3(256), 0, s, n_heavy, g->order, g->hoff, g->hadj, g->toff, g->tadj, g->dplus, g->srow, g->srow2, g->ksplit, g->trow,
                               two_sided, opos, cnt, cur, hbeg, tbeg, g->htask, g->ttask, g->tunits, totals + 2, g->spool, g->tpool, g->inline_limit, g->inline_first, g->shard_nparts, g->shard_part, cc);
        pt.mark("task lists fill");
        // work items
        const int hub_phases = cc.hub_phases();
        const int64_t slots = int64_t(hub_phases) * n_recv;
        int64_t *icnt = nullptr, *ioff = nullptr;
        if (int rc = dmalloc(&icnt, slots + 1, nullptr)) return rc;
        DevGuard g_ic{icnt};
        if (int rc = dmalloc(&ioff, slots + 1, nullptr)) return rc;
        DevGuard g_io{ioff};
        const unsigned rb = unsigned(n_recv / 256 + 1);
        hipLaunchKernelGGL(k_item_counts, dim3(rb), dim3(256), 0, s, n_recv, hbeg, cnt, cc, hub_phases, icnt);
        if (int rc = exclusive_scan_i64(icnt, ioff, slots + 1, s)) return rc;
        GMSX_HIP(hipMemcpy(&g->hitems, ioff + slots, sizeof(int64_t), hipMemcpyDeviceToHost));
        if (int rc = dmalloc(&g->hitem, g->hitems + 1, g)) return rc;
        if (n_recv > 0) hipLaunchKernelGGL(k_item_fill, dim3(rb), dim3(256), 0, s, n_recv, recv_v, opos, hbeg, ioff, g->htask, g->hoff, 0, cnt, cc, hub_phases, g->hitem);
        hipLaunchKernelGGL(k_item_counts, dim3(rb), dim3(256), 0, s, n_recv, tbeg, cnt, cc, 1, icnt);
        if (int rc = exclusive_scan_i64(icnt, ioff, n_recv + 1, s)) return rc;
        GMSX_HIP(hipMemcpy(&g->titems, ioff + n_recv, sizeof(int64_t), hipMemcpyDeviceToHost));
        if (int rc = dmalloc(&g->titem, g->titems + 1, g)) return rc;
        if (n_recv > 0) hipLaunchKernelGGL(k_item_fill, dim3(rb), dim3(256), 0, s, n_recv, recv_v, opos, tbeg, ioff, g->ttask, g->toff, 1, cnt, cc, 1, g->titem);
        if (g->hitems >= (int64_t(1) << 31) || g->titems >= (int64_t(1) << 31)) return GMSX_ERR_DEVICE_MEM;  // 32-bit queue tickets
        GMSX_HIP(hipStreamSynchronize(s));
    }
    pt.mark("work items");
    GMSX_HIP(hipStreamSynchronize(s));
    GMSX_HIP(hipGetLastError());
    // what only the BUILD read — the tail split, the second descriptors and the split points of hybrid rows — does not stay on the device
    // (16 bytes per vertex: 1.07 GB at scale 26); build_tc_once takes tc_bytes from device_bytes afterwards, so the bookkeeping follows
    {
        auto drop = [&](auto *&p, size_t bytes) {
            if (!p) return;
            (void)hipFree(p);
            p = nullptr;
            g->device_bytes -= int64_t(bytes);
        };
        const size_t nn = size_t(n > 0 ? n : 1);
        drop(g->tsplit, nn * sizeof(int32_t));
        drop(g->srow2, nn * sizeof(unsigned long long));
        // the first descriptors too: every task entry carries its own copy, and only gmsx_tc_row_histogram's what-if estimates read them
        // afterwards (GMSX_TC_KEEP_ROWS=1 keeps them for tools/tc_row_hist.py)
        const char *keep = opt("TC_KEEP_ROWS");
        if (!(keep && std::atoi(keep) != 0)) {
            drop(g->srow, nn * sizeof(unsigned long long));
            drop(g->trow, nn * sizeof(unsigned long long));
        }
        drop(g->ksplit, size_t(n + 1) * sizeof(int32_t));
    }
    return GMSX_OK;
}

static int build_tc_once(gmsx_graph *g) {
    g->tc_base_bytes = g->device_bytes;
    g->tc_building = true;
    const int rc = build_tc_sets(g);
    g->tc_building = false;
    g->tc_bytes = g->device_bytes - g->tc_base_bytes;
    if (rc) {
        free_tc(g);
        return rc;
    }
    g->tc_ready = true;
    return GMSX_OK;
}

int ensure_tc(const gmsx_graph *cg) {
    gmsx_graph *g = const_cast<gmsx_graph *>(cg);  // handles are single-threaded; the build only adds containers, nothing a caller can observe changes
    if (g->tc_ready) return GMSX_OK;
    if (const char *e = opt("TC_MEM_LIMIT_MB")) {  // test hook: the budget of the triangle-count containers
        const long long v = std::atoll(e);
        g->tc_limit_bytes = v > 0 ? int64_t(v) << 20 : 0;
    }
    int rc = build_tc_once(g);
    if (rc == GMSX_ERR_DEVICE_MEM && g->shard_nparts == 1) {
        // the containers of all pivots do not fit: halve the share of the pivots that is resident until it does (the base layout and the
        // stream rows stay; inline rows and task lists shrink with the share).  Calls then walk the passes (tc.hip).
        (void)hipGetLastError();
        for (int k = 2; k <= 4096 && rc == GMSX_ERR_DEVICE_MEM; k *= 2) {
            g->shard_part = 0;
            g->shard_nparts = k;
            rc = build_tc_once(g);
            if (rc == GMSX_OK) g->tc_passes = k;
        }
        if (rc) {
            g->shard_part = 0;
            g->shard_nparts = 1;
        }
    }
    return rc;
}

__global__ void k_shard_flags(int64_t n_items, const gmsx_tc_item *__restrict__ items, int nparts, int part, int64_t *__restrict__ flags) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i <= n_items) flags[i] = (i < n_items && shard_of(items[i].pos, nparts) == part) ? 1 : 0;
}
__global__ void k_shard_scatter(int64_t n_items, const int64_t *__restrict__ flags, const int64_t *__restrict__ slot, const gmsx_tc_item *__restrict__ items,
                                gmsx_tc_item *__restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_items && flags[i]) out[slot[i]] = items[i];
}
int tc_shard_items(const gmsx_graph *g, int part, int nparts) {
    if (g->shard_idx_part == part && g->shard_idx_nparts == nparts) return GMSX_OK;
    hipStream_t s = ctx().stream;
    (void)hipFree(g->shard_hitem);
    (void)hipFree(g->shard_titem);
    g->shard_hitem = g->shard_titem = nullptr;
    g->shard_idx_part = g->shard_idx_nparts = -1;
    auto one = [&](const gmsx_tc_item *items, int64_t n_items, gmsx_tc_item **idx, int64_t *count) -> int {
        *count = 0;
        if (n_items == 0) return GMSX_OK;
        int64_t *flags = nullptr, *slot = nullptr;
        if (int rc = dmalloc(&flags, n_items + 1, nullptr)) return rc;
        DevGuard g_f{flags};
        if (int rc = dmalloc(&slot, n_items + 1, nullptr)) return rc;
        DevGuard g_s{slot};
        hipLaunchKernelGGL(k_shard_flags, dim3(unsigned(n_items / 256 + 1)), dim3(256), 0, s, n_items, items, nparts, part, flags);
        if (int rc = exclusive_scan_i64(flags, slot, n_items + 1, s)) return rc;
        GMSX_HIP(hipMemcpy(count, slot + n_items, sizeof(int64_t), hipMemcpyDeviceToHost));
        if (int rc = dmalloc(idx, *count + 1, nullptr)) return rc;
        hipLaunchKernelGGL(k_shard_scatter, dim3(unsigned(n_items / 256 + 1)), dim3(256), 0, s, n_items, flags, slot, items, *idx);
        GMSX_HIP(hipStreamSynchronize(s));
        return GMSX_OK;
    };
    if (int rc = one(g->hitem, g->hitems, &g->shard_hitem, &g->shard_hitems)) return rc;
    if (int rc = one(g->titem, g->titems, &g->shard_titem, &g->shard_titems)) return rc;
    g->shard_idx_part = part;
    g->shard_idx_nparts = nparts;
    return GMSX_OK;
}

int ensure_tc_shard(const gmsx_graph *cg, int part, int nparts) {
    gmsx_graph *g = const_cast<gmsx_graph *>(cg);
    if (g->tc_ready && g->shard_part == part && g->shard_nparts == nparts) return GMSX_OK;
    free_tc(g);
    g->shard_part = part;
    g->shard_nparts = nparts;
    return build_tc_once(g);
}

int count_dplus_ge(const gmsx_graph *g, int32_t threshold, int64_t *out) {
    if (g->n == 0) {
        *out = 0;
        return GMSX_OK;
    }
    for (int i = 0; i < g->ge_used; ++i)
        if (g->ge_thr[i] == threshold) {
            *out = g->ge_cnt[i];
            return GMSX_OK;
        }
    hipStream_t s = ctx().stream;
    hipLaunchKernelGGL(k_count_ge, dim3(1), dim3(1), 0, s, g->n, g->order, g->dplus, threshold, reinterpret_cast<int64_t *>(g->scratch + 8));
    GMSX_HIP(hipMemcpyAsync(out, g->scratch + 8, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    GMSX_HIP(hipStreamSynchronize(s));
    if (g->ge_used < 40) {
        g->ge_thr[g->ge_used] = threshold;
        g->ge_cnt[g->ge_used] = *out;
        ++g->ge_used;
    }
    return GMSX_OK;
}

}  // namespace gmsx

using namespace gmsx;

extern "C" {

int gmsx_init(int device) {
    return gmsx::guard([&]() -> int {
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
        // one device per process (one process per GPU): the library's stream, events and every graph handle live on the first
        // device bound; re-binding to another one is refused rather than silently launching on a foreign stream
        if (c.device >= 0 && c.device != device) return GMSX_ERR_UNSUPPORTED;
        if (hipSetDevice(device) != hipSuccess) return GMSX_ERR_NO_DEVICE;  // also makes the device current for a new host thread
        if (c.device == device) return GMSX_OK;
        if (!c.own_stream) {
            if (hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking) != hipSuccess) return GMSX_ERR_NO_DEVICE;
            for (auto &e : c.ev)
                if (hipEventCreate(&e) != hipSuccess) return GMSX_ERR_NO_DEVICE;
            for (auto &st : c.side)
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return GMSX_ERR_NO_DEVICE;
            if (hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess) return GMSX_ERR_NO_DEVICE;
            for (auto &e : c.ev_join)
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return GMSX_ERR_NO_DEVICE;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
        c.compute_units = prop.multiProcessorCount;
        c.stream = c.own_stream;
        c.device = device;
        // What a process pays ONCE belongs here, not under the first upload's clock (round 5; GMSX_INIT_LAZY=1 restores the lazy behaviour): the
        // code object of the library is loaded by the first kernel launch, and the two pinned staging buffers of the uploads (128 MB) are allocated
        // and touched — together 60 … 90 ms of the first scale-26 upload before.
        if (const char *e = opt("INIT_LAZY"); !(e && std::atoi(e) != 0)) {
            hipLaunchKernelGGL(gmsx::k_warm, dim3(1), dim3(64), 0, c.stream, static_cast<unsigned long long *>(nullptr));
            if (!gmsx::h2d_staged_off()) (void)gmsx::h2d_stage();
            if (hipStreamSynchronize(c.stream) != hipSuccess) return GMSX_ERR_NO_DEVICE;
        }
        return GMSX_OK;
    });
}

int gmsx_set_stream(void *hip_stream) {
    return gmsx::guard([&]() -> int {
        if (int rc = ensure_init()) return rc;
        ctx().stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx().own_stream;
        return GMSX_OK;
    });
}

int gmsx_device_info(char *name, size_t name_len, int *compute_units, int64_t *hbm_bytes) {
    return gmsx::guard([&]() -> int {
        if (int rc = ensure_init()) return rc;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx().device) != hipSuccess) return GMSX_ERR_NO_DEVICE;
        if (name && name_len) {
            std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
        }
        if (compute_units) *compute_units = prop.multiProcessorCount;
        if (hbm_bytes) *hbm_bytes = int64_t(prop.totalGlobalMem);
        return GMSX_OK;
    });
}

int gmsx_graph_upload(int64_t n, const int64_t *offsets, const int32_t *neigh, uint32_t flags, gmsx_graph **out) {
    return gmsx::guard([&]() -> int {
        return gmsx_graph_upload_shard(n, offsets, neigh, flags, 0, 1, out);
    });
}

int gmsx_graph_upload_shard(int64_t n, const int64_t *offsets, const int32_t *neigh, uint32_t flags, int part, int nparts, gmsx_graph **out) {
    return gmsx::guard([&]() -> int {
        if (!out || n < 0 || !offsets || n > 0x7fffffffll || nparts < 1 || part < 0 || part >= nparts) return GMSX_ERR_INVALID;
        if (offsets[0] != 0 || offsets[n] < 0 || (offsets[n] > 0 && !neigh)) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        struct Free { void operator()(gmsx_graph *p) const { free_graph(p); } };
        std::unique_ptr<gmsx_graph, Free> owner(new (std::nothrow) gmsx_graph);  // freed on every way out, an exception in a build step included
        gmsx_graph *g = owner.get();
        if (!g) return GMSX_ERR_NOMEM;
        g->n = n;
        g->nnz = offsets[n];
        g->shard_part = part;
        g->shard_nparts = nparts;
        int rc = dmalloc(&g->off, n + 1, g);
        if (!rc) rc = dmalloc(&g->adj, g->nnz, g);
        if (!rc) {
            hipStream_t s = ctx().stream;
            PhaseTimer pt(s, "h2d");
            rc = staged_h2d(g->off, offsets, size_t(n + 1) * sizeof(int64_t), s);
            if (!rc && g->nnz) rc = staged_h2d(g->adj, neigh, size_t(g->nnz) * sizeof(int32_t), s);
            if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = GMSX_ERR_KERNEL;
            if (rc) (void)hipGetLastError();
            pt.mark("CSR host -> device");
        }
        // offsets must be monotone before any kernel walks the rows
        if (!rc) {
            for (int64_t i = 0; i < n; ++i)
                if (offsets[i + 1] < offsets[i]) { rc = GMSX_ERR_INVALID; break; }
        }
        if (!rc) rc = build_device_sets(g, flags);
        if (!rc && (flags & GMSX_UPLOAD_FOR_TC)) rc = ensure_tc(g);
        if (rc) return rc;
        *out = owner.release();
        return GMSX_OK;
    });
}

int gmsx_graph_upload_csr(const gmsx_csr *h, uint32_t flags, gmsx_graph **out) { return gmsx_graph_upload_csr_shard(h, flags, 0, 1, out); }

int gmsx_graph_upload_csr_shard(const gmsx_csr *h, uint32_t flags, int part, int nparts, gmsx_graph **out) {
    return gmsx::guard([&]() -> int {
        if (!h) return GMSX_ERR_INVALID;
        if (h->g.directed) return GMSX_ERR_DIRECTED;
        return gmsx_graph_upload_shard(h->g.n, h->g.off.get(), h->g.neigh.get(), flags, part, nparts, out);
    });
}

int gmsx_graph_prepare(gmsx_graph *g, uint32_t what) {
    return gmsx::guard([&]() -> int {
        if (!g || (what & ~uint32_t(GMSX_PREPARE_TC))) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        if (what & GMSX_PREPARE_TC)
            if (int rc = ensure_tc(g)) return rc;
        return GMSX_OK;
    });
}

int gmsx_graph_tc_passes(const gmsx_graph *g) { return !g ? GMSX_ERR_INVALID : (g->tc_ready ? g->tc_passes : 0); }

int gmsx_graph_free(gmsx_graph *g) {
    return gmsx::guard([&]() -> int {
        free_graph(g);
        return GMSX_OK;
    });
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
        case GMSX_ERR_COMM: return "librccl could not be loaded or an RCCL call failed";
        case GMSX_ERR_TIMEOUT: return "a peer of the communicator did not arrive in time (GMSX_COMM_TIMEOUT_S)";
        default: return "unknown gmsx status";
    }
}

}  // extern "C"
