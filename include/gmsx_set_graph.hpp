// gmsx_set_graph.hpp — C++17 adaptor that plugs the gmsx C-ABI (gmsx.h) under GMS's compile-time Set / SGraph
// concept, so drivers written against the reference's templates instantiate unchanged with the graph types below.
//
// What the reference requires of a Set (union of testing/sets.cpp and the three algorithm families):
//   gms/representations/sets/sorted_set.h:21-272     (SortedSetBase)     — mirrored by gmsx::SortedSpanSet
//   gms/representations/sets/sorted_set_ref.h:9-78   (SortedSetRefBase)  — mirrored by gmsx::SortedSpanRef (borrowed view)
//   gms/representations/sets/roaring_set.h:15-229    (RoaringSetBase)    — mirrored by gmsx::RoaringSpanSet (uint32 iteration)
//   gms/representations/graphs/set_graph.h:86-118    (SetGraph)          — mirrored by gmsx::HipGraphT<Set>:
//        HipSetGraph = HipGraphT<SortedSpanSet>, HipRoaringGraph = HipGraphT<RoaringSpanSet>, HipSetRefGraph = HipGraphT<SortedSpanRef>
//
// ZERO-COPY: a row of the graph is a *view* (pointer + count, 16 bytes) into ONE adjacency array — the caller's CSRGraph /
// gmsx_csr memory when its rows are contiguous and sorted (the reference does the same for SetGraph<SortedSetRef>,
// set_graph.h:162-168), otherwise one private copy.  No per-row allocation, no second copy of the adjacency.  A set turns
// into an owning one only when it is mutated or produced by an operator.
//
// The per-element operators below run on the host (they exist so that every generic template still compiles and
// runs); the whole-graph algorithms — the hot path — are forwarded to the device through the C-ABI:
//   gmsx::count_total(g)                 -> gmsx_tc_total          (triangle_count/parallel/total.h:7-24)
//   gmsx::vertex_count2(g, out)          -> gmsx_tc_vertex_count2  (triangle_count/parallel/vertex.h:14-49)
//   gmsx::clique_count(g, k)             -> gmsx_kclique_count     (k_clique_count_set_based.h:19-31)
//   gmsx::clique_star_count(g, k)        -> gmsx_kclique_star_count (k_clique_star_list/parallel/recursive.h:19-35, count mode)
//   gmsx::maximal_clique_count(g, rank)  -> gmsx_bk_count          (maximal_clique_enum/parallel/eppsteinPAR.h:18-53)
//   gmsx::adg_rank(g, eps, out)          -> gmsx_adg_rank          (preprocessing/parallel/degeneracy_approx_set.h:14-86)
//   gmsx::triangle_count_ordering(g,out) -> gmsx_tc_ordering       (preprocessing/parallel/triangle_count.h:11-30)
// include/gmsx_gms_glue.hpp holds the explicit specialisations that route the reference's own function names to these
// (INTEGRATION.md §2).  Header-only; link with -lgmsx.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

#include "gmsx.h"

namespace gmsx {

using SetElement = int32_t;  // sorted_set.h:25, roaring_set.h:28

namespace detail {
inline void check(int rc, const char *what) {
    if (rc != GMSX_OK) {  // the reference's convention for loader / CLI failures: message + exit (gapbs/reader.h:45,228)
        std::fprintf(stderr, "gmsx: %s failed: %s\n", what, gmsx_strerror(rc));
        std::exit(-33);
    }
}
// the four merges of sorted_set_operations.h:29-106 on raw sorted ranges; `out` may alias `a` for intersect/difference
inline size_t isect_count(const SetElement *a, size_t na, const SetElement *b, size_t nb) {
    size_t i = 0, j = 0, c = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i;
        else if (b[j] < a[i]) ++j;
        else { ++c; ++i; ++j; }
    }
    return c;
}
inline size_t isect(const SetElement *a, size_t na, const SetElement *b, size_t nb, SetElement *out) {
    size_t i = 0, j = 0, c = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i;
        else if (b[j] < a[i]) ++j;
        else { out[c++] = a[i]; ++i; ++j; }
    }
    return c;
}
inline size_t diff(const SetElement *a, size_t na, const SetElement *b, size_t nb, SetElement *out) {
    size_t i = 0, j = 0, c = 0;
    while (i < na) {
        if (j == nb || a[i] < b[j]) out[c++] = a[i++];
        else if (b[j] < a[i]) ++j;
        else { ++i; ++j; }
    }
    return c;
}
inline size_t unite(const SetElement *a, size_t na, const SetElement *b, size_t nb, SetElement *out) {
    return size_t(std::set_union(a, a + na, b, b + nb, out) - out);
}
}  // namespace detail

// ------------------------------------------------------------------------------------------------------------------
// SpanSetT: sorted duplicate-free set of int32 that is either a borrowed VIEW of caller memory (cap_ == 0) or an owning
// buffer.  16 bytes.  Move-only like the reference's sets (copying is explicit via clone(), sorted_set.h:31-39,84-87).
// IterElem is what iteration yields: int32_t for the SortedSet flavour, uint32_t for the RoaringSet flavour
// (roaring_set.h:82-105 hands out Roaring's uint32 iterators).
// ------------------------------------------------------------------------------------------------------------------
template <class IterElem>
class SpanSetT {
   public:
    using SetElement = gmsx::SetElement;
    static_assert(sizeof(IterElem) == sizeof(gmsx::SetElement), "iteration element must be 32 bits wide");
    using Container = std::vector<SetElement>;
    using const_iterator = const IterElem *;

    SpanSetT() = default;
    SpanSetT(SpanSetT &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.forget(); }
    SpanSetT &operator=(SpanSetT &&o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_;
            o.forget();
        }
        return *this;
    }
    SpanSetT(const SpanSetT &) = delete;
    SpanSetT &operator=(const SpanSetT &) = delete;
    ~SpanSetT() { release(); }

    // copies; the input need not be sorted (sorted_set.h:64-66, roaring_set.h:49-54)
    SpanSetT(const SetElement *start, size_t count) { assign_unsorted(start, count); }
    explicit SpanSetT(const Container &c) { assign_unsorted(c.data(), c.size()); }
    explicit SpanSetT(Container &&c, bool sorted = false) {
        if (sorted) assign_sorted(c.data(), c.size()); else assign_unsorted(c.data(), c.size());
    }
    SpanSetT(std::initializer_list<SetElement> il) { assign_unsorted(il.begin(), il.size()); }
    explicit SpanSetT(SetElement single) { assign_sorted(&single, 1); }

    // zero-copy view of a sorted, duplicate-free range that outlives the set (a CSR row): sorted_set_ref.h:17-20 without the
    // in-place sort — canonical rows are sorted already
    static SpanSetT borrow(const SetElement *sorted_unique, size_t count) {
        SpanSetT s;
        s.p_ = const_cast<SetElement *>(sorted_unique);
        s.n_ = uint32_t(count);
        return s;
    }
    bool borrowed() const { return cap_ == 0 && n_ != 0; }

    SpanSetT clone() const { SpanSetT s; s.assign_sorted(p_, n_); return s; }
    static SpanSetT Range(unsigned bound) {  // sorted_set.h:257-262, roaring_set.h:219-224
        SpanSetT s;
        s.reserve(bound);
        for (unsigned i = 0; i < bound; ++i) s.p_[i] = SetElement(i);
        s.n_ = bound;
        return s;
    }

    size_t cardinality() const { return n_; }
    const_iterator begin() const { return reinterpret_cast<const IterElem *>(p_); }
    const_iterator end() const { return reinterpret_cast<const IterElem *>(p_) + n_; }
    const SetElement *data() const { return p_; }
    bool contains(SetElement e) const { return std::binary_search(p_, p_ + n_, e); }
    template <class T>
    void toArray(T *out) const {  // sorted_set.h:234-239 (memcpy), roaring_set.h:190-201 (toUint32Array)
        static_assert(sizeof(T) == sizeof(SetElement), "toArray needs a 32-bit element type");
        if (n_) std::memcpy(out, p_, size_t(n_) * sizeof(SetElement));
    }
    bool operator==(const SpanSetT &o) const { return n_ == o.n_ && (n_ == 0 || std::memcmp(p_, o.p_, size_t(n_) * sizeof(SetElement)) == 0); }
    bool operator!=(const SpanSetT &o) const { return !(*this == o); }

    // ---- intersection (sorted_set.h:160-182, roaring_set.h:134-152) ----
    template <class Other>
    size_t intersect_count(const Other &o) const { return detail::isect_count(p_, n_, o.data(), o.cardinality()); }
    template <class Other>
    SpanSetT intersect(const Other &o) const {
        SpanSetT r;
        r.reserve(std::min<size_t>(n_, o.cardinality()));
        r.n_ = uint32_t(detail::isect(p_, n_, o.data(), o.cardinality(), r.p_));
        return r;
    }
    void intersect_inplace(const SpanSetT &o) {
        if (cap_) n_ = uint32_t(detail::isect(p_, n_, o.p_, o.n_, p_));  // the write index never passes the read index
        else *this = intersect(o);
    }

    // ---- union (sorted_set.h:104-158, roaring_set.h:107-132) ----
    template <class Other>
    SpanSetT union_with(const Other &o) const {
        SpanSetT r;
        r.reserve(size_t(n_) + o.cardinality());
        r.n_ = uint32_t(detail::unite(p_, n_, o.data(), o.cardinality(), r.p_));
        return r;
    }
    SpanSetT union_with(SetElement e) const { SpanSetT r = clone(); r.union_inplace(e); return r; }
    void union_inplace(const SpanSetT &o) { *this = union_with(o); }
    void union_inplace(SetElement e) {
        const size_t pos = size_t(std::lower_bound(p_, p_ + n_, e) - p_);
        if (pos < n_ && p_[pos] == e) return;
        if (cap_ == 0 || n_ == cap_) grow(std::max<size_t>(4, size_t(n_) * 2));
        std::memmove(p_ + pos + 1, p_ + pos, (size_t(n_) - pos) * sizeof(SetElement));
        p_[pos] = e;
        ++n_;
    }
    template <class Other>
    size_t union_count(const Other &o) const { return size_t(n_) + o.cardinality() - intersect_count(o); }

    // ---- difference (sorted_set.h:184-216, roaring_set.h:154-178) ----
    template <class Other>
    SpanSetT difference(const Other &o) const {
        SpanSetT r;
        r.reserve(n_);
        r.n_ = uint32_t(detail::diff(p_, n_, o.data(), o.cardinality(), r.p_));
        return r;
    }
    SpanSetT difference(SetElement e) const { SpanSetT r = clone(); r.difference_inplace(e); return r; }
    void difference_inplace(const SpanSetT &o) {
        if (cap_) n_ = uint32_t(detail::diff(p_, n_, o.p_, o.n_, p_));
        else *this = difference(o);
    }
    void difference_inplace(SetElement e) {
        const size_t pos = size_t(std::lower_bound(p_, p_ + n_, e) - p_);
        if (pos == n_ || p_[pos] != e) return;  // removing an absent element is a no-op (sets.cpp:344-357,404-408)
        if (cap_ == 0) grow(n_);
        std::memmove(p_ + pos, p_ + pos + 1, (size_t(n_) - pos - 1) * sizeof(SetElement));
        --n_;
    }
    void add(SetElement e) { union_inplace(e); }
    void remove(SetElement e) { difference_inplace(e); }

   private:
    void forget() { p_ = nullptr; n_ = cap_ = 0; }
    void release() { if (cap_) delete[] p_; forget(); }
    void reserve(size_t cap) {  // fresh owning buffer (at least one slot so that owning == cap_ > 0)
        release();
        cap_ = uint32_t(std::max<size_t>(cap, 1));
        p_ = new SetElement[cap_];
    }
    void grow(size_t cap) {  // owning buffer of >= cap that keeps the current contents
        cap = std::max<size_t>(std::max<size_t>(cap, n_), 1);
        SetElement *q = new SetElement[cap];
        if (n_) std::memcpy(q, p_, size_t(n_) * sizeof(SetElement));
        if (cap_) delete[] p_;
        p_ = q;
        cap_ = uint32_t(cap);
    }
    void assign_sorted(const SetElement *s, size_t count) {
        reserve(count);
        if (count) std::memcpy(p_, s, count * sizeof(SetElement));
        n_ = uint32_t(count);
    }
    void assign_unsorted(const SetElement *s, size_t count) {
        assign_sorted(s, count);
        std::sort(p_, p_ + n_);
        n_ = uint32_t(std::unique(p_, p_ + n_) - p_);
    }
    SetElement *p_ = nullptr;
    uint32_t n_ = 0, cap_ = 0;  // cap_ == 0: borrowed view (or empty)
};

using SortedSpanSet = SpanSetT<int32_t>;    // the SortedSet flavour   (sorted_set.h:274-276)
using RoaringSpanSet = SpanSetT<uint32_t>;  // the RoaringSet flavour  (roaring_set.h:227-229): same contents, uint32 iteration;
                                            // the Roaring containers themselves live on the device (device_graph.hpp)

// Borrowed, trivially copyable view of a sorted row whose operators return owning SortedSpanSets: the surface of
// SortedSetRefBase (sorted_set_ref.h:9-78).  Unlike the reference it neither sorts the caller's memory in its constructor
// (:17-20; rows handed to it are canonical) nor inherits the missing equality test of its contains() (:70-73).
class SortedSpanRef {
   public:
    using SetElement = gmsx::SetElement;
    using Container = std::vector<SetElement>;
    SortedSpanRef() = default;
    SortedSpanRef(const SetElement *start, size_t count) : p_(start), n_(count) {}
    static SortedSpanRef borrow(const SetElement *start, size_t count) { return SortedSpanRef(start, count); }
    size_t cardinality() const { return n_; }
    const SetElement *begin() const { return p_; }
    const SetElement *end() const { return p_ + n_; }
    const SetElement *data() const { return p_; }
    template <class Set> SortedSpanSet union_with(const Set &s) const { return as_set().union_with(s); }
    template <class Set> SortedSpanSet intersect(const Set &s) const { return as_set().intersect(s); }
    template <class Set> size_t intersect_count(const Set &s) const { return detail::isect_count(p_, n_, s.data(), s.cardinality()); }
    template <class Set> SortedSpanSet difference(const Set &s) const { return as_set().difference(s); }
    bool contains(SetElement x) const { return std::binary_search(p_, p_ + n_, x); }

   private:
    SortedSpanSet as_set() const { return SortedSpanSet::borrow(p_, n_); }
    const SetElement *p_ = nullptr;
    size_t n_ = 0;
};

// ------------------------------------------------------------------------------------------------------------------
// SGraph over a device-resident graph (SetGraph<Set>, set_graph.h:10-233).  One host adjacency (borrowed or private),
// n 16-byte row views, and the device handle.  FromCGraph uploads eagerly, so the H2D copy and the device-side container
// build are part of what the reference harness times as "GraphExec buildTime" (common/benchmark.h:105-109); on a host
// without a HIP device the upload is skipped and the first whole-graph call fails loudly instead — the generic host
// templates keep working.
// ------------------------------------------------------------------------------------------------------------------
// Upload flags FromCGraph / FromCsr / clone use when none are given.  GMSX_UPLOAD_DEFAULT builds the degree-oriented containers and
// bitsets only (what k-clique and Bron–Kerbosch need) and leaves the triangle-count task lists to the first count_total call; a
// triangle-count driver sets GMSX_UPLOAD_FOR_TC — one line in main(), or -DGMSX_ADAPTOR_UPLOAD_FLAGS=GMSX_UPLOAD_FOR_TC — so that their
// build lands in the harness's untimed "GraphExec buildTime" (common/benchmark.h:105-109) like the reference's SetGraph construction.
#ifndef GMSX_ADAPTOR_UPLOAD_FLAGS
#define GMSX_ADAPTOR_UPLOAD_FLAGS GMSX_UPLOAD_DEFAULT
#endif
inline uint32_t &default_upload_flags() {
    static uint32_t flags = uint32_t(GMSX_ADAPTOR_UPLOAD_FLAGS);
    return flags;
}
// (part, nparts) of the uploads of THIS process: a rank of a multi-GPU run sets it once (before the first FromCGraph / FromCsr), so that
// only its own shard's triangle-count containers are built (gmsx_graph_upload_shard); {0, 1} = the whole graph
inline std::pair<int, int> &default_upload_shard() {
    static std::pair<int, int> shard{0, 1};
    return shard;
}

template <class SetT>
class HipGraphT {
   public:
    using Set = SetT;

    HipGraphT() = default;
    HipGraphT(HipGraphT &&o) noexcept { *this = std::move(o); }
    HipGraphT &operator=(HipGraphT &&o) noexcept {
        if (this != &o) {
            release();
            off_own_ = std::move(o.off_own_);
            adj_own_ = std::move(o.adj_own_);
            sets_ = std::move(o.sets_);
            off_ = o.off_; adj_ = o.adj_; n_ = o.n_;
            dev_ = o.dev_; upload_rc_ = o.upload_rc_; flags_ = o.flags_; shard_ = o.shard_;
            o.dev_ = nullptr; o.off_ = nullptr; o.adj_ = nullptr; o.n_ = 0; o.shard_ = {0, 1};
        }
        return *this;
    }
    HipGraphT(const HipGraphT &) = delete;
    HipGraphT &operator=(const HipGraphT &) = delete;
    ~HipGraphT() { release(); }

    // SetGraph::FromCGraph (set_graph.h:86-89,152-181): CGraph needs num_nodes(), out_degree(u) and an iterable out_neigh(u).
    // When out_neigh(u) hands out pointers into one contiguous, row-sorted int32 array (gapbs CSRGraph, graph.h:361-364) the
    // adjacency is BORROWED — the CGraph must outlive this object, as it does in the reference harness (set_graph.h:162-168
    // makes the same assumption for SortedSetRef) — otherwise it is copied once and sorted.
    template <class CGraph>
    static HipGraphT FromCGraph(const CGraph &g) { return FromCGraph(g, default_upload_flags()); }
    template <class CGraph>
    static HipGraphT FromCGraph(const CGraph &g, uint32_t upload_flags) {
        HipGraphT r;
        r.flags_ = upload_flags;
        const int64_t n = g.num_nodes();
        r.n_ = n;
        r.off_own_.resize(size_t(n) + 1);
        r.off_own_[0] = 0;
        for (int64_t u = 0; u < n; ++u) r.off_own_[size_t(u) + 1] = r.off_own_[size_t(u)] + int64_t(g.out_degree(u));
        r.off_ = r.off_own_.data();
        const int64_t nnz = r.off_[n];
        const SetElement *base = nullptr;
        if (n > 0 && nnz > 0) base = contiguous_base(g, r.off_, n);
        if (base) {
            r.adj_ = base;
        } else {
            r.adj_own_.resize(size_t(nnz));
            for (int64_t u = 0; u < n; ++u) {
                int64_t k = r.off_[u];
                for (auto v : g.out_neigh(u)) r.adj_own_[size_t(k++)] = SetElement(v);
                std::sort(r.adj_own_.begin() + r.off_[u], r.adj_own_.begin() + r.off_[u + 1]);
            }
            r.adj_ = r.adj_own_.data();
        }
        r.make_views();
        r.try_upload();
        return r;
    }
    // borrows the arrays of a gmsx_csr (which must outlive the graph)
    static HipGraphT FromCsr(const gmsx_csr *c, uint32_t upload_flags = default_upload_flags()) {
        HipGraphT r;
        r.flags_ = upload_flags;
        r.n_ = gmsx_csr_num_nodes(c);
        r.off_ = gmsx_csr_offsets(c);
        r.adj_ = gmsx_csr_neighbors(c);
        r.make_views();
        r.try_upload();
        return r;
    }
    HipGraphT clone() const {  // set_graph.h:120-127: an independent deep copy
        HipGraphT r;
        r.flags_ = flags_;
        r.n_ = n_;
        r.off_own_.assign(off_, off_ + n_ + 1);
        r.adj_own_.assign(adj_, adj_ + off_[n_]);
        r.off_ = r.off_own_.data();
        r.adj_ = r.adj_own_.data();
        r.make_views();
        r.try_upload();
        return r;
    }

    int64_t num_nodes() const { return n_; }                                              // set_graph.h:115-118
    const Set &out_neigh(SetElement v) const { return sets_[size_t(v)]; }                 // set_graph.h:102-105
    int64_t out_degree(SetElement v) const { return off_[size_t(v) + 1] - off_[size_t(v)]; }  // set_graph.h:91-94
    bool borrows_adjacency() const { return adj_own_.empty() && n_ > 0 && off_[n_] > 0; }
    const int64_t *offsets() const { return off_; }
    const SetElement *neighbors() const { return adj_; }

    std::pair<int, int> upload_shard() const { return shard_; }  // (part, nparts) of the device copy ({0, 1} = the whole graph)
    // device handle; failures follow the reference's convention: message + exit (gapbs/reader.h:45,228; cli/cli.h:159-171) —
    // nothing is thrown across the C-ABI
    const gmsx_graph *device() const {
        if (!dev_) {
            if (upload_rc_ == GMSX_OK || upload_rc_ == GMSX_ERR_NO_DEVICE) const_cast<HipGraphT *>(this)->try_upload();  // e.g. device bound later
            if (!dev_) {
                std::fprintf(stderr, "gmsx: graph upload failed: %s\n", gmsx_strerror(upload_rc_));
                std::exit(-32);
            }
        }
        return dev_;
    }

   private:
    template <class CGraph>
    static const SetElement *contiguous_base(const CGraph &g, const int64_t *off, int64_t n) {
        using It = decltype(g.out_neigh(0).begin());
        if constexpr (std::is_pointer_v<It> && sizeof(std::remove_pointer_t<It>) == sizeof(SetElement) &&
                      std::is_integral_v<std::remove_cv_t<std::remove_pointer_t<It>>>) {
            const SetElement *base = reinterpret_cast<const SetElement *>(g.out_neigh(0).begin());
            int ok = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 4096) reduction(& : ok)
#endif
            for (int64_t u = 0; u < n; ++u) {
                const SetElement *row = reinterpret_cast<const SetElement *>(g.out_neigh(u).begin());
                if (row != base + off[u]) ok = 0;
                else
                    for (int64_t j = off[u] + 1; j < off[u + 1]; ++j) ok &= int(base[j - 1] < base[j]);
            }
            return ok ? base : nullptr;
        } else {
            (void)g; (void)off; (void)n;
            return nullptr;
        }
    }
    void make_views() {
        sets_.clear();
        sets_.reserve(size_t(n_));
        for (int64_t u = 0; u < n_; ++u) sets_.push_back(Set::borrow(adj_ + off_[u], size_t(off_[u + 1] - off_[u])));
    }
    void try_upload() {
        release_device();
        shard_ = default_upload_shard();
        upload_rc_ = gmsx_graph_upload_shard(n_, off_, adj_, flags_, shard_.first, shard_.second, &dev_);
        if (upload_rc_ != GMSX_OK) dev_ = nullptr;
    }
    void release_device() {
        if (dev_) gmsx_graph_free(dev_);
        dev_ = nullptr;
    }
    void release() { release_device(); }

    std::vector<int64_t> off_own_;
    std::vector<SetElement> adj_own_;
    std::vector<Set> sets_;
    const int64_t *off_ = nullptr;
    const SetElement *adj_ = nullptr;
    int64_t n_ = 0;
    gmsx_graph *dev_ = nullptr;
    int upload_rc_ = GMSX_OK;
    uint32_t flags_ = GMSX_UPLOAD_DEFAULT;
    std::pair<int, int> shard_{0, 1};  // (part, nparts) the device copy was uploaded with
};

using HipSetGraph = HipGraphT<SortedSpanSet>;       // the SortedSetGraph flavour       (set_graph.h:235)
using HipRoaringGraph = HipGraphT<RoaringSpanSet>;  // the RoaringGraph flavour         (set_graph.h:236)
using HipSetRefGraph = HipGraphT<SortedSpanRef>;    // the SetGraph<SortedSetRef> one   (k_clique_count_set_based.cc:38)

// ---- whole-graph algorithms: the hot path, forwarded to the device ---------------------------------------------------

// GMS::TriangleCount::Par::count_total / Seq::count_total (triangle_count/parallel/total.h:7-24, sequential/total.h:7-23)
// A graph uploaded as shard (part, nparts) of a multi-GPU run (default_upload_shard) holds the triangle-count containers of that shard
// only: count_total then returns the shard's PARTIAL count (gmsx_tc_partial) — the rank's term of the reference's reduction(+:total),
// which the caller sums over the ranks (gmsx_comm_allreduce_u64), as gmsx_driver --gpus does.
template <class S>
inline size_t count_total(const HipGraphT<S> &g) {
    uint64_t t = 0;
    if (g.upload_shard().second > 1)
        detail::check(gmsx_tc_partial(g.device(), GMSX_TC_AUTO, g.upload_shard().first, g.upload_shard().second, &t, nullptr), "gmsx_tc_partial");
    else
        detail::check(gmsx_tc_total(g.device(), GMSX_TC_AUTO, &t, nullptr), "gmsx_tc_total");
    return size_t(t);
}
// GMS::TriangleCount::Par::vertex_count2 / vertex_count2_once / Seq::vertex_count2 (parallel/vertex.h:14-49)
template <class S, class Output = std::vector<int64_t>>
inline void vertex_count2(const HipGraphT<S> &g, Output &counts) {
    counts.resize(size_t(g.num_nodes()));
    static_assert(sizeof(*counts.data()) == sizeof(int64_t), "vertex counts are 64-bit");
    detail::check(gmsx_tc_vertex_count2(g.device(), reinterpret_cast<int64_t *>(counts.data()), nullptr), "gmsx_tc_vertex_count2");
}
// CliqueCount<…, SGraph, …> (k_clique_count_set_based.h:19-31): returns k! * C_k like the reference and prints its line (:29)
template <class S>
inline size_t clique_count(const HipGraphT<S> &g, size_t k = 4) {
    uint64_t ordered = 0;
    detail::check(gmsx_kclique_count(g.device(), int(k), &ordered, nullptr, nullptr), "gmsx_kclique_count");
    std::printf("total %zu-cliques: %llu\n", k, static_cast<unsigned long long>(ordered));
    return size_t(ordered);
}
// KCliqueStar::Par::CliqueStar<SGraph, OutputMode::Count> (k_clique_star_list/parallel/recursive.h:19-35): the number of k-clique-stars —
// what `output.size()` is in count mode — printed like the reference does (:33); *star_members (optional) = the total size of the stars a
// listing would carry.  The listing itself (`CliqueStarList`) stays on the host: the generic template runs over the span sets.
template <class S>
inline int64_t clique_star_count(const HipGraphT<S> &g, int32_t k, uint64_t *star_members = nullptr) {
    uint64_t stars = 0;
    detail::check(gmsx_kclique_star_count(g.device(), int(k), &stars, star_members, nullptr), "gmsx_kclique_star_count");
    std::printf("total %d-cliques: %llu\n", int(k), static_cast<unsigned long long>(stars));
    return int64_t(stars);
}
// BkEppsteinPar::mceBench under -DBK_COUNT (eppsteinPAR.h:18-53): the maximal-clique count.  `rank` (rank format, any
// random-access container of n integers) is validated to be a permutation by the library; the count does not depend on it.
template <class S, class Ranking>
inline size_t maximal_clique_count(const HipGraphT<S> &g, const Ranking &rank) {
    uint64_t c = 0;
    const auto *r = rank.data();
    if constexpr (sizeof(*r) == sizeof(int32_t) && std::is_integral_v<std::remove_cv_t<std::remove_reference_t<decltype(*r)>>>) {
        detail::check(gmsx_bk_count(g.device(), reinterpret_cast<const int32_t *>(r), &c, nullptr), "gmsx_bk_count");
    } else {
        std::unique_ptr<int32_t[]> tmp(new int32_t[size_t(g.num_nodes())]);
        for (int64_t i = 0; i < g.num_nodes(); ++i) tmp[size_t(i)] = int32_t(rank[size_t(i)]);
        detail::check(gmsx_bk_count(g.device(), tmp.get(), &c, nullptr), "gmsx_bk_count");
    }
    return size_t(c);
}
template <class S>
inline size_t maximal_clique_count(const HipGraphT<S> &g) {
    uint64_t c = 0;
    detail::check(gmsx_bk_count(g.device(), nullptr, &c, nullptr), "gmsx_bk_count");
    return size_t(c);
}
// PpParallel::getDegeneracyOrderingApproxSGraph<averageDegree, useRankFormat> (degeneracy_approx_set.h:14-86)
template <class S, class Output>
inline void adg_rank(const HipGraphT<S> &g, double epsilon, Output &res, bool rank_format = true) {
    res.resize(size_t(g.num_nodes()));
    static_assert(sizeof(*res.data()) == sizeof(int32_t), "orderings are NodeId = int32 vectors");
    detail::check(gmsx_adg_rank(g.device(), epsilon, rank_format ? 1 : 0, reinterpret_cast<int32_t *>(res.data()), nullptr, nullptr), "gmsx_adg_rank");
}
// PpParallel::triangleCountOrdering (preprocessing/parallel/triangle_count.h:11-30): vertices by increasing per-vertex count
template <class S, class Output>
inline void triangle_count_ordering(const HipGraphT<S> &g, Output &ordering) {
    ordering.resize(size_t(g.num_nodes()));
    static_assert(sizeof(*ordering.data()) == sizeof(int32_t), "orderings are NodeId = int32 vectors");
    detail::check(gmsx_tc_ordering(g.device(), reinterpret_cast<int32_t *>(ordering.data()), nullptr), "gmsx_tc_ordering");
}

// Set::intersect / Set::difference of whole neighbourhoods for a BATCH of vertex pairs on the device (sorted_set.h:160-197): result i =
// out_neigh(u[i]) ∩ out_neigh(v[i]) (difference = false) or out_neigh(u[i]) \ out_neigh(v[i]), as sets of the graph's own flavour — what a
// listing consumer (tomita.h:51-70 with `sol`, k_clique_star_list/parallel/output.h:14-68) hands on.  One sizing call, one fill.
template <class S, class Ids>
inline std::vector<S> set_op_batch(const HipGraphT<S> &g, const Ids &u, const Ids &v, bool difference = false) {
    const int64_t np = int64_t(u.size());
    std::vector<int32_t> uu(u.begin(), u.end()), vv(v.begin(), v.end());
    std::vector<int64_t> off(size_t(np) + 1, 0);
    const int op = difference ? GMSX_SETOP_DIFFERENCE : GMSX_SETOP_INTERSECT;
    detail::check(gmsx_set_op_batch(g.device(), op, np, uu.data(), vv.data(), off.data(), nullptr, 0, nullptr), "gmsx_set_op_batch");
    std::vector<int32_t> ids(size_t(off[size_t(np)]) + 1);
    detail::check(gmsx_set_op_batch(g.device(), op, np, uu.data(), vv.data(), off.data(), ids.data(), off[size_t(np)], nullptr), "gmsx_set_op_batch");
    std::vector<S> out;
    out.reserve(size_t(np));
    for (int64_t i = 0; i < np; ++i) out.emplace_back(ids.data() + off[size_t(i)], size_t(off[size_t(i) + 1] - off[size_t(i)]));  // (owning copies)
    return out;
}

}  // namespace gmsx
