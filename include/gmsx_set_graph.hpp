// gmsx_set_graph.hpp — C++17 adaptor that plugs the gmsx C-ABI (gmsx.h) under GMS's compile-time Set / SGraph
// concept, so drivers written against the reference's templates instantiate unchanged with `gmsx::HipSetGraph`.
//
// What the reference requires of a Set (union of testing/sets.cpp and the three algorithm families):
//   gms/representations/sets/sorted_set.h:21-272   (SortedSetBase)  — the surface mirrored by gmsx::SortedSpanSet
//   gms/representations/graphs/set_graph.h:86-118  (SetGraph)       — the surface mirrored by gmsx::HipSetGraph
// The per-element operators below run on the host (they exist so that every generic template still compiles and
// runs); the whole-graph algorithms — the hot path — are forwarded to the device through the C-ABI:
//   gmsx::count_total(const HipSetGraph&)        -> gmsx_tc_total        (triangle_count/parallel/total.h:7-24)
//   gmsx::vertex_count2(const HipSetGraph&, out) -> gmsx_tc_vertex_count2 (triangle_count/parallel/vertex.h:14-49)
//   gmsx::clique_count(const HipSetGraph&, k)    -> gmsx_kclique_count   (k_clique_count_set_based.h:19-31)
//   gmsx::maximal_clique_count(const HipSetGraph&, rank) -> gmsx_bk_count (maximal_clique_enum/parallel/eppsteinPAR.h:18-53)
// INTEGRATION.md shows the explicit specialisations a GMS maintainer adds so that the reference's own function names
// resolve to these.  Header-only; link with -lgmsx.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "gmsx.h"

namespace gmsx {

using SetElement = int32_t;  // sorted_set.h:25, roaring_set.h:28

// Owning sorted set of int32 with the reference's Set surface.  Move-only like the reference (copying is explicit via
// clone(), sorted_set.h:31-39,84-87).  Construction accepts unsorted input (sorted_set.h:64-66).
class SortedSpanSet {
   public:
    using SetElement = gmsx::SetElement;
    using Container = std::vector<SetElement>;
    using const_iterator = Container::const_iterator;

    SortedSpanSet() = default;
    SortedSpanSet(SortedSpanSet &&) noexcept = default;
    SortedSpanSet &operator=(SortedSpanSet &&) noexcept = default;
    SortedSpanSet(const SortedSpanSet &) = delete;
    SortedSpanSet &operator=(const SortedSpanSet &) = delete;

    SortedSpanSet(const SetElement *start, size_t count) : v_(start, start + count) { normalise(); }
    explicit SortedSpanSet(const Container &c) : v_(c) { normalise(); }
    explicit SortedSpanSet(Container &&c, bool sorted = false) : v_(std::move(c)) {
        if (!sorted) normalise();
    }
    SortedSpanSet(std::initializer_list<SetElement> il) : v_(il) { normalise(); }
    explicit SortedSpanSet(SetElement single) : v_(1, single) {}

    SortedSpanSet clone() const { return SortedSpanSet(Container(v_), true); }
    static SortedSpanSet Range(int bound) {  // sorted_set.h:257-262
        Container c(static_cast<size_t>(std::max(bound, 0)));
        for (int i = 0; i < bound; ++i) c[size_t(i)] = i;
        return SortedSpanSet(std::move(c), true);
    }

    size_t cardinality() const { return v_.size(); }
    const_iterator begin() const { return v_.begin(); }
    const_iterator end() const { return v_.end(); }
    const SetElement *data() const { return v_.data(); }
    bool contains(SetElement e) const { return std::binary_search(v_.begin(), v_.end(), e); }
    void toArray(SetElement *out) const { std::copy(v_.begin(), v_.end(), out); }
    bool operator==(const SortedSpanSet &o) const { return v_ == o.v_; }
    bool operator!=(const SortedSpanSet &o) const { return v_ != o.v_; }

    // ---- intersection (sorted_set.h:160-182) ----
    size_t intersect_count(const SortedSpanSet &o) const {
        size_t c = 0;
        auto a = v_.begin(), ae = v_.end();
        auto b = o.v_.begin(), be = o.v_.end();
        while (a != ae && b != be) {
            if (*a < *b) ++a;
            else if (*b < *a) ++b;
            else { ++c; ++a; ++b; }
        }
        return c;
    }
    SortedSpanSet intersect(const SortedSpanSet &o) const {
        Container r;
        r.reserve(std::min(v_.size(), o.v_.size()));
        std::set_intersection(v_.begin(), v_.end(), o.v_.begin(), o.v_.end(), std::back_inserter(r));
        return SortedSpanSet(std::move(r), true);
    }
    void intersect_inplace(const SortedSpanSet &o) { *this = intersect(o); }

    // ---- union (sorted_set.h:104-158) ----
    SortedSpanSet union_with(const SortedSpanSet &o) const {
        Container r;
        r.reserve(v_.size() + o.v_.size());
        std::set_union(v_.begin(), v_.end(), o.v_.begin(), o.v_.end(), std::back_inserter(r));
        return SortedSpanSet(std::move(r), true);
    }
    SortedSpanSet union_with(SetElement e) const {
        SortedSpanSet r = clone();
        r.union_inplace(e);
        return r;
    }
    void union_inplace(const SortedSpanSet &o) { *this = union_with(o); }
    void union_inplace(SetElement e) {
        auto it = std::lower_bound(v_.begin(), v_.end(), e);
        if (it == v_.end() || *it != e) v_.insert(it, e);
    }
    size_t union_count(const SortedSpanSet &o) const { return v_.size() + o.v_.size() - intersect_count(o); }

    // ---- difference (sorted_set.h:184-216) ----
    SortedSpanSet difference(const SortedSpanSet &o) const {
        Container r;
        r.reserve(v_.size());
        std::set_difference(v_.begin(), v_.end(), o.v_.begin(), o.v_.end(), std::back_inserter(r));
        return SortedSpanSet(std::move(r), true);
    }
    SortedSpanSet difference(SetElement e) const {
        SortedSpanSet r = clone();
        r.difference_inplace(e);
        return r;
    }
    void difference_inplace(const SortedSpanSet &o) { *this = difference(o); }
    void difference_inplace(SetElement e) {
        auto it = std::lower_bound(v_.begin(), v_.end(), e);
        if (it != v_.end() && *it == e) v_.erase(it);
    }
    void add(SetElement e) { union_inplace(e); }
    void remove(SetElement e) { difference_inplace(e); }

   private:
    void normalise() {
        std::sort(v_.begin(), v_.end());
        v_.erase(std::unique(v_.begin(), v_.end()), v_.end());
    }
    Container v_;
};

// SGraph over a device-resident graph.  Keeps a host copy of the CSR (so out_neigh() works for the generic
// templates) and uploads lazily on the first whole-graph call.
class HipSetGraph {
   public:
    using Set = SortedSpanSet;

    HipSetGraph() = default;
    HipSetGraph(HipSetGraph &&o) noexcept { *this = std::move(o); }
    HipSetGraph &operator=(HipSetGraph &&o) noexcept {
        release();
        off_ = std::move(o.off_);
        adj_ = std::move(o.adj_);
        sets_ = std::move(o.sets_);
        dev_ = o.dev_;
        o.dev_ = nullptr;
        return *this;
    }
    HipSetGraph(const HipSetGraph &) = delete;
    HipSetGraph &operator=(const HipSetGraph &) = delete;
    ~HipSetGraph() { release(); }

    // SetGraph::FromCGraph (set_graph.h:86-89): CGraph needs num_nodes(), out_degree(u) and an iterable out_neigh(u)
    template <class CGraph>
    static HipSetGraph FromCGraph(const CGraph &g) {
        HipSetGraph r;
        const int64_t n = g.num_nodes();
        r.off_.resize(size_t(n) + 1);
        r.off_[0] = 0;
        for (int64_t u = 0; u < n; ++u) r.off_[size_t(u) + 1] = r.off_[size_t(u)] + int64_t(g.out_degree(u));
        r.adj_.resize(size_t(r.off_[size_t(n)]));
        for (int64_t u = 0; u < n; ++u) {
            int64_t k = r.off_[size_t(u)];
            for (auto v : g.out_neigh(u)) r.adj_[size_t(k++)] = SetElement(v);
            std::sort(r.adj_.begin() + r.off_[size_t(u)], r.adj_.begin() + r.off_[size_t(u) + 1]);
        }
        r.make_sets();
        return r;
    }
    static HipSetGraph FromCsr(const gmsx_csr *c) {
        HipSetGraph r;
        const int64_t n = gmsx_csr_num_nodes(c), nnz = gmsx_csr_num_edges_directed(c);
        r.off_.assign(gmsx_csr_offsets(c), gmsx_csr_offsets(c) + n + 1);
        r.adj_.assign(gmsx_csr_neighbors(c), gmsx_csr_neighbors(c) + nnz);
        r.make_sets();
        return r;
    }

    int64_t num_nodes() const { return int64_t(off_.size()) - 1; }                       // set_graph.h:115-118
    const Set &out_neigh(SetElement v) const { return sets_[size_t(v)]; }                 // set_graph.h:102-105
    int64_t out_degree(SetElement v) const { return off_[size_t(v) + 1] - off_[size_t(v)]; }  // set_graph.h:91-94

    // device handle (uploads on first use).  Failures follow the reference's convention: message + exit
    // (e.g. gapbs/reader.h:45,228; cli/cli.h:159-171) — nothing is thrown across the C-ABI.
    const gmsx_graph *device() const {
        if (!dev_) {
            const int rc = gmsx_graph_upload(num_nodes(), off_.data(), adj_.data(), GMSX_UPLOAD_DEFAULT, &dev_);
            if (rc != GMSX_OK) {
                std::fprintf(stderr, "gmsx: graph upload failed: %s\n", gmsx_strerror(rc));
                std::exit(-32);
            }
        }
        return dev_;
    }

   private:
    void make_sets() {
        const int64_t n = num_nodes();
        sets_.reserve(size_t(n));
        for (int64_t u = 0; u < n; ++u)
            sets_.emplace_back(SortedSpanSet::Container(adj_.begin() + off_[size_t(u)], adj_.begin() + off_[size_t(u) + 1]), true);
    }
    void release() {
        if (dev_) gmsx_graph_free(dev_);
        dev_ = nullptr;
    }
    std::vector<int64_t> off_;
    std::vector<SetElement> adj_;
    std::vector<Set> sets_;
    mutable gmsx_graph *dev_ = nullptr;
};

namespace detail {
inline void check(int rc, const char *what) {
    if (rc != GMSX_OK) {
        std::fprintf(stderr, "gmsx: %s failed: %s\n", what, gmsx_strerror(rc));
        std::exit(-33);
    }
}
}  // namespace detail

// GMS::TriangleCount::Par::count_total<HipSetGraph> (triangle_count/parallel/total.h:7-24)
inline size_t count_total(const HipSetGraph &g) {
    uint64_t t = 0;
    detail::check(gmsx_tc_total(g.device(), GMSX_TC_AUTO, &t, nullptr), "gmsx_tc_total");
    return size_t(t);
}
// GMS::TriangleCount::Par::vertex_count2<HipSetGraph> (triangle_count/parallel/vertex.h:14-27)
template <class Output = std::vector<int64_t>>
inline void vertex_count2(const HipSetGraph &g, Output &counts) {
    counts.resize(size_t(g.num_nodes()));
    detail::check(gmsx_tc_vertex_count2(g.device(), counts.data(), nullptr), "gmsx_tc_vertex_count2");
}
// CliqueCount<…, HipSetGraph, …> (k_clique_count_set_based.h:19-31): returns k! * C_k like the reference
inline size_t clique_count(const HipSetGraph &g, size_t k = 4) {
    uint64_t ordered = 0;
    detail::check(gmsx_kclique_count(g.device(), int(k), &ordered, nullptr, nullptr), "gmsx_kclique_count");
    std::printf("total %zu-cliques: %llu\n", k, static_cast<unsigned long long>(ordered));  // the reference prints this line (:29)
    return size_t(ordered);
}
// BkEppsteinPar::mceBench<HipSetGraph> under -DBK_COUNT (eppsteinPAR.h:18-53): the maximal-clique count
template <class Ranking>
inline size_t maximal_clique_count(const HipSetGraph &g, const Ranking &rank) {
    std::vector<int32_t> r(size_t(g.num_nodes()));
    for (int64_t i = 0; i < g.num_nodes(); ++i) r[size_t(i)] = int32_t(rank[size_t(i)]);
    uint64_t c = 0;
    detail::check(gmsx_bk_count(g.device(), r.data(), &c, nullptr), "gmsx_bk_count");
    return size_t(c);
}

}  // namespace gmsx
