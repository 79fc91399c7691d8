// CPU test of the C++ adaptor (include/gmsx_set_graph.hpp): gmsx::SortedSpanSet / RoaringSpanSet / SortedSpanRef must behave like the reference's
// Set concept on the literal cases of testing/sets.cpp (values restated from SURVEY §8b), and gmsx::HipSetGraph must
// expose the SGraph surface.  With -DWITH_REFERENCE the reference's own generic algorithm templates are instantiated
// over HipSetGraph on the host (no device call) and compared with the reference's SortedSetGraph.
#include <cassert>
#include <cstdio>
#include <type_traits>
#include <vector>

#ifdef WITH_REFERENCE
#include "gms/third_party/gapbs/benchmark.h"
#include <gms/common/cli/cli.h>
#include <gms/common/types.h>
#include <gms/representations/graphs/set_graph.h>
#include <gms/algorithms/set_based/triangle_count/triangle_count.h>
#include <gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h>
#include <gms/algorithms/set_based/maximal_clique_enum/bron_kerbosch.h>
#endif

#include "gmsx_set_graph.hpp"

#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

template <class Set> static std::vector<int> vec(const Set &s) { return std::vector<int>(s.begin(), s.end()); }

// the literal cases of testing/sets.cpp, for one set flavour (the reference runs them as a typed test over its set types, sets.cpp:15-24)
template <class S>
static int set_cases() {
    // construction is order-insensitive and from unsorted input (sets.cpp:39,80-94)
    CHECK(S({2, 4, 8}) == S({4, 2, 8}));
    const int raw[] = {1, 5, 2, 7, 9, 0, 3};
    CHECK(S(raw, 7).cardinality() == 7);
    CHECK(S(raw, 0).cardinality() == 0);
    CHECK(S(5).cardinality() == 1 && S(5).contains(5));
    // intersect / intersect_count symmetric and consistent (sets.cpp:108-141)
    const S e{}, a{1, 2, 3}, b{4, 5, 6}, c{1, 2, 3, 4, 5}, d{3, 4, 5, 6, 7}, f{1, 2, 3, 4, 5, 6, 7}, h{2, 4, 6, 8};
    CHECK(e.intersect_count(e) == 0 && e.intersect_count(a) == 0 && a.intersect_count(e) == 0 && a.intersect_count(b) == 0);
    CHECK(vec(c.intersect(d)) == (std::vector<int>{3, 4, 5}) && c.intersect_count(d) == 3 && d.intersect_count(c) == 3);
    CHECK(vec(f.intersect(h)) == (std::vector<int>{2, 4, 6}) && vec(h.intersect(f)) == (std::vector<int>{2, 4, 6}));
    CHECK(a.intersect(a) == a && a.intersect_count(a) == 3);
    {   // intersect_inplace must not modify its argument (sets.cpp:144-158)
        S x = c.clone(), y = d.clone();
        x.intersect_inplace(y);
        CHECK(vec(x) == (std::vector<int>{3, 4, 5}) && y == d);
    }
    // union (sets.cpp:194-292): union_count 0,3,6,7,5
    CHECK(e.union_count(e) == 0 && e.union_count(a) == 3 && a.union_count(b) == 6 && c.union_count(d) == 7 && c.union_count(S{3, 4, 5}) == 5);
    CHECK(vec(a.union_with(b)) == (std::vector<int>{1, 2, 3, 4, 5, 6}));
    {
        S x = a.union_with(9).union_with(0);
        CHECK(vec(x) == (std::vector<int>{0, 1, 2, 3, 9}));
        x.union_inplace(9);  // idempotent
        CHECK(x.cardinality() == 5);
        x.union_inplace(b);
        CHECK(x.cardinality() == 8);
    }
    // difference both directions (sets.cpp:311-357): {1..5}\{3,4,5,6,8}={1,2}; reverse {6,8}; absent element is a no-op
    const S g{3, 4, 5, 6, 8};
    CHECK(vec(c.difference(g)) == (std::vector<int>{1, 2}) && vec(g.difference(c)) == (std::vector<int>{6, 8}));
    CHECK(vec(c.difference(3)) == (std::vector<int>{1, 2, 4, 5}) && c.difference(42) == c);
    {
        S x = c.clone();
        x.difference_inplace(g);
        CHECK(vec(x) == (std::vector<int>{1, 2}));
        x.remove(7);
        CHECK(x.cardinality() == 2);
        x.add(0); x.add(9); x.add(1);  // front / end / existing (sets.cpp:404-429)
        CHECK(vec(x) == (std::vector<int>{0, 1, 2, 9}));
    }
    CHECK(c.contains(1) && c.contains(5) && !c.contains(0) && !c.contains(6) && !e.contains(0));
    CHECK(S::Range(0).cardinality() == 0 && vec(S::Range(5)) == (std::vector<int>{0, 1, 2, 3, 4}));
    {
        int buf[3] = {-1, -1, -1};
        e.toArray(buf);
        CHECK(buf[0] == -1);  // empty set leaves the buffer untouched (sets.cpp:475-492)
        a.toArray(buf);
        CHECK(buf[0] + buf[1] + buf[2] == 6);
    }
    CHECK(c != d && !(c == d));

    // ---- views (the zero-copy row form): read-only ops work in place, the first mutation detaches a private copy ----
    {
        const int32_t row[] = {1, 3, 5, 7, 9};
        S v = S::borrow(row, 5);
        CHECK(v.borrowed() && v.data() == row && v.cardinality() == 5 && v.contains(7) && !v.contains(8));
        CHECK(v.intersect_count(c) == 3 && vec(v.intersect(c)) == (std::vector<int>{1, 3, 5}) && vec(v.difference(c)) == (std::vector<int>{7, 9}));
        S w = v.clone();
        CHECK(!w.borrowed() && w == v && w.data() != row);
        v.union_inplace(4);  // detaches
        CHECK(!v.borrowed() && vec(v) == (std::vector<int>{1, 3, 4, 5, 7, 9}) && row[2] == 5);
        S x = S::borrow(row, 5);
        x.remove(3);
        CHECK(vec(x) == (std::vector<int>{1, 5, 7, 9}) && row[1] == 3);
        S y = S::borrow(row, 5);
        y.intersect_inplace(c);
        CHECK(vec(y) == (std::vector<int>{1, 3, 5}) && row[3] == 7);
        S z = S::borrow(row, 5);
        z.difference_inplace(c);
        CHECK(vec(z) == (std::vector<int>{7, 9}));
        S m = std::move(z);
        CHECK(vec(m) == (std::vector<int>{7, 9}) && z.cardinality() == 0);
        for (int i = 0; i < 100; ++i) m.add(100 - i);  // growth path
        CHECK(m.cardinality() == 100 && m.contains(1) && m.contains(100) && m.contains(7));
    }
    return 0;
}

int main() {
    if (set_cases<gmsx::SortedSpanSet>()) return 1;
    if (set_cases<gmsx::RoaringSpanSet>()) return 1;
    using S = gmsx::SortedSpanSet;
    static_assert(sizeof(S) == 16 && sizeof(gmsx::SortedSpanRef) == 16, "a row view is pointer + count");
    static_assert(std::is_same_v<decltype(*gmsx::RoaringSpanSet().begin()), const uint32_t &>, "Roaring flavour iterates uint32 (roaring_set.h:82-105)");
    static_assert(std::is_same_v<decltype(*S().begin()), const int32_t &>, "SortedSet flavour iterates SetElement");
    {   // SortedSetRef surface (sorted_set_ref.h:9-78): borrowed, operators return owning sets
        const int32_t row[] = {1, 2, 3, 4, 5}, other[] = {3, 4, 5, 6, 8};
        gmsx::SortedSpanRef r(row, 5), o(other, 5);
        CHECK(r.cardinality() == 5 && r.begin() == row && r.intersect_count(o) == 3);
        CHECK(vec(r.intersect(o)) == (std::vector<int>{3, 4, 5}) && vec(r.difference(o)) == (std::vector<int>{1, 2}));
        CHECK(vec(r.union_with(o)).size() == 7 && r.contains(5) && !r.contains(6));
        CHECK(vec(r.intersect(S{2, 3, 9})) == (std::vector<int>{2, 3}));
    }

    // SGraph surface on a tiny CSR through the C-ABI host substrate (no device call)
    const int32_t src[] = {0, 1, 2, 2}, dst[] = {1, 2, 0, 3};
    gmsx_csr *csr = nullptr;
    CHECK(gmsx_csr_from_edges(-1, 4, src, dst, 1, GMSX_RELABEL_NEVER, &csr) == GMSX_OK);
    {
        gmsx::HipSetGraph hg = gmsx::HipSetGraph::FromCsr(csr);
        CHECK(hg.num_nodes() == 4 && hg.out_degree(2) == 3 && vec(hg.out_neigh(2)) == (std::vector<int>{0, 1, 3}));
        CHECK(hg.out_neigh(0).intersect_count(hg.out_neigh(1)) == 1);
        // zero-copy: the rows are views into the gmsx_csr arrays, no second adjacency
        CHECK(hg.borrows_adjacency() && hg.neighbors() == gmsx_csr_neighbors(csr) && hg.out_neigh(2).data() == gmsx_csr_neighbors(csr) + gmsx_csr_offsets(csr)[2]);
        gmsx::HipSetGraph deep = hg.clone();
        CHECK(!deep.borrows_adjacency() && deep.num_nodes() == 4 && deep.out_neigh(2) == hg.out_neigh(2));
        gmsx::HipRoaringGraph rg = gmsx::HipRoaringGraph::FromCsr(csr);
        unsigned sum = 0;
        for (auto v : rg.out_neigh(2)) sum += v;  // uint32 iteration
        CHECK(sum == 4 && rg.out_neigh(2).intersect_count(rg.out_neigh(0)) == 1);
        if (std::getenv("GMSX_TEST_DEVICE_SET_OPS")) {  // (instantiated for every flavour on every build; executed only where a device is: the GPU adaptor test)
            auto both = gmsx::set_op_batch(hg, std::vector<int>{0, 2}, std::vector<int>{1, 0});
            auto left = gmsx::set_op_batch(rg, std::vector<int>{2}, std::vector<int>{0}, true);
            CHECK(both.size() == 2 && vec(both[0]) == (std::vector<int>{2}) && vec(both[1]) == (std::vector<int>{1}));
            CHECK(left.size() == 1 && left[0].cardinality() == 2 && left[0].contains(0) && left[0].contains(3));
        }
        gmsx::HipSetRefGraph fg = gmsx::HipSetRefGraph::FromCsr(csr);
        CHECK(fg.out_neigh(2).cardinality() == 3 && fg.out_neigh(2).intersect_count(fg.out_neigh(0)) == 1);
        // a graph uploaded as a shard stays that shard when it is moved (count_total of a moved rank graph asked gmsx_tc_total for the
        // WHOLE graph on a sharded upload); the moved-from object is a whole-graph placeholder again
        gmsx::default_upload_shard() = {1, 2};
        gmsx::HipSetGraph sharded = gmsx::HipSetGraph::FromCsr(csr);
        gmsx::default_upload_shard() = {0, 1};
        CHECK(sharded.upload_shard() == (std::pair<int, int>{1, 2}));
        gmsx::HipSetGraph moved(std::move(sharded));
        CHECK(moved.upload_shard() == (std::pair<int, int>{1, 2}) && sharded.upload_shard() == (std::pair<int, int>{0, 1}) && moved.num_nodes() == 4);
        gmsx::HipSetGraph assigned;
        assigned = std::move(moved);
        CHECK(assigned.upload_shard() == (std::pair<int, int>{1, 2}) && moved.upload_shard() == (std::pair<int, int>{0, 1}));
    }
    gmsx_csr_free(csr);

#ifdef WITH_REFERENCE
    {   // the reference's generic templates instantiate over HipSetGraph and agree with its own SortedSetGraph (host only)
        char a0[] = "t", a1[] = "-g", a2[] = "kronecker", a3[] = "8", a4[] = "--deg", a5[] = "16";
        char *argv[] = {a0, a1, a2, a3, a4, a5};
        auto [args, cg] = GMS::CLI::Parser().parse_and_load(6, argv);
        (void)args;
        auto ref = SortedSetGraph::FromCGraph(cg);
        auto mine = gmsx::HipSetGraph::FromCGraph(cg);
        CHECK(mine.num_nodes() == ref.num_nodes());
        // zero-copy from the reference's CSRGraph: rows are views into ITS neighbour array (gapbs/graph.h:361-364)
        CHECK(mine.borrows_adjacency() && mine.out_neigh(0).data() == cg.out_neigh(0).begin());
        CHECK(GMS::TriangleCount::Seq::count_total(mine) == GMS::TriangleCount::Seq::count_total(ref));
        CHECK(GMS::TriangleCount::Par::count_total(mine) == 10479);  // tests/golden/graphs.json kronecker-8
        std::vector<int64_t> c1, c2;
        GMS::TriangleCount::Par::vertex_count2(mine, c1);
        GMS::TriangleCount::Par::vertex_count2(ref, c2);
        CHECK(c1 == c2);
        CHECK(RecursiveStepCliqueCount(mine, 3, mine.out_neigh(0)) == RecursiveStepCliqueCount(ref, 3, ref.out_neigh(0)));
        pvector<NodeId> rank(mine.num_nodes());
        PpParallel::getDegreeOrdering<gmsx::HipSetGraph, true, pvector<NodeId>>(mine, rank);
        BK_CLIQUE_COUNTER = 0;
        BkEppsteinPar::mceBench<gmsx::HipSetGraph>(mine, rank);
        CHECK(BK_CLIQUE_COUNTER == 1808);  // tests/golden/graphs.json kronecker-8 bk
        // the Roaring flavour through the same generic templates (the BK driver instantiates RoaringGraph, …bron_kerbosch.cc:84-91)
        auto rmine = gmsx::HipRoaringGraph::FromCGraph(cg);
        CHECK(GMS::TriangleCount::Par::count_total(rmine) == 10479);
        BK_CLIQUE_COUNTER = 0;
        BkEppsteinPar::mceBench<gmsx::HipRoaringGraph>(rmine, rank);
        CHECK(BK_CLIQUE_COUNTER == 1808);
        // ADG on the host through the generic template over our sets == over the reference's (1 thread: deterministic ties)
        omp_set_num_threads(1);
        pvector<NodeId> r1(mine.num_nodes()), r2(mine.num_nodes());
        PpParallel::getDegeneracyOrderingApproxSGraph<PpParallel::boundary_function::averageDegree, true, gmsx::HipSetGraph, pvector<NodeId>>(mine, r1, 0.001);
        PpParallel::getDegeneracyOrderingApproxSGraph<PpParallel::boundary_function::averageDegree, true, SortedSetGraph, pvector<NodeId>>(ref, r2, 0.001);
        for (int64_t i = 0; i < mine.num_nodes(); ++i) CHECK(r1[i] == r2[i]);
        // SetGraph<SortedSetRef> flavour in the k-clique recursion (k_clique_count_set_based.cc:38)
        auto fmine = gmsx::HipSetRefGraph::FromCGraph(cg);
        CHECK(RecursiveStepCliqueCount(fmine, 3, fmine.out_neigh(0)) == RecursiveStepCliqueCount(ref, 3, ref.out_neigh(0)));
    }
#endif
    std::printf("set concept ok\n");
    return 0;
}
