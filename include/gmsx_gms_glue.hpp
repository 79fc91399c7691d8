// gmsx_gms_glue.hpp — the reference-side glue: explicit specialisations that route GraphMineSuite's
// OWN function names to the gmsx device kernels for the three gmsx graph flavours.  This is the one header a GMS maintainer
// adds (e.g. as gms/representations/graphs/hip_set_graph.h, or by `#include <gmsx_gms_glue.hpp>` with -I<gmsx>/include); the
// drivers then gain one line each (`benchmark_suite<HipSetGraph>(…)`, `runEppstein<HipRoaringGraph>(…)`, one
// `BenchmarkKernel(…CliqueCount<…HipSetGraph…>…)`), exactly like the existing set types.
//
// It includes reference headers (<gms/...>), so it compiles only inside a GMS build (-I<gms root>); nothing of the reference
// is stored in this repository.  tests/test_reference_drivers.py compiles the three real drivers with it.
//
// Routed (SGraph ∈ {HipSetGraph, HipRoaringGraph, HipSetRefGraph}):
//   GMS::TriangleCount::Par::count_total / Seq::count_total            gms/algorithms/set_based/triangle_count/parallel/total.h:7-24, sequential/total.h:7-23
//   GMS::TriangleCount::Par::vertex_count2 / vertex_count2_once / Seq::vertex_count2     parallel/vertex.h:14-49, sequential/vertex.h:14-26
//   CliqueCount<Set, SGraph, Set2>                                      gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h:19-31
//   BkEppsteinPar::mceBench<SGraph>   (count builds only, see below)    gms/algorithms/set_based/maximal_clique_enum/parallel/eppsteinPAR.h:18-53
//   PpParallel::getDegeneracyOrderingApproxSGraph<averageDegree, …>     gms/algorithms/preprocessing/parallel/degeneracy_approx_set.h:14-86
//   PpParallel::triangleCountOrdering<SGraph>                           gms/algorithms/preprocessing/parallel/triangle_count.h:11-30
// Everything not listed (Verify::*, BkTomita::mce, getDegreeOrdering, getDegeneracyOrderingMatula, …) keeps instantiating the
// reference's generic templates over the gmsx host sets — that is what makes the harness's `-v` verifiers independent host
// recounts (triangle_count/verifier.h:13-42, maximal_clique_enum/verifier.h:41-49).
#pragma once

#include <gmsx_set_graph.hpp>

#include <gms/algorithms/preprocessing/preprocessing.h>
#include <gms/algorithms/preprocessing/parallel/triangle_count.h>
#include <gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h>
#include <gms/algorithms/set_based/maximal_clique_enum/bron_kerbosch.h>
#include <gms/algorithms/set_based/triangle_count/triangle_count.h>

using HipSetGraph = gmsx::HipSetGraph;          // SGraph concept of set_graph.h:86-118, SortedSet flavour
using HipRoaringGraph = gmsx::HipRoaringGraph;  // RoaringSet flavour (uint32 iteration, roaring_set.h:82-105)
using HipSetRefGraph = gmsx::HipSetRefGraph;    // SetGraph<SortedSetRef> flavour (borrowed rows, sorted_set_ref.h:9-78)

#define GMSX_GLUE_TC(SGRAPH)                                                                                                       \
    namespace GMS::TriangleCount::Par {                                                                                            \
    template <> inline size_t count_total<SGRAPH>(const SGRAPH &g) { return gmsx::count_total(g); }                                \
    template <> inline void vertex_count2<SGRAPH, std::vector<int64_t>>(const SGRAPH &g, std::vector<int64_t> &c) { gmsx::vertex_count2(g, c); }      \
    template <> inline void vertex_count2_once<SGRAPH, std::vector<int64_t>>(const SGRAPH &g, std::vector<int64_t> &c) { gmsx::vertex_count2(g, c); } \
    template <> inline void vertex_count2<SGRAPH, pvector<int64_t>>(const SGRAPH &g, pvector<int64_t> &c) { gmsx::vertex_count2(g, c); }              \
    template <> inline void vertex_count2_once<SGRAPH, pvector<int64_t>>(const SGRAPH &g, pvector<int64_t> &c) { gmsx::vertex_count2(g, c); }         \
    }                                                                                                                              \
    namespace GMS::TriangleCount::Seq {                                                                                            \
    template <> inline size_t count_total<SGRAPH>(const SGRAPH &g) { return gmsx::count_total(g); }                                \
    template <> inline void vertex_count2<SGRAPH, std::vector<int64_t>>(const SGRAPH &g, std::vector<int64_t> &c) { gmsx::vertex_count2(g, c); }      \
    }                                                                                                                              \
    namespace PpParallel {                                                                                                         \
    template <> inline void triangleCountOrdering<SGRAPH>(const SGRAPH &g, std::vector<NodeId> &ordering) { gmsx::triangle_count_ordering(g, ordering); } \
    }

// BK_CLIQUE_COUNTER (helper.h:15) is what -DBK_COUNT builds print and verify (helper.h:127-133, verifier.h:72-78).  The device path COUNTS and
// returns an empty `sol`; that is the reference's own behaviour only in a count build — with MINEBENCH_TEST the reference fills `sol` (tomita.h:79-84)
// and its tests compare the set of cliques (testing/bron_kerbosch.cpp:117-127), and without BK_COUNT nothing reads the counter.  So mceBench is routed
// to the device ONLY under -DBK_COUNT without MINEBENCH_TEST (how the reference builds its three Bron–Kerbosch drivers,
// maximal_clique_enum/CMakeLists.txt:8-10); in every other build the specialisation does not exist and the call instantiates the reference's
// generic template over the gmsx host sets — a listing caller gets its cliques (from the host), never an empty vector that looks like "no cliques".
#if defined(BK_COUNT) && !defined(MINEBENCH_TEST)
#define GMSX_GLUE_BK_ROUTED 1
#define GMSX_GLUE_BK(SGRAPH)                                                                                                       \
    namespace BkEppsteinPar {                                                                                                      \
    template <> inline std::vector<SGRAPH::Set> mceBench<SGRAPH, SGRAPH::Set>(const SGRAPH &g, const pvector<NodeId> &ordering) { \
        BK_CLIQUE_COUNTER = gmsx::maximal_clique_count(g, ordering);                                                               \
        return {};                                                                                                                 \
    }                                                                                                                              \
    }
#else
#define GMSX_GLUE_BK_ROUTED 0
#define GMSX_GLUE_BK(SGRAPH)
#endif

#define GMSX_GLUE_ADG(SGRAPH)                                                                                                      \
    namespace PpParallel {                                                                                                         \
    template <>                                                                                                                    \
    inline void getDegeneracyOrderingApproxSGraph<boundary_function::averageDegree, true, SGRAPH, pvector<NodeId>>(               \
        const SGRAPH &g, pvector<NodeId> &res, const double epsilon) {                                                             \
        gmsx::adg_rank(g, epsilon, res, true);                                                                                     \
    }                                                                                                                              \
    template <>                                                                                                                    \
    inline void getDegeneracyOrderingApproxSGraph<boundary_function::averageDegree, false, SGRAPH, std::vector<NodeId>>(          \
        const SGRAPH &g, std::vector<NodeId> &res, const double epsilon) {                                                         \
        gmsx::adg_rank(g, epsilon, res, false);                                                                                    \
    }                                                                                                                              \
    }

// CliqueCount builds its SGraph inside the timed region (k_clique_count_set_based.h:22); so does this — with the lean upload (DAG containers
// + bitsets only: no triangle-count task lists), whatever the driver's default flags are
#define GMSX_GLUE_KC(SET, SGRAPH, SET2)                                                                                            \
    template <> inline size_t CliqueCount<SET, SGRAPH, SET2>(CSRGraph & g, size_t k) {                                             \
        return gmsx::clique_count(SGRAPH::FromCGraph(g, GMSX_UPLOAD_DEFAULT), k);                                                  \
    }

GMSX_GLUE_TC(HipSetGraph)
GMSX_GLUE_TC(HipRoaringGraph)
GMSX_GLUE_TC(HipSetRefGraph)
GMSX_GLUE_BK(HipSetGraph)
GMSX_GLUE_BK(HipRoaringGraph)
GMSX_GLUE_ADG(HipSetGraph)
GMSX_GLUE_ADG(HipRoaringGraph)
GMSX_GLUE_KC(gmsx::SortedSpanSet, HipSetGraph, gmsx::SortedSpanSet)        // like <SortedSet, SortedSetGraph, SortedSet>         (k_clique_count_set_based.cc:42)
GMSX_GLUE_KC(gmsx::RoaringSpanSet, HipRoaringGraph, gmsx::RoaringSpanSet)  // like <RoaringSet, RoaringGraph, RoaringSet>          (:34)
GMSX_GLUE_KC(gmsx::SortedSpanSet, HipSetRefGraph, gmsx::SortedSpanRef)     // like <SortedSet, SetGraph<SortedSetRef>, SortedSetRef> (:38)

#undef GMSX_GLUE_TC
#undef GMSX_GLUE_BK
#undef GMSX_GLUE_ADG
#undef GMSX_GLUE_KC
