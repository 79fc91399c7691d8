// TEST: include/gmsx_gms_glue.hpp routes BkEppsteinPar::mceBench to the device ONLY in the reference's count build (-DBK_COUNT without
// MINEBENCH_TEST).  Compiled by tests/test_reference_drivers.py against the reference tree where it lies, three times:
//   -DBK_COUNT -DEXPECT_ROUTED=1                    the specialisation exists (how the reference builds its drivers, maximal_clique_enum/CMakeLists.txt:8-10)
//   -DBK_COUNT -DMINEBENCH_TEST -DEXPECT_ROUTED=0   a listing build: mceBench<HipSetGraph> is the reference's GENERIC template over the gmsx host sets and
//                                                   returns the cliques themselves (tomita.h:79-84) — run here, on the host, without a GPU
//   -DEXPECT_ROUTED=0                               neither macro: generic template again
#include "gms/third_party/gapbs/benchmark.h"
#include <gms/common/cli/cli.h>
#include <gms/common/types.h>
#include <gms/representations/graphs/set_graph.h>
#include <gms/algorithms/set_based/maximal_clique_enum/bron_kerbosch.h>
#include <gmsx_gms_glue.hpp>

#include <cstdio>
#include <string>
#include <vector>

static_assert(GMSX_GLUE_BK_ROUTED == EXPECT_ROUTED, "mceBench routing does not follow BK_COUNT / MINEBENCH_TEST");

template <class SGraph>
static size_t listed(const CSRGraph &g) {
    SGraph sg = SGraph::FromCGraph(g);
    pvector<NodeId> rank(sg.num_nodes());
    PpParallel::getDegreeOrdering<SGraph, true, pvector<NodeId>>(sg, rank);
    return BkEppsteinPar::mceBench<SGraph>(sg, rank).size();
}

int main() {
#if defined(MINEBENCH_TEST)
    std::vector<std::string> argv_s = {"glue", "-g", "kronecker", "8", "--deg", "16"};
    std::vector<char *> argv;
    for (auto &s : argv_s) argv.push_back(const_cast<char *>(s.c_str()));
    GMS::CLI::Parser parser;
    GMS::CLI::Args args = parser.parse((int)argv.size(), argv.data());
    CSRGraph g = args.load_graph();
    const size_t want = listed<RoaringGraph>(g);           // the reference over its own sets
    const size_t got = listed<HipSetGraph>(g);             // the same generic template over the gmsx host sets (no device involved)
    const size_t got_r = listed<HipRoaringGraph>(g);
    std::printf("listed %zu %zu %zu\n", want, got, got_r);
    return want > 0 && got == want && got_r == want ? 0 : 1;
#else
    std::printf("routed %d\n", GMSX_GLUE_BK_ROUTED);
    return 0;
#endif
}
