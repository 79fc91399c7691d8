// gmsx_driver — command-line driver with the reference drivers' flags and output lines, running the device path
// through include/gmsx_set_graph.hpp (C++ adaptor) -> include/gmsx.h (C-ABI).
//
// Mirrors (paths relative to the spcl/gms tree):
//   flags        gms/common/cli/cli.h:75-155   -g kronecker|uniform <scale> [--deg d] | -f file ; -v ; -n trials ; -t threads ; -p name=value
//   loading      gms/common/cli/cli.h:157-184  (generate/load, reject directed, relabel when WorthRelabelling)
//   trial loop   gms/common/benchmark.h:96-137 ("GraphExec buildTime", "Trial Time", "Verification", "@@@ …", "Average Time")
//   kernels      triangle_count.cc:22-48, k_clique_count_set_based.cc:27-47, maximal_clique_enum_bron_kerbosch.cc:59-93
// Usage:  gmsx_driver <tc|vertex|kclique|bk> [reference flags]      e.g.  gmsx_driver tc -g kronecker 20 --deg 16 -n 3 -v
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "gmsx_set_graph.hpp"

namespace {

struct Args {  // gms/common/cli/args.h:17-107 defaults
    std::string kernel;
    bool verify = false;
    int64_t trials = 3, threads = 0;
    std::string file, gen;
    int scale = -1, deg = 16;
    int clique_size = 4;
    int error = 0;
};

struct Timer {  // gapbs/timer.h:18-47
    std::chrono::high_resolution_clock::time_point t0, t1;
    void Start() { t0 = std::chrono::high_resolution_clock::now(); }
    void Stop() { t1 = std::chrono::high_resolution_clock::now(); }
    double Seconds() const { return std::chrono::duration<double>(t1 - t0).count(); }
};
void PrintTime(const std::string &s, double sec) { std::printf("%-21s%3.5lf\n", (s + ":").c_str(), sec); }          // gapbs/util.h:31-33
void PrintLabel(const std::string &l, const std::string &v) { std::printf("%-21s%7s\n", (l + ":").c_str(), v.c_str()); }  // util.h:27-29

Args parse(int argc, char **argv) {
    Args a;
    if (argc < 2) { a.error = 101; return a; }
    a.kernel = argv[1];
    for (int i = 2; i < argc; ++i) {
        const std::string f = argv[i];
        auto need = [&](int k) { if (i + k >= argc) { a.error = 100; return false; } return true; };
        if (f == "-v" || f == "--verify") a.verify = true;
        else if (f == "-n" || f == "--num-trials") { if (!need(1)) break; a.trials = std::atoll(argv[++i]); }
        else if (f == "-t" || f == "--threads") { if (!need(1)) break; a.threads = std::atoll(argv[++i]); }
        else if (f == "-f" || f == "--file") { if (!need(1)) break; a.file = argv[++i]; }
        else if (f == "-g" || f == "--gen") { if (!need(2)) break; a.gen = argv[++i]; a.scale = std::atoi(argv[++i]); }
        else if (f == "--deg") { if (!need(1)) break; a.deg = std::atoi(argv[++i]); }
        else if (f == "-p" || f == "--param") {
            if (!need(1)) break;
            const std::string kv = argv[++i];
            if (kv.rfind("clique-size=", 0) == 0) a.clique_size = std::atoi(kv.c_str() + 12);
            else if (kv.rfind("cs=", 0) == 0) a.clique_size = std::atoi(kv.c_str() + 3);
            else a.error = 100;
        } else a.error = 100;
    }
    if (!a.error && a.file.empty() && a.gen.empty()) a.error = 101;  // cli/cli.h:131-133
    if (!a.error && !a.gen.empty() && a.gen != "kronecker" && a.gen != "uniform") a.error = 100;
    return a;
}

void usage(const char *argv0) {
    std::printf("usage: %s <tc|vertex|kclique|bk> (-g kronecker|uniform <scale> [--deg d] | -f file.{el,sg}) [-v] [-n trials] [-t threads] "
                "[-p clique-size=k]\n", argv0);
}

}  // namespace

int main(int argc, char **argv) {
    Args args = parse(argc, argv);
    if (args.error) { usage(argv[0]); return args.error; }  // the reference exits with 100 / 101 (cli/cli.h:122-133,159-160)
    if (args.kernel != "tc" && args.kernel != "vertex" && args.kernel != "kclique" && args.kernel != "bk") { usage(argv[0]); return 100; }

    // ---- parse_and_load ------------------------------------------------------------------------------------
    Timer t;
    gmsx_csr *csr = nullptr;
    t.Start();
    int rc;
    if (!args.file.empty()) rc = gmsx_csr_load(args.file.c_str(), 1, GMSX_RELABEL_AUTO, &csr);
    else rc = gmsx_csr_generate(args.gen == "uniform" ? GMSX_GEN_UNIFORM : GMSX_GEN_KRONECKER, args.scale, args.deg, GMSX_RELABEL_AUTO,
                                int(args.threads), &csr);
    t.Stop();
    if (rc != GMSX_OK) {
        std::printf("could not load the graph: %s\n", gmsx_strerror(rc));
        return rc == GMSX_ERR_DIRECTED ? 100 : 2;
    }
    PrintTime("Load Time", t.Seconds());
    const int64_t n = gmsx_csr_num_nodes(csr), m = gmsx_csr_num_edges(csr);
    // CSRGraph::PrintStats (gapbs/graph.h:270-277)
    std::cout << "Graph has " << n << " nodes and " << m << " undirected edges for degree: " << (n ? m / n : 0) << std::endl;
    if (gmsx_init(-1) != GMSX_OK) { std::printf("no HIP device: this driver has no CPU path\n"); return 3; }

    // ---- BenchmarkKernelBk -----------------------------------------------------------------------------------
    t.Start();
    gmsx::HipSetGraph g = gmsx::HipSetGraph::FromCsr(csr);
    (void)g.device();  // upload + device-side set construction
    t.Stop();
    PrintTime("GraphExec buildTime", t.Seconds());

    std::string label;
    double total = 0;
    for (int64_t it = 0; it < args.trials; ++it) {
        uint64_t result = 0;
        std::vector<int64_t> counts;
        t.Start();
        if (args.kernel == "tc") { result = gmsx::count_total(g); label = "tc-total-par-HipSetGraph"; }
        else if (args.kernel == "vertex") { gmsx::vertex_count2(g, counts); label = "tc-vertex-count2-par-HipSetGraph"; }
        else if (args.kernel == "kclique") { result = gmsx::clique_count(g, size_t(args.clique_size)); label = "HipSet HipSetGraph"; }
        else { std::vector<int32_t> rank(size_t(n), 0); result = gmsx::maximal_clique_count(g, rank); label = "BK-GMS-DEG"; }
        t.Stop();
        const double trial = t.Seconds();
        total += trial;
        PrintTime("Trial Time", trial);
        if (args.kernel == "tc") std::printf("triangles: %" PRIu64 "\n", result);
        if (args.kernel == "bk") std::printf("The Number of maximal clique counted: %" PRIu64 "\n", result);  // helper.h:120-133
        if (args.verify) {
            t.Start();
            bool ok = true;
            if (args.kernel == "tc") {  // verifier.h:13-42 recounts with an independent formulation; here: the full-row kernel
                uint64_t again = 0;
                ok = gmsx_tc_total(g.device(), GMSX_TC_FULL, &again, nullptr) == GMSX_OK && again == result;
            } else if (args.kernel == "vertex") {  // verifier.h:44-85: 3*T == Σ counts / 2
                uint64_t tri = 0, sum = 0;
                for (int64_t c : counts) sum += uint64_t(c);
                ok = gmsx_tc_total(g.device(), GMSX_TC_AUTO, &tri, nullptr) == GMSX_OK && sum == 6 * tri;
            } else if (args.kernel == "kclique") {  // k_clique_count_set_based.cc:15-21 re-runs the same function
                uint64_t again = 0;
                ok = gmsx_kclique_count(g.device(), args.clique_size, &again, nullptr, nullptr) == GMSX_OK && again == result;
            } else {
                uint64_t again = 0;
                ok = gmsx_bk_count(g.device(), nullptr, &again, nullptr) == GMSX_OK && again == result;
            }
            t.Stop();
            const std::string mark = ok ? "PASS" : "FAIL";
            PrintLabel("Verification", mark);
            PrintTime("Verification Time", t.Seconds());
            std::cout << "@@@ " << trial << " " << mark << " " << t.Seconds() << " " << label << std::endl;
        } else {
            std::cout << "@@@ " << trial << " " << label << std::endl;
        }
    }
    PrintTime("Average Time", total / double(args.trials > 0 ? args.trials : 1));
    gmsx_csr_free(csr);
    return 0;
}
