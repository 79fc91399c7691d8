// Internal declarations shared by the host substrate and the device layer of libgmsx.
#pragma once
#include <cstdint>
#include <memory>

#include "gmsx.h"

namespace gmsx {

// Host CSR with 64-bit offsets and 32-bit ids (the reference keeps row POINTERS, gapbs/graph.h:361-364;
// offsets are what the device wants and what the .sg file stores).
struct Csr {
    int64_t n = 0;
    int64_t nnz = 0;
    std::unique_ptr<int64_t[]> off;    // n + 1
    std::unique_ptr<int32_t[]> neigh;  // nnz
    bool directed = false;
};

bool worth_relabelling(const Csr &g);
int relabel_by_degree(const Csr &g, Csr &out);

}  // namespace gmsx

struct gmsx_csr {
    gmsx::Csr g;
    bool relabelled = false;
};
