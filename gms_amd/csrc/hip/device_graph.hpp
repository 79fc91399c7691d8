// Device-resident graph of libgmsx (gfx950).  Shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gmsx.h"

// Layout in HBM (all arrays hipMalloc'd once at upload, read-only afterwards):
//   off  int64[n+1], adj  int32[nnz]   full symmetric CSR, rows ascending       (the reference's CSRGraph rows)
//   doff int64[n+1], dadj int32[m]     degree-oriented DAG: N+(u) = { v in N(u) : (deg v, v) > (deg u, u) },
//                                      rows ascending; every undirected edge appears exactly once
//   order int32[n]                     vertices by decreasing d+ (work-sorted launch order, heavy first)
struct gmsx_graph {
    int64_t n = 0, nnz = 0, m = 0;
    int64_t *off = nullptr;
    int32_t *adj = nullptr;
    int64_t *doff = nullptr;
    int32_t *dadj = nullptr;
    int32_t *order = nullptr;
    int32_t max_dplus = 0;
    int32_t max_deg = 0;
    // positions in `order` where d+ drops below a threshold (host copy): bin_end[i] = #vertices with d+ >= kBinThr[i]
    static constexpr int kBins = 8;
    int64_t bin_end[kBins] = {0};
    unsigned long long *scratch = nullptr;  // device: a few u64 accumulators
    uint64_t alg_elements = 0;              // Σ_{u<v}(d_u+d_v), computed on the device at upload
    int64_t device_bytes = 0;
};

namespace gmsx {

// d+ thresholds of the launch bins (descending): a vertex with d+ >= kBinThr[i] and < kBinThr[i-1] is in bin i
static constexpr int32_t kBinThr[gmsx_graph::kBins] = {8192, 2048, 512, 128, 64, 16, 2, 0};

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;       // stream all launches go to
    hipStream_t own_stream = nullptr;   // created by gmsx_init
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int compute_units = 0;
};
Ctx &ctx();
int ensure_init();

#define GMSX_HIP(call)                                                    \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) {                                           \
            (void)hipGetLastError();                                      \
            return e_ == hipErrorOutOfMemory ? GMSX_ERR_DEVICE_MEM : GMSX_ERR_KERNEL; \
        }                                                                 \
    } while (0)

}  // namespace gmsx
