// gmsx_hbm_read_probe: what a read-only stream reaches on THIS device — the measured ceiling the roofline fractions of bench.py are also quoted
// against (`frac_of_measured_stream_ceiling`), next to the 8 TB/s specification peak.  A buffer far larger than the 256 MB Infinity Cache is read
// with 16-byte loads, four independent ones in flight per lane, in the in-order sweep of a grid-stride loop (the access pattern of the counting
// kernels' row streams at their best); the time is taken with HIP events on the library's stream.  Diagnostic only: no reference counterpart.
#include "device_graph.hpp"

namespace gmsx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_read_probe(const u32x4 *__restrict__ buf, int64_t n16, unsigned long long *sink) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(buf + i), b = __builtin_nontemporal_load(buf + i + stride), c = __builtin_nontemporal_load(buf + i + 2 * stride),
                    d = __builtin_nontemporal_load(buf + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(buf + i);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9E3779B1u) atomicAdd(sink, 1ull);  // keeps the loads alive; the buffer is zero-filled, so it never fires
}

}  // namespace gmsx

using namespace gmsx;

extern "C" int gmsx_hbm_read_probe(int64_t bytes, int iterations, double *gbps) {
    return gmsx::guard([&]() -> int {
        if (!gbps || bytes < (int64_t(1) << 20) || iterations < 1 || iterations > 100000) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;
        Ctx &c = ctx();
        hipStream_t s = c.stream;
        const int64_t n16 = bytes / 16;
        uint4 *buf = nullptr;
        unsigned long long *sink = nullptr;
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&buf), size_t(n16) * 16));
        struct Free { void *p; ~Free() { (void)hipFree(p); } } f0{buf};
        GMSX_HIP(hipMalloc(reinterpret_cast<void **>(&sink), 8));
        Free f1{sink};
        GMSX_HIP(hipMemsetAsync(buf, 0, size_t(n16) * 16, s));
        GMSX_HIP(hipMemsetAsync(sink, 0, 8, s));
        const int cus = c.compute_units > 0 ? c.compute_units : 256;
        const dim3 grid(unsigned(cus * 4)), block(512);  // 2048 lanes per CU = every wave slot, four 16-byte loads each in flight
        hipLaunchKernelGGL(k_read_probe, grid, block, 0, s, reinterpret_cast<const u32x4 *>(buf), n16, sink);  // warm-up (code object, TLB)
        GMSX_HIP(hipEventRecord(c.ev[0], s));
        for (int it = 0; it < iterations; ++it) hipLaunchKernelGGL(k_read_probe, grid, block, 0, s, reinterpret_cast<const u32x4 *>(buf), n16, sink);
        GMSX_HIP(hipEventRecord(c.ev[1], s));
        GMSX_HIP(hipEventSynchronize(c.ev[1]));
        GMSX_HIP(hipGetLastError());
        float ms = 0.f;
        GMSX_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        if (!(ms > 0.f)) return GMSX_ERR_KERNEL;
        *gbps = double(n16) * 16.0 * iterations / (double(ms) * 1e-3) / 1e9;
        return GMSX_OK;
    });
}
