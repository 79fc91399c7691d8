// tools/probes/h2d_probe.hip — how should gmsx_graph_upload move 8.9 GB of caller-owned PAGEABLE memory to the device?
//   hipcc --offload-arch=gfx950 -O2 -fopenmp tools/probes/h2d_probe.hip -o /tmp/h2d_probe && /tmp/h2d_probe [GiB]
// (a) one hipMemcpyAsync from the pageable buffer (round 4), (b) hipHostRegister + copy + unregister, (c) two pinned staging buffers filled by an
// OpenMP memcpy while the other is on the wire.  Each on a fresh, first-touched buffer and again on the same (warm) one.
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const size_t bytes = size_t((argc > 1 ? std::atof(argv[1]) : 8.0) * (1ull << 30));
    char *dev = nullptr;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s, s2;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&s2));
    auto fresh = [&]() {
        char *p = static_cast<char *>(std::malloc(bytes));
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)bytes; i += 4096) p[i] = char(i >> 12);
        return p;
    };
    std::printf("threads %d, %.1f GiB\n", omp_get_max_threads(), double(bytes) / (1ull << 30));
    for (int variant = 0; variant < 3; ++variant) {
        char *host = fresh();
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            if (variant == 0) {
                CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
            } else if (variant == 1) {
                CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
                const double t1 = now();
                CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
                const double t2 = now();
                CK(hipHostUnregister(host));
                std::printf("    register %.3f s, copy %.3f s (%.1f GB/s), unregister %.3f s\n", t1 - t0, t2 - t1, bytes / (t2 - t1) / 1e9, now() - t2);
            } else {
                const size_t chunk = 64ull << 20;
                static char *pin[2] = {nullptr, nullptr};
                static hipEvent_t ev[2];
                if (!pin[0]) {
                    CK(hipHostMalloc(reinterpret_cast<void **>(&pin[0]), chunk, hipHostMallocDefault));
                    CK(hipHostMalloc(reinterpret_cast<void **>(&pin[1]), chunk, hipHostMallocDefault));
                    CK(hipEventCreate(&ev[0]));
                    CK(hipEventCreate(&ev[1]));
                }
                int b = 0;
                bool used[2] = {false, false};
                for (size_t off = 0; off < bytes; off += chunk, b ^= 1) {
                    const size_t n = std::min(chunk, bytes - off);
                    if (used[b]) CK(hipEventSynchronize(ev[b]));
#pragma omp parallel
                    {
                        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
                        const size_t per = (n + nt - 1) / nt, lo = std::min(n, per * t), hi = std::min(n, lo + per);
                        std::memcpy(pin[b] + lo, host + off + lo, hi - lo);
                    }
                    CK(hipMemcpyAsync(dev + off, pin[b], n, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[b], s));
                    used[b] = true;
                }
                CK(hipStreamSynchronize(s));
            }
            const double dt = now() - t0;
            std::printf("%s %s: %.3f s = %.1f GB/s\n", variant == 0 ? "(a) pageable hipMemcpyAsync" : variant == 1 ? "(b) hipHostRegister + copy" : "(c) pinned double buffer + OpenMP memcpy",
                        rep == 0 ? "first touch by the runtime" : "again", dt, bytes / dt / 1e9);
        }
        std::free(host);
    }
    return 0;
}
