// tools/probes/kc4_mfma_probe.hip — the matrix-core k = 4 count (gms_amd/csrc/hip/kc4_mfma.hpp) alone: random strictly-lower-triangular bit matrices in the
// pool layout, the kernel's sum against the host's AND + popcount sum, and its time on NMAT matrices of one width.
// build: hipcc --offload-arch=gfx950 -O3 -Igms_amd/csrc/hip tools/probes/kc4_mfma_probe.hip -o /tmp/kc4_mfma_probe
// run:   /tmp/kc4_mfma_probe [d density nmat]...      (default: a few shapes)
#include "kc4_mfma.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

using namespace gmsx;

static unsigned long long host_count(const std::vector<uint32_t> &m, int d, int WS) {
    unsigned long long t = 0;
    const int W = (d + 31) >> 5;
    for (int i = 1; i < d; ++i)
        for (int j = 0; j < i; ++j)
            if (m[size_t(i) * WS + (j >> 5)] >> (j & 31) & 1u)
                for (int w = 0; w < W; ++w) t += __builtin_popcount(m[size_t(i) * WS + w] & m[size_t(j) * WS + w]);
    return t;
}

template <int T, int NT, int DBG = 0, int PF = 1>
static void run(const char *name, const uint32_t *dpoolm, size_t slot_words, const int32_t *dd, int nmat, unsigned long long want, double macs, int G = 1, int gridmul = 1) {
    unsigned long long *acc;
    CK(hipMalloc(&acc, 64 * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long got = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipMemset(acc, 0, 64 * 16 * 8));
        CK(hipEventRecord(e0));
        k_kc4_mfma<T, NT, DBG, PF><<<256 * gridmul, NT>>>(dpoolm, slot_words, dd, nmat, G, acc, 64, 16);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        std::vector<unsigned long long> h(64 * 16);
        CK(hipMemcpy(h.data(), acc, 64 * 16 * 8, hipMemcpyDeviceToHost));
        got = 0;
        for (auto x : h) got += x;
    }
    printf("  %-18s G %2d %8.3f ms   sum %llu %s   %.0f T useful bit-MAC/s\n", name, G, best, got, got == want ? "OK" : "WRONG", macs / best / 1e9);
    CK(hipFree(acc));
}

static void shape(int d, double dens, int nmat) {
    const int WS = kc4m_stride(d);
    const size_t slot_words = size_t(d) * WS;
    std::mt19937 rng(d * 7 + nmat);
    const int ndist = 3;  // distinct matrices, repeated over the pool
    std::vector<std::vector<uint32_t>> ms(ndist, std::vector<uint32_t>(slot_words, 0u));
    std::vector<unsigned long long> cnt(ndist);
    std::vector<int> dsz(ndist);
    for (int v = 0; v < ndist; ++v) {
        dsz[v] = d - v * 37 > 8 ? d - v * 37 : d;  // ragged widths in one pool: the stride follows d
        const int dv = dsz[v], WSv = kc4m_stride(dv);
        ms[v].assign(slot_words, 0xffffffffu);  // rows / words the BUILD does not write stay garbage
        for (int i = 0; i < dv; ++i) {
            for (int w = 0; w < WSv; ++w) ms[v][size_t(i) * WSv + w] = 0u;
            for (int j = 0; j < i; ++j)
                if ((rng() & 0xffff) < dens * 65536.0) ms[v][size_t(i) * WSv + (j >> 5)] |= 1u << (j & 31);
        }
        cnt[v] = host_count(ms[v], dv, WSv);
    }
    uint32_t *pool;
    int32_t *dd;
    CK(hipMalloc(&pool, slot_words * 4 * size_t(nmat)));
    CK(hipMalloc(&dd, size_t(nmat) * 4));
    std::vector<int32_t> hd(nmat);
    unsigned long long want = 0;
    double macs = 0;
    for (int q = 0; q < nmat; ++q) {
        const int v = q % ndist;
        CK(hipMemcpy(pool + size_t(q) * slot_words, ms[v].data(), slot_words * 4, hipMemcpyHostToDevice));
        hd[q] = dsz[v];
        want += cnt[v];
        macs += double(dsz[v]) * dsz[v] * dsz[v] / 6.0;
    }
    CK(hipMemcpy(dd, hd.data(), size_t(nmat) * 4, hipMemcpyHostToDevice));
    printf("d = %d density %.2f, %d matrices (%.1f MB each)\n", d, dens, nmat, slot_words * 4 / 1e6);
    for (int G : {1, 2, 4, 8, 16, 32}) run<2, 1024>("2x2 wg1024", pool, slot_words, dd, nmat, want, macs, G);
    for (int G : {1, 8}) run<2, 512>("2x2 wg512", pool, slot_words, dd, nmat, want, macs, G);
    for (int G : {1, 8}) run<2, 512>("2x2 wg512 x2", pool, slot_words, dd, nmat, want, macs, G, 2);
    for (int G : {1, 8}) run<4, 256>("4x4 wg256", pool, slot_words, dd, nmat, want, macs, G);
    run<2, 1024, 1>("2x2 wg1024 noepi", pool, slot_words, dd, nmat, want, macs, 8);
    run<2, 1024, 3>("2x2 wg1024 noexp", pool, slot_words, dd, nmat, want, macs, 8);
    CK(hipFree(pool));
    CK(hipFree(dd));
}

int main(int argc, char **argv) {
    if (argc >= 4) {
        for (int a = 1; a + 2 < argc; a += 3) shape(atoi(argv[a]), atof(argv[a + 1]), atoi(argv[a + 2]));
        return 0;
    }
    shape(70, 0.5, 7);
    shape(200, 0.3, 1024);
    shape(600, 0.5, 2048);
    shape(1200, 0.5, 1024);
    shape(1800, 0.5, 1024);
    shape(3000, 0.7, 512);
    return 0;
}
