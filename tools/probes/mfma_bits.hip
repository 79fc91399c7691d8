// tools/probes/mfma_bits.hip — can the k = 4 count  Σ_{i>j} L_ij (L Lᵀ)_ij  of a dense bit matrix run on the matrix cores?
// (1) exactness: C[i][j] = popc(rowA_i & rowB_j) from bit rows expanded in registers to i8 (v_mfma_i32_32x32x32_i8) and to fp4 e2m1
//     (v_mfma_scale_f32_32x32x64_f8f6f4, scales 1.0), against the host's popcounts — asymmetric random rows;
// (2) rate: a TI x TJ register-blocked loop (fragments from LDS words, expanded per K-step) on every CU, in bit-MACs per second.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bits.hip -o /tmp/mfma_bits ; run: /tmp/mfma_bits
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

// ---- expansions: every product of an A element and the B element of the same bit is exactly 1 ----
// i8: lane half h takes the bits 4h+t (+8b) of the word: dword t = bytes b = 0..3 of value bit(8b + 4h + t)
__device__ __forceinline__ v4i exp_i8(uint32_t w, int h) {
    const uint32_t s = w >> (4 * h);
    v4i r;
    r[0] = int(s & 0x01010101u);
    r[1] = int((s >> 1) & 0x01010101u);
    r[2] = int((s >> 2) & 0x01010101u);
    r[3] = int((s >> 3) & 0x01010101u);
    return r;
}
// fp4 e2m1: 0b0001 = 0.5, 0b0010 = 1.0, 0b0100 = 2.0.  A side: bit 4b+t of the word in nibble b of dword t as 0.5 / 1 / 2 / 2
__device__ __forceinline__ v8i exp_fp4_a(uint32_t w) {
    v8i r = {};
    r[0] = int(w & 0x11111111u);
    r[1] = int(w & 0x22222222u);
    r[2] = int(w & 0x44444444u);
    r[3] = int((w >> 1) & 0x44444444u);
    return r;
}
// B side: the same bits as 2 / 1 / 0.5 / 0.5
__device__ __forceinline__ v8i exp_fp4_b(uint32_t w) {
    v8i r = {};
    r[0] = int((w << 2) & 0x44444444u);
    r[1] = int(w & 0x22222222u);
    const uint32_t s = w >> 2;
    r[2] = int(s & 0x11111111u);
    r[3] = int((s >> 1) & 0x11111111u);
    return r;
}

// ---- (1) exactness ----
// rows: a[32] words (i8, K = 32) / a[32][2] words (fp4, K = 64)
__global__ void k_check_i8(const uint32_t *a, const uint32_t *b, int *out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(exp_i8(a[r], h), exp_i8(b[r], h), c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) out[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];  // row = (reg&3) + 8 (reg>>2) + 4 (lane>>5), col = lane & 31
}
template <int SCALE_MODE>
__global__ void k_check_fp4(const uint32_t *a, const uint32_t *b, float *out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v16f c = {};
    constexpr int sc = SCALE_MODE == 0 ? 0x7f7f7f7f : 0;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(exp_fp4_a(a[2 * r + h]), exp_fp4_b(b[2 * r + h]), c, 4, 4, 0, sc, 0, sc);
    for (int g = 0; g < 16; ++g) out[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

// ---- (2) rate ----
// A workgroup holds ROWS bit rows of KW words in LDS (stride KW + 1); every wave accumulates a (32 TI) x (32 TJ) block over the KW words, REP times.
template <int MODE /* 0 i8, 1 fp4 */, int TI, int TJ, int NT>
__global__ __launch_bounds__(NT) void k_rate(const uint32_t *src, int KW, int rep, float *sink) {
    extern __shared__ uint32_t lds[];
    const int ROWS = 32 * (TI + TJ);
    const int S = KW + 1;
    for (int x = threadIdx.x; x < ROWS * S; x += NT) lds[x] = src[x % 4096] * 2654435761u + x;
    __syncthreads();
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5;
    float total = 0.f;
    for (int it = 0; it < rep; ++it) {
        if constexpr (MODE == 0) {
            v16i acc[TI][TJ] = {};
            for (int c = 0; c < KW; ++c) {
                v4i fa[TI], fb[TJ];
#pragma unroll
                for (int a = 0; a < TI; ++a) fa[a] = exp_i8(lds[(32 * a + r) * S + c], h);
#pragma unroll
                for (int b = 0; b < TJ; ++b) fb[b] = exp_i8(lds[(32 * (TI + b) + r) * S + c], h);
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
#pragma unroll
                    for (int g = 0; g < 16; ++g) total += float(acc[a][b][g]);
        } else {
            v16f acc[TI][TJ] = {};
            for (int c = 0; c + 1 < KW; c += 2) {
                v8i fa[TI], fb[TJ];
#pragma unroll
                for (int a = 0; a < TI; ++a) fa[a] = exp_fp4_a(lds[(32 * a + r) * S + c + h]);
#pragma unroll
                for (int b = 0; b < TJ; ++b) fb[b] = exp_fp4_b(lds[(32 * (TI + b) + r) * S + c + h]);
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
#pragma unroll
                    for (int g = 0; g < 16; ++g) total += acc[a][b][g];
        }
    }
    if (total == 12345.678f) sink[0] = total;
}

template <int MODE, int TI, int TJ, int NT>
static void rate(const char *name, const uint32_t *dsrc, float *dsink) {
    const int KW = 64, rep = 200, grid = 256 * 4;
    const size_t ldsb = size_t(32 * (TI + TJ)) * (KW + 1) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rate<MODE, TI, TJ, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsb)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_rate<MODE, TI, TJ, NT><<<grid, NT, ldsb>>>(dsrc, KW, 2, dsink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<MODE, TI, TJ, NT><<<grid, NT, ldsb>>>(dsrc, KW, rep, dsink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = double(grid) * (NT / 64) * rep * double(TI * TJ) * 32.0 * 32.0 * (KW * 32.0);
    printf("rate %-28s wg %4d lds %6zu B: %8.3f ms  %7.1f T bit-MAC/s  (%.0f per clk per SIMD at 2.4 GHz)\n", name, NT, ldsb, ms, macs / ms / 1e9,
           macs / (ms * 1e-3) / 1024.0 / 2.4e9);
}

int main() {
    std::vector<uint32_t> a(64), b(64);
    srand(7);
    for (auto &x : a) x = uint32_t(rand()) ^ (uint32_t(rand()) << 11);
    for (auto &x : b) x = uint32_t(rand()) ^ (uint32_t(rand()) << 13);
    uint32_t *da, *db;
    int *oi;
    float *of;
    CK(hipMalloc(&da, 256));
    CK(hipMalloc(&db, 256));
    CK(hipMalloc(&oi, 4096));
    CK(hipMalloc(&of, 4096));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    std::vector<int> hi(1024);
    std::vector<float> hf(1024);
    k_check_i8<<<1, 64>>>(da, db, oi);
    CK(hipMemcpy(hi.data(), oi, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) bad += hi[i * 32 + j] != __builtin_popcount(a[i] & b[j]);
    printf("i8  32x32x32: %d wrong of 1024\n", bad);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) k_check_fp4<0><<<1, 64>>>(da, db, of);
        else k_check_fp4<1><<<1, 64>>>(da, db, of);
        CK(hipMemcpy(hf.data(), of, 4096, hipMemcpyDeviceToHost));
        bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                bad += hf[i * 32 + j] != float(__builtin_popcount(a[2 * i] & b[2 * j]) + __builtin_popcount(a[2 * i + 1] & b[2 * j + 1]));
        printf("fp4 32x32x64 (scale operand %s): %d wrong of 1024   [0][0] = %g want %d\n", mode == 0 ? "0x7f7f7f7f" : "0", bad, hf[0],
               __builtin_popcount(a[0] & b[0]) + __builtin_popcount(a[1] & b[1]));
    }
    uint32_t *dsrc;
    float *dsink;
    CK(hipMalloc(&dsrc, 4096 * 4));
    CK(hipMalloc(&dsink, 64));
    std::vector<uint32_t> src(4096);
    for (auto &x : src) x = uint32_t(rand());
    CK(hipMemcpy(dsrc, src.data(), 4096 * 4, hipMemcpyHostToDevice));
    rate<0, 2, 2, 1024>("i8 2x2", dsrc, dsink);
    rate<0, 2, 2, 512>("i8 2x2", dsrc, dsink);
    rate<0, 4, 2, 512>("i8 4x2", dsrc, dsink);
    rate<1, 2, 2, 1024>("fp4 2x2", dsrc, dsink);
    rate<1, 2, 2, 512>("fp4 2x2", dsrc, dsink);
    rate<1, 4, 2, 512>("fp4 4x2", dsrc, dsink);
    rate<1, 4, 2, 256>("fp4 4x2", dsrc, dsink);
    rate<1, 2, 1, 1024>("fp4 2x1", dsrc, dsink);
    rate<1, 1, 1, 1024>("fp4 1x1", dsrc, dsink);
    return 0;
}
