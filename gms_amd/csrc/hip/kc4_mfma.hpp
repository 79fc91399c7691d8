// kc4_mfma.hpp — the k = 4 count of a dense local bit-matrix on the matrix cores (round 6).
//
// The reference's RecursiveStepCliqueCount at its last two levels (gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.h:5-17:
// `isect.intersect_count(N(v))` under `isect = cand.intersect(N(u))`) is, on the local adjacency bit-matrix L of a pivot (strictly lower triangular,
// d x d, row i = the members below member i that are its neighbours),
//
//        count  =  Σ_{i > j, L_ij = 1} | row_i ∩ row_j |  =  Σ_{i,j} L_ij · (L Lᵀ)_ij .
//
// Rounds 1-6 evaluated it pair by pair (AND + popcount per 32-bit word, one lane per pair or per neighbour).  For the pivots of d+ in the hundreds and
// thousands of an RMAT graph L is DENSE (the hubs' neighbourhoods are nearly cliques), so the sum is a masked bit-GEMM, and gfx950's matrix cores take
// 32 x 32 x 64 one-bit products per `v_mfma_scale_f32_32x32x64_f8f6f4` in 32 cycles — 2 048 per clock and SIMD against ~80 for the AND + popcount
// loop at its measured rate.  Exact: every bit becomes an fp4 (e2m1) element 0.5 / 1 / 2 on one side and 2 / 1 / 0.5 on the other, every product of two
// set bits is exactly 1.0, the accumulators are f32 sums of at most 8 192 ones (tools/probes/mfma_bits.hip checks the instruction against popcounts).
//
// Layout the kernel reads (written by the BUILD kernels of kclique.hip): matrix q at pool + q * slot_words; d = dpool[q]; row stride
// WSg = kc4m_stride(d) words (a multiple of 8: 32-byte rows, 16-byte loads); row i holds bits j < i only, words up to WSg zero-filled.  Rows >= d
// are never written: their loads are clamped to row d - 1 (finite garbage in accumulators that the mask then drops).
//
// A wave owns a (32 T) x (32 T) block (bi >= bj) of the product — T = 2 in the library: 2 x 2 accumulator tiles, four waves per SIMD.  Per chunk of 8 column words
// each lane loads 16 bytes of "its" row of every tile (lane l: row l & 31, words 4 (l >> 5) …) — four K-steps of fragments straight from global memory /
// L2, no LDS staging.  The J rows are the instruction's A operand, the I rows its B operand: an accumulator's COLUMN (the lane) is then a row i of L, so the
// mask bits L_ij of a tile are one word of the lane's own row.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace gmsx {

typedef int kc4m_v8i __attribute__((ext_vector_type(8)));
typedef float kc4m_v16f __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ int kc4m_stride(int d) { return (((d + 31) >> 5) + 7) & ~7; }  // words per row in the pool

// A-operand side: bit 4b + t of the word -> nibble b of dword t, as 0.5 / 1 / 2 / 2
__device__ __forceinline__ kc4m_v8i kc4m_exp_a(uint32_t w) {
    kc4m_v8i r = {};
    r[0] = int(w & 0x11111111u);
    r[1] = int(w & 0x22222222u);
    r[2] = int(w & 0x44444444u);
    r[3] = int((w >> 1) & 0x44444444u);
    return r;
}
// B-operand side: the same bits as 2 / 1 / 0.5 / 0.5
__device__ __forceinline__ kc4m_v8i kc4m_exp_b(uint32_t w) {
    kc4m_v8i r = {};
    r[0] = int((w << 2) & 0x44444444u);
    r[1] = int(w & 0x22222222u);
    const uint32_t s = w >> 2;
    r[2] = int(s & 0x11111111u);
    r[3] = int((s >> 1) & 0x11111111u);
    return r;
}

// DBG (tools/probes/kc4_mfma_probe.hip only; wrong sums): 1 = no mask epilogue, 2 = every chunk re-reads the first one (L1 / TA side), 3 = no expansion (raw words as fragments)
// Full 16-byte chunks: the four words a lane holds feed four K-steps, and K-step S takes bit class S of ALL four words (nibble position S of every nibble) —
// one weight per instruction (0.5 / 1 / 2 / 2), the same expansion on both sides (one AND per dword; class 3 a shift more), and the instruction's block scales
// (E8M0: 2^(byte - 127), one byte per lane) put the product back to 1: 2 x 2 for the 0.5s, 1 x 1, 0.5 x 0.5 for the 2s.  20 VALU per tile role and chunk
// against 4 x (5 + 7) / 2 = 24 with the mixed-weight expansions above, which the short tails keep.
template <int S>
__device__ __forceinline__ kc4m_v8i kc4m_class(uint4 w) {
    kc4m_v8i r = {};
    if constexpr (S < 3) {
        constexpr uint32_t M = 0x11111111u << S;
        r[0] = int(w.x & M); r[1] = int(w.y & M); r[2] = int(w.z & M); r[3] = int(w.w & M);
    } else {
        r[0] = int((w.x >> 1) & 0x44444444u); r[1] = int((w.y >> 1) & 0x44444444u); r[2] = int((w.z >> 1) & 0x44444444u); r[3] = int((w.w >> 1) & 0x44444444u);
    }
    return r;
}
template <int S>
__device__ __forceinline__ int kc4m_class_scale() { return S == 0 ? int(0x80808080u) : S == 1 ? 0x7f7f7f7f : 0x7e7e7e7e; }

template <int T, int DBG = 0, int PF = 1>  // tiles per block side: block = (32 T) x (32 T); PF: chunks of fragment loads in flight behind the one being multiplied
struct Kc4mBlock {
    // one block (bi >= bj) of matrix `m` (row stride WS words, d rows); returns Σ L_ij (L Lᵀ)_ij over the block
    static __device__ __forceinline__ uint32_t run(const uint32_t *__restrict__ m, int WS, int d, int bi, int bj, int lane) {
        const int r = lane & 31, h = lane >> 5;
        const int I0 = 32 * T * bi, J0 = 32 * T * bj;
        uint32_t offJ[T], offI[T];  // word offsets of the lane's rows (a matrix is far below 2^32 words)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            offJ[t] = uint32_t(min(J0 + 32 * t + r, d - 1)) * uint32_t(WS) + 4u * h;
            offI[t] = uint32_t(min(I0 + 32 * t + r, d - 1)) * uint32_t(WS) + 4u * h;
        }
        kc4m_v16f acc[T][T];
        // columns k < J0 + 32 T only (row j has no bit at or above j): words [0, nw), nw = T (bj + 1) — chunks of 8 words (16 bytes per lane: lane half h holds
        // words 4 h … 4 h + 3, K-step s multiplies word 4 h + s of every row), then 4 words (8 bytes per lane, two K-steps) and 2 words (one K-step): no
        // K-step multiplies columns that cannot hold a bit.  Which 64 columns a K-step covers is free as long as both operands agree.
        const int nw = T * (bj + 1), nfull = nw >> 3;
        // one K-step: fragments from one word per lane and tile; ZERO: it starts the accumulators (C = 0 in the instruction)
        auto kstep = [&](auto zero_tag, const uint32_t (&xj)[T], const uint32_t (&xi)[T]) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            kc4m_v8i fa[T], fb[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if constexpr (DBG == 3) {
                    fa[t] = kc4m_v8i{int(xj[t]), int(xj[t]), int(xi[t]), int(xi[t]), 0, 0, 0, 0};
                    fb[t] = kc4m_v8i{int(xi[t]), int(xj[t]), int(xi[t]), int(xj[t]), 0, 0, 0, 0};
                } else {
                    fa[t] = kc4m_exp_a(xj[t]);
                    fb[t] = kc4m_exp_b(xi[t]);
                }
            }
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    kc4m_v16f c0;
                    if constexpr (ZERO) {
#pragma unroll
                        for (int g = 0; g < 16; ++g) c0[g] = 0.f;
                    } else {
                        c0 = acc[a][b];
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[b], fb[a], c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
        };
        // one K-step of a full chunk: bit class S of the four words of every lane (kc4m_class)
        auto cstep = [&](auto zero_tag, auto s_tag, const uint4 (&cj)[T], const uint4 (&ci)[T]) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            constexpr int S = decltype(s_tag)::value;
            kc4m_v8i fa[T], fb[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if constexpr (DBG == 3) {
                    fa[t] = kc4m_v8i{int(cj[t].x), int(cj[t].y), int(cj[t].z), int(cj[t].w), 0, 0, 0, 0};
                    fb[t] = kc4m_v8i{int(ci[t].x), int(ci[t].y), int(ci[t].z), int(ci[t].w), 0, 0, 0, 0};
                } else {
                    fa[t] = kc4m_class<S>(cj[t]);
                    fb[t] = kc4m_class<S>(ci[t]);
                }
            }
            const int sc = kc4m_class_scale<S>();
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    kc4m_v16f c0;
                    if constexpr (ZERO) {
#pragma unroll
                        for (int g = 0; g < 16; ++g) c0[g] = 0.f;
                    } else {
                        c0 = acc[a][b];
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[b], fb[a], c0, 4, 4, 0, sc, 0, sc);
                }
        };
        // the short tail's words are on their way from the start
        const uint32_t tb = uint32_t(nfull) * 8u - 4u * h;  // (offJ / offI carry + 4 h for the 16-byte chunks; the sums below wrap in 32 bits BEFORE they meet the pointer)
        uint2 t2j[T], t2i[T];
        uint32_t t1j[T], t1i[T];
        if (nw & 4) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                t2j[t] = *reinterpret_cast<const uint2 *>(m + (offJ[t] + tb + 2u * h));
                t2i[t] = *reinterpret_cast<const uint2 *>(m + (offI[t] + tb + 2u * h));
            }
        }
        if (nw & 2) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                t1j[t] = m[uint32_t(offJ[t] + tb + uint32_t(nw & 4) + h)];
                t1i[t] = m[uint32_t(offI[t] + tb + uint32_t(nw & 4) + h)];
            }
        }
        if (nfull > 0) {
            uint4 wj[T], wi[T], mj[T], mi[T];  // the chunk being multiplied; PF = 2: the one behind it
#pragma unroll
            for (int t = 0; t < T; ++t) {
                wj[t] = *reinterpret_cast<const uint4 *>(m + offJ[t]);
                wi[t] = *reinterpret_cast<const uint4 *>(m + offI[t]);
            }
            if constexpr (PF == 2) {
                const uint32_t c1 = uint32_t(min(1, nfull - 1)) * 8u;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    mj[t] = *reinterpret_cast<const uint4 *>(m + offJ[t] + c1);
                    mi[t] = *reinterpret_cast<const uint4 *>(m + offI[t] + c1);
                }
            }
            for (int c = 0; c < nfull; ++c) {
                uint4 nj[T], ni[T];
                const uint32_t cn = DBG == 2 ? 0u : uint32_t(min(c + PF, nfull - 1)) * 8u;  // (the last trips reload the last chunk: no branch around the loads)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    nj[t] = *reinterpret_cast<const uint4 *>(m + offJ[t] + cn);
                    ni[t] = *reinterpret_cast<const uint4 *>(m + offI[t] + cn);
                }
                // (the scheduler sinks these loads to the end of the trip — their results are only copied there — so a chunk waits for its own loads; pinning
                //  them here with a sched_barrier, one or two chunks ahead, was SLOWER (0.96 against 0.81 ms, 1 024 matrices of d+ = 1 800): the loop is bound by
                //  the bytes its fragment loads pull through the fabric, not by their latency — see the teams below)
                if (c == 0) cstep(std::true_type{}, std::integral_constant<int, 0>{}, wj, wi);
                else cstep(std::false_type{}, std::integral_constant<int, 0>{}, wj, wi);
                cstep(std::false_type{}, std::integral_constant<int, 1>{}, wj, wi);
                cstep(std::false_type{}, std::integral_constant<int, 2>{}, wj, wi);
                cstep(std::false_type{}, std::integral_constant<int, 3>{}, wj, wi);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if constexpr (PF == 2) {
                        wj[t] = mj[t];
                        wi[t] = mi[t];
                        mj[t] = nj[t];
                        mi[t] = ni[t];
                    } else {
                        wj[t] = nj[t];
                        wi[t] = ni[t];
                    }
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
        }
        if (nw & 4) {
            uint32_t xj[T], xi[T];
#pragma unroll
            for (int t = 0; t < T; ++t) { xj[t] = t2j[t].x; xi[t] = t2i[t].x; }
            kstep(std::false_type{}, xj, xi);
#pragma unroll
            for (int t = 0; t < T; ++t) { xj[t] = t2j[t].y; xi[t] = t2i[t].y; }
            kstep(std::false_type{}, xj, xi);
        }
        if (nw & 2) kstep(std::false_type{}, t1j, t1i);
        // the mask: accumulator (a, b), register g of lane (n = lane & 31, hh = lane >> 5) is the pair  i = I0 + 32 a + n,  j = J0 + 32 b + (g & 3) + 8 (g >> 2) + 4 hh,
        // and L_ij is bit (j & 31) of word (J0 / 32 + b) of row i — the lane's own row
        float sum = 0.f;  // at most (32 T)^2 x 8 192 / 64 per lane: exact in f32
        if constexpr (DBG == 1) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) sum += acc[a][b][0];
            return uint32_t(sum);
        }
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int i = I0 + 32 * a + r;
            uint32_t mw[T];
            const uint32_t *mp = m + size_t(min(i, d - 1)) * WS + T * bj;
            if constexpr (T == 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(mp);
                mw[0] = q.x; mw[1] = q.y; mw[2] = q.z; mw[3] = q.w;
            } else {
#pragma unroll
                for (int b = 0; b < T; ++b) mw[b] = mp[b];
            }
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const uint32_t w = (i < d ? mw[b] : 0u) >> (4 * h);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int keep = __builtin_amdgcn_sbfe(int(w), (g & 3) + 8 * (g >> 2), 1);  // 0 or -1
                    sum += __int_as_float(__float_as_int(acc[a][b][g]) & keep);
                }
            }
        }
        return uint32_t(sum);
    }
};

// TEAMS.  A block re-reads its 2 x 32 T rows once per K-chunk and a matrix is re-read ~nb / 2 times in all.  With one matrix per workgroup the 32 CUs of an XCD
// are on 32 matrices at once — 16 MB at d+ = 1 800 against 4 MB of L2 — and the fragment loads run at what the fabric behind the L2 delivers (5.8 TB/s; with
// every chunk re-reading the first one the kernel was a third faster).  So G workgroups that share an XCD — the dispatcher deals consecutive workgroup ids
// round the XCDs, blockIdx mod 8 labels the ones that share one; a wrong guess costs speed, never the count — form a team on ONE matrix: team k takes the
// matrices k, k + nteams, … of the launch (they are sorted by width: equal shares), member g the blocks g, g + G, … of each, its 16 waves through a ticket in
// LDS.  No global atomic (a returning atomic on one address completes every ~11 ns chip-wide: per-XCD block tickets were tried and cost more than the loads).
// NT = 1 024: four waves per SIMD at 128 registers hide the most of the loads (512 threads: + 5 … 10 %).
template <int T, int NT, int DBG = 0, int PF = 1>
__global__ __launch_bounds__(NT) void k_kc4_mfma(const uint32_t *__restrict__ pool, size_t slot_words, const int32_t *__restrict__ dpool, int nmat, int G,
                                                unsigned long long *__restrict__ acc, int acc_slots, int acc_stride) {
    __shared__ int s_blk;
    __shared__ unsigned long long red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // gridDim.x is a multiple of 8 G (host): teams = gridDim.x / G
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3, g = y % G, team = x + 8 * (y / G), nteams = int(gridDim.x) / G;
    unsigned long long total = 0;
    for (int q = team; q < nmat; q += nteams) {
        if (tid == 0) s_blk = 0;
        __syncthreads();
        const int d = dpool[q];
        const uint32_t *m = pool + size_t(q) * slot_words;
        const int WS = kc4m_stride(d);
        const int nb = (d + 32 * T - 1) / (32 * T), ntri = nb * (nb + 1) / 2;
        while (true) {
            int task = 0;
            if (lane == 0) task = atomicAdd(&s_blk, 1);
            task = __builtin_amdgcn_readfirstlane(task) * G + g;
            if (task >= ntri) break;
            const int tt = ntri - 1 - task;  // far blocks first: the long ones
            int bi = int((__builtin_sqrtf(8.0f * float(tt) + 1.0f) - 1.0f) * 0.5f);
            while (bi * (bi + 1) / 2 > tt) --bi;
            while ((bi + 1) * (bi + 2) / 2 <= tt) ++bi;
            const int bj = tt - bi * (bi + 1) / 2;
            total += Kc4mBlock<T, DBG, PF>::run(m, WS, d, bi, bj, lane);
        }
        __syncthreads();  // every wave is done with s_blk
    }
    for (int s = 32; s > 0; s >>= 1) total += __shfl_down(total, s);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < NT / 64; ++w) t += red[w];
        if (t) atomicAdd(&acc[(blockIdx.x & (acc_slots - 1)) * acc_stride], t);
    }
}

}  // namespace gmsx
