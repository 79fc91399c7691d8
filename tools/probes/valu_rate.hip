// Probe (MI355X): cycles of SIMD time per wave64 VALU instruction, by opcode, with 8 waves per SIMD (2048 workgroups of 256 threads,
// every CU full) and independent chains — is integer VALU 2 or 4 cycles per wave-instruction?  Reports ns per wave-instruction per SIMD
// and the same in cycles at the clock measured with s_memtime / s_memrealtime.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, unsigned long long *clk) {
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 77u, d = b + 1234567u;
    float fa = a * 1e-9f, fb = b * 1e-9f, fc = 1.0001f, fd = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(fa), "+v"(fb) : "v"(fc), "v"(fd));) }
        if (OP == 1) { REP16(asm volatile("v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 2) { REP16(asm volatile("v_bfe_u32 %0, %0, %2, 1\n\tv_bfe_u32 %1, %1, %3, 1" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 3) { REP16(asm volatile("v_lshrrev_b32 %0, 5, %0\n\tv_lshrrev_b32 %1, 5, %1" : "+v"(a), "+v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 5) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 6) { REP16(asm volatile("v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %3, %1" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 7) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %2\n\tv_lshl_or_b32 %1, %1, 1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 8) { REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (OP == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&fa) : "v"(*(double *)&fc));) }  // 1 instr
        if (OP == 10) { REP16(asm volatile("s_and_b32 s20, s20, s21\n\ts_add_u32 s22, s22, s23" ::: "s20", "s22", "scc");) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + uint32_t(fa + fb);
}
int main() {
    uint32_t *out; unsigned long long *clk, h[2];
    const int blocks = 256 * 8, iters = 4000;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, 16);
    const char *names[] = {"v_fma_f32", "v_and_b32", "v_bfe_u32", "v_lshrrev_b32", "v_add_u32", "v_mul_lo_u32", "v_bcnt_u32_b32", "v_lshl_or_b32", "v_cndmask_b32", "v_pk_fma_f32", "s_and/s_add"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int op = 0; op <= 10; ++op) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            switch (op) {
#define L(n) case n: hipLaunchKernelGGL(k<n>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10)
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = double(h[0]) / (double(h[1]) * 10.0);  // s_memrealtime ticks at 100 MHz
        const double per_wave = (op == 9 ? 16.0 : 32.0) * iters;          // instructions per wave
        const double waves_per_simd = blocks * 4.0 / 1024.0;
        const double ns = best * 1e6 / (per_wave * waves_per_simd);
        printf("%-16s %8.3f ms  clock %.2f GHz  %.3f ns = %.2f cycles per wave-instruction per SIMD\n", names[op], best, ghz, ns, ns * ghz);
    }
    return 0;
}
