// Probe (MI355X): LDS-DMA staging as the persistent triangle kernel uses it — per-lane global source, wave-uniform LDS base in M0,
// exec-masked lanes, sources that are only 4- / 8-byte aligned; builtin and inline-asm forms, dword and dwordx4.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ void glds4_builtin(const uint32_t *src, uint32_t *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds, 4, 0, 0);
}
__device__ __forceinline__ void glds4_asm(const uint32_t *src, uint32_t *lds) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds16_asm(const uint32_t *src, uint32_t *lds) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int MODE>
__global__ void k(const uint32_t *src, uint32_t *out, int ndw) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) buf[i] = 0xdeadbeefu;
    __syncthreads();
    if (MODE < 2) {
        for (int base = wave * 64; base < ndw; base += 256) {
            const int i = base + lane;
            if (i < ndw) {
                if (MODE == 0) glds4_builtin(src + i, buf + base);
                else glds4_asm(src + i, buf + base);
            }
        }
    } else {  // 16 bytes per lane: ndw rounded down to whole 16-byte pieces
        for (int base = wave * 256; base < ndw; base += 1024) {
            const int i = base + 4 * lane;
            if (i + 3 < ndw) glds16_asm(src + i, buf + base);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = buf[i];
}
int main() {
    uint32_t h[1032], *d, *o;
    for (int i = 0; i < 1032; i++) h[i] = i * 3 + 1;
    (void)hipMalloc(&d, 4096 + 64); (void)hipMalloc(&o, 4096);
    int bad = 0;
    for (int mis : {0, 4, 8, 12})
        for (int mode = 0; mode < 3; mode++)
            for (int ndw : {1, 4, 63, 64, 65, 116, 232, 500, 1000, 1024}) {
                (void)hipMemcpy((char *)d + mis, h, 4096, hipMemcpyHostToDevice);
                const uint32_t *s = (const uint32_t *)((char *)d + mis);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, s, o, ndw);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, s, o, ndw);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, s, o, ndw);
                uint32_t r[1024];
                (void)hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
                const int lim = mode == 2 ? (ndw & ~3) : ndw;
                int b = 0;
                for (int i = 0; i < 1024; i++) {
                    const uint32_t want = i < lim ? h[i] : 0xdeadbeefu;
                    if (r[i] != want) { if (b < 2) printf("  mis %d mode %d ndw %d i %d got %x want %x\n", mis, mode, ndw, i, r[i], want); b++; }
                }
                if (b) printf("mis %d mode %d ndw %d: %d bad\n", mis, mode, ndw, b);
                bad += b;
            }
    printf("glds probe: bad=%d\n", bad);
    return bad != 0;
}
