// LDS DMA on gfx950: where global_load_lds_dwordx4 / _dword put a lane's data, and that s_waitcnt vmcnt(N) with N = the
// number of NEWER DMA loads is enough to read an older one (in-order return), with ordinary loads in between.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void *gp, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma4(const void *gp, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gp), "s"(lds_byte_addr) : "memory");
}
__global__ void k(const uint32_t *__restrict__ g, uint32_t *__restrict__ out, int n_iter) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[4][256 + 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)&ring[0][0];
    // test 1: placement.  lane l loads 16 bytes from g + 4 * (63 - l) (reversed), slot 0; 4 bytes from g + 1000 + 2 l, behind it
    dma16(g + 4 * (63 - lane), base);
    dma4(g + 1000 + 2 * lane, base + 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < 5; ++i) out[lane * 5 + i] = ring[0][lane + 64 * i];
    // test 2: a ring three deep, waits with exact counts, an ordinary load in every iteration
    uint32_t acc = 0, bad = 0;
    const uint32_t *p = g + 4096;
    dma16(p + 4 * lane, base + 0 * 1280);
    dma16(p + 256 + 4 * lane, base + 1 * 1280);
    for (int i = 0; i < n_iter; ++i) {
        dma16(p + 256 * (i + 2) + 4 * lane, base + ((i + 2) % 3) * 1280);
        const uint32_t w = g[(i * 64 + lane) & 4095];  // compiler-tracked
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // newer than slot i's DMA: DMA i+1, DMA i+2, the ordinary load
        __builtin_amdgcn_wave_barrier();
        const uint32_t v = ring[0][(i % 3) * 320 + 4 * (lane ^ 1)];
        bad += (v != (uint32_t)(4096 + 256 * i + 4 * (lane ^ 1))) ? 1u : 0u;
        acc += w;
        __builtin_amdgcn_wave_barrier();
    }
    out[320 + lane] = bad;
    out[384 + lane] = acc;
}
int main() {
    std::vector<uint32_t> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
    uint32_t *g, *o;
    hipMalloc(&g, h.size() * 4);
    hipMalloc(&o, 4096);
    hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(o, 0xff, 4096);
    const int n_iter = 10000;
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, g, o, n_iter);
    hipDeviceSynchronize();
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad1 = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 5; ++i) {
            const uint32_t got = r[l * 5 + i], idx = l + 64 * i;  // ring word idx
            uint32_t want;
            if (idx < 256) want = 4 * (63 - idx / 4) + idx % 4; else want = 1000 + 2 * (idx - 256);
            if (got != want) { if (bad1 < 8) printf("placement: word %u got %u want %u\n", idx, got, want); ++bad1; }
        }
    uint32_t bad2 = 0;
    for (int l = 0; l < 64; ++l) bad2 += r[320 + l];
    printf("placement mismatches %d; ring mismatches %u of %d\n", bad1, bad2, 64 * n_iter);
    return (bad1 || bad2) ? 1 : 0;
}
