// Micro-benchmark (round 4): issue interval of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD by the register class of
// its operands (A/B in VGPRs or AGPRs, C/D in VGPRs or AGPRs), MFMAs only, four accumulator tiles in rotation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define MF(ACC, AC, CC, F) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+" CC(ACC) : AC(a[F]), "v"(b[F]));
template <int KIND>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters, unsigned long long *clk) {
    __shared__ float one_block_per_cu[24 * 1024];
    one_block_per_cu[threadIdx.x] = 0.0f;
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = acc3[i] = threadIdx.x * 1e-3f + i;
    bf16x8 a[6], b[6];
    for (int f = 0; f < 6; ++f)
        for (int i = 0; i < 8; ++i) a[f][i] = (short)(0x3f80 + f), b[f][i] = (short)(0x3f80 + threadIdx.x % 7 + f);
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) { MF(acc0, "a", "v", 0) MF(acc1, "a", "v", 1) MF(acc2, "a", "v", 0) MF(acc3, "a", "v", 1) }
            if (KIND == 1) { MF(acc0, "v", "v", 0) MF(acc1, "v", "v", 1) MF(acc2, "v", "v", 0) MF(acc3, "v", "v", 1) }
            if (KIND == 2) { MF(acc0, "v", "a", 0) MF(acc1, "v", "a", 1) MF(acc2, "v", "a", 0) MF(acc3, "v", "a", 1) }
            if (KIND == 3) { MF(acc0, "a", "a", 0) MF(acc1, "a", "a", 1) MF(acc2, "a", "a", 0) MF(acc3, "a", "a", 1) }
            if (KIND == 4) { MF(acc0, "v", "v", 0) MF(acc0, "v", "v", 1) MF(acc0, "v", "v", 0) MF(acc0, "v", "v", 1) }  // one chain
            if (KIND == 5) { MF(acc0, "a", "v", 0) MF(acc1, "a", "v", 1) MF(acc2, "a", "v", 2) MF(acc3, "a", "v", 3) MF(acc0, "a", "v", 4) MF(acc1, "a", "v", 5) MF(acc2, "a", "v", 0) MF(acc3, "a", "v", 1) }
            if (KIND == 6) { MF(acc0, "a", "v", 0) MF(acc0, "a", "v", 1) MF(acc0, "a", "v", 2) MF(acc0, "a", "v", 3) MF(acc0, "a", "v", 4) MF(acc0, "a", "v", 5) MF(acc1, "a", "v", 0) MF(acc1, "a", "v", 1) }
            if (KIND == 7) { MF(acc0, "a", "v", 0) __builtin_amdgcn_sched_barrier(0); MF(acc1, "a", "v", 1) __builtin_amdgcn_sched_barrier(0); MF(acc2, "a", "v", 2) __builtin_amdgcn_sched_barrier(0); MF(acc3, "a", "v", 3) __builtin_amdgcn_sched_barrier(0); }
        }
    }
    float s = one_block_per_cu[(threadIdx.x * 7) & 255];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0, clk[1] = wall_clock64() - w0;
}

template <int KIND>
void run(const char *name, int blocks) {
    const int iters = 4000;
    float *out;
    unsigned long long *clk, hclk[2];
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    kern<KIND><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e0);
    kern<KIND><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);
    const int per_iter = (KIND == 5 || KIND == 6) ? 32 : 16;
    printf("%-28s %3d workgroups: %6.2f ns per MFMA = %5.1f core cycles at %4.0f MHz\n", name, blocks, ms * 1e6 / iters / per_iter, ms * 1e6 / iters / per_iter * mhz * 1e-3, mhz);
    (void)hipFree(out), (void)hipFree(clk);
}

int main() {
    for (int blocks : {256, 32}) {
        run<0>("A in AGPRs, C/D in VGPRs", blocks);
        run<1>("A in VGPRs, C/D in VGPRs", blocks);
        run<2>("A in VGPRs, C/D in AGPRs", blocks);
        run<3>("A in AGPRs, C/D in AGPRs", blocks);
        run<4>("all VGPRs, ONE chain", blocks);
        run<5>("six operand sets, rotating", blocks);
        run<6>("six operand sets, chains", blocks);
        run<7>("sched_barrier between", blocks);
    }
    return 0;
}
