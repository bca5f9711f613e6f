// Micro-benchmark (round 4): a lone wave per SIMD running the screen's pattern -- MFMAs with NV VALU instructions in every
// gap between two of them.  Does the matrix pipe's time hide behind the VALU instructions of the SAME wave?
//   SHAPE 32: v_mfma_f32_32x32x16_bf16 (16 accumulator registers)     SHAPE 16: v_mfma_f32_16x16x32_bf16 (4)
//   DEP 1: chains of six dependent MFMAs on one accumulator tile (ring of four), as the screen issues them
//   DEP 0: consecutive MFMAs go to different tiles (ring of four)
//   DEP 2: the chains of TWO tiles interleaved over two phases (A B A B ...), the other two tiles read by the VALU
//   VALU: tag, tag, med3, min3, min on the tile finished a phase ago (what reduce_hg does)
// Prints ns per gap (one MFMA + NV VALU) from the wall clock of 256 x 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int DEP, int NV, bool MF>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters, unsigned long long *clk) {
    __shared__ float one_block_per_cu[24 * 1024];  // 96 KB: two of these workgroups do not fit one CU (the kernel needs few registers,
    one_block_per_cu[threadIdx.x] = 0.0f;          // and two workgroups on a CU would put two waves on a SIMD -- the first run of this file measured that)
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[4];
    f32x4 acs[4];
    // (initialisation fully unrolled: left as loops the compiler indexes the vectors at run time through v_readlane /
    // v_writelane sequences, and a wave that has run those issues its MFMAs at 52-63 cycles instead of 32 for the rest
    // of its life -- ab/rc_v2 against rc_v5 of this round's notes; the first two runs of this file measured that)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = threadIdx.x * 1e-3f + i;
#pragma unroll
        for (int i = 0; i < 4; ++i) acs[t][i] = threadIdx.x * 1e-3f + i;
    }
    bf16x8 a[6], b[6];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[f][i] = (short)(0x3f80 + i + f), b[f][i] = (short)(0x3f80 + threadIdx.x % 7 + f);
    float q1 = 1e30f, q2 = 1e30f;
    unsigned mask = 0xFFFFFFC0u;
    asm volatile("" : "+s"(mask));
    // every register index below is a compile-time constant (GAP is expanded 24 times: a loop over tiles and MFMAs left
    // the element indices to run time and the "benchmark" measured v_cndmask chains)
#define ROUND(TR, E0, E1, J)                                                                          \
    {                                                                                                 \
        const float src0 = SHAPE == 32 ? acc[TR][(E0) & 15] : acs[TR][(E0) & 3];                      \
        const float src1 = SHAPE == 32 ? acc[TR][(E1) & 15] : acs[TR][(E1) & 3];                      \
        const float pa = __uint_as_float((__float_as_uint(src0) & mask) | (unsigned)((J) + 1));       \
        const float pb = __uint_as_float((__float_as_uint(src1) & mask) | (unsigned)((J) + 2));       \
        const float tm = __builtin_amdgcn_fmed3f(q1, pa, pb);                                         \
        q1 = __builtin_fminf(__builtin_fminf(q1, pa), pb);                                            \
        q2 = __builtin_fminf(q2, tm);                                                                 \
        asm volatile("" ::"v"(q1), "v"(q2));                                                          \
    }
#define GAP(I, F)                                                                                                              \
    {                                                                                                                          \
        constexpr int t = DEP == 1 ? (((I) + 2) & 3) : DEP == 2 ? ((2 * ((I) / 2) + 2 + ((F) & 1)) & 3) : (((I) + 2 + (F)) & 3); \
        constexpr int tr = DEP == 1 ? ((I) & 3) : DEP == 2 ? ((2 * ((I) / 2) + ((F) & 1)) & 3) : (((I) + (F) + 2) & 3);        \
        if (MF && SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(a[F]), "v"(b[F])); \
        if (MF && SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acs[t]) : "a"(a[F]), "v"(b[F])); \
        if (NV > 0) ROUND(tr, 2 * (F), 2 * (F) + 1, 0)                                                                         \
        if (NV > 1) ROUND(tr, 2 * (F) + 2, 2 * (F) + 3, 1)                                                                     \
        if (NV > 2) ROUND(tr, 2 * (F) + 4, 2 * (F) + 5, 2)                                                                     \
        if (NV > 3) ROUND(tr, 2 * (F) + 6, 2 * (F) + 7, 3)                                                                     \
        if (NV > 4) ROUND(tr, 2 * (F) + 8, 2 * (F) + 9, 4)                                                                     \
        if (NV > 5) ROUND(tr, 2 * (F) + 10, 2 * (F) + 11, 5)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    }
#define PHASE(I) GAP(I, 0) GAP(I, 1) GAP(I, 2) GAP(I, 3) GAP(I, 4) GAP(I, 5)
    for (int it = 0; it < iters; ++it) {
        PHASE(0) PHASE(1) PHASE(2) PHASE(3)
    }
    float s = q1 + q2 + one_block_per_cu[(threadIdx.x * 7) & 255];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[t][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acs[t][i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0, clk[1] = wall_clock64() - w0;
}

template <int SHAPE, int DEP, int NV, bool MF>
void run() {
    const int iters = 2000, blocks = 256;
    float *out;
    unsigned long long *clk, hclk[2];
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    kern<SHAPE, DEP, NV, MF><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e0);
    kern<SHAPE, DEP, NV, MF><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);  // wall_clock64 counts 100 MHz
    printf("shape %2d %s %s  rounds per gap %d : %7.2f ns per gap = %6.1f core cycles at %4.0f MHz (clock64 / wall_clock64)\n", SHAPE, MF ? "MFMA" : "no  ",
           DEP == 1 ? "chained    " : DEP == 2 ? "two chains " : "independent", NV, ms * 1e6 / iters / 24, ms * 1e6 / iters / 24 * mhz * 1e-3, mhz);
    (void)hipFree(out);
}

template <int SHAPE, int DEP>
void sweep() {
    run<SHAPE, DEP, 0, true>(), run<SHAPE, DEP, 1, true>(), run<SHAPE, DEP, 2, true>(), run<SHAPE, DEP, 3, true>(), run<SHAPE, DEP, 4, true>(),
        run<SHAPE, DEP, 6, true>();
}

int main() {
    run<32, 1, 1, false>(), run<32, 1, 2, false>(), run<32, 1, 4, false>(), run<32, 1, 6, false>();
    sweep<32, 1>(), sweep<32, 2>(), sweep<32, 0>();
    return 0;
}
