// Device self-test of the one non-architectural assumption in the bf16-split screen's error
// bound (DESIGN.md 4.1 "screen soundness"): a bf16 MFMA adds its K products and the C input into
// an f32 result with an error of at most kBf16AssumedUlps (32) * 2^-24 * (|C| + sum|a*b|).  The ISA guide documents
// the fp32 MFMA as an fmaf chain but not the bf16 datapath, so the library measures it once per
// process on the device it runs on -- v_mfma_f32_32x32x16_bf16 (X32 engine) and
// v_mfma_f32_16x16x32_bf16 (16x16 variants) against exact f64 sums over seven adversarial
// families -- and refuses the bf16 engine (AUTO falls back to the fp32 MFMA screen) if the worst
// ratio exceeds HALF the assumed constant.
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "kernels.hpp"

namespace vqhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline float bf16_value(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ inline uint16_t make_bf16(uint32_t sign, int exp2, uint32_t mant7) {  // (1 + mant7/128) * 2^exp2
    return (uint16_t)((sign << 15) | ((uint32_t)(exp2 + 127) << 7) | (mant7 & 127u));
}

// element (trial, row/col index i, k) of operand `which` (0 = A, 1 = B); C value of (trial, i, j)
__device__ uint16_t gen_operand(uint32_t trial, int which, uint32_t i, uint32_t kk, uint32_t K) {
    const uint32_t fam = trial % 7, s = 8 + (trial / 7) % 27, pos = (trial / 189) % K;
    const uint64_t h = mix64(((uint64_t)trial << 32) ^ ((uint64_t)which << 24) ^ ((uint64_t)i << 8) ^ kk);
    const uint32_t mant = (uint32_t)(h & 127u), sign = (uint32_t)((h >> 7) & 1u);
    switch (fam) {
        case 0:  // random values over 40 binades
            return make_bf16(sign, (int)((h >> 8) % 41) - 20, mant);
        case 1:  // one dominant product, the rest 2^-s below, alternating signs
        case 3:  // the same with C cancelling the dominant product (see gen_c)
            if (kk == pos) return make_bf16(0, 0, which ? 7 : 5);
            return make_bf16(kk & 1u, which ? -(int)(s - s / 2) : -(int)(s / 2), mant);
        case 2:  // dominant C, small products
            return make_bf16(sign, which ? -(int)(s - s / 2) : -(int)(s / 2), mant);
        case 4:  // same sign, same binade: plain accumulation
            return make_bf16(0, 0, mant);
        case 5:  // one dominant product, the rest POSITIVE and 2^-s below (every aligned addend truncates the same way)
            if (kk == pos) return make_bf16(0, 0, which ? 7 : 5);
            return make_bf16(0, which ? -(int)(s - s / 2) : -(int)(s / 2), mant);
        default: {  // three exponent levels per operand, mostly positive: several dominant addends per adder group
            const uint32_t level = (uint32_t)((h >> 12) % 3);
            return make_bf16(((h >> 20) & 7u) == 0 ? 1u : 0u, -(int)(level * (s / 4)), mant);
        }
    }
}
__device__ float gen_c(uint32_t trial, uint32_t i, uint32_t j) {
    const uint32_t fam = trial % 7;
    const uint64_t h = mix64(((uint64_t)trial << 32) ^ 0xC0000000ull ^ ((uint64_t)i << 8) ^ j);
    if (fam == 2) return 1.0f + (float)(h & 1023u) / 1024.0f;
    if (fam == 3) return -(1.0f + 5.0f / 128) * (1.0f + 7.0f / 128);
    if (fam == 0) return ((h & 1u) ? -1.0f : 1.0f) * __uint_as_float(((uint32_t)(107 + (h >> 8) % 41) << 23) | (uint32_t)((h >> 20) & 0x7FFFFFu));
    if (fam == 6) return ((h >> 3) & 1u) ? ldexpf(1.0f + (float)(h & 1023u) / 1024.0f, 1 - (int)((h >> 12) % 3)) : 0.0f;
    return 0.0f;
}

template <int SHAPE>  // 32: 32x32x16, 16: 16x16x32
__global__ __launch_bounds__(64) void k_selftest_bf16(uint32_t trials, uint32_t *worst_bits) {
    constexpr uint32_t MN = SHAPE, K = (SHAPE == 32) ? 16 : 32, KB = K / 8;  // k-blocks of 8 per lane group
    const uint32_t lane = threadIdx.x, idx = lane % MN, kb = lane / MN;
    float worst = 0.0f;
    for (uint32_t t = blockIdx.x; t < trials; t += gridDim.x) {
        bf16x8 a, b;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a[q] = (short)gen_operand(t, 0, idx, 8 * kb + q, K);
            b[q] = (short)gen_operand(t, 1, idx, 8 * kb + q, K);
        }
        (void)KB;
        if constexpr (SHAPE == 32) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = gen_c(t, (r & 3) + 8 * (r >> 2) + 4 * kb, idx);
            const f32x16 dres = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                double exact = (double)c[r], mag = fabs((double)c[r]);
                for (uint32_t kk = 0; kk < K; ++kk) {
                    const double p = (double)bf16_value(gen_operand(t, 0, row, kk, K)) *
                                     (double)bf16_value(gen_operand(t, 1, idx, kk, K));
                    exact += p;
                    mag += fabs(p);
                }
                const double ratio = fabs((double)dres[r] - exact) / (5.9604644775390625e-08 * mag + 1e-300);
                worst = fmaxf(worst, (float)ratio);
            }
        } else {
            f32x4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = gen_c(t, 4 * kb + r, idx);
            const f32x4 dres = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t row = 4 * kb + r;
                double exact = (double)c[r], mag = fabs((double)c[r]);
                for (uint32_t kk = 0; kk < K; ++kk) {
                    const double p = (double)bf16_value(gen_operand(t, 0, row, kk, K)) *
                                     (double)bf16_value(gen_operand(t, 1, idx, kk, K));
                    exact += p;
                    mag += fabs(p);
                }
                const double ratio = fabs((double)dres[r] - exact) / (5.9604644775390625e-08 * mag + 1e-300);
                worst = fmaxf(worst, (float)ratio);
            }
        }
    }
    if (!(worst == worst)) worst = __builtin_inff();
    atomicMax(worst_bits, __float_as_uint(worst));  // non-negative floats order like their bits
}

// Probe of ONE v_mfma_f32_32x32x16_bf16: trial t computes d[t] = C + sum_k a[t][k] * b[t][k] as the
// diagonal element (t % 32, t % 32) of a 32x32 tile, i.e. 32 independent trials per instruction (row i
// of A and column i of B belong to trial i; the off-diagonal products are computed and ignored).
// Operand layout: lane l holds A[i = l % 32][k = 8 (l / 32) .. + 7] and B[k = 8 (l / 32) .. + 7][j = l % 32];
// accumulator register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l / 32)][l % 32].
__global__ __launch_bounds__(64) void k_mfma_probe_32x32x16(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                            const float *__restrict__ c, uint64_t trials,
                                                            float *__restrict__ d) {
    const uint32_t lane = threadIdx.x, idx = lane & 31, kb = lane >> 5;
    for (uint64_t t0 = (uint64_t)blockIdx.x * 32; t0 < trials; t0 += (uint64_t)gridDim.x * 32) {
        const uint64_t t = t0 + idx;
        const bool live = t < trials;
        bf16x8 av, bv;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            av[q] = live ? (short)a[t * 16 + 8 * kb + q] : (short)0;
            bv[q] = live ? (short)b[t * 16 + 8 * kb + q] : (short)0;
        }
        f32x16 cv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * kb;
            cv[r] = (row == idx && live) ? c[t] : 0.0f;
        }
        const f32x16 dres = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, cv, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * kb;
            if (row == idx && live) d[t] = dres[r];
        }
    }
}

std::mutex g_mu;
int g_state = 0;  // 0 not run, 1 trusted, 2 refused
float g_ratio32 = -1.0f, g_ratio16 = -1.0f;

}  // namespace

// worst measured |D - exact| / (2^-24 (|C| + sum|ab|)) for both bf16 MFMA shapes; cached per process
int bf16_mfma_selftest(float *ratio32, float *ratio16, int *trusted) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_state == 0) {
        uint32_t *dev = nullptr;
        VQ_HIP(hipMalloc(&dev, 8));
        VQ_HIP(hipMemset(dev, 0, 8));
        const uint32_t trials = 7 * 27 * 32 * 4;  // every (family, shift, dominant position), 4 draws
        hipLaunchKernelGGL(k_selftest_bf16<32>, dim3(512), dim3(64), 0, nullptr, trials, dev);
        hipLaunchKernelGGL(k_selftest_bf16<16>, dim3(512), dim3(64), 0, nullptr, trials, dev + 1);
        uint32_t bits[2] = {0, 0};
        const hipError_t e = hipMemcpy(bits, dev, 8, hipMemcpyDeviceToHost);
        (void)hipFree(dev);
        if (e != hipSuccess) return fail(VQHIP_ERR_RUNTIME, "bf16 MFMA self-test: %s", hipGetErrorString(e));
        memcpy(&g_ratio32, &bits[0], 4);
        memcpy(&g_ratio16, &bits[1], 4);
        // the soundness proof budgets kBf16AssumedUlps * 2^-24 per MFMA; demand a factor 2 of head room
        g_state = (g_ratio32 <= kBf16AssumedUlps / 2 && g_ratio16 <= kBf16AssumedUlps / 2) ? 1 : 2;
    }
    if (ratio32) *ratio32 = g_ratio32;
    if (ratio16) *ratio16 = g_ratio16;
    if (trusted) *trusted = (g_state == 1) ? 1 : 0;
    return VQHIP_OK;
}

// host buffers in and out; the model of the instruction's adder (tests/mfma_model.py) is checked against this
int mfma_bf16_probe(const uint16_t *a, const uint16_t *b, const float *c, uint64_t trials, float *d) {
    if (trials == 0) return VQHIP_OK;
    DevBuf da, db, dc, dd;
    VQ_TRY(da.alloc((size_t)trials * 32));
    VQ_TRY(db.alloc((size_t)trials * 32));
    VQ_TRY(dc.alloc((size_t)trials * 4));
    VQ_TRY(dd.alloc((size_t)trials * 4));
    VQ_HIP(hipMemcpy(da.p, a, (size_t)trials * 32, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(db.p, b, (size_t)trials * 32, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dc.p, c, (size_t)trials * 4, hipMemcpyHostToDevice));
    uint64_t blocks = (trials + 31) / 32;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_mfma_probe_32x32x16, dim3((uint32_t)blocks), dim3(64), 0, nullptr, da.as<uint16_t>(),
                       db.as<uint16_t>(), dc.as<float>(), trials, dd.as<float>());
    VQ_LAUNCH_CHECK("k_mfma_probe_32x32x16");
    VQ_HIP(hipMemcpy(d, dd.p, (size_t)trials * 4, hipMemcpyDeviceToHost));
    return VQHIP_OK;
}

}  // namespace vqhip
