// Adversarial probe of v_mfma_f32_16x16x32_bf16 accumulation: one dominant addend (a product
// or C) plus many addends just below the dominant one's rounding granularity, at every relative
// magnitude 2^-s.  A shared-exponent "align and truncate" datapath would lose them entirely.
// Reports max |D - exact| / (2^-24 * (|C| + sum|a*b|)) per family.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const uint16_t *A, const uint16_t *B, const float *C, float *D, int trials) {
    int lane = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (short)A[(t * 16 + (lane & 15)) * 32 + 8 * (lane >> 4) + j];
            b[j] = (short)B[(t * 32 + 8 * (lane >> 4) + j) * 16 + (lane & 15)];
        }
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[(t * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)];
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[(t * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = d[r];
    }
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }  // exact inputs only
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    // trial t = (family f, shift s, position pos of the dominant product, sign pattern)
    struct Case { int fam, s, pos, sg; };
    std::vector<Case> cases;
    for (int fam = 0; fam < 4; ++fam)
        for (int s = 8; s <= 34; ++s)
            for (int pos = 0; pos < 32; pos += 5)
                for (int sg = 0; sg < 3; ++sg) cases.push_back({fam, s, pos, sg});
    const int T = (int)cases.size();
    std::vector<uint16_t> A(T * 512, 0), B(T * 512, 0);
    std::vector<float> C(T * 256, 0.f), D(T * 256);
    for (int t = 0; t < T; ++t) {
        const Case &c = cases[t];
        for (int r = 0; r < 16; ++r)
            for (int col = 0; col < 16; ++col) {
                // every output element of the trial sees the same pattern (A rows identical, B cols identical)
            }
        float small_a = ldexpf(1.0f + 1.0f / 128, -(c.s / 2)), small_b = ldexpf(1.0f + 3.0f / 128, -(c.s - c.s / 2));
        for (int kk = 0; kk < 32; ++kk) {
            float av, bv;
            if ((c.fam == 0 || c.fam == 2) && kk == c.pos) { av = 1.0f + 5.0f / 128; bv = 1.0f + 7.0f / 128; }  // dominant product
            else { av = small_a; bv = small_b; }
            float sign = 1.f;
            if (c.sg == 1) sign = (kk & 1) ? -1.f : 1.f;
            if (c.sg == 2 && kk != c.pos) sign = -1.f;
            if (c.fam == 3 && kk >= 16) { av = 0.f; }  // only half the products
            for (int r = 0; r < 16; ++r) A[(t * 16 + r) * 32 + kk] = f2bf(av * sign);
            for (int col = 0; col < 16; ++col) B[(t * 32 + kk) * 16 + col] = f2bf(bv);
        }
        float cval = (c.fam == 1 || c.fam == 3) ? (1.0f + 1.0f / 1024) : ((c.fam == 2) ? -(1.0f + 5.0f / 128) * (1.0f + 7.0f / 128) : 0.f);
        for (int i = 0; i < 256; ++i) C[t * 256 + i] = cval;
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    k<<<256, 64>>>(dA, dB, dC, dD, T);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst[4] = {0, 0, 0, 0}; int worst_s[4] = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        double exact = C[t * 256], mag = fabs(exact);
        for (int kk = 0; kk < 32; ++kk) {
            double p = (double)bf2f(A[(t * 16) * 32 + kk]) * (double)bf2f(B[(t * 32 + kk) * 16]);
            exact += p; mag += fabs(p);
        }
        double err = fabs((double)D[t * 256] - exact);
        double ratio = err / (ldexp(1.0, -24) * mag);
        if (ratio > worst[cases[t].fam]) { worst[cases[t].fam] = ratio; worst_s[cases[t].fam] = cases[t].s; }
    }
    const char *names[4] = {"dominant product + 31 small", "dominant C + 32 small", "cancelling C/product + 31 small", "dominant C + 16 small"};
    for (int f = 0; f < 4; ++f) printf("family %d (%s): max err/(2^-24*(|C|+sum|ab|)) = %.3f at s=%d\n", f, names[f], worst[f], worst_s[f]);
    return 0;
}
