// Probe: accumulation error of v_mfma_f32_16x16x32_bf16 on gfx950 relative to exact arithmetic.
// For each trial a 16x32 A, 32x16 B (bf16) and 16x16 C (f32) are drawn; D is compared with the
// exact (f64) value.  Reports max |D-exact| / (2^-24 * (|C| + sum|a*b|)) per input family.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t *A, const uint16_t *B, const float *C, float *D, int trials) {
    // A: [trial][16][32], B: [trial][32][16], C/D: [trial][16][16]
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
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int T = 20000;
    std::mt19937_64 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_int_distribution<int> ex(-12, 12);
    for (int family = 0; family < 5; ++family) {
        std::vector<uint16_t> A(T * 512), B(T * 512);
        std::vector<float> C(T * 256), D(T * 256);
        for (int t = 0; t < T; ++t) {
            for (int i = 0; i < 512; ++i) {
                float sa = 1.f, sb = 1.f;
                if (family == 1) { sa = ldexpf(1.f, ex(rng)); sb = ldexpf(1.f, ex(rng)); }
                if (family == 3) { sa = ldexpf(1.f, -9 * (i % 3)); }
                A[t * 512 + i] = f2bf(nd(rng) * sa);
                B[t * 512 + i] = f2bf(nd(rng) * sb);
            }
            for (int i = 0; i < 256; ++i) C[t * 256 + i] = (family == 4) ? 0.f : nd(rng) * (family == 2 ? 1e4f : 1.f);
            if (family == 2) {  // cancellation: rows of A paired +x / -x with equal B -> sum tiny vs terms
                for (int r = 0; r < 16; ++r)
                    for (int kk = 0; kk < 32; kk += 2) {
                        A[(t * 16 + r) * 32 + kk + 1] = A[(t * 16 + r) * 32 + kk] ^ 0x8000;
                        for (int c = 0; c < 16; ++c) B[(t * 32 + kk + 1) * 16 + c] = B[(t * 32 + kk) * 16 + c];
                    }
                for (int r = 0; r < 16; ++r) A[(t * 16 + r) * 32 + 31] = f2bf(1e-3f);  // small survivor
            }
        }
        uint16_t *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        k<<<256, 64>>>(dA, dB, dC, dD, T);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double worst_abs = 0, worst_rel_res = 0;
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    double exact = C[(t * 16 + r) * 16 + c], mag = fabs(exact);
                    for (int kk = 0; kk < 32; ++kk) {
                        double p = (double)bf2f(A[(t * 16 + r) * 32 + kk]) * (double)bf2f(B[(t * 32 + kk) * 16 + c]);
                        exact += p; mag += fabs(p);
                    }
                    double err = fabs((double)D[(t * 16 + r) * 16 + c] - exact);
                    double u = ldexp(1.0, -24);
                    if (err / (u * mag) > worst_abs) worst_abs = err / (u * mag);
                    if (fabs(exact) > 0 && err / (u * fabs(exact)) > worst_rel_res) worst_rel_res = err / (u * fabs(exact));
                }
        printf("family %d: max err/(2^-24*(|C|+sum|ab|)) = %.4f   max err/(2^-24*|exact|) = %.4f\n", family, worst_abs, worst_rel_res);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    return 0;
}
