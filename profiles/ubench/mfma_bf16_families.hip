// Per-family / per-shift worst accumulation error of v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_16x16x32_bf16 in units of 2^-24 (|C| + sum|ab|): structured families that try to make
// every small addend lose as much as possible when it is aligned to a dominant addend.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_families mfma_bf16_families.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// one trial = one MFMA whose rows/cols all carry the same pattern: A[i][k] = a[k], B[k][j] = b[k], C = c
struct Trial { uint16_t a[32], b[32]; float c; };

template <int SHAPE>
__global__ void run(const Trial *tr, int n, float *out) {
    constexpr int K = SHAPE == 32 ? 16 : 32;
    const int lane = threadIdx.x, kb = lane / SHAPE;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        bf16x8 a, b;
        for (int q = 0; q < 8; ++q) { a[q] = (short)tr[t].a[(8 * kb + q) % K]; b[q] = (short)tr[t].b[(8 * kb + q) % K]; }
        if constexpr (SHAPE == 32) {
            f32x16 c; for (int r = 0; r < 16; ++r) c[r] = tr[t].c;
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
            if (lane == 0) out[t] = d[0];
        } else {
            f32x4 c; for (int r = 0; r < 4; ++r) c[r] = tr[t].c;
            f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            if (lane == 0) out[t] = d[0];
        }
    }
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    for (int shape : {32, 16}) {
        const int K = shape == 32 ? 16 : 32;
        std::vector<Trial> tr; std::vector<int> fam, sh;
        // family 0: dominant C = 1.xxx, all K products = +m * 2^-s with m sweeping the 16-bit product significand
        // family 1: dominant product, K-1 products = +m * 2^-s, C = 0
        // family 2: dominant C negative, products positive (result small: cancellation)
        // family 3: dominant C, products alternate sign
        // family 4: C = 0, products geometric 2^-(s*i/K) all positive
        for (int f = 0; f < 5; ++f)
            for (int s = 1; s <= 40; ++s)
                for (int mv = 0; mv < 64; ++mv) {
                    Trial t; memset(&t, 0, sizeof t);
                    float ma = 1.0f + (float)((mv * 37) % 128) / 128.0f, mb = 1.0f + (float)((mv * 91 + 5) % 128) / 128.0f;
                    for (int k = 0; k < K; ++k) {
                        float av = ldexpf(ma, -(s / 2)), bv = ldexpf(mb, -(s - s / 2));
                        if (f == 1 && k == (mv % K)) { av = 1.0f + 5.0f / 128; bv = 1.0f + 7.0f / 128; }
                        if (f == 3 && (k & 1)) av = -av;
                        if (f == 4) { av = ldexpf(ma, -((s * k) / K) / 2); bv = ldexpf(mb, -(((s * k) / K) - ((s * k) / K) / 2)); }
                        t.a[k] = f2bf(av); t.b[k] = f2bf(bv);
                    }
                    t.c = (f == 0 || f == 3) ? 1.0f + (float)mv / 64.0f : (f == 2 ? -(1.0f + (float)mv / 64.0f) : 0.0f);
                    tr.push_back(t); fam.push_back(f); sh.push_back(s);
                }
        Trial *d; float *o; const int n = (int)tr.size();
        hipMalloc(&d, n * sizeof(Trial)); hipMalloc(&o, n * 4);
        hipMemcpy(d, tr.data(), n * sizeof(Trial), hipMemcpyHostToDevice);
        if (shape == 32) run<32><<<256, 64>>>(d, n, o); else run<16><<<256, 64>>>(d, n, o);
        std::vector<float> out(n); hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
        double worst[5][41] = {};
        for (int t = 0; t < n; ++t) {
            double exact = tr[t].c, mag = fabs(exact);
            for (int k = 0; k < K; ++k) { double p = (double)bf2f(tr[t].a[k]) * (double)bf2f(tr[t].b[k]); exact += p; mag += fabs(p); }
            double r = fabs((double)out[t] - exact) / (ldexp(1.0, -24) * mag);
            if (r > worst[fam[t]][sh[t]]) worst[fam[t]][sh[t]] = r;
        }
        printf("shape %dx%dx%d\n", shape, shape, K);
        for (int f = 0; f < 5; ++f) {
            printf(" family %d:", f);
            for (int s = 1; s <= 40; ++s) printf(" %.2f", worst[f][s]);
            printf("\n");
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
