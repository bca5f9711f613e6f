// Micro-benchmark: does VALU work overlap with f32 / bf16 MFMA on gfx950?
// One workgroup of WAVES*64 threads per CU; each wave runs ITER iterations of
// {NM MFMAs + NV independent VALU ops}.  Prints shader cycles per iteration (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>  // KIND 0: none, 1: f32 16x16x4, 2: bf16 16x16x32
__global__ void kern(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f80 + threadIdx.x % 7); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    float lim = 1e30f;
    asm volatile("" : "+s"(lim));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_amdgcn_fmed3f(v[j % 8], v[(j + 1) % 8], lim);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int NV>
void run(int waves_per_cu, const char *name) {
    int iters = 2000, blocks = 256;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * waves_per_cu * 64 * 4);
    hipMalloc(&cyc, blocks * waves_per_cu * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<KIND, NV><<<blocks, waves_per_cu * 64>>>(out, cyc, iters);
    hipEventRecord(e0);
    kern<KIND, NV><<<blocks, waves_per_cu * 64>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * waves_per_cu);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= h.size();
    // per iteration of 4 x {1 MFMA + NV VALU}
    printf("%-10s NV=%d waves/SIMD=%d : %8.1f memtime-ticks/iter (4 MFMA + %d VALU)  wall %.3f ms -> %.1f ns/iter\n",
           name, NV, waves_per_cu / 4, avg / iters, 4 * NV, ms, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc);
}

#define RUNALL(KIND, NAME) \
    for (int w : {4, 8}) { run<KIND, 0>(w, NAME); run<KIND, 2>(w, NAME); run<KIND, 4>(w, NAME); run<KIND, 6>(w, NAME); run<KIND, 8>(w, NAME); run<KIND, 12>(w, NAME);}

int main() {
    RUNALL(0, "valu-only")
    RUNALL(1, "f32-mfma")
    RUNALL(2, "bf16-mfma")
    return 0;
}
