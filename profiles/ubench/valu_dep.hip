// Micro-benchmark (round 4): latency of DEPENDENT v_add_f32 in a lone wave (one wave per SIMD): N independent chains of
// dependent additions, addends in VGPRs.  cycles per addition = cycles per step / N when the chains interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters, unsigned long long *clk) {
    __shared__ float one_block_per_cu[24 * 1024];
    one_block_per_cu[threadIdx.x] = 0.0f;
    float s[N], x[8];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = threadIdx.x * 0.5f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 0.001f * (threadIdx.x + i);
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = s[i] + x[(k + i) & 7];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" : "+v"(s[i]));
    }
    const unsigned long long c1 = clock64();
    float t = one_block_per_cu[(threadIdx.x * 7) & 255];
#pragma unroll
    for (int i = 0; i < N; ++i) t += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int N>
void run() {
    const int iters = 2000;
    float *out;
    unsigned long long *clk, h;
    (void)hipMalloc(&out, 256 * 256 * 4), (void)hipMalloc(&clk, 8);
    kern<N><<<256, 256>>>(out, iters, clk);
    kern<N><<<256, 256>>>(out, iters, clk);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%d independent chain(s): %.2f core cycles per step of %d addition(s) = %.2f per addition\n", N, (double)h / iters / 64, N, (double)h / iters / 64 / N);
    (void)hipFree(out), (void)hipFree(clk);
}
int main() {
    run<1>(), run<2>(), run<3>(), run<4>();
    return 0;
}
