// Micro-benchmark: does the 9 % of scattered 128-byte WRITES k_fs_fold mixes into its read stream (a tile's summaries:
// 32 lines, one per column, 250 KB apart) cost the read stream bandwidth?  Pattern A of row_read_patterns.hip (the fold's
// access shape: 8 lanes per 128-byte line, 16 consecutive rows per lane, 16 KB per wave-item, next item's loads in flight
// while the current one is consumed), 64 fma per loaded float4, plus per item:
//   W0  nothing written
//   W1  8 lines of 128 bytes, each in another column plane (stride n / 64 * 16 bytes): the fold's layout [column][segment]
//   W2  the same 1 KB as one contiguous run per item: a layout [tile][column][segment]
//   W3  like W1 but through non-temporal stores
//   hipcc --offload-arch=gfx950 -O3 row_read_write_mix.hip -o bin/row_read_write_mix && bin/row_read_write_mix [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WM>
__global__ __launch_bounds__(64) void k_mix(const float* __restrict__ X, uint32_t n, float4* __restrict__ S, size_t plane4, float* __restrict__ out, int work) {
    const uint32_t lane = threadIdx.x, d = 128;
    const uint32_t n_items = (n / 128) * 4;
    float acc = 0.f;
    float4 v[16], w[16];
    auto addr = [&](uint32_t item, int i) -> const float4* {
        const uint32_t rb = item / 4, cb = item % 4, g = lane >> 3, q = lane & 7;
        return reinterpret_cast<const float4*>(X + (size_t)(rb * 128 + 16 * g + i) * d + cb * 32 + 4 * q);
    };
    uint32_t item = blockIdx.x;
    if (item >= n_items) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *addr(item, i);
    for (;;) {
        const uint32_t next = item + gridDim.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = v[i];
        if (next < n_items) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = *addr(next, i);
        }
        for (int r = 0; r < work; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_fmaf(w[i].x, w[i].y, acc) + w[i].z * w[i].w;
        }
        if (WM) {
            // 64 lanes x 16 bytes = 1 KB per item: lane (c, p) = (lane / 8, lane % 8) writes piece p of column c's line
            const uint32_t c = lane >> 3, p = lane & 7, rb = item / 4, cb = item % 4;
            const float4 val = make_float4(acc, w[0].x, w[1].y, w[2].z);
            if (WM == 2) {
                S[(size_t)item * 64 + lane] = val;
            } else {
                float4* dst = S + (size_t)(cb * 8 + c) * plane4 + (size_t)rb * 8 + p;
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v vv = {val.x, val.y, val.z, val.w};
                if (WM == 3) __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(dst)); else *dst = val;
            }
        }
        if (next >= n_items) break;
        item = next;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t n = 1u << 20;
    float* X; float* out; float4* S;
    hipMalloc(&X, (size_t)n * 128 * 4); hipMalloc(&out, 4);
    const size_t plane4 = (size_t)(n / 128) * 8 + 64;  // float4 per column plane (8 pieces per 128-row block)
    hipMalloc(&S, (size_t)32 * plane4 * 16 + ((size_t)n / 128 * 4 * 64 * 16));
    hipMemset(X, 0, (size_t)n * 128 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * wpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {4, 16, 24, 32, 40}) {
        for (int wm = 0; wm < 4; ++wm) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (wm == 0) hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(64), 0, 0, X, n, S, plane4, out, work);
                if (wm == 1) hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(64), 0, 0, X, n, S, plane4, out, work);
                if (wm == 2) hipLaunchKernelGGL(k_mix<2>, dim3(grid), dim3(64), 0, 0, X, n, S, plane4, out, work);
                if (wm == 3) hipLaunchKernelGGL(k_mix<3>, dim3(grid), dim3(64), 0, 0, X, n, S, plane4, out, work);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("waves/CU %d work %2d writes W%d: %.1f us  read %.2f TB/s\n", wpc, work, wm, best * 1e3, 0.536870912 / best);
        }
    }
    return 0;
}
