// Micro-benchmark: how fast can 1M x 128 f32 rows (512 MB) be streamed through registers with the access shapes the
// TSVQ column-sum kernels could use?  One-wave workgroups, persistent, 16 independent 16-byte loads per lane and item
// (16 KB per wave-item), the next item's loads issued before the current one is consumed.
//   A  8 lanes per row (one 128-byte line), 8 row groups x 16 consecutive rows, 4 waves side by side cover a row
//   B  32 lanes per row (the whole 512-byte row), 2 rows per instruction, 32 consecutive rows per item
//   C  linear: the wave reads 16 KB contiguous
//   hipcc --offload-arch=gfx950 -O3 row_read_patterns.hip -o row_read_patterns && ./row_read_patterns [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(64) void k_read(const float* __restrict__ X, uint32_t n, float* __restrict__ out, int work) {
    const uint32_t lane = threadIdx.x;
    const uint32_t d = 128;
    uint32_t n_items = (PAT == 0) ? (n / 128) * 4 : (n / 32);
    float acc = 0.f;
    float4 v[16], w[16];
    auto addr = [&](uint32_t item, int i) -> const float4* {
        if (PAT == 0) {
            const uint32_t rb = item / 4, cb = item % 4, g = lane >> 3, q = lane & 7;
            return reinterpret_cast<const float4*>(X + (size_t)(rb * 128 + 16 * g + i) * d + cb * 32 + 4 * q);
        } else if (PAT == 1) {
            const uint32_t g = lane >> 5, q = lane & 31;
            return reinterpret_cast<const float4*>(X + (size_t)(item * 32 + 16 * g + i) * d + 4 * q);
        } else {
            return reinterpret_cast<const float4*>(X + (size_t)item * 4096 + (size_t)i * 256 + 4 * lane);
        }
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
        if (next >= n_items) break;
        item = next;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t n = 1u << 20;
    float* X; float* out;
    hipMalloc(&X, (size_t)n * 128 * 4); hipMalloc(&out, 4);
    hipMemset(X, 0, (size_t)n * 128 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * wpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 4, 16}) {
        for (int pat = 0; pat < 3; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k_read<0>, dim3(grid), dim3(64), 0, 0, X, n, out, work);
                if (pat == 1) hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(64), 0, 0, X, n, out, work);
                if (pat == 2) hipLaunchKernelGGL(k_read<2>, dim3(grid), dim3(64), 0, 0, X, n, out, work);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("waves/CU %d work %2d pattern %c: %.1f us  %.2f TB/s\n", wpc, work, "ABC"[pat], best * 1e3, 0.536870912 / best);
        }
    }
    return 0;
}
