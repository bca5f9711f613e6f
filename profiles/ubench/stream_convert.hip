// Micro-benchmark: f16 -> f32 streaming conversion (2 bytes in, 4 bytes out per element; k_dequant_f16), 1M x 128 elements:
// how much of a plain copy's rate does the loop shape cost?
//   V0  the library's kernel: grid-stride, one 16-byte load and two 16-byte stores per iteration, 8 workgroups per CU
//   V1  four independent loads per iteration, then the eight stores
//   V2  V1 with non-temporal stores
//   V3  V1 with non-temporal loads and stores
//   V4  one group per thread, no loop (grid = groups / 256)
//   hipcc --offload-arch=gfx950 -O3 stream_convert.hip -o bin/stream_convert && bin/stream_convert
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cvt8(u4v v, f4v& a, f4v& b) {
    a.x = __half2float(__ushort_as_half((unsigned short)(v.x & 0xffffu)));
    a.y = __half2float(__ushort_as_half((unsigned short)(v.x >> 16)));
    a.z = __half2float(__ushort_as_half((unsigned short)(v.y & 0xffffu)));
    a.w = __half2float(__ushort_as_half((unsigned short)(v.y >> 16)));
    b.x = __half2float(__ushort_as_half((unsigned short)(v.z & 0xffffu)));
    b.y = __half2float(__ushort_as_half((unsigned short)(v.z >> 16)));
    b.z = __half2float(__ushort_as_half((unsigned short)(v.w & 0xffffu)));
    b.w = __half2float(__ushort_as_half((unsigned short)(v.w >> 16)));
}

template <int V>
__global__ __launch_bounds__(256) void k_cvt(const u4v* __restrict__ in, uint64_t groups, f4v* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    if (V == 0 || V == 4) {
        for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
            f4v a, b;
            cvt8(in[g], a, b);
            out[2 * g] = a;
            out[2 * g + 1] = b;
            if (V == 4) break;
        }
    } else {
        for (uint64_t g0 = (uint64_t)blockIdx.x * 256 + threadIdx.x; g0 < groups; g0 += 4 * stride) {
            u4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t g = g0 + u * stride;
                if (g < groups) v[u] = (V == 3) ? __builtin_nontemporal_load(in + g) : in[g];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t g = g0 + u * stride;
                if (g < groups) {
                    f4v a, b;
                    cvt8(v[u], a, b);
                    if (V >= 2) {
                        __builtin_nontemporal_store(a, out + 2 * g);
                        __builtin_nontemporal_store(b, out + 2 * g + 1);
                    } else {
                        out[2 * g] = a;
                        out[2 * g + 1] = b;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_copy(const f4v* __restrict__ in, uint64_t n4, f4v* __restrict__ out) {
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += (uint64_t)gridDim.x * 256) out[g] = in[g];
}

int main() {
    const uint64_t count = 128ull << 20, groups = count / 8;
    u4v* in; f4v* out;
    hipMalloc(&in, count * 2); hipMalloc(&out, count * 4);
    hipMemset(in, 0x3c, count * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {8, 16, 32}) {
        const int grid = p.multiProcessorCount * wpc;
        for (int v = 0; v < 5; ++v) {
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(e0);
                const int gr = (v == 4) ? (int)(groups / 256) : grid;
                if (v == 0) hipLaunchKernelGGL(k_cvt<0>, dim3(gr), dim3(256), 0, 0, in, groups, out);
                if (v == 1) hipLaunchKernelGGL(k_cvt<1>, dim3(gr), dim3(256), 0, 0, in, groups, out);
                if (v == 2) hipLaunchKernelGGL(k_cvt<2>, dim3(gr), dim3(256), 0, 0, in, groups, out);
                if (v == 3) hipLaunchKernelGGL(k_cvt<3>, dim3(gr), dim3(256), 0, 0, in, groups, out);
                if (v == 4) hipLaunchKernelGGL(k_cvt<4>, dim3(gr), dim3(256), 0, 0, in, groups, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 1 && ms < best) best = ms;
            }
            printf("workgroups/CU %2d V%d: %.1f us  %.2f TB/s\n", wpc, v, best * 1e3, (double)count * 6 / best / 1e9);
        }
    }
    // the yardstick: a copy of 256 MB inside `out` (512 MB moved)
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_copy, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, (const f4v*)out, count * 2 / 16, out + count * 2 / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 1 && ms < best) best = ms;
    }
    printf("copy of 256 MB (512 MB moved): %.1f us  %.2f TB/s\n", best * 1e3, (double)count * 4 / best / 1e9);
    return 0;
}
