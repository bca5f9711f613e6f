// k_misc.hip -- output gathers and the synthetic data generator (gfx950).  All HBM-bound.
#include <hip/hip_fp16.h>

#include "kernels.hpp"

namespace vqhip {
namespace {

// splitmix64 finaliser on a per-element counter: element (row, col) of the synthetic matrix
// depends only on (seed, global row, col), so any shard of any size reproduces the same data
// on the device and on the host.  24 random bits -> exact multiples of 2^-24 in [0, 1).
__host__ __device__ inline float synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t d) {
    uint64_t z = seed + (row * (uint64_t)d + col + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
}

__global__ __launch_bounds__(256) void k_synth_uniform(float *__restrict__ X, uint64_t n, uint32_t d,
                                                       uint64_t seed, uint64_t row_offset) {
    const uint64_t total = n * d;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * 256) {
        const uint64_t row = e / d;
        const uint32_t col = (uint32_t)(e - row * d);
        X[e] = synth_value(seed, row_offset + row, col, d);
    }
}

// out[row][s*sd + t] = f16(codebook[s][codes[row][s]][t]), src/pq.rs:193-195.
// VEC output elements per lane (8 -> one 16-byte store; the codebook rows come from L1/L2).
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_f16(const float *__restrict__ cb, uint32_t m,
                                                    uint32_t k, uint32_t sd,
                                                    const uint8_t *__restrict__ codes, uint64_t n,
                                                    uint16_t *__restrict__ out) {
    const uint32_t d = m * sd;
    const uint32_t gpr = d / VEC;  // groups per row
    const uint64_t total = n * gpr;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * 256) {
        const uint64_t row = e / gpr;
        const uint32_t col = (uint32_t)(e - row * gpr) * VEC;
        const uint32_t s = col / sd, t = col - s * sd;
        const uint32_t code = load_code(codes, row * m + s, k);
        const float *src = cb + ((size_t)s * k + code) * sd + t;
        if constexpr (VEC == 8) {
            const float4 a = *reinterpret_cast<const float4 *>(src);
            const float4 b = *reinterpret_cast<const float4 *>(src + 4);
            uint4 o;
            o.x = (uint32_t)__half_as_ushort(__float2half_rn(a.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(a.y)) << 16);
            o.y = (uint32_t)__half_as_ushort(__float2half_rn(a.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(a.w)) << 16);
            o.z = (uint32_t)__half_as_ushort(__float2half_rn(b.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(b.y)) << 16);
            o.w = (uint32_t)__half_as_ushort(__float2half_rn(b.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(b.w)) << 16);
            *reinterpret_cast<uint4 *>(out + row * d + col) = o;
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) out[row * d + col + i] = __half_as_ushort(__float2half_rn(src[i]));
        }
    }
}

// out[row][s*sd + t] = codebook[s][codes[row][s]][t]: 4*D bytes out + m bytes in per row, HBM-bound (the codebook rows
// come from L1 / L2).  VEC = 4: one 16-byte store per lane, the row / column split in 32-bit arithmetic per block tile
// (the scalar form paid a 64-bit division per ELEMENT: 0.9 TB/s).
template <int VEC>
__global__ __launch_bounds__(256) void k_decode_f32(const float *__restrict__ cb, uint32_t m,
                                                    uint32_t k, uint32_t sd,
                                                    const uint8_t *__restrict__ codes, uint64_t n,
                                                    float *__restrict__ out) {
    const uint32_t d = m * sd;
    const uint32_t gpr = d / VEC;  // groups per row
    const uint64_t total = n * gpr;
    for (uint64_t base = (uint64_t)blockIdx.x * 256; base < total; base += (uint64_t)gridDim.x * 256) {
        const uint64_t row0 = base / gpr;                       // (uniform: scalar unit)
        const uint32_t g = (uint32_t)(base - row0 * gpr) + threadIdx.x;
        const uint32_t dr = g / gpr;
        const uint64_t row = row0 + dr;
        if (row >= n) continue;
        const uint32_t col = (g - dr * gpr) * VEC;
        const uint32_t s = col / sd, t = col - s * sd;
        const float *src = cb + ((size_t)s * k + load_code(codes, row * m + s, k)) * sd + t;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4 *>(out + row * d + col) = *reinterpret_cast<const float4 *>(src);
        } else {
            out[row * d + col] = src[0];
        }
    }
}

// f16 bits -> f32 (exact, src/pq.rs:208): 2 bytes in + 4 bytes out per element, HBM-bound; eight elements per lane
// (one 16-byte load, two 16-byte stores), scalar tail
__global__ __launch_bounds__(256) void k_dequant_f16(const uint16_t *__restrict__ in, uint64_t count,
                                                     float *__restrict__ out) {
    const uint64_t groups = count / 8;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned) {
        for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * 256) {
            const uint4 v = *reinterpret_cast<const uint4 *>(in + g * 8);
            float4 a, b;
            a.x = __half2float(__ushort_as_half((unsigned short)(v.x & 0xffffu)));
            a.y = __half2float(__ushort_as_half((unsigned short)(v.x >> 16)));
            a.z = __half2float(__ushort_as_half((unsigned short)(v.y & 0xffffu)));
            a.w = __half2float(__ushort_as_half((unsigned short)(v.y >> 16)));
            b.x = __half2float(__ushort_as_half((unsigned short)(v.z & 0xffffu)));
            b.y = __half2float(__ushort_as_half((unsigned short)(v.z >> 16)));
            b.z = __half2float(__ushort_as_half((unsigned short)(v.w & 0xffffu)));
            b.w = __half2float(__ushort_as_half((unsigned short)(v.w >> 16)));
            *reinterpret_cast<float4 *>(out + g * 8) = a;
            *reinterpret_cast<float4 *>(out + g * 8 + 4) = b;
        }
    }
    for (uint64_t e = (aligned ? groups * 8 : 0) + (uint64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (uint64_t)gridDim.x * 256)
        out[e] = __half2float(__ushort_as_half(in[e]));
}

// One item per thread: the kernels' grid-stride loops run once (they only matter past 2^30 workgroups).  A persistent
// grid of 8 workgroups per CU looping over its items converted f16 -> f32 at 4.45 TB/s where this form reaches 5.9
// (profiles/ubench/stream_convert.hip): in a loop the wait for the next load (in-order vmcnt) is also a wait for the
// stores of the iteration before.  Non-temporal loads / stores: 3.2-3.6 TB/s, not used.
uint32_t stream_grid(uint64_t total) {
    uint64_t b = (total + 255) / 256;
    if (b > (1ull << 30)) b = 1ull << 30;
    if (b < 1) b = 1;
    return (uint32_t)b;
}
// (k_decode_f32 -- one code byte in, a gather from L2, 16 bytes out -- keeps the persistent grid: 0.132-0.136 ms against
// 0.139 with one item per thread at 1M x 128)
uint32_t persistent_grid(uint64_t total) {
    const uint64_t cap = (uint64_t)num_cus() * 8;
    const uint64_t b = (total + 255) / 256;
    return (uint32_t)std::max<uint64_t>(1, std::min(b, cap));
}

}  // namespace

// The same with the codebooks in LDS (m k sd floats <= 36864: C2's 128 KB): the gather leaves the memory pipe, which then
// carries the code bytes in and one 16-byte store per lane out.  One workgroup of 1024 per CU, U items per lane and trip:
// their code bytes requested together, then their table reads, then their stores -- the stores of a trip sit in front of
// the next trip's loads on the in-order counter, U of them per wait instead of one.
template <int U>
__global__ __launch_bounds__(1024) void k_decode_f32_lds(const float *__restrict__ cb, uint32_t m, uint32_t k, uint32_t sd,
                                                         const uint8_t *__restrict__ codes, uint64_t n, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds_cb[];
    const uint32_t cb_floats = m * k * sd;  // (a multiple of 4: sd is)
    for (uint32_t e = 4 * threadIdx.x; e < cb_floats; e += 4096) *reinterpret_cast<float4 *>(lds_cb + e) = *reinterpret_cast<const float4 *>(cb + e);
    __syncthreads();
    const uint32_t d = m * sd, gpr = d / 4;
    const uint64_t total = n * gpr;
    for (uint64_t base = (uint64_t)blockIdx.x * (1024 * U); base < total; base += (uint64_t)gridDim.x * (1024 * U)) {
        const uint64_t row0 = base / gpr;  // (uniform: scalar unit)
        const uint32_t g0 = (uint32_t)(base - row0 * gpr) + threadIdx.x;
        uint32_t code[U], col[U];
        uint64_t row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + (uint32_t)u * 1024u, dr = g / gpr;
            row[u] = min(row0 + dr, n - 1);  // (clamped: loads never under a test; the store is)
            col[u] = (g - dr * gpr) * 4;
            code[u] = load_code(codes, row[u] * m + col[u] / sd, k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(code[u]));  // all U code loads requested before the first is used
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t s = col[u] / sd, t = col[u] - s * sd;
            v[u] = *reinterpret_cast<const float4 *>(lds_cb + ((size_t)s * k + code[u]) * sd + t);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + (uint64_t)u * 1024u + threadIdx.x < total) *reinterpret_cast<float4 *>(out + row[u] * d + col[u]) = v[u];
    }
}

// k_gather_f16 with the codebooks in LDS (as k_decode_f32_lds): eight f16 per lane and item, U items per lane and trip
// (U = 2: 62 us per 1M x 128; 1: 74, 4: 66, 8: 68.5, 16: 170; the decode: U = 4: 101.6 us; 2: 110, 8: 108, 16: 112)
template <int U>
__global__ __launch_bounds__(1024) void k_gather_f16_lds(const float *__restrict__ cb, uint32_t m, uint32_t k, uint32_t sd,
                                                         const uint8_t *__restrict__ codes, uint64_t n, uint16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds_cb[];
    const uint32_t cb_floats = m * k * sd;
    for (uint32_t e = 4 * threadIdx.x; e < cb_floats; e += 4096) *reinterpret_cast<float4 *>(lds_cb + e) = *reinterpret_cast<const float4 *>(cb + e);
    __syncthreads();
    const uint32_t d = m * sd, gpr = d / 8;
    const uint64_t total = n * gpr;
    for (uint64_t base = (uint64_t)blockIdx.x * (1024 * U); base < total; base += (uint64_t)gridDim.x * (1024 * U)) {
        const uint64_t row0 = base / gpr;  // (uniform: scalar unit)
        const uint32_t g0 = (uint32_t)(base - row0 * gpr) + threadIdx.x;
        uint32_t code[U], col[U];
        uint64_t row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + (uint32_t)u * 1024u, dr = g / gpr;
            row[u] = min(row0 + dr, n - 1);
            col[u] = (g - dr * gpr) * 8;
            code[u] = load_code(codes, row[u] * m + col[u] / sd, k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(code[u]));
        uint4 o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t s = col[u] / sd, t = col[u] - s * sd;
            const float *src = lds_cb + ((size_t)s * k + code[u]) * sd + t;
            const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
            o[u].x = (uint32_t)__half_as_ushort(__float2half_rn(a.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(a.y)) << 16);
            o[u].y = (uint32_t)__half_as_ushort(__float2half_rn(a.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(a.w)) << 16);
            o[u].z = (uint32_t)__half_as_ushort(__float2half_rn(b.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(b.y)) << 16);
            o[u].w = (uint32_t)__half_as_ushort(__float2half_rn(b.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(b.w)) << 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + (uint64_t)u * 1024u + threadIdx.x < total) *reinterpret_cast<uint4 *>(out + row[u] * d + col[u]) = o[u];
    }
}

int launch_gather_f16(const CodebookView &cb, const uint8_t *codes, uint64_t n, uint16_t *f16_out,
                      hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    const bool vec8 = (cb.sd % 8 == 0) && ((reinterpret_cast<uintptr_t>(f16_out) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(cb.cb) & 15) == 0);
    const size_t cb_bytes = (size_t)cb.m * cb.k * cb.sd * 4;
    static const char *lds_env = getenv("VQHIP_DECODE_LDS");  // =0: the gather from L2 (A/B)
    if (vec8 && cb_bytes <= 144 * 1024 && n * (uint64_t)(cb.m * cb.sd / 8) >= (1u << 19) && (uint64_t)cb.m * cb.sd / 8 + 8192 < (1ull << 31) &&
        !(lds_env && lds_env[0] == '0')) {
        constexpr int U = 2;
        static PerDeviceOnce attr;
        if (attr.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather_f16_lds<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            attr.done();
        }
        const uint64_t trips = (n * (uint64_t)(cb.m * cb.sd / 8) + 1024 * U - 1) / (1024 * U);
        hipLaunchKernelGGL(k_gather_f16_lds<U>, dim3((uint32_t)std::min<uint64_t>(trips, (uint64_t)num_cus())), dim3(1024), cb_bytes, stream, cb.cb, cb.m,
                           cb.k, cb.sd, codes, n, f16_out);
        VQ_LAUNCH_CHECK("k_gather_f16_lds");
        return VQHIP_OK;
    }
    if (vec8)
        hipLaunchKernelGGL(k_gather_f16<8>, dim3(stream_grid(n * cb.m * cb.sd / 8)), dim3(256), 0,
                           stream, cb.cb, cb.m, cb.k, cb.sd, codes, n, f16_out);
    else
        hipLaunchKernelGGL(k_gather_f16<1>, dim3(stream_grid(n * cb.m * cb.sd)), dim3(256), 0, stream,
                           cb.cb, cb.m, cb.k, cb.sd, codes, n, f16_out);
    VQ_LAUNCH_CHECK("k_gather_f16");
    return VQHIP_OK;
}

int launch_decode_f32(const CodebookView &cb, const uint8_t *codes, uint64_t n, float *out,
                      hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    const bool vec4 = (cb.sd % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((reinterpret_cast<uintptr_t>(cb.cb) & 15) == 0) &&
                      (uint64_t)cb.m * cb.sd / 4 + 256 < (1ull << 31);
    const size_t cb_bytes = (size_t)cb.m * cb.k * cb.sd * 4;
    static const char *lds_env = getenv("VQHIP_DECODE_LDS");  // =0: the gather from L2 (A/B)
    if (vec4 && cb_bytes <= 144 * 1024 && n * (uint64_t)(cb.m * cb.sd / 4) >= (1u << 20) && !(lds_env && lds_env[0] == '0')) {
        constexpr int U = 4;
        static PerDeviceOnce attr;
        if (attr.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_f32_lds<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            attr.done();
        }
        const uint64_t trips = (n * (uint64_t)(cb.m * cb.sd / 4) + 1024 * U - 1) / (1024 * U);
        hipLaunchKernelGGL(k_decode_f32_lds<U>, dim3((uint32_t)std::min<uint64_t>(trips, (uint64_t)num_cus())), dim3(1024), cb_bytes, stream, cb.cb, cb.m,
                           cb.k, cb.sd, codes, n, out);
    } else if (vec4)
        hipLaunchKernelGGL(k_decode_f32<4>, dim3(persistent_grid(n * cb.m * cb.sd / 4)), dim3(256), 0, stream, cb.cb, cb.m, cb.k, cb.sd, codes, n, out);
    else
        hipLaunchKernelGGL(k_decode_f32<1>, dim3(persistent_grid(n * cb.m * cb.sd)), dim3(256), 0, stream, cb.cb, cb.m, cb.k, cb.sd, codes, n, out);
    VQ_LAUNCH_CHECK("k_decode_f32");
    return VQHIP_OK;
}

int launch_dequant_f16(const uint16_t *in, uint64_t count, float *out, hipStream_t stream) {
    if (count == 0) return VQHIP_OK;
    hipLaunchKernelGGL(k_dequant_f16, dim3(stream_grid((count + 7) / 8)), dim3(256), 0, stream, in, count, out);
    VQ_LAUNCH_CHECK("k_dequant_f16");
    return VQHIP_OK;
}

int launch_synth_uniform(float *X, uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset,
                         hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    hipLaunchKernelGGL(k_synth_uniform, dim3(stream_grid(n * d)), dim3(256), 0, stream, X, n, d,
                       seed, row_offset);
    VQ_LAUNCH_CHECK("k_synth_uniform");
    return VQHIP_OK;
}

void synth_uniform_host(float *out, uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset) {
    for (uint64_t r = 0; r < n; ++r)
        for (uint32_t c = 0; c < d; ++c) out[r * d + c] = synth_value(seed, row_offset + r, c, d);
}

}  // namespace vqhip
