// Latency path for the reference-shaped per-vector calls (`quantize(&[f32])`, src/pq.rs:167-199 and
// src/tsvq.rs:239-255): for a handful of rows the batch pipeline (upload, screen, re-check, gather,
// read-back: ~12 stream operations) costs 60-130 us per call.  Here ONE kernel reads the rows from
// mapped pinned host memory, decides them in the reference's exact arithmetic and writes codes /
// leaf ids and the f16 reconstruction straight back to pinned host memory.
#include <hip/hip_fp16.h>

#include "common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

// Distance::compute, scalar kernels, runtime length (src/core/distance.rs:76-82, 94, 107-119); for
// cosine `na` / `nb` are the two norms (sqrt of the sequential sums of squares)
__device__ float small_distance(int metric, const float *__restrict__ x, const float *__restrict__ c, uint32_t n,
                                float na, float nb) {
    if (metric == VQHIP_SQUARED_EUCLIDEAN || metric == VQHIP_EUCLIDEAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            const float diff = x[t] - c[t];
            const float sq = diff * diff;
            acc = acc + sq;
        }
        return metric == VQHIP_EUCLIDEAN ? sqrtf(acc) : acc;
    }
    if (metric == VQHIP_MANHATTAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            const float diff = x[t] - c[t];
            acc = acc + fabsf(diff);
        }
        return acc;
    }
    float dot = -0.0f;
    for (uint32_t t = 0; t < n; ++t) {
        const float p = x[t] * c[t];
        dot = dot + p;
    }
    return vq_cosine_finish(metric, dot, na, nb);
}

// one wave per (row, subspace): lane l scans centroids l, l+64, ... ascending with strict '<', the 64
// partial winners merge on (distance, index); a NaN distance at centroid 0 blocks every later
// `dist < best` (src/pq.rs:184-190) -> code 0
__global__ __launch_bounds__(64) void k_pq_encode_small(const float *__restrict__ rows, uint32_t d, uint32_t m,
                                                        uint32_t k, uint32_t sd, int metric,
                                                        const float *__restrict__ cb, const float *__restrict__ cnsqrt,
                                                        uint8_t *__restrict__ codes, uint16_t *__restrict__ f16_out) {
    extern __shared__ float xs[];  // [sd]
    const uint32_t row = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const float *xrow = rows + (size_t)row * d + (size_t)s * sd;
    for (uint32_t t = lane; t < sd; t += 64) xs[t] = xrow[t];
    __syncthreads();
    float na = 0.0f;
    if (vq_is_cos(metric)) {
        float sa = -0.0f;
        for (uint32_t t = 0; t < sd; ++t) {
            const float p = xs[t] * xs[t];
            sa = sa + p;
        }
        na = sqrtf(sa);
    }
    const float *cbs = cb + (size_t)s * k * sd;
    const uint32_t NONE = 0xFFFFFFFFu;
    float bd = __builtin_inff();
    uint32_t bj = NONE;
    bool d0_nan = false;
    for (uint32_t j = lane; j < k; j += 64) {
        const float dist = small_distance(metric, xs, cbs + (size_t)j * sd, sd, na,
                                          vq_is_cos(metric) ? cnsqrt[(size_t)s * k + j] : 0.0f);
        const bool isnan_d = dist != dist;
        if (j == 0) d0_nan = isnan_d;
        if (!isnan_d && (bj == NONE || dist < bd)) {
            bd = dist;
            bj = j;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const uint32_t oj = (uint32_t)__shfl_xor((int)bj, off);
        const bool take = (oj != NONE) && (bj == NONE || od < bd || (od == bd && oj < bj));
        bd = take ? od : bd;
        bj = take ? oj : bj;
    }
    const bool blocked = __shfl((int)d0_nan, 0) != 0;
    const uint32_t best = (blocked || bj == NONE) ? 0u : bj;
    if (codes && lane == 0) store_code(codes, (size_t)row * m + s, best, k);
    if (f16_out) {
        const float *c = cbs + (size_t)best * sd;
        for (uint32_t t = lane; t < sd; t += 64)
            f16_out[(size_t)row * d + (size_t)s * sd + t] = __half_as_ushort(__float2half_rn(c[t]));  // pq.rs:192-196
    }
}

// one wave per row walks the tree (find_leaf, src/tsvq.rs:117-132): the per-dimension terms of both
// child distances are formed by the lanes in parallel, the two sums are then carried in dimension
// order through v_readlane, i.e. with the reference's sequential roundings
constexpr uint32_t kSmallMaxChunks = 32;  // d <= 2048

template <uint32_t NCH>  // 64-dimension chunks held per lane (fully unrolled: the row stays in registers)
__global__ __launch_bounds__(64) void k_tsvq_encode_small(const float *__restrict__ rows, uint32_t d, int metric,
                                                          const float *__restrict__ centroids,
                                                          const float *__restrict__ cnorm,
                                                          const int32_t *__restrict__ left,
                                                          const int32_t *__restrict__ right,
                                                          int32_t *__restrict__ leaf_out, uint16_t *__restrict__ f16_out) {
    const uint32_t row = blockIdx.x, lane = threadIdx.x;
    constexpr uint32_t nch = NCH;
    float x[NCH];
#pragma unroll
    for (uint32_t i = 0; i < nch; ++i) {
        const uint32_t t = i * 64 + lane;
        x[i] = (t < d) ? rows[(size_t)row * d + t] : 0.0f;
    }
    float na = 0.0f;
    if (vq_is_cos(metric)) {
        float sa = -0.0f;
#pragma unroll
        for (uint32_t i = 0; i < nch; ++i) {
            if (i * 64 >= d) break;
            const float p = x[i] * x[i];
            const uint32_t cnt = min(64u, d - i * 64);
            for (uint32_t l = 0; l < cnt; ++l) sa = sa + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), (int)l));
        }
        na = sqrtf(sa);
    }
    int32_t node = 0;
    for (;;) {
        const int32_t l = left[node], r = right[node];
        if (l >= 0 && r >= 0) {
            const float *cl = centroids + (size_t)l * d, *cr = centroids + (size_t)r * d;
            float al = -0.0f, ar = -0.0f;
#pragma unroll
            for (uint32_t i = 0; i < nch; ++i) {
                if (i * 64 >= d) break;
                const uint32_t t = i * 64 + lane;
                float tl = 0.0f, tr = 0.0f;
                if (t < d) {
                    const float a = cl[t], b = cr[t];
                    if (vq_is_cos(metric)) {
                        tl = x[i] * a;
                        tr = x[i] * b;
                    } else {
                        const float d1 = x[i] - a, d2 = x[i] - b;
                        if (metric == VQHIP_MANHATTAN) {
                            tl = fabsf(d1);
                            tr = fabsf(d2);
                        } else {
                            tl = d1 * d1;
                            tr = d2 * d2;
                        }
                    }
                }
                const uint32_t cnt = min(64u, d - i * 64);
                if (cnt == 64) {  // full chunk: straight-line chain, no loop branches on the critical path
#pragma unroll
                    for (int q = 0; q < 64; ++q) {
                        al = al + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tl), q));
                        ar = ar + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tr), q));
                    }
                } else {
                    for (uint32_t q = 0; q < cnt; ++q) {
                        al = al + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tl), (int)q));
                        ar = ar + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tr), (int)q));
                    }
                }
            }
            float dl = al, dr = ar;
            if (metric == VQHIP_EUCLIDEAN) {
                dl = sqrtf(al);
                dr = sqrtf(ar);
            } else if (vq_is_cos(metric)) {
                dl = vq_cosine_finish(metric, al, na, cnorm[l]);
                dr = vq_cosine_finish(metric, ar, na, cnorm[r]);
            }
            node = (dl <= dr) ? l : r;  // left on ties, tsvq.rs:122
        } else if (l >= 0) {
            node = l;
        } else if (r >= 0) {
            node = r;
        } else {
            break;
        }
    }
    if (leaf_out && lane == 0) leaf_out[row] = node;
    if (f16_out) {
        const float *c = centroids + (size_t)node * d;
        for (uint32_t t = lane; t < d; t += 64) f16_out[(size_t)row * d + t] = __half_as_ushort(__float2half_rn(c[t]));
    }
}

}  // namespace

int launch_pq_encode_small(const float *rows_dev, uint32_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, int metric,
                           const float *cb, const float *cnsqrt, uint8_t *codes_dev, uint16_t *f16_dev,
                           hipStream_t stream) {
    hipLaunchKernelGGL(k_pq_encode_small, dim3(n, m), dim3(64), (size_t)sd * 4, stream, rows_dev, d, m, k, sd, metric, cb,
                       cnsqrt, codes_dev, f16_dev);
    VQ_LAUNCH_CHECK("k_pq_encode_small");
    return VQHIP_OK;
}

bool tsvq_small_supported(uint32_t d) { return d <= 64 * kSmallMaxChunks; }

int launch_tsvq_encode_small(const float *rows_dev, uint32_t n, uint32_t d, int metric, const float *centroids,
                             const float *cnorm, const int32_t *left, const int32_t *right, int32_t *leaf_dev,
                             uint16_t *f16_dev, hipStream_t stream) {
    const uint32_t nch = (d + 63) / 64;
#define VQ_SMALL_TSVQ(NCHV)                                                                                        \
    hipLaunchKernelGGL(k_tsvq_encode_small<NCHV>, dim3(n), dim3(64), 0, stream, rows_dev, d, metric, centroids, cnorm, \
                       left, right, leaf_dev, f16_dev)
    if (nch <= 1) VQ_SMALL_TSVQ(1);
    else if (nch <= 2) VQ_SMALL_TSVQ(2);
    else if (nch <= 4) VQ_SMALL_TSVQ(4);
    else if (nch <= 8) VQ_SMALL_TSVQ(8);
    else if (nch <= 16) VQ_SMALL_TSVQ(16);
    else VQ_SMALL_TSVQ(32);
#undef VQ_SMALL_TSVQ
    VQ_LAUNCH_CHECK("k_tsvq_encode_small");
    return VQHIP_OK;
}

}  // namespace vqhip
