// k_update.hip -- centroid update of one Lloyd iteration (gfx950).
//
// Replaces src/core/vector.rs:432-447 (bucket rows by cluster, per-cluster mean, 1e-6
// convergence test).  Rows are never bucketed: each row's sub-vectors are added straight
// into per-cluster accumulators selected by the assignment code.
//
//   accumulate : HBM-bound segmented sum.  One workgroup per (row chunk, subspace chunk);
//                the chunk's accumulators [subspaces][k][sub_dim] f32 + counts live in LDS
//                (up to 152 KiB of the CU's 160 KiB) and are updated with LDS float atomics;
//                X is read once, 16 B per lane, fully coalesced; codes come from L2.
//                Each workgroup stores its accumulators as one partial slab (plain stores,
//                no global float atomics).
//   reduce     : fixed-order f64 combination of the partial slabs -> slab [m][k][sd+1]
//                (last column = member count).  This slab is what row-sharded multi-GPU
//                training all-reduces.
//   finalize   : mean = sum / count, `changed` iff |new-old| >= 1e-6 in some component of a
//                non-empty cluster (vector.rs:232-240, 444-446); empty clusters keep their
//                centroid (the caller reseeds, vector.rs:448-452).
//
// Algorithmic bytes per row: 4*d + m.  Roofline: HBM.
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr uint32_t kAccBlock = 1024;
constexpr uint32_t kLdsBudgetWords = 38912;  // 152 KiB of the CU's 160 KiB

template <int VEC>
__global__ __launch_bounds__(kAccBlock) void k_accumulate(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd,
    uint32_t spc, uint64_t rows_per_chunk, const uint8_t *__restrict__ codes,
    const uint8_t *__restrict__ active, float *__restrict__ partial_sums,
    uint32_t *__restrict__ partial_counts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t s0 = blockIdx.y * spc;
    const uint32_t ns = (m - s0 < spc) ? (m - s0) : spc;
    const uint32_t W = ns * sd;            // floats of a row handled by this workgroup
    float *sums = lds;                     // [ns][k][sd]
    uint32_t *cnts = reinterpret_cast<uint32_t *>(lds + (size_t)spc * k * sd);  // [ns][k]
    const uint32_t words = ns * k * sd;
    for (uint32_t e = threadIdx.x; e < words; e += kAccBlock) sums[e] = 0.0f;
    for (uint32_t e = threadIdx.x; e < ns * k; e += kAccBlock) cnts[e] = 0u;
    __syncthreads();

    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_chunk;
    uint64_t r1 = r0 + rows_per_chunk;
    if (r1 > n) r1 = n;
    if (r0 < r1) {
        const uint32_t ipr = W / VEC;  // items per row
        const uint32_t items = (uint32_t)(r1 - r0) * ipr;
        for (uint32_t it = threadIdx.x; it < items; it += kAccBlock) {
            const uint32_t rr = it / ipr, q = it - rr * ipr;
            const uint32_t col = q * VEC;
            const uint32_t ls = col / sd, t = col - ls * sd;
            const uint32_t s = s0 + ls;
            if (active && !active[s]) continue;
            const uint64_t row = r0 + rr;
            const uint32_t code = codes[row * m + s];
            const float *px = X + row * d + (size_t)s0 * sd + col;
            float *dst = sums + ((size_t)ls * k + code) * sd + t;
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(px);
                atomicAdd(dst + 0, v.x);
                atomicAdd(dst + 1, v.y);
                atomicAdd(dst + 2, v.z);
                atomicAdd(dst + 3, v.w);
            } else {
                atomicAdd(dst, px[0]);
            }
            if (t == 0) atomicAdd(&cnts[ls * k + code], 1u);
        }
    }
    __syncthreads();
    // partial slab of this row chunk: sums [m][k][sd], counts [m][k]
    float *ps = partial_sums + ((size_t)blockIdx.x * m + s0) * k * sd;
    for (uint32_t e = threadIdx.x; e < words; e += kAccBlock) ps[e] = sums[e];
    uint32_t *pc = partial_counts + ((size_t)blockIdx.x * m + s0) * k;
    for (uint32_t e = threadIdx.x; e < ns * k; e += kAccBlock) pc[e] = cnts[e];
}

__global__ __launch_bounds__(256) void k_reduce_partials(
    const float *__restrict__ partial_sums, const uint32_t *__restrict__ partial_counts,
    uint32_t n_row_chunks, uint32_t m, uint32_t k, uint32_t sd, const uint8_t *__restrict__ active,
    double *__restrict__ slab) {
    const uint32_t total = m * k * (sd + 1);
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const uint32_t t = e % (sd + 1), sj = e / (sd + 1), s = sj / k;
    double acc = 0.0;
    if (!active || active[s]) {
        if (t < sd) {
            const size_t stride = (size_t)m * k * sd;
            const float *p = partial_sums + (size_t)sj * sd + t;
            for (uint32_t c = 0; c < n_row_chunks; ++c) acc += (double)p[c * stride];
        } else {
            const size_t stride = (size_t)m * k;
            const uint32_t *p = partial_counts + sj;
            unsigned long long cnt = 0;
            for (uint32_t c = 0; c < n_row_chunks; ++c) cnt += p[c * stride];
            acc = (double)cnt;
        }
    }
    slab[e] = acc;
}

__global__ __launch_bounds__(256) void k_finalize(uint32_t m, uint32_t k, uint32_t sd,
                                                  const double *__restrict__ slab,
                                                  const uint8_t *__restrict__ active,
                                                  float *__restrict__ centroids,
                                                  uint32_t *__restrict__ counts,
                                                  uint32_t *__restrict__ changed, int exact_div) {
    const uint32_t sj = blockIdx.x * 256 + threadIdx.x;
    if (sj >= m * k) return;
    const uint32_t s = sj / k;
    const bool act = !active || active[s];
    const double *row = slab + (size_t)sj * (sd + 1);
    const double cnt = row[sd];
    if (counts) counts[sj] = act ? (uint32_t)cnt : 0u;
    if (!act || !(cnt > 0.0)) return;
    float *c = centroids + (size_t)sj * sd;
    bool moved = false;
    const float EPSILON = 1e-6f;  // vector.rs:439
    const float nf = (float)cnt;   // indices.len() as f32, vector.rs:373
    for (uint32_t t = 0; t < sd; ++t) {
        float nv;
        if (exact_div) nv = (float)row[t] / nf;  // row[t] holds an exact f32 value
        else nv = (float)(row[t] / cnt);
        const float diff = nv - c[t];
        if (!(fabsf(diff) < EPSILON)) moved = true;
        c[t] = nv;
    }
    if (moved) atomicOr(&changed[s], 1u);
}

__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ X, uint32_t d,
                                                     uint32_t m, uint32_t k, uint32_t sd,
                                                     const uint64_t *__restrict__ rows,
                                                     float *__restrict__ centroids) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * k * sd) return;
    const uint32_t t = e % sd, sj = e / sd, s = sj / k;
    centroids[e] = X[rows[sj] * d + (size_t)s * sd + t];
}

}  // namespace

int plan_update(uint32_t m, uint32_t k, uint32_t sd, uint64_t n, UpdatePlan *p) {
    p->m = m;
    p->k = k;
    p->sd = sd;
    const uint64_t per_sub = (uint64_t)k * (sd + 1);
    if (per_sub > kLdsBudgetWords)
        return fail(VQHIP_ERR_UNSUPPORTED,
                    "k*(sub_dim+1)=%llu accumulators exceed one CU's LDS (%u words)",
                    (unsigned long long)per_sub, kLdsBudgetWords);
    uint32_t spc = (uint32_t)(kLdsBudgetWords / per_sub);
    if (spc > m) spc = m;
    p->subs_per_chunk = spc;
    p->n_sub_chunks = (m + spc - 1) / spc;
    uint32_t target = (uint32_t)num_cus();
    uint32_t rc = target / p->n_sub_chunks;
    if (rc < 1) rc = 1;
    // at least ~512 rows per chunk so that the slab write-out stays a small fraction
    uint64_t max_rc = (n + 511) / 512;
    if (max_rc < 1) max_rc = 1;
    if (rc > max_rc) rc = (uint32_t)max_rc;
    p->n_row_chunks = rc;
    p->partial_floats = (size_t)m * k * sd;
    p->partial_counts = (size_t)m * k;
    return VQHIP_OK;
}

int launch_accumulate(const UpdatePlan &p, const float *X, uint64_t n, uint32_t d,
                      const uint8_t *codes, const uint8_t *active, float *partial_sums,
                      uint32_t *partial_counts, hipStream_t stream) {
    const uint64_t rows_per_chunk = (n + p.n_row_chunks - 1) / p.n_row_chunks;
    if (rows_per_chunk * (uint64_t)p.subs_per_chunk * p.sd >= (1ull << 32))
        return fail(VQHIP_ERR_UNSUPPORTED, "row chunk too large for 32-bit item index");
    const size_t lds_bytes = ((size_t)p.subs_per_chunk * p.k * (p.sd + 1)) * 4;
    dim3 grid(p.n_row_chunks, p.n_sub_chunks);
    const bool vec4 = (p.sd % 4 == 0) && (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    if (vec4) {
        static bool attr_set4 = false;
        if (!attr_set4) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set4 = true;
        }
        hipLaunchKernelGGL(k_accumulate<4>, grid, dim3(kAccBlock), lds_bytes, stream, X, n, d, p.m,
                           p.k, p.sd, p.subs_per_chunk, rows_per_chunk, codes, active, partial_sums,
                           partial_counts);
    } else {
        static bool attr_set1 = false;
        if (!attr_set1) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set1 = true;
        }
        hipLaunchKernelGGL(k_accumulate<1>, grid, dim3(kAccBlock), lds_bytes, stream, X, n, d, p.m,
                           p.k, p.sd, p.subs_per_chunk, rows_per_chunk, codes, active, partial_sums,
                           partial_counts);
    }
    VQ_LAUNCH_CHECK("k_accumulate");
    return VQHIP_OK;
}

int launch_reduce_partials(const UpdatePlan &p, const float *partial_sums,
                           const uint32_t *partial_counts, const uint8_t *active, double *slab,
                           hipStream_t stream) {
    const uint32_t total = p.m * p.k * (p.sd + 1);
    hipLaunchKernelGGL(k_reduce_partials, dim3((total + 255) / 256), dim3(256), 0, stream,
                       partial_sums, partial_counts, p.n_row_chunks, p.m, p.k, p.sd, active, slab);
    VQ_LAUNCH_CHECK("k_reduce_partials");
    return VQHIP_OK;
}

int launch_finalize(uint32_t m, uint32_t k, uint32_t sd, const double *slab, const uint8_t *active,
                    float *centroids, uint32_t *counts, uint32_t *changed, int exact_div,
                    hipStream_t stream) {
    VQ_HIP(hipMemsetAsync(changed, 0, (size_t)m * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_finalize, dim3((m * k + 255) / 256), dim3(256), 0, stream, m, k, sd, slab,
                       active, centroids, counts, changed, exact_div);
    VQ_LAUNCH_CHECK("k_finalize");
    return VQHIP_OK;
}

int launch_gather_rows(const float *X, uint32_t d, uint32_t m, uint32_t k, uint32_t sd,
                       const uint64_t *rows, float *centroids, hipStream_t stream) {
    const uint32_t total = m * k * sd;
    hipLaunchKernelGGL(k_gather_rows, dim3((total + 255) / 256), dim3(256), 0, stream, X, d, m, k,
                       sd, rows, centroids);
    VQ_LAUNCH_CHECK("k_gather_rows");
    return VQHIP_OK;
}

size_t exact_sums_workspace_bytes(uint32_t, uint32_t, uint64_t) { return 0; }

int launch_exact_sums(uint32_t, uint32_t, uint32_t, const float *, uint64_t, uint32_t,
                      const uint8_t *, const uint8_t *, void *, size_t, double *, hipStream_t) {
    return fail(VQHIP_ERR_UNSUPPORTED, "exact (reference-order) centroid sums are not built yet");
}

}  // namespace vqhip
