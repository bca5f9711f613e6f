// k_exact.hip -- exact nearest-centroid scan in the reference's arithmetic (gfx950).
//
// Every distance is evaluated with the reference's operation order: sequential f32,
// one rounding per operation, no fused multiply-add (src/core/vector.rs:135-143,
// src/core/distance.rs:76-82, 94, 107-119), and the argmin keeps the FIRST minimum
// (strict '<', src/core/vector.rs:355-361, src/pq.rs:185-191).  This kernel is
//   (1) the re-check stage behind the MFMA screen (k_screen.hip) -- it alone decides the
//       rows the screen could not prove, and
//   (2) the assignment engine for shapes/metrics without an MFMA form (Manhattan, cosine,
//       sub_dim not a multiple of 4, k > 256-tile budget).
// It also prepares the per-codebook constants both engines need.
//
// Roofline: VALU.  3 ops per (row, centroid, dim) for L2 (sub, mul, add; they cannot fuse),
// algorithmic bytes 4*d in + m out per row.  Centroids are wave-uniform (one subspace per
// workgroup) so they arrive through the scalar cache and feed VALU ops as SGPR operands.
#include "kernels.hpp"
#include <type_traits>

// The reference never contracts a*b+c (Rust has no implicit FMA).  Belt and braces with
// -ffp-contract=off on the command line.
#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr int kExactBlock = 256;

template <int METRIC, int SD, bool GENERIC>
__device__ __forceinline__ uint32_t scan_one(const float *__restrict__ xrow, uint32_t sd_rt,
                                             const float *__restrict__ cbs,
                                             const float *__restrict__ cnsq, uint32_t k) {
    const uint32_t sd = GENERIC ? sd_rt : (uint32_t)SD;
    float x[GENERIC ? 1 : SD];
    if constexpr (!GENERIC) {
#pragma unroll
        for (int t = 0; t < SD; ++t) x[t] = xrow[t];
    }
    float na = 0.0f;
    if constexpr (vq_is_cos(METRIC)) {
        float sa = -0.0f;  // `.sum()` folds from -0.0 (Rust 1.85); invisible after sqrt/compare
        if constexpr (GENERIC) {
            for (uint32_t t = 0; t < sd; ++t) {
                float p = xrow[t] * xrow[t];
                sa = sa + p;
            }
        } else {
#pragma unroll
            for (int t = 0; t < SD; ++t) {
                float p = x[t] * x[t];
                sa = sa + p;
            }
        }
        na = sqrtf(sa);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    }
    uint32_t best = 0;
    float best_dist = 0.0f;
    for (uint32_t j = 0; j < k; ++j) {
        const float *c = cbs + (size_t)j * sd;
        float dist;
        if constexpr (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
            float acc = 0.0f;
            if constexpr (GENERIC) {
                for (uint32_t t = 0; t < sd; ++t) {
                    float diff = xrow[t] - c[t];
                    float sq = diff * diff;
                    acc = acc + sq;
                }
            } else {
#pragma unroll
                for (int t = 0; t < SD; ++t) {
                    float diff = x[t] - c[t];
                    float sq = diff * diff;
                    acc = acc + sq;
                }
            }
            dist = (METRIC == VQHIP_EUCLIDEAN) ? sqrtf(acc) : acc;
        } else if constexpr (METRIC == VQHIP_MANHATTAN) {
            float acc = 0.0f;
            if constexpr (GENERIC) {
                for (uint32_t t = 0; t < sd; ++t) {
                    float diff = xrow[t] - c[t];
                    acc = acc + fabsf(diff);
                }
            } else {
#pragma unroll
                for (int t = 0; t < SD; ++t) {
                    float diff = x[t] - c[t];
                    acc = acc + fabsf(diff);
                }
            }
            dist = acc;
        } else {  // cosine, src/core/distance.rs:107-119
            float dot = -0.0f;
            if constexpr (GENERIC) {
                for (uint32_t t = 0; t < sd; ++t) {
                    float p = xrow[t] * c[t];
                    dot = dot + p;
                }
            } else {
#pragma unroll
                for (int t = 0; t < SD; ++t) {
                    float p = x[t] * c[t];
                    dot = dot + p;
                }
            }
            const float nb = cnsq[j];  // sqrt(sum c^2): depends on c only, hoisted
            dist = vq_cosine_finish(METRIC, dot, na, nb);
        }
        if (j == 0) {
            best_dist = dist;
        } else if (dist < best_dist) {  // strict: first minimum wins; NaN never wins
            best_dist = dist;
            best = j;
        }
    }
    return best;
}

template <int METRIC, int SD, bool GENERIC>
__global__ __launch_bounds__(kExactBlock) void k_assign_exact(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd_rt,
    const float *__restrict__ cb, const float *__restrict__ cnsqrt,
    const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows,
    const uint32_t *__restrict__ wl_count, uint64_t wl_stride, uint8_t *__restrict__ codes) {
    const uint32_t sd = GENERIC ? sd_rt : (uint32_t)SD;
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint64_t count = wl_rows ? (uint64_t)wl_count[s] : n;
    const float *cbs = cb + (size_t)s * k * sd;
    const float *cnsq = cnsqrt ? cnsqrt + (size_t)s * k : nullptr;
    for (uint64_t i = (uint64_t)blockIdx.x * kExactBlock + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * kExactBlock) {
        const uint64_t row = wl_rows ? (uint64_t)wl_rows[(size_t)s * wl_stride + i] : i;
        const float *xrow = X + row * d + (size_t)s * sd;
        uint32_t best = scan_one<METRIC, SD, GENERIC>(xrow, sd, cbs, cnsq, k);
        store_code(codes, row * m + s, best, k);
    }
}

// Exact scan for ANY sub_dim (the engine behind sub_dims without a compile-time instantiation, and behind
// lbg_quantize on whole vectors, m = 1: src/core/vector.rs:390-395).  A lane cannot keep an arbitrary-length
// sub-vector in registers, and re-reading it from memory once per centroid (what the run-time-length loop of
// scan_one does) costs 4*k bytes of uncoalesced traffic per row element.  Here a workgroup owns 64 rows: the rows
// are staged through LDS 32 dimensions at a time (coalesced 128-byte reads, transposed so that lane r reads its own
// row conflict-free), the running distance of every (row, centroid) pair lives in LDS ([256 centroids][64 rows],
// 64 KB), and wave q advances the pairs of centroids j = q, q+4, ... by one chunk: acc -> 32 x (x - c, squared,
// added) -> acc, centroid elements wave-uniform scalar loads.  Every pair still sees its additions in ascending t
// from the reference's start value, so the bits are scan_one's; the rows are read from HBM once per 256 centroids.
constexpr uint32_t kTiledRows = 64, kTiledDims = 32, kTiledCentroids = 256;
template <int METRIC>
__global__ __launch_bounds__(256) void k_assign_exact_tiled(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd,
    const float *__restrict__ cb, const float *__restrict__ cnsqrt, const uint32_t *__restrict__ sub_list,
    uint8_t *__restrict__ codes) {
    constexpr uint32_t TR = kTiledRows, TC = kTiledDims, KG = kTiledCentroids;
    __shared__ float acc[KG][TR];
    __shared__ float xs[TC][TR + 1];
    __shared__ float na_s[TR];  // cosine: |x| of the workgroup's rows
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint32_t r = threadIdx.x & 63;
    const uint32_t q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t row0 = (uint64_t)blockIdx.x * TR;
    const float *cbs = cb + (size_t)s * k * sd;
    const float *cnsq = cnsqrt ? cnsqrt + (size_t)s * k : nullptr;
    const float init = (vq_is_cos(METRIC)) ? -0.0f : 0.0f;  // `.sum()` folds from -0.0, the L2 / L1 loops from 0.0
    float sa = -0.0f, na = 0.0f;  // cosine: |x|^2 chain of this lane's row (wave 0 only)
    uint32_t best = 0;
    float best_dist = 0.0f;
    for (uint32_t kg0 = 0; kg0 < k; kg0 += KG) {
        const uint32_t kn = min(KG, k - kg0);
        for (uint32_t jj = q; jj < kn; jj += 4) acc[jj][r] = init;
        for (uint32_t t0 = 0; t0 < sd; t0 += TC) {
            const uint32_t tc = min(TC, sd - t0);
            __syncthreads();  // the previous chunk's readers are done with xs (and acc is initialised)
#pragma unroll
            for (uint32_t e = 0; e < TR * TC / 256; ++e) {
                const uint32_t idx = threadIdx.x + 256 * e, row = idx / TC, col = idx % TC;
                const uint64_t grow = row0 + row;
                xs[col][row] = (grow < n && col < tc) ? X[grow * d + (size_t)s * sd + t0 + col] : 0.0f;
            }
            __syncthreads();
            float xr[TC];
#pragma unroll
            for (uint32_t t = 0; t < TC; ++t) xr[t] = xs[t][r];
            if (vq_is_cos(METRIC) && kg0 == 0 && q == 0) {
#pragma unroll
                for (uint32_t t = 0; t < TC; ++t)
                    if (t < tc) {
                        const float p = xr[t] * xr[t];
                        sa = sa + p;
                    }
            }
            // one (row, centroid) pair advanced by N dimensions, N a compile-time constant: only the reference's own
            // operations, no per-dimension test (a run-time `t < tc` inside the unrolled loop cost two scalar
            // instructions per dimension and kept the short last chunk at a third of the full chunks' rate)
            auto advance = [&](float a, const float *__restrict__ c, auto nc) {
                constexpr uint32_t N = decltype(nc)::value;
#pragma unroll
                for (uint32_t t = 0; t < N; ++t) {
                    if constexpr (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
                        const float diff = xr[t] - c[t];
                        const float sq = diff * diff;
                        a = a + sq;
                    } else if constexpr (METRIC == VQHIP_MANHATTAN) {
                        const float diff = xr[t] - c[t];
                        a = a + fabsf(diff);
                    } else {
                        const float p = xr[t] * c[t];
                        a = a + p;
                    }
                }
                return a;
            };
            auto run_chunk = [&](auto nc) {
                constexpr uint32_t N = decltype(nc)::value;
                uint32_t jj = q;  // wave-uniform
                if constexpr (N == TC) {
                    for (; jj + 4 < kn; jj += 8) {  // two centroids per trip: both scalar loads are in flight together
                        const float *c0 = cbs + (size_t)(kg0 + jj) * sd + t0, *c1 = c0 + (size_t)4 * sd;
                        const float a0 = advance(acc[jj][r], c0, nc);
                        const float a1 = advance(acc[jj + 4][r], c1, nc);
                        acc[jj][r] = a0;
                        acc[jj + 4][r] = a1;
                    }
                }
                for (; jj < kn; jj += 4) acc[jj][r] = advance(acc[jj][r], cbs + (size_t)(kg0 + jj) * sd + t0, nc);
            };
#define VQ_TC(NV) \
    case NV: run_chunk(std::integral_constant<uint32_t, NV>{}); break;
            switch (tc) {
                VQ_TC(1) VQ_TC(2) VQ_TC(3) VQ_TC(4) VQ_TC(5) VQ_TC(6) VQ_TC(7) VQ_TC(8)
                VQ_TC(9) VQ_TC(10) VQ_TC(11) VQ_TC(12) VQ_TC(13) VQ_TC(14) VQ_TC(15) VQ_TC(16)
                VQ_TC(17) VQ_TC(18) VQ_TC(19) VQ_TC(20) VQ_TC(21) VQ_TC(22) VQ_TC(23) VQ_TC(24)
                VQ_TC(25) VQ_TC(26) VQ_TC(27) VQ_TC(28) VQ_TC(29) VQ_TC(30) VQ_TC(31)
            default: run_chunk(std::integral_constant<uint32_t, TC>{}); break;
            }
#undef VQ_TC
        }
        if (vq_is_cos(METRIC) && kg0 == 0 && q == 0) na_s[r] = sqrtf(sa);  // |x|, shared with the other waves
        __syncthreads();
        if constexpr (METRIC == VQHIP_EUCLIDEAN || vq_is_cos(METRIC)) {
            // accumulators -> distances, every wave its own centroids (the division and the square root are the
            // expensive part of a short sub-vector's scan; one wave doing all of them serialised the workgroup)
            if (vq_is_cos(METRIC)) na = na_s[r];
            for (uint32_t jj = q; jj < kn; jj += 4) {
                const float a = acc[jj][r];
                float dist;
                if constexpr (METRIC == VQHIP_EUCLIDEAN) {
                    dist = sqrtf(a);
                } else {  // src/core/distance.rs:107-119
                    const float nb = cnsq[kg0 + jj];
                    dist = vq_cosine_finish(METRIC, a, na, nb);
                }
                acc[jj][r] = dist;
            }
            __syncthreads();
        }
        if (q == 0) {  // this group's centroids in ascending order, strict '<' (scan_one's rule)
            for (uint32_t jj = 0; jj < kn; ++jj) {
                const uint32_t j = kg0 + jj;
                const float dist = acc[jj][r];
                if (j == 0) {
                    best_dist = dist;
                } else if (dist < best_dist) {
                    best_dist = dist;
                    best = j;
                }
            }
        }
        __syncthreads();  // acc is re-initialised by the next centroid group
    }
    if (q == 0 && row0 + r < n) store_code(codes, (row0 + r) * m + s, best, k);
}

// Exact distance of one row to one centroid, compile-time length (same op order as scan_one).
template <int METRIC, int SD>
__device__ __forceinline__ float exact_dist_fixed(const float (&x)[SD], const float *__restrict__ c,
                                                  float na, float nb) {
    if constexpr (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < SD; ++t) {
            float diff = x[t] - c[t];
            float sq = diff * diff;
            acc = acc + sq;
        }
        return (METRIC == VQHIP_EUCLIDEAN) ? sqrtf(acc) : acc;
    } else if constexpr (METRIC == VQHIP_MANHATTAN) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < SD; ++t) {
            float diff = x[t] - c[t];
            acc = acc + fabsf(diff);
        }
        return acc;
    } else {
        float dot = -0.0f;
#pragma unroll
        for (int t = 0; t < SD; ++t) {
            float p = x[t] * c[t];
            dot = dot + p;
        }
        return vq_cosine_finish(METRIC, dot, na, nb);
    }
}

// Re-check stage behind the MFMA screen: ONE WAVE per work-list entry.  Lane l evaluates
// centroids l, l+64, l+128, ... exactly (ascending, strict '<'), then the 64 partial winners
// are merged with "smaller distance, then smaller index" -- which equals the reference's
// sequential first-minimum scan as long as centroid 0's distance is not NaN; if it is, no
// later `dist < best` can ever be true and the answer is 0 (src/pq.rs:184-190).
template <int METRIC, int SD>
__global__ __launch_bounds__(1024) void k_recheck_wave(
    const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k,
    const float *__restrict__ cb, const float *__restrict__ cnsqrt,
    const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows,
    const uint32_t *__restrict__ wl_count, uint64_t wl_stride, const uint32_t *__restrict__ wl_seg,
    uint32_t n_seg, uint8_t *__restrict__ codes) {
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint32_t lane = threadIdx.x & 63;
    // two work-list forms: one global list per subspace (count in wl_count[s]), or wave-private
    // segments written without atomics by the X32 screen (wl_seg[s][seg] = {first slot, count});
    // in the segmented form workgroup blockIdx.x owns segment blockIdx.x
    uint32_t count, first, wave, n_waves;
    if (wl_seg) {
        const uint32_t *sg = wl_seg + ((size_t)s * n_seg + blockIdx.x) * 2;
        first = sg[0];
        count = sg[1];
        wave = threadIdx.x >> 6;
        n_waves = blockDim.x >> 6;
    } else {
        first = 0;
        count = wl_count[s];
        wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        n_waves = gridDim.x * (blockDim.x >> 6);
    }
    const float *cbs = cb + (size_t)s * k * SD;
    const float *cnsq = cnsqrt ? cnsqrt + (size_t)s * k : nullptr;
    const uint32_t NONE = 0xFFFFFFFFu;
    for (uint32_t i = wave; i < count; i += n_waves) {
        const uint64_t row = wl_rows[(size_t)s * wl_stride + first + i];
        const float *xrow = X + row * d + (size_t)s * SD;
        float x[SD];
#pragma unroll
        for (int t = 0; t < SD; ++t) x[t] = xrow[t];
        float na = 0.0f;
        if constexpr (vq_is_cos(METRIC)) {
            float sa = -0.0f;
#pragma unroll
            for (int t = 0; t < SD; ++t) {
                float p = x[t] * x[t];
                sa = sa + p;
            }
            na = sqrtf(sa);
        }
        float bd = __builtin_inff();
        uint32_t bj = NONE;
        bool d0_nan = false;
        for (uint32_t j = lane; j < k; j += 64) {
            const float dist = exact_dist_fixed<METRIC, SD>(x, cbs + (size_t)j * SD, na,
                                                            vq_is_cos(METRIC) ? cnsq[j] : 0.0f);
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
        if (lane == 0) store_code(codes, row * m + s, (blocked || bj == NONE) ? 0u : bj, k);
    }
}

// (distance, index) minimum over the wave without LDS traffic: DPP exchanges inside the 16-lane rows, v_readlane
// across them.  "Smaller distance, then smaller index" is commutative and associative, so any tree gives the
// reference's first minimum.  The result is wave-uniform.  (__shfl_xor is ds_bpermute: ~100 cycles a step, and
// its lgkmcnt waits would also serialise the scalar prefetch of the next entry's row.)
template <int CTRL>
__device__ __forceinline__ void argmin_dpp_step(float &bd, uint32_t &bj) {
    // a lane without a candidate carries (+inf, 0xFFFFFFFF), which loses every comparison below by itself
    const float od = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bd), __float_as_int(bd), CTRL, 0xF, 0xF, false));
    const uint32_t oj = (uint32_t)__builtin_amdgcn_update_dpp((int)bj, (int)bj, CTRL, 0xF, 0xF, false);
    const bool take = (od < bd) | ((od == bd) & (oj < bj));
    bd = take ? od : bd;
    bj = take ? oj : bj;
}
__device__ __forceinline__ void argmin_wave(float &bd, uint32_t &bj) {
    argmin_dpp_step<0xB1>(bd, bj);   // quad_perm [1,0,3,2]
    argmin_dpp_step<0x4E>(bd, bj);   // quad_perm [2,3,0,1]
    argmin_dpp_step<0x141>(bd, bj);  // row_half_mirror
    argmin_dpp_step<0x140>(bd, bj);  // row_mirror: every lane of a 16-lane row now holds the row's minimum
    float rd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), 0));
    uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)bj, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float od = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), r));
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)bj, r);
        const bool take = (od < rd) | ((od == rd) & (oj < rj));
        rd = take ? od : rd;
        rj = take ? oj : rj;
    }
    bd = rd;
    bj = rj;
}

// Re-check over the X32 screen's wave-private segments with the sub-codebook RESIDENT IN REGISTERS: lane l keeps
// centroids l, l+64, ... (KPL of them, KPL*SD <= 128 VGPRs) for the whole launch, so an entry costs its own
// 4*SD-byte row -- a uniform address, fetched one entry ahead -- instead of the 4*k*SD bytes of centroids that
// k_recheck_wave pulls through the L1 per entry (16 KB at k=256, sub_dim 16: clustered data, 2.7 % of the rows
// re-checked, spent 0.27 ms there).  A segment is shared by `parts` waves (contiguous shares).  Same arithmetic,
// same merge and NaN rule as k_recheck_wave.
template <int METRIC, int SD, int KPL>
__global__ __launch_bounds__(256) void k_recheck_resident(
    const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k,
    const float *__restrict__ cb, const float *__restrict__ cnsqrt,
    const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows, uint64_t wl_stride,
    const uint32_t *__restrict__ wl_seg, const uint32_t *__restrict__ wl_count, uint32_t n_seg, uint32_t parts,
    uint8_t *__restrict__ codes) {
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4;
    const float *cbs = cb + (size_t)s * k * SD;
    const uint32_t NONE = 0xFFFFFFFFu;
    // squared-L2 / Euclidean with an even KPL: two centroids per instruction on the packed-f32 VALU ops
    // (v_pk_add_f32 / v_pk_mul_f32 round each half like the scalar ops; the row comes in as a broadcast SGPR)
    constexpr bool PK = (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) && (KPL % 2 == 0);
    typedef float f2 __attribute__((ext_vector_type(2)));
    float c[PK ? 1 : KPL][SD];
    f2 cp[PK ? KPL / 2 : 1][SD];
    float cn[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const uint32_t j = lane + 64u * i;
        const float *src = cbs + (size_t)(j < k ? j : 0u) * SD;
#pragma unroll
        for (int t = 0; t < SD; t += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + t);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (PK) {
                    if (i % 2 == 0) cp[i / 2][t + u].x = e[u];
                    else cp[i / 2][t + u].y = e[u];
                } else {
                    c[i][t + u] = e[u];
                }
            }
        }
        cn[i] = (vq_is_cos(METRIC)) ? cnsqrt[(size_t)s * k + (j < k ? j : 0u)] : 0.0f;
    }
    const uint32_t *rows_s = wl_rows + (size_t)s * wl_stride;
    for (uint32_t unit = wave; unit < n_seg * parts; unit += n_waves) {
        const uint32_t seg = unit / parts, part = unit - seg * parts;
        // segmented lists, or (wl_seg == nullptr, n_seg == 1) the subspace's single list shared by `parts` waves
        const uint32_t *sg = wl_seg ? wl_seg + ((size_t)s * n_seg + seg) * 2 : nullptr;
        const uint32_t seg_first = sg ? sg[0] : 0u, seg_count = sg ? sg[1] : wl_count[s];
        const uint32_t share = (seg_count + parts - 1) / parts;
        const uint32_t lo = part * share, hi = min(seg_count, lo + share);
        for (uint32_t base = lo; base < hi; base += 64) {
            const uint32_t nb = min(64u, hi - base);
            // the batch's row ids, one per lane; entry e's id is read with v_readlane
            const uint32_t my_row = rows_s[seg_first + base + min(lane, nb - 1)];
            auto fetch = [&](uint32_t e, float (&x)[SD]) {
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, (int)e);
                const float *xrow = X + (size_t)row * d + (size_t)s * SD;  // wave-uniform address
#pragma unroll
                for (int t = 0; t < SD; t += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(xrow + t);
                    x[t] = v.x, x[t + 1] = v.y, x[t + 2] = v.z, x[t + 3] = v.w;
                }
            };
            float xn[SD];
            fetch(0, xn);
            for (uint32_t e = 0; e < nb; ++e) {
                float x[SD];
#pragma unroll
                for (int t = 0; t < SD; ++t) x[t] = xn[t];
                if (e + 1 < nb) fetch(e + 1, xn);  // the next entry's row, in flight during this one's arithmetic
                float na = 0.0f;
                if constexpr (vq_is_cos(METRIC)) {
                    float sa = -0.0f;
#pragma unroll
                    for (int t = 0; t < SD; ++t) {
                        float p = x[t] * x[t];
                        sa = sa + p;
                    }
                    na = sqrtf(sa);
                }
                float bd = __builtin_inff();
                uint32_t bj = NONE;
                bool d0_nan = false;
                auto consider = [&](int i, float dist) {  // centroid lane + 64 i, ascending i
                    const uint32_t j = lane + 64u * i;
                    const bool isnan_d = dist != dist;
                    if (i == 0) d0_nan = isnan_d && (lane == 0);
                    // selects, not branches (the four short-circuit tests cost more than the 48 flops they guard)
                    const bool better = (j < k) & !isnan_d & ((bj == NONE) | (dist < bd));
                    bd = better ? dist : bd;
                    bj = better ? j : bj;
                };
                if constexpr (PK) {
#pragma unroll
                    for (int p = 0; p < KPL / 2; ++p) {
                        f2 acc = {0.0f, 0.0f};
#pragma unroll
                        for (int t = 0; t < SD; ++t) {  // same three roundings per element as exact_dist_fixed
                            const f2 xx = {x[t], x[t]};
                            const f2 diff = xx - cp[p][t];
                            const f2 sq = diff * diff;
                            acc = acc + sq;
                        }
                        consider(2 * p, (METRIC == VQHIP_EUCLIDEAN) ? sqrtf(acc.x) : acc.x);
                        consider(2 * p + 1, (METRIC == VQHIP_EUCLIDEAN) ? sqrtf(acc.y) : acc.y);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < KPL; ++i) consider(i, exact_dist_fixed<METRIC, SD>(x, c[i], na, cn[i]));
                }
                argmin_wave(bd, bj);
                const bool blocked = __builtin_amdgcn_readlane((int)d0_nan, 0) != 0;
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, (int)e);
                if (lane == 0) store_code(codes, (size_t)row * m + s, (blocked || bj == NONE) ? 0u : bj, k);
            }
        }
    }
}

// Re-check for long sub-vectors (the screens of sub_dim 48 .. 128 leave one list per subspace): k_recheck_wave pulls
// the whole sub-codebook through the L1 for every entry (128 KB at sub_dim 128, k = 256).  Here a workgroup of 16
// waves takes 16 entries at a time and stages the centroids through LDS in tiles of 64, once for all 16: lane l of
// every wave evaluates centroid (tile + l) for its wave's entry, the entry's row sits in LDS too (broadcast reads).
// Same arithmetic and merge rule as k_recheck_wave.
template <int METRIC, int SD>
__global__ __launch_bounds__(1024) void k_recheck_tiled(
    const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k, const float *__restrict__ cb,
    const float *__restrict__ cnsqrt, const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows,
    const uint32_t *__restrict__ wl_count, uint64_t wl_stride, uint8_t *__restrict__ codes) {
    constexpr uint32_t TCN = 64, PITCH = SD + 4;  // 16-byte rows; the pad spreads the lanes' rows over the banks
    __shared__ __attribute__((aligned(16))) float ct[TCN][PITCH];
    __shared__ __attribute__((aligned(16))) float xs[16][SD];
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t count = wl_count[s];
    const float *cbs = cb + (size_t)s * k * SD;
    const uint32_t NONE = 0xFFFFFFFFu;
    for (uint32_t e0 = blockIdx.x * 16; e0 < count; e0 += gridDim.x * 16) {  // uniform per workgroup
        const uint32_t e = e0 + wave;
        const bool valid = e < count;
        const uint32_t row = valid ? wl_rows[(size_t)s * wl_stride + e] : 0u;
        __syncthreads();  // the previous batch is done with xs and ct
        if (valid)
            for (uint32_t t = lane; t < SD; t += 64) xs[wave][t] = X[(size_t)row * d + (size_t)s * SD + t];
        float na = 0.0f;
        float bd = __builtin_inff();
        uint32_t bj = NONE;
        bool d0_nan = false;
        for (uint32_t j0 = 0; j0 < k; j0 += TCN) {
            __syncthreads();  // xs written / previous tile consumed
            for (uint32_t idx = threadIdx.x; idx < TCN * (SD / 4); idx += 1024) {
                const uint32_t r = idx / (SD / 4), q = idx % (SD / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + r < k) v = *reinterpret_cast<const float4 *>(cbs + (size_t)(j0 + r) * SD + 4 * q);
                *reinterpret_cast<float4 *>(&ct[r][4 * q]) = v;
            }
            __syncthreads();
            if (vq_is_cos(METRIC) && j0 == 0) {
                float sa = -0.0f;
                for (uint32_t t = 0; t < SD; ++t) {
                    const float p = xs[wave][t] * xs[wave][t];
                    sa = sa + p;
                }
                na = sqrtf(sa);
            }
            const uint32_t j = j0 + lane;
            float acc = (vq_is_cos(METRIC)) ? -0.0f : 0.0f;
#pragma unroll 4
            for (uint32_t t4 = 0; t4 < SD / 4; ++t4) {
                const float4 xv = *reinterpret_cast<const float4 *>(&xs[wave][4 * t4]);
                const float4 cv = *reinterpret_cast<const float4 *>(&ct[lane][4 * t4]);
                const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, ce[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
                        const float diff = xe[u] - ce[u];
                        const float sq = diff * diff;
                        acc = acc + sq;
                    } else if (METRIC == VQHIP_MANHATTAN) {
                        const float diff = xe[u] - ce[u];
                        acc = acc + fabsf(diff);
                    } else {
                        const float p = xe[u] * ce[u];
                        acc = acc + p;
                    }
                }
            }
            float dist = acc;
            if (METRIC == VQHIP_EUCLIDEAN) dist = sqrtf(acc);
            if (vq_is_cos(METRIC)) {
                const float nb = (j < k) ? cnsqrt[(size_t)s * k + j] : 1.0f;
                dist = vq_cosine_finish(METRIC, acc, na, nb);
            }
            const bool isnan_d = dist != dist;
            if (j == 0) d0_nan = isnan_d;
            const bool better = (j < k) & !isnan_d & ((bj == NONE) | (dist < bd));  // ascending j within the lane
            bd = better ? dist : bd;
            bj = better ? j : bj;
        }
        argmin_wave(bd, bj);
        const bool blocked = __builtin_amdgcn_readlane((int)d0_nan, 0) != 0;
        if (valid && lane == 0) store_code(codes, (size_t)row * m + s, (blocked || bj == NONE) ? 0u : bj, k);
    }
}

// k_recheck_tiled for a run-time sub_dim (a multiple of 4; the widths of the wide screen without a compile-time
// instantiation): the centroid tile is staged 64 dimensions at a time, the running distance of (entry, centroid)
// stays in the lane across the dimension chunks, the entries' rows sit in dynamic LDS.
template <int METRIC>
__global__ __launch_bounds__(1024) void k_recheck_tiled_any(
    const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const float *__restrict__ cb,
    const float *__restrict__ cnsqrt, const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows,
    const uint32_t *__restrict__ wl_count, uint64_t wl_stride, uint8_t *__restrict__ codes) {
    constexpr uint32_t TCN = 64, TD = 64, PITCH = TD + 4;
    __shared__ __attribute__((aligned(16))) float ct[TCN][PITCH];
    extern __shared__ __attribute__((aligned(16))) float xs_any[];  // [16][sd]
    const uint32_t s = sub_list ? sub_list[blockIdx.y] : blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t count = wl_count[s];
    const float *cbs = cb + (size_t)s * k * sd;
    float *xw = xs_any + (size_t)wave * sd;
    const uint32_t NONE = 0xFFFFFFFFu;
    for (uint32_t e0 = blockIdx.x * 16; e0 < count; e0 += gridDim.x * 16) {
        const uint32_t e = e0 + wave;
        const bool valid = e < count;
        const uint32_t row = valid ? wl_rows[(size_t)s * wl_stride + e] : 0u;
        __syncthreads();
        if (valid)
            for (uint32_t t = lane; t < sd; t += 64) xw[t] = X[(size_t)row * d + (size_t)s * sd + t];
        __syncthreads();
        float na = 0.0f;
        if (vq_is_cos(METRIC)) {
            float sa = -0.0f;
            for (uint32_t t = 0; t < sd; ++t) {
                const float p = xw[t] * xw[t];
                sa = sa + p;
            }
            na = sqrtf(sa);
        }
        float bd = __builtin_inff();
        uint32_t bj = NONE;
        bool d0_nan = false;
        for (uint32_t j0 = 0; j0 < k; j0 += TCN) {
            const uint32_t j = j0 + lane;
            float acc = (vq_is_cos(METRIC)) ? -0.0f : 0.0f;
            for (uint32_t t0 = 0; t0 < sd; t0 += TD) {
                const uint32_t td = min(TD, sd - t0);  // a multiple of 4
                __syncthreads();  // the previous chunk is consumed
                for (uint32_t idx = threadIdx.x; idx < TCN * (TD / 4); idx += 1024) {
                    const uint32_t r = idx / (TD / 4), q = idx % (TD / 4);
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j0 + r < k && 4 * q < td) v = *reinterpret_cast<const float4 *>(cbs + (size_t)(j0 + r) * sd + t0 + 4 * q);
                    *reinterpret_cast<float4 *>(&ct[r][4 * q]) = v;
                }
                __syncthreads();
                for (uint32_t t4 = 0; t4 < td / 4; ++t4) {
                    const float4 xv = *reinterpret_cast<const float4 *>(&xw[t0 + 4 * t4]);
                    const float4 cv = *reinterpret_cast<const float4 *>(&ct[lane][4 * t4]);
                    const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, ce[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
                            const float diff = xe[u] - ce[u];
                            const float sq = diff * diff;
                            acc = acc + sq;
                        } else if (METRIC == VQHIP_MANHATTAN) {
                            const float diff = xe[u] - ce[u];
                            acc = acc + fabsf(diff);
                        } else {
                            const float p = xe[u] * ce[u];
                            acc = acc + p;
                        }
                    }
                }
            }
            float dist = acc;
            if (METRIC == VQHIP_EUCLIDEAN) dist = sqrtf(acc);
            if (vq_is_cos(METRIC)) {
                const float nb = (j < k) ? cnsqrt[(size_t)s * k + j] : 1.0f;
                dist = vq_cosine_finish(METRIC, acc, na, nb);
            }
            const bool isnan_d = dist != dist;
            if (j == 0) d0_nan = isnan_d;
            const bool better = (j < k) & !isnan_d & ((bj == NONE) | (dist < bd));
            bd = better ? dist : bd;
            bj = better ? j : bj;
        }
        argmin_wave(bd, bj);
        const bool blocked = __builtin_amdgcn_readlane((int)d0_nan, 0) != 0;
        if (valid && lane == 0) store_code(codes, (size_t)row * m + s, (blocked || bj == NONE) ? 0u : bj, k);
    }
}

// One workgroup per subspace: squared norms, the screen's A-operand image, flags.
__global__ __launch_bounds__(256) void k_prepare_codebook(const float *__restrict__ cb, uint32_t m,
                                                          uint32_t k, uint32_t sd, uint32_t nt,
                                                          uint32_t ks, float *__restrict__ prepA,
                                                          float *__restrict__ prepCn,
                                                          float *__restrict__ meta,
                                                          float *__restrict__ cnsqrt) {
    __shared__ float s_max[256];
    __shared__ int s_bad[256];
    const uint32_t s = blockIdx.x;
    const float *cbs = cb + (size_t)s * k * sd;
    float lmax = 0.0f;
    int bad = 0;
    const uint32_t kpad = nt * 16;
    for (uint32_t j = threadIdx.x; j < (kpad > k ? kpad : k); j += blockDim.x) {
        if (j < k) {
            float acc = -0.0f;
            for (uint32_t t = 0; t < sd; ++t) {
                float v = cbs[(size_t)j * sd + t];
                float p = v * v;
                acc = acc + p;
                if (!(fabsf(v) <= 3.0e38f)) bad = 1;  // NaN or Inf
            }
            if (!(acc <= 3.0e38f)) bad = 1;
            if (cnsqrt) cnsqrt[(size_t)s * k + j] = sqrtf(acc);
            if (prepCn) prepCn[(size_t)s * kpad + j] = acc + 0.0f;
            lmax = fmaxf(lmax, acc);
        } else if (prepCn) {
            prepCn[(size_t)s * kpad + j] = __builtin_inff();  // padding never wins
        }
    }
    s_max[threadIdx.x] = lmax;
    s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + off]);
            s_bad[threadIdx.x] |= s_bad[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && meta) {
        // margin coefficient: see DESIGN.md "screen soundness".  (8*sd+16)*2^-24 covers
        // (6*sd+8)(1+eps) from the error analysis with ~1.4x slack; a non-finite codebook
        // gets +inf so that every row of the subspace goes to the exact re-check.
        const float u = 5.9604644775390625e-08f;  // 2^-24
        float coef = (8.0f * (float)sd + 16.0f) * u;
        meta[s * 4 + 0] = sqrtf(s_max[0]) * 1.0000005f + 1e-30f;
        meta[s * 4 + 1] = s_bad[0] ? __builtin_inff() : coef;
        // bf16-split screen (k_screen_bf16.hip): + 2 * kBf16AssumedUlps per MFMA (both ends of the
        // gap; the constant is checked on the device by k_selftest.hip) + 16 for the dropped cross terms
        float coef16 = __builtin_inff();
        if (sd % 4 == 0 && sd >= 4 && sd <= 32) {
            const uint32_t dpg = sd / 4, pp = 8 / dpg, nm = (6 + pp - 1) / pp;
            coef16 = (8.0f * (float)sd + 16.0f + 2.0f * kBf16AssumedUlps * (float)nm + 16.0f) * u;
        } else {
            // no 16x16 variant for this sub_dim; the X32 kernels (also the zero-padded ones) read [3] only as
            // "codebook finite?" (cosine)
            const uint32_t nm32 = (6 * ((sd + 1) / 2) + 7) / 8;
            coef16 = (8.0f * (float)sd + 16.0f + 2.0f * kBf16AssumedUlps * (float)nm32 + 16.0f) * u;
        }
        meta[s * 4 + 2] = s_bad[0] ? __builtin_inff() : coef16;
        // X32 variant packs a 6-bit index into the low mantissa bits: |perturbation| < 2^-18 |s|
        meta[s * 4 + 3] = s_bad[0] ? __builtin_inff() : coef16 + 128.0f * u;
    }
    if (prepA) {
        // A operand of v_mfma_f32_16x16x4_f32: lane l holds A[row = l&15][k = l>>4].  Tile i,
        // k-step q, lane l  <-  -2 * C[16i + (l&15)][ks*(l>>4) + q]  (k order permuted the same
        // way as the B operand the screen kernel loads; any order is a valid dot product).
        const uint32_t total = nt * ks * 64;
        for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
            uint32_t lane = e & 63, q = (e >> 6) % ks, i = (e >> 6) / ks;
            uint32_t j = 16 * i + (lane & 15);
            uint32_t t = ks * (lane >> 4) + q;
            float v = (j < k) ? -2.0f * cbs[(size_t)j * sd + t] : 0.0f;
            prepA[(size_t)s * total + e] = v;
        }
    }
}

// Distance::compute for one pair, runtime length (src/core/distance.rs:48-64)
__device__ float exact_distance_rt(int metric, const float *__restrict__ a,
                                   const float *__restrict__ b, uint32_t n) {
    if (metric == VQHIP_SQUARED_EUCLIDEAN || metric == VQHIP_EUCLIDEAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            float diff = a[t] - b[t];
            float sq = diff * diff;
            acc = acc + sq;
        }
        return metric == VQHIP_EUCLIDEAN ? sqrtf(acc) : acc;
    }
    if (metric == VQHIP_MANHATTAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            float diff = a[t] - b[t];
            acc = acc + fabsf(diff);
        }
        return acc;
    }
    float dot = -0.0f, sa = -0.0f, sb = -0.0f;
    for (uint32_t t = 0; t < n; ++t) {
        float p = a[t] * b[t];
        dot = dot + p;
    }
    for (uint32_t t = 0; t < n; ++t) {
        float p = a[t] * a[t];
        sa = sa + p;
    }
    for (uint32_t t = 0; t < n; ++t) {
        float p = b[t] * b[t];
        sb = sb + p;
    }
    const float na = sqrtf(sa), nb = sqrtf(sb);
    return vq_cosine_finish(metric, dot, na, nb);
}

__global__ __launch_bounds__(256) void k_distance_batch(int metric, const float *__restrict__ a,
                                                        const float *__restrict__ b, uint64_t n,
                                                        uint32_t d, float *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        out[i] = exact_distance_rt(metric, a + i * d, b + i * d, d);
}

template <int METRIC>
int dispatch_exact(const CodebookView &cb, const AssignArgs &a, bool wl, dim3 grid,
                   hipStream_t stream) {
    const uint32_t *wlr = wl ? a.wl_rows : nullptr;
    const uint32_t *wlc = wl ? a.wl_count : nullptr;
    if (wl) {
        const bool seg = (a.wl_seg != nullptr && a.n_seg > 0);
        const dim3 wgrid(seg ? a.n_seg : (uint32_t)num_cus() * 4, a.n_sub);
        const uint32_t *wls = seg ? a.wl_seg : nullptr;
        if constexpr (METRIC != VQHIP_MANHATTAN)
        if (cb.k <= 256 && ((seg && (cb.sd == 8 || cb.sd == 12 || cb.sd == 16 || cb.sd == 24)) || (!seg && cb.sd == 32))) {
            // register-resident sub-codebook; enough (segment, part) units for ~2 waves per SIMD.  The grouped
            // sub_dim-32 screen and the fp32 screen leave one list per subspace: n_seg = 1, many parts.
            const uint32_t kpl = cb.k <= 64 ? 1u : cb.k <= 128 ? 2u : 4u;
            const uint32_t target = (uint32_t)num_cus() * 16;
            const uint32_t nseg = seg ? a.n_seg : 1u;
            uint32_t parts = target / std::max(1u, nseg * a.n_sub);
            parts = std::max(parts, 1u);
            if (seg) parts = std::min(parts, 4u);
            const dim3 rgrid((nseg * parts + 3) / 4, a.n_sub);
#define VQ_RESIDENT(SDV, KPLV)                                                                                      \
    if (cb.sd == SDV && kpl == KPLV) {                                                                              \
        hipLaunchKernelGGL((k_recheck_resident<METRIC, SDV, KPLV>), rgrid, dim3(256), 0, stream, a.X, a.d, cb.m, cb.k, \
                           cb.cb, cb.cnsqrt, a.sub_list, wlr, a.wl_stride, wls, wlc, nseg, parts, a.codes);        \
        VQ_LAUNCH_CHECK("k_recheck_resident");                                                                      \
        return VQHIP_OK;                                                                                            \
    }
            VQ_RESIDENT(8, 1) VQ_RESIDENT(8, 2) VQ_RESIDENT(8, 4)
            VQ_RESIDENT(12, 1) VQ_RESIDENT(12, 2) VQ_RESIDENT(12, 4)
            VQ_RESIDENT(16, 1) VQ_RESIDENT(16, 2) VQ_RESIDENT(16, 4)
            VQ_RESIDENT(24, 1) VQ_RESIDENT(24, 2) VQ_RESIDENT(24, 4)
            VQ_RESIDENT(32, 1) VQ_RESIDENT(32, 2) VQ_RESIDENT(32, 4)
#undef VQ_RESIDENT
        }
        if (!seg && (cb.sd == 48 || cb.sd == 64 || (cb.sd >= 72 && cb.sd <= 192))) {
            const dim3 tgrid((uint32_t)num_cus(), a.n_sub);
#define VQ_RECHECK_TILED(SDV)                                                                                        \
    if (cb.sd == SDV) {                                                                                              \
        hipLaunchKernelGGL((k_recheck_tiled<METRIC, SDV>), tgrid, dim3(1024), 0, stream, a.X, a.d, cb.m, cb.k, cb.cb, \
                           cb.cnsqrt, a.sub_list, wlr, wlc, a.wl_stride, a.codes);                                    \
        VQ_LAUNCH_CHECK("k_recheck_tiled");                                                                          \
        return VQHIP_OK;                                                                                             \
    }
            VQ_RECHECK_TILED(48) VQ_RECHECK_TILED(64) VQ_RECHECK_TILED(72) VQ_RECHECK_TILED(80) VQ_RECHECK_TILED(96)
            VQ_RECHECK_TILED(100) VQ_RECHECK_TILED(112) VQ_RECHECK_TILED(120) VQ_RECHECK_TILED(128)
            VQ_RECHECK_TILED(160) VQ_RECHECK_TILED(192)
#undef VQ_RECHECK_TILED
            // any other multiple of 4 in the wide screen's range
            hipLaunchKernelGGL((k_recheck_tiled_any<METRIC>), tgrid, dim3(1024), (size_t)16 * cb.sd * 4, stream, a.X, a.d, cb.m,
                               cb.k, cb.sd, cb.cb, cb.cnsqrt, a.sub_list, wlr, wlc, a.wl_stride, a.codes);
            VQ_LAUNCH_CHECK("k_recheck_tiled_any");
            return VQHIP_OK;
        }
#define VQ_RECHECK_CASE(SDV)                                                                   \
    case SDV:                                                                                  \
        hipLaunchKernelGGL((k_recheck_wave<METRIC, SDV>), wgrid, dim3(seg ? 1024 : 256), 0, stream, a.X, a.d, \
                           cb.m, cb.k, cb.cb, cb.cnsqrt, a.sub_list, wlr, wlc, a.wl_stride,     \
                           wls, a.n_seg, a.codes);                                                           \
        VQ_LAUNCH_CHECK("k_recheck_wave");                                                     \
        return VQHIP_OK;
        switch (cb.sd) {
            VQ_RECHECK_CASE(4)
            VQ_RECHECK_CASE(5)
            VQ_RECHECK_CASE(6)
            VQ_RECHECK_CASE(7)
            VQ_RECHECK_CASE(8)
            VQ_RECHECK_CASE(9)
            VQ_RECHECK_CASE(10)
            VQ_RECHECK_CASE(11)
            VQ_RECHECK_CASE(12)
            VQ_RECHECK_CASE(13)
            VQ_RECHECK_CASE(14)
            VQ_RECHECK_CASE(15)
            VQ_RECHECK_CASE(16)
            VQ_RECHECK_CASE(17)
            VQ_RECHECK_CASE(18)
            VQ_RECHECK_CASE(19)
            VQ_RECHECK_CASE(20)
            VQ_RECHECK_CASE(21)
            VQ_RECHECK_CASE(22)
            VQ_RECHECK_CASE(23)
            VQ_RECHECK_CASE(24)
            VQ_RECHECK_CASE(25)
            VQ_RECHECK_CASE(26)
            VQ_RECHECK_CASE(27)
            VQ_RECHECK_CASE(28)
            VQ_RECHECK_CASE(29)
            VQ_RECHECK_CASE(30)
            VQ_RECHECK_CASE(31)
            VQ_RECHECK_CASE(32)
            VQ_RECHECK_CASE(33)
            VQ_RECHECK_CASE(34)
            VQ_RECHECK_CASE(35)
            VQ_RECHECK_CASE(36)
            VQ_RECHECK_CASE(37)
            VQ_RECHECK_CASE(38)
            VQ_RECHECK_CASE(39)
            VQ_RECHECK_CASE(40)
            VQ_RECHECK_CASE(41)
            VQ_RECHECK_CASE(42)
            VQ_RECHECK_CASE(43)
            VQ_RECHECK_CASE(44)
            VQ_RECHECK_CASE(45)
            VQ_RECHECK_CASE(46)
            VQ_RECHECK_CASE(47)
            VQ_RECHECK_CASE(48)
            VQ_RECHECK_CASE(49)
            VQ_RECHECK_CASE(50)
            VQ_RECHECK_CASE(51)
            VQ_RECHECK_CASE(52)
            VQ_RECHECK_CASE(53)
            VQ_RECHECK_CASE(54)
            VQ_RECHECK_CASE(55)
            VQ_RECHECK_CASE(56)
            VQ_RECHECK_CASE(57)
            VQ_RECHECK_CASE(58)
            VQ_RECHECK_CASE(59)
            VQ_RECHECK_CASE(60)
            VQ_RECHECK_CASE(61)
            VQ_RECHECK_CASE(62)
            VQ_RECHECK_CASE(63)
            VQ_RECHECK_CASE(64)
            VQ_RECHECK_CASE(96)
            VQ_RECHECK_CASE(128)
        default: break;
        }
#undef VQ_RECHECK_CASE
    }
#define VQ_EXACT_CASE(SDV)                                                                    \
    case SDV:                                                                                  \
        hipLaunchKernelGGL((k_assign_exact<METRIC, SDV, false>), grid, dim3(kExactBlock), 0,   \
                           stream, a.X, a.n, a.d, cb.m, cb.k, cb.sd, cb.cb, cb.cnsqrt,         \
                           a.sub_list, wlr, wlc, a.wl_stride, a.codes);                        \
        break;
    switch (cb.sd) {
        VQ_EXACT_CASE(1)
        VQ_EXACT_CASE(2)
        VQ_EXACT_CASE(3)
        VQ_EXACT_CASE(4)
        VQ_EXACT_CASE(5)
        VQ_EXACT_CASE(6)
        VQ_EXACT_CASE(7)
        VQ_EXACT_CASE(8)
        VQ_EXACT_CASE(9)
        VQ_EXACT_CASE(10)
        VQ_EXACT_CASE(11)
        VQ_EXACT_CASE(12)
        VQ_EXACT_CASE(13)
        VQ_EXACT_CASE(14)
        VQ_EXACT_CASE(15)
        VQ_EXACT_CASE(16)
        VQ_EXACT_CASE(17)
        VQ_EXACT_CASE(18)
        VQ_EXACT_CASE(19)
        VQ_EXACT_CASE(20)
        VQ_EXACT_CASE(21)
        VQ_EXACT_CASE(22)
        VQ_EXACT_CASE(23)
        VQ_EXACT_CASE(24)
        VQ_EXACT_CASE(25)
        VQ_EXACT_CASE(26)
        VQ_EXACT_CASE(27)
        VQ_EXACT_CASE(28)
        VQ_EXACT_CASE(29)
        VQ_EXACT_CASE(30)
        VQ_EXACT_CASE(31)
        VQ_EXACT_CASE(32)
        VQ_EXACT_CASE(40)
        VQ_EXACT_CASE(48)
        VQ_EXACT_CASE(64)
        VQ_EXACT_CASE(96)
        VQ_EXACT_CASE(128)
    default:
        if (!wl) {  // any other sub_dim: rows staged through LDS (k_assign_exact_tiled)
            const dim3 tgrid((uint32_t)((a.n + kTiledRows - 1) / kTiledRows), a.n_sub);
            hipLaunchKernelGGL((k_assign_exact_tiled<METRIC>), tgrid, dim3(256), 0, stream, a.X, a.n, a.d, cb.m, cb.k,
                               cb.sd, cb.cb, cb.cnsqrt, a.sub_list, a.codes);
            VQ_LAUNCH_CHECK("k_assign_exact_tiled");
            return VQHIP_OK;
        }
        hipLaunchKernelGGL((k_assign_exact<METRIC, 1, true>), grid, dim3(kExactBlock), 0, stream,
                           a.X, a.n, a.d, cb.m, cb.k, cb.sd, cb.cb, cb.cnsqrt, a.sub_list, wlr, wlc,
                           a.wl_stride, a.codes);
    }
#undef VQ_EXACT_CASE
    VQ_LAUNCH_CHECK("k_assign_exact");
    return VQHIP_OK;
}

}  // namespace

int launch_prepare_codebook(const CodebookView &v, float *prepA, float *prepCn, float *meta,
                            float *cnsqrt, hipStream_t stream) {
    if (v.m == 0) return VQHIP_OK;
    hipLaunchKernelGGL(k_prepare_codebook, dim3(v.m), dim3(256), 0, stream, v.cb, v.m, v.k, v.sd,
                       v.nt, v.ks, prepA, prepCn, meta, cnsqrt);
    VQ_LAUNCH_CHECK("k_prepare_codebook");
    return VQHIP_OK;
}

int launch_distance_batch(int metric, const float *a, const float *b, uint64_t n, uint32_t d,
                          float *out, hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > (uint64_t)num_cus() * 8) blocks = (uint64_t)num_cus() * 8;
    hipLaunchKernelGGL(k_distance_batch, dim3((uint32_t)blocks), dim3(256), 0, stream, metric, a, b,
                       n, d, out);
    VQ_LAUNCH_CHECK("k_distance_batch");
    return VQHIP_OK;
}

int launch_assign_exact(const CodebookView &cb, const AssignArgs &a, bool use_worklist,
                        hipStream_t stream) {
    if (a.n == 0 || a.n_sub == 0) return VQHIP_OK;
    if (vq_is_cos(a.metric) && !cb.cnsqrt)
        return fail(VQHIP_ERR_FAILURE, "cosine assignment needs prepared centroid norms");
    // work-list launches do not know their size on the host: a fixed grid strides over it
    uint64_t items = use_worklist ? (uint64_t)num_cus() * 4 * kExactBlock : a.n;
    uint32_t gx = (uint32_t)((items + kExactBlock - 1) / kExactBlock);
    uint32_t cap = (uint32_t)num_cus() * 16;
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    dim3 grid(gx, a.n_sub);
    switch (a.metric) {
    case VQHIP_SQUARED_EUCLIDEAN:
        return dispatch_exact<VQHIP_SQUARED_EUCLIDEAN>(cb, a, use_worklist, grid, stream);
    case VQHIP_EUCLIDEAN: return dispatch_exact<VQHIP_EUCLIDEAN>(cb, a, use_worklist, grid, stream);
    case VQHIP_MANHATTAN: return dispatch_exact<VQHIP_MANHATTAN>(cb, a, use_worklist, grid, stream);
    case VQHIP_COSINE: return dispatch_exact<VQHIP_COSINE>(cb, a, use_worklist, grid, stream);
    case VQHIP_COSINE_UNCLAMPED: return dispatch_exact<VQHIP_COSINE_UNCLAMPED>(cb, a, use_worklist, grid, stream);
    default: return fail(VQHIP_ERR_INVALID_INPUT, "unknown metric %d", a.metric);
    }
}

}  // namespace vqhip
