// Asymmetric distance search over stored PQ codes ("code-based storage + ADC").
// The reference has no counterpart (it keeps f16 reconstructions, src/pq.rs:165-199); the semantics
// are the ones include/vqhip.h states (the tests hold a CPU statement of the same definition):
//   t_s(q, j)  = the reference's per-subspace distance between the query's sub-vector and centroid j
//                (squared L2: Vector::distance2, src/core/vector.rs:135-143; L1: distance.rs:85-95),
//   D(q, i)    = t_0(q, code[i][0]) + t_1(...) + ... in subspace order, f32,
//   result     = the topk rows by (D, row index) ascending; Euclidean reports sqrt(D).
// Two schedules, one result (launch_adc_search_fast / launch_adc_search; the caller picks, vqhip_pq_adc_search_device):
//   * one scan (round 6; n >= 32768, topk <= 256): tables interleaved over a batch of 8 queries (k_adc_lut_i), a threshold
//     per query from a sample of the rows (k_adc_thresh), ONE pass over the codes that keeps the rows at or below it
//     (k_adc_scan_thr: LDS-bound on the table reads), the exact top-k of those (k_adc_sort_thr); a query whose
//     threshold let fewer than topk or more than 8192 rows pass is flagged and repeated by
//   * the full pass: the table t (k_adc_lut), a byte-gather scan of the codes with the tables of 8 queries in LDS that
//     writes every D(q, i) (k_adc_scan), a histogram cut + candidate sort (k_adc_pick_bin / _collect / _sort_out) and an
//     exact per-query radix select where the cut is too dense (k_adc_topk).
#include "common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

__device__ __forceinline__ uint32_t adc_key(float f) {  // order-preserving; NaN sorts last
    const uint32_t b = __float_as_uint(f);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float adc_unkey(uint32_t k) {
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// lut[q][s][j]
// also bounds[q] = {sum_s min_j t_s, sum_s max_j t_s}: the range every D(q, .) lies in (used only to
// place the histogram bins of the candidate filter, so the float atomics' order does not matter)
__global__ __launch_bounds__(256) void k_adc_lut(const float *__restrict__ queries, uint32_t nq, uint32_t m,
                                                 uint32_t k, uint32_t sd, const float *__restrict__ cb, int l1,
                                                 float *__restrict__ lut, float *__restrict__ bounds) {
    __shared__ float s_lo[256], s_hi[256];
    const uint32_t q = blockIdx.x, s = blockIdx.y;
    const float *x = queries + ((size_t)q * m + s) * sd;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (uint32_t j = threadIdx.x; j < k; j += 256) {
        const float *c = cb + ((size_t)s * k + j) * sd;
        float acc = l1 ? 0.0f : -0.0f;
        for (uint32_t t = 0; t < sd; ++t) {
            const float diff = x[t] - c[t];
            if (l1) {
                acc = acc + fabsf(diff);
            } else {
                const float sq = diff * diff;
                acc = acc + sq;
            }
        }
        lut[((size_t)q * m + s) * k + j] = acc;
        lo = fminf(lo, acc);
        hi = fmaxf(hi, acc);
    }
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + off]);
            s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&bounds[2 * q + 0], s_lo[0]);
        atomicAdd(&bounds[2 * q + 1], s_hi[0]);
    }
    (void)nq;
}

constexpr uint32_t kAdcQB = 8;      // queries per scan pass (their tables share the LDS)
constexpr uint32_t kAdcBins = 512;  // linear bins over [bounds lo, hi] for the candidate filter
constexpr uint32_t kAdcCand = 8192; // candidates the fast top-k path sorts in LDS

// monotone (non-decreasing in d) bin of a distance; NaN and out-of-range values go to the last bin
__device__ __forceinline__ uint32_t adc_bin(float dval, float lo, float scale) {
    const float t = (dval - lo) * scale;
    return (t >= 0.0f && t < (float)(kAdcBins - 1)) ? (uint32_t)t : ((t < 0.0f) ? 0u : kAdcBins - 1);
}

// dist[qq][i] for the queries q0 .. q0+nqb-1
// blockIdx.y = batch of kAdcQB queries inside a group (round 6: the batches of a group run in ONE set of launches; the
// per-query arrays -- lut, bounds, dist, hist -- are laid out over the group's queries, batch b owns queries 8b .. 8b+7)
__global__ __launch_bounds__(256) void k_adc_scan(const uint8_t *__restrict__ codes, uint64_t n, uint32_t m,
                                                  uint32_t k, const float *__restrict__ lut, uint32_t nq_group, uint32_t qb,
                                                  const float *__restrict__ bounds, float *__restrict__ dist,
                                                  uint32_t *__restrict__ hist) {
    extern __shared__ float lds_lut[];  // [nqb][m][k], then the block's histograms [nqb][kAdcBins]
    const uint32_t tab = m * k;
    const uint32_t q_first = blockIdx.y * qb;
    const uint32_t nqb = min(qb, nq_group - q_first);
    lut += (size_t)q_first * tab;
    bounds += 2 * q_first;
    dist += (size_t)q_first * n;
    hist += (size_t)q_first * kAdcBins;
    uint32_t *lds_hist = reinterpret_cast<uint32_t *>(lds_lut + (size_t)nqb * tab);
    for (uint32_t e = threadIdx.x; e < nqb * tab; e += 256) lds_lut[e] = lut[e];
    for (uint32_t e = threadIdx.x; e < nqb * kAdcBins; e += 256) lds_hist[e] = 0u;
    float blo[kAdcQB], bsc[kAdcQB];
#pragma unroll
    for (uint32_t qq = 0; qq < kAdcQB; ++qq) {
        const float lo = (qq < nqb) ? bounds[2 * qq] : 0.0f, hi = (qq < nqb) ? bounds[2 * qq + 1] : 1.0f;
        blo[qq] = lo;
        bsc[qq] = (hi > lo) ? (float)kAdcBins / (hi - lo) : 0.0f;
    }
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        float acc[kAdcQB];
#pragma unroll
        for (uint32_t qq = 0; qq < kAdcQB; ++qq) acc[qq] = 0.0f;
        if (k <= 256 && (m & 7u) == 0 && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0) {
            // one-byte codes, rows of whole 8-byte words: a row's codes in m / 8 loads instead of m byte loads (same order of
            // the additions: subspace 0 first)
            for (uint32_t s8 = 0; s8 < m; s8 += 8) {
                const uint2 w = *reinterpret_cast<const uint2 *>(codes + i * m + s8);
#pragma unroll
                for (uint32_t b = 0; b < 8; ++b) {
                    const uint32_t s = s8 + b;
                    const uint32_t off = s * k + (((b < 4 ? w.x : w.y) >> (8 * (b & 3))) & 255u);
#pragma unroll
                    for (uint32_t qq = 0; qq < kAdcQB; ++qq)
                        if (qq < nqb) acc[qq] = (s == 0) ? lds_lut[qq * tab + off] : acc[qq] + lds_lut[qq * tab + off];
                }
            }
        } else {
            for (uint32_t s = 0; s < m; ++s) {
                const uint32_t off = s * k + load_code(codes, i * m + s, k);  // one byte per code, two above 256 centroids
#pragma unroll
                for (uint32_t qq = 0; qq < kAdcQB; ++qq)
                    if (qq < nqb) acc[qq] = (s == 0) ? lds_lut[qq * tab + off] : acc[qq] + lds_lut[qq * tab + off];
            }
        }
#pragma unroll
        for (uint32_t qq = 0; qq < kAdcQB; ++qq)
            if (qq < nqb) {
                dist[(size_t)qq * n + i] = acc[qq];
                atomicAdd(&lds_hist[qq * kAdcBins + adc_bin(acc[qq], blo[qq], bsc[qq])], 1u);
            }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nqb * kAdcBins; e += 256)
        if (lds_hist[e]) atomicAdd(&hist[e], lds_hist[e]);
}

// fast top-k, step 1: the bin that holds the k-th smallest value; sel[q] = {bin, candidates up to it}
__global__ __launch_bounds__(64) void k_adc_pick_bin(const uint32_t *__restrict__ hist, uint32_t topk,
                                                     uint32_t *__restrict__ sel) {
    const uint32_t q = blockIdx.x;
    if (threadIdx.x != 0) return;
    uint32_t cum = 0, b = 0;
    for (; b < kAdcBins; ++b) {
        cum += hist[q * kAdcBins + b];
        if (cum >= topk) break;
    }
    sel[2 * q + 0] = b;
    sel[2 * q + 1] = cum;
}

// step 2: every row whose bin is <= the selected one becomes a candidate (key, row); any order
__global__ __launch_bounds__(256) void k_adc_collect(const float *__restrict__ dist, uint64_t n,
                                                     const float *__restrict__ bounds, const uint32_t *__restrict__ sel,
                                                     unsigned long long *__restrict__ cand, uint32_t *__restrict__ cand_n) {
    const uint32_t q = blockIdx.y;
    if (sel[2 * q + 1] > kAdcCand) return;  // too dense: the exact radix select handles this query
    const float lo = bounds[2 * q], hi = bounds[2 * q + 1];
    const float scale = (hi > lo) ? (float)kAdcBins / (hi - lo) : 0.0f;
    const uint32_t bmax = sel[2 * q];
    const float *dq = dist + (size_t)q * n;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float dv = dq[i];
        if (adc_bin(dv, lo, scale) <= bmax) {
            const uint32_t pos = atomicAdd(&cand_n[q], 1u);
            if (pos < kAdcCand) cand[(size_t)q * kAdcCand + pos] = ((unsigned long long)adc_key(dv) << 32) | (uint32_t)i;
        }
    }
}

// step 3: sort the candidates by (key, row) in LDS, emit the first topk
__global__ __launch_bounds__(1024) void k_adc_sort_out(const unsigned long long *__restrict__ cand,
                                                       const uint32_t *__restrict__ sel, uint32_t topk, int take_sqrt,
                                                       uint32_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    extern __shared__ unsigned long long sort_buf[];  // [kAdcCand]
    const uint32_t q = blockIdx.x, cnt = sel[2 * q + 1];
    if (cnt > kAdcCand) return;  // handled by k_adc_topk
    uint32_t len = 1024;
    while (len < cnt) len <<= 1;
    for (uint32_t e = threadIdx.x; e < len; e += 1024) sort_buf[e] = (e < cnt) ? cand[(size_t)q * kAdcCand + e] : ~0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= len; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < len; t += 1024) {
                const uint32_t partner = t ^ stride;
                if (partner > t) {
                    const bool up = (t & size) == 0;
                    const unsigned long long a = sort_buf[t], b = sort_buf[partner];
                    if ((a > b) == up) {
                        sort_buf[t] = b;
                        sort_buf[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x < topk) {
        const unsigned long long w = sort_buf[threadIdx.x];
        float dv = adc_unkey((uint32_t)(w >> 32));
        if (take_sqrt) dv = sqrtf(dv);
        idx_out[(size_t)q * topk + threadIdx.x] = (uint32_t)w;
        dist_out[(size_t)q * topk + threadIdx.x] = dv;
    }
}

// exact top-k of one query's distances: radix select of the k-th key, ordered collection (ties by
// row index), bitonic sort of the <= 1024 winners by (key, index)
__global__ __launch_bounds__(1024) void k_adc_topk(const float *__restrict__ dist, uint64_t n, uint32_t topk, int take_sqrt,
                                                   const uint32_t *__restrict__ sel, uint32_t *__restrict__ idx_out,
                                                   float *__restrict__ dist_out) {
    __shared__ uint32_t hist[256];
    if (sel && sel[2 * blockIdx.x + 1] <= kAdcCand) return;  // the candidate path produced this query's result
    __shared__ uint32_t s_prefix, s_rank, s_count;
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long win[1024];
    const float *dq = dist + (size_t)blockIdx.x * n;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_prefix = 0;
        s_rank = topk - 1;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint64_t i = tid; i < n; i += 1024) {
            const uint32_t key = adc_key(dq[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rank = s_rank, b = 0;
            for (; b < 255; ++b) {
                if (rank < hist[b]) break;
                rank -= hist[b];
            }
            s_rank = rank;
            s_prefix = prefix | (b << shift);
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;         // the k-th smallest key
    const uint32_t need_eq = s_rank + 1;  // how many keys == T belong to the result (lowest row indices)
    if (tid == 0) s_count = 0;
    __syncthreads();
    // ordered collection: chunks of 1024 rows, block prefix sums keep row order
    uint32_t eq_taken = 0;  // replicated in every thread (uniform updates)
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + tid;
        uint32_t key = 0xFFFFFFFFu;
        bool less = false, eq = false;
        if (i < n) {
            key = adc_key(dq[i]);
            less = key < T;
            eq = key == T;
        }
        // ranks among this chunk's `eq` rows and among its selected rows (wave scan + wave sums)
        const uint64_t eqm = __ballot(eq);
        const uint32_t lane = tid & 63, wv = tid >> 6;
        const uint32_t eq_before_w = __popcll(eqm & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(eqm);
        __syncthreads();
        uint32_t eq_before = eq_before_w, eq_total = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wv) eq_before += wsum[w];
            eq_total += wsum[w];
        }
        __syncthreads();
        const bool take = less || (eq && (eq_taken + eq_before < need_eq));
        const uint64_t tm = __ballot(take);
        const uint32_t t_before_w = __popcll(tm & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(tm);
        __syncthreads();
        uint32_t t_before = t_before_w, t_total = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wv) t_before += wsum[w];
            t_total += wsum[w];
        }
        const uint32_t pos = s_count + t_before;
        if (take && pos < 1024) win[pos] = ((unsigned long long)key << 32) | (uint32_t)i;
        __syncthreads();
        if (tid == 0) s_count += t_total;
        {
            const uint32_t remaining = need_eq - eq_taken;  // eq_taken <= need_eq always
            eq_taken += eq_total < remaining ? eq_total : remaining;
        }
        __syncthreads();
        if (s_count >= topk) break;  // uniform
    }
    const uint32_t got = s_count < topk ? s_count : topk;
    for (uint32_t e = tid; e < 1024; e += 1024)
        if (e >= got) win[e] = ~0ull;
    __syncthreads();
    // bitonic sort of 1024 (key, index) pairs
    for (uint32_t size = 2; size <= 1024; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t partner = tid ^ stride;
            if (partner > tid) {
                const bool up = (tid & size) == 0;
                const unsigned long long a = win[tid], b = win[partner];
                if ((a > b) == up) {
                    win[tid] = b;
                    win[partner] = a;
                }
            }
            __syncthreads();
        }
    }
    if (tid < topk) {
        const unsigned long long w = win[tid];
        const bool valid = tid < got;
        float dv = adc_unkey((uint32_t)(w >> 32));
        if (take_sqrt) dv = sqrtf(dv);
        idx_out[(size_t)blockIdx.x * topk + tid] = valid ? (uint32_t)w : 0xFFFFFFFFu;
        dist_out[(size_t)blockIdx.x * topk + tid] = valid ? dv : __uint_as_float(0x7FC00000u);
    }
}


// ---- round 6: one pass over the codes, nothing but candidates written ------------------------------------------------
// The pass above writes every D(q, i) (4 n bytes per query) and reads it again to collect the candidates: 512 MB of
// traffic for 64 queries over 1M rows whose codes are 8 MB, plus eight LDS atomics per row and batch for the histograms.
// Here a THRESHOLD per query comes first, from a sample of the rows (k_adc_thresh), and the scan keeps only the rows at or
// below it: (key, row) pairs appended to the query's candidate list.  Any threshold gives the exact answer as long as at
// least `topk` and at most kAdcCand rows pass (every row below the topk-th smallest D, and every tie with it, is <= T);
// k_adc_sort_thr flags the queries where that fails and the caller sends those through the pass above.
// Tables are interleaved over the batch's queries, [s][j][QB]: one LDS address per (row, subspace) yields all QB terms
// (QB / 4 ds_read_b128 instead of QB ds_read_b32 and their offsets).
__host__ __device__ __forceinline__ uint32_t adc_tabp(uint32_t m, uint32_t k, uint32_t qb) { return (m * k * qb + 3u) & ~3u; }  // floats per batch of tables (16-byte units)
template <uint32_t QB>
__global__ __launch_bounds__(256) void k_adc_lut_i(const float *__restrict__ queries, uint32_t nq, uint32_t m, uint32_t k,
                                                   uint32_t sd, const float *__restrict__ cb, int l1, float *__restrict__ lut) {
    const uint32_t q = blockIdx.x, s = blockIdx.y, batch = q / QB, qq = q % QB;
    const bool real = q < nq;  // the last batch is padded with zero tables
    // (the query's sub-vector through LDS: `queries` may be pinned host memory, read once per workgroup then)
    __shared__ float xs[256];
    const float *xg = queries + ((size_t)(real ? q : 0) * m + s) * sd;
    for (uint32_t t = threadIdx.x; t < min(sd, 256u); t += 256) xs[t] = xg[t];
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < k; j += 256) {
        const float *c = cb + ((size_t)s * k + j) * sd;
        float acc = l1 ? 0.0f : -0.0f;
        for (uint32_t t = 0; t < sd; ++t) {
            const float diff = (t < 256u ? xs[t] : xg[t]) - c[t];
            if (l1) {
                acc = acc + fabsf(diff);
            } else {
                const float sq = diff * diff;
                acc = acc + sq;
            }
        }
        lut[(size_t)batch * adc_tabp(m, k, QB) + ((size_t)s * k + j) * QB + qq] = real ? acc : 0.0f;
    }
}

// QL consecutive table entries (QL queries' terms of one (subspace, code)) from float offset `at`
template <uint32_t QL>
__device__ __forceinline__ void adc_terms(const float *__restrict__ lds, uint32_t at, float (&v)[QL]) {
    if constexpr (QL == 1) {
        v[0] = lds[at];
    } else if constexpr (QL == 2) {
        const float2 a = *reinterpret_cast<const float2 *>(lds + at);
        v[0] = a.x, v[1] = a.y;
    } else {
#pragma unroll
        for (uint32_t h = 0; h < QL / 4; ++h) {
            const float4 a = *reinterpret_cast<const float4 *>(lds + at + 4 * h);
            v[4 * h] = a.x, v[4 * h + 1] = a.y, v[4 * h + 2] = a.z, v[4 * h + 3] = a.w;
        }
    }
}

// D(q, i) of QL of the batch's qb queries (those from `first` on) for row i, subspace 0 first (the order of k_adc_scan and
// of the oracle)
template <uint32_t QL>
__device__ __forceinline__ void adc_row(const uint8_t *__restrict__ codes, uint64_t i, uint32_t m, uint32_t k, bool words,
                                        const float *__restrict__ lds, uint32_t qb, uint32_t first, float (&acc)[QL]) {
    float v[QL];
    if (words) {  // one-byte codes, rows of whole 8-byte words
        for (uint32_t s8 = 0; s8 < m; s8 += 8) {
            const uint2 w = *reinterpret_cast<const uint2 *>(codes + i * m + s8);
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const uint32_t s = s8 + b;
                adc_terms<QL>(lds, (s * k + (((b < 4 ? w.x : w.y) >> (8 * (b & 3))) & 255u)) * qb + first, v);
#pragma unroll
                for (uint32_t qq = 0; qq < QL; ++qq) acc[qq] = (s == 0) ? v[qq] : acc[qq] + v[qq];
            }
        }
    } else {
        for (uint32_t s = 0; s < m; ++s) {
            adc_terms<QL>(lds, (s * k + load_code(codes, i * m + s, k)) * qb + first, v);
#pragma unroll
            for (uint32_t qq = 0; qq < QL; ++qq) acc[qq] = (s == 0) ? v[qq] : acc[qq] + v[qq];
        }
    }
}

__device__ __forceinline__ void adc_tables_to_lds(float *__restrict__ lds, const float *__restrict__ src, uint32_t floats, uint32_t nt) {
    for (uint32_t e = 4 * threadIdx.x; e < floats; e += 4 * nt)  // (adc_tabp floats: whole 16-byte units)
        *reinterpret_cast<float4 *>(lds + e) = *reinterpret_cast<const float4 *>(src + e);
}

// The sample behind the thresholds: G workgroups per batch of queries, every thread `rpt` of 1024 G rpt evenly spaced rows;
// wmins[q][16 g + wave] = the wave's minimum of D(q, .).  The scan takes the j-th smallest of a query's 16 G minima as
// its threshold: for j well below 16 G that is close to the j-th smallest of all S = 1024 G rpt sampled distances, the
// j / S quantile -- about n j / S rows at or below it (the host picks G, rpt, j for 1024-4096 of them).  One workgroup
// per batch (round 6's first form) spent 40 us reading its own CU's LDS; the tables' 64 KB per workgroup are the cost now.
constexpr uint32_t kAdcStage = 128;    // candidates a workgroup stages per query before it appends them to the query's list
constexpr uint32_t kAdcCntStride = 64;  // the queries' list counters sit 256 bytes apart: 64 of them in two cache lines took
                                        // every append of every workgroup through one L2 channel (~10 ns each: 320 us of a 64-query scan)
constexpr uint32_t kAdcMaxG = 64;  // sampler workgroups per batch (16 kAdcMaxG minima per query at most)
template <uint32_t QB>
__global__ __launch_bounds__(1024) void k_adc_thresh(const uint8_t *__restrict__ codes, uint64_t n, uint32_t m, uint32_t k,
                                                     const float *__restrict__ lut, uint32_t rpt, float *__restrict__ wmins,
                                                     uint32_t *__restrict__ cand_n) {
    extern __shared__ float lds_lut[];  // [m][k][QB]
    const uint32_t tab = adc_tabp(m, k, QB), batch = blockIdx.y, G = gridDim.x;
    adc_tables_to_lds(lds_lut, lut + (size_t)batch * tab, tab, 1024);
    // (the candidate lists' counters start at zero: one launch less than a memset in front)
    if (blockIdx.x == 0 && threadIdx.x < QB) cand_n[(size_t)(batch * QB + threadIdx.x) * kAdcCntStride] = 0u;
    __syncthreads();
    const bool words = k <= 256 && (m & 7u) == 0 && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0;
    float mn[QB];
#pragma unroll
    for (uint32_t qq = 0; qq < QB; ++qq) mn[qq] = __builtin_inff();
    const uint64_t total = 1024ull * G * rpt;
    for (uint32_t u = 0; u < rpt; ++u) {
        const uint64_t i = (((uint64_t)u * G + blockIdx.x) * 1024 + threadIdx.x) * n / total;
        float acc[QB];
        adc_row<QB>(codes, i, m, k, words, lds_lut, QB, 0u, acc);
#pragma unroll
        for (uint32_t qq = 0; qq < QB; ++qq) mn[qq] = fminf(mn[qq], acc[qq]);  // (NaN never wins)
    }
#pragma unroll
    for (uint32_t qq = 0; qq < QB; ++qq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn[qq] = fminf(mn[qq], __shfl_xor(mn[qq], off));
        if ((threadIdx.x & 63u) == 0) wmins[(size_t)(batch * QB + qq) * (16 * kAdcMaxG) + 16 * blockIdx.x + (threadIdx.x >> 6)] = mn[qq];
    }
}

// the j-th smallest (j >= 1) of `count` <= 16 kAdcMaxG values, by one wave: every lane holds up to 16, j rounds of "wave
// minimum, its first holder drops one copy"
__device__ __forceinline__ float adc_jth_smallest(const float *__restrict__ vals, uint32_t count, uint32_t j, uint32_t lane) {
    float v[16];
#pragma unroll
    for (uint32_t u = 0; u < 16; ++u) v[u] = (64 * u + lane < count) ? vals[64 * u + lane] : __builtin_inff();
    float t = __builtin_inff();
    for (uint32_t r = 0; r < j; ++r) {
        float mine = v[0];
#pragma unroll
        for (uint32_t u = 1; u < 16; ++u) mine = fminf(mine, v[u]);
        t = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t = fminf(t, __shfl_xor(t, off));
        const uint64_t holders = __ballot(mine == t);
        if (holders && lane == (uint32_t)__builtin_ctzll(holders)) {
            bool dropped = false;
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u)
                if (!dropped && v[u] == t) v[u] = __builtin_inff(), dropped = true;
        }
    }
    return t;  // (+inf when fewer than j finite values exist: every row passes, the list overflows, the query is repeated)
}

// LQ lanes per row: lane (row, part) holds QL = QB / LQ of the batch's queries, ONE 16-byte read per subspace when LQ > 1.
// A row per lane (LQ = 1) reads QB floats at a random 4 QB-byte slot per lane and subspace: the LDS serves eight lanes'
// 16 bytes per clock only when they fall on eight different 16-byte bank groups, and eight random ones pile ~2.7 deep
// (measured: 35 % of the LDS rate).  With the LQ lanes of a row on consecutive 16-byte pieces of one entry a pass of eight
// lanes touches 8 / LQ random entries instead of eight.
template <uint32_t QB, uint32_t LQ, uint32_t NT>
__global__ __launch_bounds__(NT) void k_adc_scan_thr(const uint8_t *__restrict__ codes, uint64_t n, uint32_t m, uint32_t k,
                                                     const float *__restrict__ lut, uint32_t nq, const float *__restrict__ wmins,
                                                     uint32_t n_wmins, uint32_t order,
                                                     unsigned long long *__restrict__ cand, uint32_t *__restrict__ cand_n) {
    static_assert(NT >= 64 * QB, "one wave per query places its threshold");
    static_assert(QB % LQ == 0 && (LQ == 1 || QB / LQ == 4), "a lane holds all of the batch's queries or four of them");
    constexpr uint32_t QL = QB / LQ, RPB = NT / LQ;  // queries per lane, rows per pass of the workgroup
    extern __shared__ float lds_lut[];  // [m][k][QB], then the staged candidates [QB][kAdcStage] (8 bytes each) and their counts [QB]
    const uint32_t tab = adc_tabp(m, k, QB), batch = blockIdx.y, q_first = batch * QB;
    unsigned long long *stage = reinterpret_cast<unsigned long long *>(lds_lut + tab);
    uint32_t *stage_n = reinterpret_cast<uint32_t *>(stage + QB * kAdcStage);
    __shared__ uint32_t s_base[QB];
    __shared__ float s_T[QB];
    adc_tables_to_lds(lds_lut, lut + (size_t)batch * tab, tab, NT);
    if (threadIdx.x < QB) stage_n[threadIdx.x] = 0u;
    if (threadIdx.x < 64 * QB) {  // wave qq: the threshold of the batch's query qq (padding queries: nothing passes)
        const uint32_t qq = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const float t = (q_first + qq < nq) ? adc_jth_smallest(wmins + (size_t)(q_first + qq) * (16 * kAdcMaxG), n_wmins, order, lane) : -__builtin_inff();
        if (lane == 0) s_T[qq] = t;
    }
    __syncthreads();
    const uint32_t part = threadIdx.x % LQ, first = part * QL;
    float T[QL];
#pragma unroll
    for (uint32_t qq = 0; qq < QL; ++qq) T[qq] = s_T[first + qq];
    const bool words = k <= 256 && (m & 7u) == 0 && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0;
    for (uint64_t i = (uint64_t)blockIdx.x * RPB + threadIdx.x / LQ; i < n; i += (uint64_t)gridDim.x * RPB) {
        float acc[QL];
        adc_row<QL>(codes, i, m, k, words, lds_lut, QB, first, acc);
        bool any = false;
#pragma unroll
        for (uint32_t qq = 0; qq < QL; ++qq) any |= acc[qq] <= T[qq];
        if (any) {  // ~1e-3 of the rows per query
#pragma unroll
            for (uint32_t qq = 0; qq < QL; ++qq)
                if (acc[qq] <= T[qq]) {
                    const uint32_t bq = first + qq;
                    const unsigned long long e = ((unsigned long long)adc_key(acc[qq]) << 32) | (uint32_t)i;
                    const uint32_t sp = atomicAdd(&stage_n[bq], 1u);
                    if (sp < kAdcStage) {
                        stage[bq * kAdcStage + sp] = e;
                    } else {  // stage full (ties piled on one value): straight to the list
                        const uint32_t pos = atomicAdd(&cand_n[(size_t)(q_first + bq) * kAdcCntStride], 1u);
                        if (pos < kAdcCand) cand[(size_t)(q_first + bq) * kAdcCand + pos] = e;
                    }
                }
        }
    }
    __syncthreads();
    // one append per (workgroup, query)
    if (threadIdx.x < QB) {
        const uint32_t c = min(stage_n[threadIdx.x], kAdcStage);
        s_base[threadIdx.x] = c ? atomicAdd(&cand_n[(size_t)(q_first + threadIdx.x) * kAdcCntStride], c) : 0u;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < QB * kAdcStage; e += NT) {
        const uint32_t qq = e / kAdcStage, j = e % kAdcStage;
        if (j < min(stage_n[qq], kAdcStage)) {
            const uint32_t pos = s_base[qq] + j;
            if (pos < kAdcCand) cand[(size_t)(q_first + qq) * kAdcCand + pos] = stage[qq * kAdcStage + j];
        }
    }
}

// the candidates of a query sorted by (key, row) in LDS, the first topk out; redo[q] = 1 where the threshold let fewer
// than topk or more than kAdcCand rows pass (the caller repeats those queries with the full pass)
__global__ __launch_bounds__(1024) void k_adc_sort_thr(const unsigned long long *__restrict__ cand,
                                                       const uint32_t *__restrict__ cand_n, uint32_t topk, int take_sqrt,
                                                       uint32_t *__restrict__ idx_out, float *__restrict__ dist_out,
                                                       uint32_t *__restrict__ redo, int force_redo) {
    extern __shared__ unsigned long long sort_buf[];  // [kAdcCand]
    const uint32_t q = blockIdx.x, cnt = cand_n[(size_t)q * kAdcCntStride];
    if (threadIdx.x == 0) redo[gridDim.x + q] = cnt;  // (diagnostics: candidates the threshold let pass)
    if (cnt > kAdcCand || cnt < topk || force_redo) {
        if (threadIdx.x == 0) redo[q] = 1u;
        return;
    }
    if (threadIdx.x == 0) redo[q] = 0u;
    const unsigned long long *cq = cand + (size_t)q * kAdcCand;
    if (topk <= 64) {
        // No block-wide sort (a bitonic network over >= 1024 slots is 55-66 barriers: 30 us for ~1000 candidates).  Every
        // wave keeps the 64 smallest of its share: a chunk of 64 sorted by a shuffle network, min'ed against the reversed
        // running best (the 64 smallest of the 128, as a bitonic sequence), six merge steps; wave 0 then folds the sixteen
        // sorted runs the same way.  One barrier.
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        auto sort64 = [&](unsigned long long x) {
#pragma unroll
            for (uint32_t kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    const unsigned long long p = __shfl_xor(x, (int)j);
                    const bool keep_min = ((lane & j) == 0) == ((lane & kk) == 0);
                    x = keep_min ? (x < p ? x : p) : (x < p ? p : x);
                }
            return x;
        };
        auto fold = [&](unsigned long long best, unsigned long long sorted_run) {  // both ascending -> the 64 smallest, ascending
            const unsigned long long r = __shfl(sorted_run, (int)(63u - lane));
            unsigned long long x = best < r ? best : r;
#pragma unroll
            for (uint32_t j = 32; j > 0; j >>= 1) {
                const unsigned long long p = __shfl_xor(x, (int)j);
                x = ((lane & j) == 0) ? (x < p ? x : p) : (x < p ? p : x);
            }
            return x;
        };
        unsigned long long best = ~0ull;
        for (uint32_t c = wv * 64; c < cnt; c += 1024) best = fold(best, sort64((c + lane < cnt) ? cq[c + lane] : ~0ull));
        // the sixteen runs folded pairwise (four rounds; one wave folding fifteen runs in turn was 6 of the kernel's 13 us)
        for (uint32_t stride = 1; stride < 16; stride <<= 1) {
            if ((wv & (2 * stride - 1)) == stride) sort_buf[wv * 64 + lane] = best;
            __syncthreads();
            if ((wv & (2 * stride - 1)) == 0) best = fold(best, sort_buf[(wv + stride) * 64 + lane]);
        }
        if (wv == 0 && lane < topk) {
            float dv = adc_unkey((uint32_t)(best >> 32));
            if (take_sqrt) dv = sqrtf(dv);
            idx_out[(size_t)q * topk + lane] = (uint32_t)best;
            dist_out[(size_t)q * topk + lane] = dv;
        }
        return;
    }
    uint32_t len = 1024;
    while (len < cnt) len <<= 1;
    for (uint32_t e = threadIdx.x; e < len; e += 1024) sort_buf[e] = (e < cnt) ? cq[e] : ~0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= len; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < len; t += 1024) {
                const uint32_t partner = t ^ stride;
                if (partner > t) {
                    const bool up = (t & size) == 0;
                    const unsigned long long a = sort_buf[t], b = sort_buf[partner];
                    if ((a > b) == up) {
                        sort_buf[t] = b;
                        sort_buf[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x < topk) {
        const unsigned long long w = sort_buf[threadIdx.x];
        float dv = adc_unkey((uint32_t)(w >> 32));
        if (take_sqrt) dv = sqrtf(dv);
        idx_out[(size_t)q * topk + threadIdx.x] = (uint32_t)w;
        dist_out[(size_t)q * topk + threadIdx.x] = dv;
    }
}

template <uint32_t QB, uint32_t LQ, uint32_t NT>
int adc_fast_launch(const float *cb, uint32_t m, uint32_t k, uint32_t sd, int l1, int take_sqrt, const uint8_t *codes, uint64_t n,
                    const float *queries_dev, uint32_t nq, uint32_t topk, float *lut_ws, void *state_ws,
                    unsigned long long *cand_ws, uint32_t *idx_out_dev, float *dist_out_dev, uint32_t *redo_dev,
                    hipStream_t stream, uint32_t G, uint32_t rpt, uint32_t order, int force_redo) {
    const size_t tab_b = (size_t)adc_tabp(m, k, QB) * 4;
    static PerDeviceOnce attr;
    if (attr.needed()) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_adc_scan_thr<QB, LQ, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_adc_thresh<QB>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_adc_sort_thr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kAdcCand * 8)));
        attr.done();
    }
    const uint32_t batches = (nq + QB - 1) / QB;
    // state: the sampler's wave minima [batches QB][16 kAdcMaxG] f32 | the list counters, 256 bytes apart
    float *wmins = reinterpret_cast<float *>(state_ws);
    uint32_t *cand_n = reinterpret_cast<uint32_t *>(wmins + (size_t)batches * QB * 16 * kAdcMaxG);
    hipLaunchKernelGGL(k_adc_lut_i<QB>, dim3(batches * QB, m), dim3(256), 0, stream, queries_dev, nq, m, k, sd, cb, l1, lut_ws);
    VQ_LAUNCH_CHECK("k_adc_lut_i");
    hipLaunchKernelGGL(k_adc_thresh<QB>, dim3(G, batches), dim3(1024), tab_b, stream, codes, n, m, k, lut_ws, rpt, wmins, cand_n);
    VQ_LAUNCH_CHECK("k_adc_thresh");
    const size_t scan_lds = tab_b + QB * kAdcStage * 8 + QB * 4;
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / NT, (158 * 1024) / scan_lds));
    uint64_t blocks = (n + NT / LQ - 1) / (NT / LQ);
    const uint64_t cap = std::max<uint64_t>(1, ((uint64_t)num_cus() * per_cu + batches - 1) / batches);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_adc_scan_thr<QB, LQ, NT>), dim3((uint32_t)blocks, batches), dim3(NT), scan_lds, stream, codes, n, m, k, lut_ws, nq,
                       wmins, 16 * G, order, cand_ws, cand_n);
    VQ_LAUNCH_CHECK("k_adc_scan_thr");
    hipLaunchKernelGGL(k_adc_sort_thr, dim3(nq), dim3(1024), (size_t)kAdcCand * 8, stream, cand_ws, cand_n, topk, take_sqrt, idx_out_dev, dist_out_dev, redo_dev, force_redo);
    VQ_LAUNCH_CHECK("k_adc_sort_thr");
    return VQHIP_OK;
}

}  // namespace

// queries_dev [nq][m*sd]; workspaces sized for a group of `qgroup` queries (adc_query_group): lut_ws >= qgroup*m*k floats,
// dist_ws >= qgroup*n floats, state_ws >= adc_state_bytes(qgroup), cand_ws >= adc_cand_bytes(qgroup); outputs on the device
int launch_adc_search(const float *cb, uint32_t m, uint32_t k, uint32_t sd, int metric, const uint8_t *codes, uint64_t n,
                      const float *queries_dev, uint32_t nq, uint32_t topk, float *lut_ws, float *dist_ws,
                      void *state_ws, unsigned long long *cand_ws, uint32_t *idx_out_dev, float *dist_out_dev,
                      hipStream_t stream, uint32_t qgroup) {
    if (vq_is_cos(metric))
        return fail(VQHIP_ERR_UNSUPPORTED, "cosine distance is not a sum over subspaces: no ADC form");
    if (topk == 0 || topk > 1024 || topk > n) return fail(VQHIP_ERR_INVALID_INPUT, "topk must be in [1, min(n, 1024)]");
    // queries per scan pass: as many tables as the LDS holds, at most kAdcQB
    const size_t lds_budget = 150 * 1024;
    uint32_t qb = (uint32_t)std::min<size_t>(kAdcQB, lds_budget / (((size_t)m * k + kAdcBins) * 4));
    if (qb == 0) return fail(VQHIP_ERR_UNSUPPORTED, "one ADC table (m=%u, k=%u) exceeds the LDS", m, k);
    static PerDeviceOnce attr;
    if (attr.needed()) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_adc_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   150 * 1024));
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_adc_sort_out),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kAdcCand * 8)));
        attr.done();
    }
    const int l1 = metric == VQHIP_MANHATTAN ? 1 : 0;
    const int take_sqrt = metric == VQHIP_EUCLIDEAN ? 1 : 0;
    // the scan: every workgroup first copies the batch's tables (64 KB at m = 8, k = 256, 8 queries) into LDS -- two
    // workgroups per CU is what that LDS allows, and 2048 of them spent the pass re-reading tables (128 MB of L2 traffic for
    // 8 MB of codes: 70 us per pass at 1M rows)
    const size_t scan_lds = (size_t)qb * ((size_t)m * k + kAdcBins) * 4;
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(8, (150 * 1024) / std::max<size_t>(scan_lds, 1)));
    // A GROUP of up to `qgroup` queries (the caller's workspaces: adc_query_group(n)) goes through one set of launches: the
    // six kernels of a pass are dependent and tiny (5-25 us each, 110 us per pass whatever the work), so eight passes one
    // after the other cost eight times that for 64 queries
    const uint32_t qg = std::max(qb, qgroup / qb * qb);
    // small state per query of the group: bounds [2] f32 | hist [bins] u32 | sel [2] u32 | cand_n u32
    float *bounds = reinterpret_cast<float *>(state_ws);
    uint32_t *hist = reinterpret_cast<uint32_t *>(bounds + 2 * (size_t)qg);
    uint32_t *sel = hist + (size_t)qg * kAdcBins;
    uint32_t *cand_n = sel + 2 * (size_t)qg;
    const size_t state_bytes = adc_state_bytes(qg);
    for (uint32_t q0 = 0; q0 < nq; q0 += qg) {
        const uint32_t nqg = (nq - q0 < qg) ? nq - q0 : qg;
        const uint32_t batches = (nqg + qb - 1) / qb;
        uint64_t blocks = (n + 255) / 256;
        const uint64_t cap = std::max<uint64_t>(1, (uint64_t)num_cus() * per_cu / batches);
        if (blocks > cap) blocks = cap;
        VQ_HIP(hipMemsetAsync(state_ws, 0, state_bytes, stream));
        hipLaunchKernelGGL(k_adc_lut, dim3(nqg, m), dim3(256), 0, stream, queries_dev + (size_t)q0 * m * sd, nqg, m, k, sd,
                           cb, l1, lut_ws, bounds);
        VQ_LAUNCH_CHECK("k_adc_lut");
        hipLaunchKernelGGL(k_adc_scan, dim3((uint32_t)blocks, batches), dim3(256), (size_t)qb * (m * k + kAdcBins) * 4, stream,
                           codes, n, m, k, lut_ws, nqg, qb, bounds, dist_ws, hist);
        VQ_LAUNCH_CHECK("k_adc_scan");
        // top-k: candidates below a histogram cut, sorted in LDS; dense cuts fall back to the radix select
        hipLaunchKernelGGL(k_adc_pick_bin, dim3(nqg), dim3(64), 0, stream, hist, topk, sel);
        hipLaunchKernelGGL(k_adc_collect, dim3(64, nqg), dim3(256), 0, stream, dist_ws, n, bounds, sel, cand_ws, cand_n);
        hipLaunchKernelGGL(k_adc_sort_out, dim3(nqg), dim3(1024), (size_t)kAdcCand * 8, stream, cand_ws, sel, topk, take_sqrt,
                           idx_out_dev + (size_t)q0 * topk, dist_out_dev + (size_t)q0 * topk);
        hipLaunchKernelGGL(k_adc_topk, dim3(nqg), dim3(1024), 0, stream, dist_ws, n, topk, take_sqrt, sel,
                           idx_out_dev + (size_t)q0 * topk, dist_out_dev + (size_t)q0 * topk);
        VQ_LAUNCH_CHECK("k_adc_topk");
    }
    return VQHIP_OK;
}

size_t adc_state_bytes(uint32_t qgroup) { return (size_t)qgroup * (2 + kAdcBins + 2 + 1) * 4; }
size_t adc_cand_bytes(uint32_t qgroup) { return (size_t)qgroup * kAdcCand * 8; }
uint32_t adc_query_batch() { return kAdcQB; }
// queries that share one set of launches: up to 64, as long as their distance rows (4 n bytes each) stay under 1 GB
uint32_t adc_query_group(uint64_t n, uint32_t nq) {
    uint64_t g = (1ull << 30) / std::max<uint64_t>(4 * n, 1);
    g = std::min<uint64_t>(std::max<uint64_t>(g / kAdcQB * kAdcQB, kAdcQB), 64);
    return (uint32_t)std::min<uint64_t>(g, ((uint64_t)nq + kAdcQB - 1) / kAdcQB * kAdcQB);
}

// ---- the threshold pass (round 6) ----
// eligible: enough rows for a sample to place a threshold, few enough results for the candidate lists
bool adc_fast_eligible(uint32_t m, uint32_t k, uint64_t n, uint32_t topk) {
    static const char *env = getenv("VQHIP_ADC_FAST");
    if (env && env[0] == '0') return false;
    return n >= 32768 && topk <= 256 && (size_t)m * k * 4 <= 150 * 1024;
}
// queries per scan batch: the largest power of two up to eight whose interleaved tables fit the LDS.  (Sixteen per batch,
// four lanes per row, one workgroup of 1024 per CU: scan 59.8 us against ~57 for 64 queries over 1M rows, and the
// sampler's workgroups load 128 KB of tables each -- 14.6 us against 7.9: not kept.)
uint32_t adc_fast_batch(uint32_t m, uint32_t k, uint32_t nq) {
    (void)nq;
    uint32_t qb = 8;
    while (qb > 1 && (size_t)m * k * qb * 4 > 150 * 1024) qb >>= 1;
    return qb;
}
size_t adc_fast_lut_bytes(uint32_t m, uint32_t k, uint32_t nq) {
    const uint32_t qb = adc_fast_batch(m, k, nq);
    return (size_t)((nq + qb - 1) / qb) * (((size_t)m * k * qb + 3) & ~(size_t)3) * 4;
}
size_t adc_fast_state_bytes(uint32_t m, uint32_t k, uint32_t nq) {
    const uint32_t qb = adc_fast_batch(m, k, nq);
    return (size_t)((nq + qb - 1) / qb) * qb * (4 * 16 * 64 + 4 * 64);  // the sampler's wave minima (16 kAdcMaxG per query) + the list counters 256 bytes apart (kAdcCntStride)
}
size_t adc_fast_cand_bytes(uint32_t m, uint32_t k, uint32_t nq) {
    const uint32_t qb = adc_fast_batch(m, k, nq);
    return (size_t)((nq + qb - 1) / qb) * qb * kAdcCand * 8;
}
// all nq queries in one set of five launches; redo_dev[q] = 1 where the caller must repeat query q with launch_adc_search
int launch_adc_search_fast(const float *cb, uint32_t m, uint32_t k, uint32_t sd, int metric, const uint8_t *codes, uint64_t n,
                           const float *queries_dev, uint32_t nq, uint32_t topk, float *lut_ws, void *state_ws,
                           unsigned long long *cand_ws, uint32_t *idx_out_dev, float *dist_out_dev, uint32_t *redo_dev,
                           hipStream_t stream) {
    if (vq_is_cos(metric))
        return fail(VQHIP_ERR_UNSUPPORTED, "cosine distance is not a sum over subspaces: no ADC form");
    if (topk == 0 || topk > 1024 || topk > n) return fail(VQHIP_ERR_INVALID_INPUT, "topk must be in [1, min(n, 1024)]");
    const int l1 = metric == VQHIP_MANHATTAN ? 1 : 0, take_sqrt = metric == VQHIP_EUCLIDEAN ? 1 : 0;
    // candidates wanted: 16 topk, at least 1024 (kAdcCand = 8192 is the list's size): the j-th smallest of S sampled
    // distances sits near the j / S quantile, n j / S rows.  j = 8 (its rank among all rows scatters by 1 / sqrt(j))
    // and S = 8 n / want rows, 1024 per sampler workgroup; past 64 workgroups x 16 rows per thread j comes down instead
    const double want = std::min<double>(std::max<double>(16.0 * topk, 1024.0), 4096.0);
    uint32_t order = 8;
    const double S = std::min<double>((double)order * (double)n / want, (double)n);
    uint32_t G = (uint32_t)std::min<double>(std::max<double>(std::ceil(S / 1024.0), 1.0), (double)kAdcMaxG);
    uint32_t rpt = (uint32_t)std::min<double>(std::max<double>(std::ceil(S / (1024.0 * G)), 1.0), 16.0);
    while ((uint64_t)1024 * G * rpt > n && rpt > 1) --rpt;
    while ((uint64_t)1024 * G * rpt > n && G > 1) --G;
    order = (uint32_t)std::min<double>(std::max<double>(std::floor(want * (1024.0 * G * rpt) / (double)n + 0.5), 1.0), 8.0);
    static const char *force_env = getenv("VQHIP_TEST_ADC_REDO");  // tests: every query flagged, the caller's repeat path runs
    const int force_redo = (force_env && force_env[0] == '1') ? 1 : 0;
    static const char *lq_env = getenv("VQHIP_ADC_LQ1");  // =1: a row per lane at eight queries per batch (A/B)
    const bool lq1 = lq_env && lq_env[0] == '1';
    // (workgroups of 1024, one per CU, half the table loads: 113 us against 105 per 64-query call -- not kept)
#define VQ_ADC_FAST(QB, LQ, NT)                                                                                                     \
    return adc_fast_launch<QB, LQ, NT>(cb, m, k, sd, l1, take_sqrt, codes, n, queries_dev, nq, topk, lut_ws, state_ws, cand_ws,     \
                                       idx_out_dev, dist_out_dev, redo_dev, stream, G, rpt, order, force_redo);
    switch (adc_fast_batch(m, k, nq)) {
        case 8:
            if (lq1) VQ_ADC_FAST(8, 1, 512)
            VQ_ADC_FAST(8, 2, 512)
        case 4: VQ_ADC_FAST(4, 1, 512)
        case 2: VQ_ADC_FAST(2, 1, 512)
        case 1: VQ_ADC_FAST(1, 1, 512)
    }
#undef VQ_ADC_FAST
    return fail(VQHIP_ERR_RUNTIME, "adc_fast_batch");
}

}  // namespace vqhip
