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
#include <algorithm>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr uint32_t kAccBlock = 1024;
constexpr uint32_t kLdsBudgetWords = 38912;  // 152 KiB of the CU's 160 KiB

// LDS image of one workgroup: for each of its `spc` subspaces k rows of (sd + 1) floats (the
// pad word spreads the k rows over all 32 LDS banks: with stride sd = 16 every row would start
// on bank 0 or 16 and a wave's 64 atomics would pile onto 8 banks), then spc*k counts.
template <int VEC>
__global__ __launch_bounds__(kAccBlock) void k_accumulate(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd,
    uint32_t spc, uint32_t kr, uint64_t rows_per_chunk, const uint8_t *__restrict__ codes,
    const uint8_t *__restrict__ active, float *__restrict__ partial_sums,
    uint32_t *__restrict__ partial_counts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // cluster range of this workgroup: [k_lo, k_lo + kn).  kr == k (one range) whenever a whole subspace fits
    // the LDS; otherwise blockIdx.z walks the ranges and rows whose code lies outside are skipped.
    const uint32_t k_lo = blockIdx.z * kr;
    const uint32_t kn = (k - k_lo < kr) ? (k - k_lo) : kr;
    const uint32_t s0 = blockIdx.y * spc;
    const uint32_t ns = (m - s0 < spc) ? (m - s0) : spc;
    const uint32_t W = ns * sd;  // floats of a row handled by this workgroup
    const uint32_t rstride = sd + 1;
    float *sums = lds;                                                              // [ns][kr][sd+1]
    uint32_t *cnts = reinterpret_cast<uint32_t *>(lds + (size_t)spc * kr * rstride);  // [ns][kr]
    for (uint32_t e = threadIdx.x; e < ns * kr * rstride; e += kAccBlock) sums[e] = 0.0f;
    for (uint32_t e = threadIdx.x; e < ns * kr; e += kAccBlock) cnts[e] = 0u;
    __syncthreads();

    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_chunk;
    uint64_t r1 = r0 + rows_per_chunk;
    if (r1 > n) r1 = n;
    // a thread keeps ONE column group of the row slice for the whole chunk and walks down the
    // rows: no per-item index arithmetic in the loop
    const uint32_t ipr = W / VEC;                  // items per row
    const uint32_t rpp = kAccBlock / ipr;          // rows per pass (>= 1: W <= 4096 words)
    const uint32_t q = threadIdx.x % ipr, rr = threadIdx.x / ipr;
    const uint32_t col = q * VEC, ls = col / sd, t = col - ls * sd;
    const uint32_t s = s0 + ls;
    const bool live = (rr < rpp) && (!active || active[s]);
    if (live) {
        float *base = sums + (size_t)ls * kr * rstride + t;
        uint32_t *cbase = cnts + ls * kr;
        const float *px = X + (size_t)s0 * sd + col;
        uint64_t row = r0 + rr;
        // two rows in flight per thread
        for (; row + rpp < r1; row += 2 * (uint64_t)rpp) {
            const uint64_t rowb = row + rpp;
            // codes relative to the range; out-of-range (unsigned wrap included) rows are another workgroup's
            const uint32_t ca = load_code(codes, row * m + s, k) - k_lo, cb2 = load_code(codes, rowb * m + s, k) - k_lo;
            const bool ina = ca < kn, inb = cb2 < kn;
            if constexpr (VEC == 4) {
                const float4 va = *reinterpret_cast<const float4 *>(px + row * d);
                const float4 vb = *reinterpret_cast<const float4 *>(px + rowb * d);
                float *da = base + ca * rstride, *db = base + cb2 * rstride;
                if (ina) {
                    atomicAdd(da + 0, va.x);
                    atomicAdd(da + 1, va.y);
                    atomicAdd(da + 2, va.z);
                    atomicAdd(da + 3, va.w);
                }
                if (inb) {
                    atomicAdd(db + 0, vb.x);
                    atomicAdd(db + 1, vb.y);
                    atomicAdd(db + 2, vb.z);
                    atomicAdd(db + 3, vb.w);
                }
            } else {
                if (ina) atomicAdd(base + ca * rstride, px[row * d]);
                if (inb) atomicAdd(base + cb2 * rstride, px[rowb * d]);
            }
            if (t == 0) {
                if (ina) atomicAdd(cbase + ca, 1u);
                if (inb) atomicAdd(cbase + cb2, 1u);
            }
        }
        for (; row < r1; row += rpp) {
            const uint32_t ca = load_code(codes, row * m + s, k) - k_lo;
            if (ca >= kn) continue;
            if constexpr (VEC == 4) {
                const float4 va = *reinterpret_cast<const float4 *>(px + row * d);
                float *da = base + ca * rstride;
                atomicAdd(da + 0, va.x);
                atomicAdd(da + 1, va.y);
                atomicAdd(da + 2, va.z);
                atomicAdd(da + 3, va.w);
            } else {
                atomicAdd(base + ca * rstride, px[row * d]);
            }
            if (t == 0) atomicAdd(cbase + ca, 1u);
        }
    }
    __syncthreads();
    // partial slab of this row chunk: sums [m][k][sd] (un-padded), counts [m][k]
    float *ps = partial_sums + ((size_t)blockIdx.x * m + s0) * k * sd;
    for (uint32_t e = threadIdx.x; e < ns * kn * sd; e += kAccBlock) {
        const uint32_t rowi = e / sd, tt = e - rowi * sd, lsi = rowi / kn, j = rowi - lsi * kn;
        ps[((size_t)lsi * k + k_lo + j) * sd + tt] = sums[((size_t)lsi * kr + j) * rstride + tt];
    }
    uint32_t *pcnt = partial_counts + ((size_t)blockIdx.x * m + s0) * k;
    for (uint32_t e = threadIdx.x; e < ns * kn; e += kAccBlock) {
        const uint32_t lsi = e / kn, j = e - lsi * kn;
        pcnt[(size_t)lsi * k + k_lo + j] = cnts[lsi * kr + j];
    }
}

// Wave-owned accumulation (the fast path).  LDS float atomics run at ~1 lane per 3 cycles on
// gfx950 (k_accumulate above is LDS-bound at 0.8 TB/s), so this kernel uses none: wave w of a
// workgroup OWNS the accumulators of subspace s0+w ([k][sd] f32 + [k] counts in LDS) and is the
// only writer.  One wave step covers RPS = 64/KS consecutive rows (KS = sd/4 lanes of 16 B per
// row).  Rows of the same step that share a cluster would collide in a plain read-modify-write,
// so each row gets rank = number of EARLIER rows of the step with the same code, and the step
// runs rounds r = 0, 1, ...: rows of rank r add themselves with ds_read_b128 / add / ds_write_b128.
// Every cluster therefore receives its rows in ascending row order: the chunk's partial sum is
// the reference's sequential f32 sum over the chunk (src/core/vector.rs:376-380), bit for bit
// and run to run.
// VW floats per lane: 4 (16-byte parts, sub_dim a multiple of 4), 2 (even sub_dims) or 1 (odd ones): the sub_dims
// that the zero-padded screen serves (5..23) keep the atomic-free update this way.
template <int VW> struct AccPart;
template <> struct AccPart<4> {
    using T = float4;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ T add(T a, const T &x) {
        a.x = a.x + x.x;
        a.y = a.y + x.y;
        a.z = a.z + x.z;
        a.w = a.w + x.w;
        return a;
    }
};
template <> struct AccPart<2> {
    using T = float2;
    static __device__ __forceinline__ T zero() { return make_float2(0.f, 0.f); }
    static __device__ __forceinline__ T add(T a, const T &x) {
        a.x = a.x + x.x;
        a.y = a.y + x.y;
        return a;
    }
};
template <> struct AccPart<1> {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.0f; }
    static __device__ __forceinline__ T add(T a, const T &x) { return a + x; }
};

template <int KS, int VW = 4>
__global__ __launch_bounds__(512) void k_accumulate_owned(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, uint32_t k,
    uint32_t waves_per_block, uint64_t rows_per_chunk, const uint8_t *__restrict__ codes,
    const uint8_t *__restrict__ active, float *__restrict__ partial_sums,
    uint32_t *__restrict__ partial_counts, uint32_t row_split) {
    // row_split > 1 (few subspaces): waves_per_block / row_split subspaces per workgroup, and row_split waves per
    // subspace, each adding its own part of the chunk's rows into its own slab -- a full workgroup where m alone
    // would leave one or two waves (m = 1 used to fall back to the LDS-atomic kernel, whose sums depend on the order
    // the atomics happen to land in)
    using Part = typename AccPart<VW>::T;
    constexpr uint32_t SD = KS * VW;
    constexpr uint32_t RPS = 64 / KS;  // rows per step; lanes >= RPS*KS idle when KS is not a power of two
    constexpr bool POW2 = (KS & (KS - 1)) == 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t subs_per_block = waves_per_block / row_split;
    const uint32_t s = blockIdx.y * subs_per_block + wave % subs_per_block, part = wave / subs_per_block;
    if (s >= m) return;                       // no barrier below: waves are independent
    if (active && !active[s]) return;
    const uint32_t per_wave = (k * (SD + 1) + 3u) & ~3u;  // keeps every wave's base 16-byte aligned
    float *sums = lds + (size_t)wave * per_wave;           // [k][SD] then [k] counts
    uint32_t *cnts = reinterpret_cast<uint32_t *>(sums + (size_t)k * SD);
    for (uint32_t e = lane; e < k * SD; e += 64) sums[e] = 0.0f;
    for (uint32_t e = lane; e < k; e += 64) cnts[e] = 0u;

    const uint32_t p = lane / KS, g = lane % KS;
    const bool lane_on = lane < RPS * KS;
    // 16-byte slot of part gg of cluster c: xor swizzle (power-of-two KS) or rotation, so that the
    // KS parts of different clusters spread over the banks
    auto swz = [](uint32_t gg, uint32_t c) { return POW2 ? (gg ^ (c & (KS - 1))) : ((gg + c) % KS); };
    const uint64_t rows_per_part = (rows_per_chunk + row_split - 1) / row_split;
    uint64_t r0 = (uint64_t)blockIdx.x * rows_per_chunk + part * rows_per_part;
    uint64_t r1 = min((uint64_t)blockIdx.x * rows_per_chunk + min((uint64_t)(part + 1) * rows_per_part, rows_per_chunk), n);
    if (r0 > r1) r0 = r1;
    const uint32_t slab = blockIdx.x * row_split + part;
    const float *px = X + (size_t)s * SD + VW * g;
    auto load_x = [&](uint64_t row) {
        return (lane_on && row < r1) ? *reinterpret_cast<const Part *>(px + row * d) : AccPart<VW>::zero();
    };
    auto load_c = [&](uint64_t row) { return (lane_on && row < r1) ? load_code(codes, row * m + s, k) : 0xFFFFFFFFu; };

    // HBM latency (~2 us under load) against one 1-KB load per wave limited the first version to
    // 2.1 TB/s; batches of PF steps are double-buffered so a wave keeps 2*PF KB in flight
    constexpr uint32_t PF = 4;
    Part xb[PF];
    uint32_t cb_[PF];
#pragma unroll
    for (uint32_t i = 0; i < PF; ++i) {
        xb[i] = load_x(r0 + i * RPS + p);
        cb_[i] = load_c(r0 + i * RPS + p);
    }
    for (uint64_t base = r0; base < r1; base += (uint64_t)PF * RPS) {
        Part xc[PF];
        uint32_t cc[PF];
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) {
            xc[i] = xb[i];
            cc[i] = cb_[i];
        }
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) {
            xb[i] = load_x(base + (uint64_t)(PF + i) * RPS + p);
            cb_[i] = load_c(base + (uint64_t)(PF + i) * RPS + p);
        }
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) {
            const Part x = xc[i];
            const uint32_t code = cc[i];
            const bool valid = code != 0xFFFFFFFFu;
            // rank among the step's rows (row q's code sits in lane q*KS)
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t q = 0; q + 1 < RPS; ++q) {
                const uint32_t cq = (uint32_t)__builtin_amdgcn_readlane((int)code, (int)(q * KS));
                rank += (q < p && cq == code) ? 1u : 0u;
            }
            // xor-swizzled 16-byte slot so that the KS parts of different clusters spread over banks
            Part *slot = reinterpret_cast<Part *>(sums + (size_t)(valid ? code : 0u) * SD) + swz(g, code);
            uint32_t pending = valid ? 1u : 0u;
            for (uint32_t r = 0; __any(pending != 0); ++r) {
                if (pending && rank == r) {
                    *slot = AccPart<VW>::add(*slot, x);
                    if (g == 0) cnts[code] += 1u;
                    pending = 0;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // partial slab of this (row chunk, subspace): un-swizzle on the way out
    float *ps = partial_sums + ((size_t)slab * m + s) * k * SD;
    for (uint32_t e = lane; e < k * KS; e += 64) {
        const uint32_t j = e / KS, gg = e % KS;
        const Part v = reinterpret_cast<const Part *>(sums + (size_t)j * SD)[swz(gg, j)];
        reinterpret_cast<Part *>(ps + (size_t)j * SD)[gg] = v;
    }
    uint32_t *pcnt = partial_counts + ((size_t)slab * m + s) * k;
    for (uint32_t e = lane; e < k; e += 64) pcnt[e] = cnts[e];
}

// The fused screen (k_assign_screen_bf16_x32<..., ACC>) sums the rows it proves; the rows it hands to the exact
// re-check are listed per (subspace, screen chunk) in row order (wl_seg / wl_rows).  Once the re-check has written
// their codes, one wave per (patch chunk, subspace) walks its share of the segments in order and adds the listed
// rows exactly like k_accumulate_owned (rank rounds, ascending row order inside a cluster) into one more partial
// slab.  ~0.1 % of the rows on uniform data, a few per cent on tightly clustered data.
template <int KS>
__global__ __launch_bounds__(64) void k_accumulate_listed(
    const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k, const uint8_t *__restrict__ codes,
    const uint32_t *__restrict__ sub_list, const uint32_t *__restrict__ wl_rows, uint64_t wl_stride,
    const uint32_t *__restrict__ wl_seg, uint32_t n_seg, uint32_t segs_per_patch, uint32_t first_chunk,
    float *__restrict__ partial_sums, uint32_t *__restrict__ partial_counts) {
    constexpr uint32_t SD = KS * 4, RPS = 64 / KS;
    constexpr bool POW2 = (KS & (KS - 1)) == 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t s = sub_list[blockIdx.y];
    float *sums = lds;
    uint32_t *cnts = reinterpret_cast<uint32_t *>(sums + (size_t)k * SD);
    for (uint32_t e = lane; e < k * SD; e += 64) sums[e] = 0.0f;
    for (uint32_t e = lane; e < k; e += 64) cnts[e] = 0u;
    const uint32_t p = lane / KS, g = lane % KS;
    const bool lane_on = lane < RPS * KS;
    auto swz = [](uint32_t gg, uint32_t c) { return POW2 ? (gg ^ (c & (KS - 1))) : ((gg + c) % KS); };
    const uint32_t seg0 = blockIdx.x * segs_per_patch;
    const uint32_t seg1 = min(n_seg, seg0 + segs_per_patch);
    // The lists are short (a handful of rows per segment on uniform data) and every entry is three dependent loads
    // deep (segment header -> row number -> code and sub-vector): walked one segment at a time the wave spent its
    // life waiting (33 us for 9 K rows).  So 64 segment headers are read at once, their entries numbered through a
    // wave prefix sum, and the loads of 8 steps (128 entries) are issued level by level before any row is added.
    constexpr int NB = 8;  // steps per batch
    for (uint32_t g0 = seg0; g0 < seg1; g0 += 64) {
        const uint32_t nq = min(64u, seg1 - g0);
        uint32_t first = 0, count = 0;
        if (lane < nq) {
            const uint2 hd = *reinterpret_cast<const uint2 *>(wl_seg + ((size_t)s * n_seg + g0 + lane) * 2);
            first = hd.x;
            count = hd.y;
        }
        uint32_t incl = count;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if ((int)lane >= off) incl += up;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        const uint32_t shift = first - (incl - count);  // entry e of this segment sits at wl_rows[.. + e + shift]
        for (uint32_t b0 = 0; b0 < total; b0 += NB * RPS) {
            uint32_t row[NB], code[NB];
            float4 x[NB];
            bool valid[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t e = b0 + (uint32_t)u * RPS + p;
                valid[u] = lane_on && e < total;
                // segment of entry e: the first q with incl_q > e (a binary search over the lanes' prefix sums: six shuffles; walking
                // the 64 headers by v_readlane was ~320 instructions per entry, most of the kernel's 10 us at C2); its
                // `shift` turns e into a slot of the list
                uint32_t lo = 0, hi = nq - 1u;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t im = (uint32_t)__shfl((int)incl, (int)mid);
                    const bool right = !(e < im) && lo < hi;  // (converged lanes stay where they are)
                    const bool left = (e < im) && lo < hi;
                    lo = right ? mid + 1u : lo;
                    hi = left ? mid : hi;
                }
                const uint32_t sh = (uint32_t)__shfl((int)shift, (int)lo);
                // (clamped, never under a test: `valid ? load : 0` comes out of the compiler as a branch around the load with
                // s_waitcnt vmcnt(0) behind it -- the eight entries of a level were eight memory round trips in a row)
                row[u] = wl_rows[(size_t)s * wl_stride + (valid[u] ? e + sh : 0u)];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(row[u]));  // all eight row numbers requested before the first is used
#pragma unroll
            for (int u = 0; u < NB; ++u) row[u] = valid[u] ? row[u] : 0u;  // (a lane without an entry read a word nobody wrote: row 0 is a valid address)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                code[u] = (uint32_t)codes[(size_t)row[u] * m + s];
                x[u] = *reinterpret_cast<const float4 *>(X + (size_t)row[u] * d + (size_t)s * SD + 4 * g);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                asm volatile("" : "+v"(code[u]), "+v"(x[u].x), "+v"(x[u].y), "+v"(x[u].z), "+v"(x[u].w));
                if (!valid[u]) code[u] = 0xFFFFFFFFu, x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (b0 + (uint32_t)u * RPS >= total) break;
                uint32_t rank = 0;
#pragma unroll
                for (uint32_t q = 0; q + 1 < RPS; ++q) {
                    const uint32_t cq = (uint32_t)__builtin_amdgcn_readlane((int)code[u], (int)(q * KS));
                    rank += (q < p && cq == code[u]) ? 1u : 0u;
                }
                float4 *slot = reinterpret_cast<float4 *>(sums + (size_t)(valid[u] ? code[u] : 0u) * SD) + swz(g, code[u]);
                uint32_t pending = valid[u] ? 1u : 0u;
                for (uint32_t r = 0; __any(pending != 0); ++r) {
                    if (pending && rank == r) {
                        float4 t = *slot;
                        t.x = t.x + x[u].x;
                        t.y = t.y + x[u].y;
                        t.z = t.z + x[u].z;
                        t.w = t.w + x[u].w;
                        *slot = t;
                        if (g == 0) cnts[code[u]] += 1u;
                        pending = 0;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    // slab layout of the fused path: [chunk][position in the active list][k][SD]
    const size_t slab_id = (size_t)(first_chunk + blockIdx.x) * gridDim.y + blockIdx.y;
    float *ps = partial_sums + slab_id * k * SD;
    for (uint32_t e = lane; e < k * KS; e += 64) {
        const uint32_t j = e / KS, gg = e % KS;
        reinterpret_cast<float4 *>(ps + (size_t)j * SD)[gg] = reinterpret_cast<const float4 *>(sums + (size_t)j * SD)[swz(gg, j)];
    }
    uint32_t *pcnt = partial_counts + slab_id * k;
    for (uint32_t e = lane; e < k; e += 64) pcnt[e] = cnts[e];
}

// 32 slab elements x 8 chunk groups per workgroup; every element's chunks are summed in a
// fixed order (group-strided, then groups ascending), so the result is reproducible.
constexpr uint32_t kRedGroups = 8;
__global__ __launch_bounds__(256) void k_reduce_partials(
    const float *__restrict__ partial_sums, const uint32_t *__restrict__ partial_counts,
    uint32_t n_row_chunks, uint32_t m, uint32_t k, uint32_t sd, const uint8_t *__restrict__ active,
    double *__restrict__ slab) {
    __shared__ double part[kRedGroups][32];
    const uint32_t total = m * k * (sd + 1);
    const uint32_t el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const uint32_t e = blockIdx.x * 32 + el;
    double acc = 0.0;
    if (e < total) {
        const uint32_t t = e % (sd + 1), sj = e / (sd + 1), s = sj / k;
        if (!active || active[s]) {
            if (t < sd) {
                const size_t stride = (size_t)m * k * sd;
                const float *p = partial_sums + (size_t)sj * sd + t;
                for (uint32_t c = grp; c < n_row_chunks; c += kRedGroups) acc += (double)p[c * stride];
            } else {
                const size_t stride = (size_t)m * k;
                const uint32_t *p = partial_counts + sj;
                unsigned long long cnt = 0;
                for (uint32_t c = grp; c < n_row_chunks; c += kRedGroups) cnt += p[c * stride];
                acc = (double)cnt;
            }
        }
    }
    part[grp][el] = acc;
    __syncthreads();
    if (grp == 0 && e < total) {
        double r = part[0][el];
        for (uint32_t g = 1; g < kRedGroups; ++g) r += part[g][el];
        slab[e] = r;
    }
}

// the same for the fused path's slabs [chunk][position in the active subspace list][k][sd]; sub_pos[s] = position or -1
__global__ __launch_bounds__(256) void k_reduce_partials_pos(
    const float *__restrict__ partial_sums, const uint32_t *__restrict__ partial_counts,
    uint32_t n_chunks, uint32_t n_sub, const int32_t *__restrict__ sub_pos, uint32_t m, uint32_t k, uint32_t sd,
    double *__restrict__ slab, const uint8_t *__restrict__ gate_active, const uint32_t *__restrict__ gate_halt,
    uint32_t *__restrict__ clear_changed) {
    __shared__ double part[kRedGroups][32];
    if (gate_halt && *gate_halt) return;  // paused run: the slab keeps the pausing iteration's sums
    // device-driven run: the `changed` flags of the iteration are cleared here (k_finalize sets them; a paused run keeps
    // those of the iteration that paused it) -- a kernel of its own for m words was one launch in ten at small sizes
    if (clear_changed && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < m; i += 256) clear_changed[i] = 0u;
    const uint32_t total = m * k * (sd + 1);
    const uint32_t el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const uint32_t e = blockIdx.x * 32 + el;
    double acc = 0.0;
    bool skip = false;
    if (e < total) {
        const uint32_t t = e % (sd + 1), sj = e / (sd + 1), s = sj / k, j = sj - s * k;
        const int32_t pos = sub_pos[s];
        skip = gate_active && !gate_active[s];  // converged inside a device-driven run: its slab stays as it is
        if (pos >= 0 && !skip) {
            if (t < sd) {
                const size_t stride = (size_t)n_sub * k * sd;
                const float *p = partial_sums + ((size_t)pos * k + j) * sd + t;
                // six partials requested at a time, added in the same order as one by one (the f64 sum's order is part
                // of the result's low bits): the loop was a chain of memory latencies
                for (uint32_t c0 = grp; c0 < n_chunks; c0 += 6 * kRedGroups) {
                    float v[6];
#pragma unroll
                    for (uint32_t u = 0; u < 6; ++u) v[u] = p[min(c0 + u * kRedGroups, n_chunks - 1u) * stride];  // (clamped: nothing under a test)
#pragma unroll
                    for (uint32_t u = 0; u < 6; ++u)
                        if (c0 + u * kRedGroups < n_chunks) acc += (double)v[u];
                }
            } else {
                const size_t stride = (size_t)n_sub * k;
                const uint32_t *p = partial_counts + (size_t)pos * k + j;
                unsigned long long cnt = 0;
                for (uint32_t c0 = grp; c0 < n_chunks; c0 += 6 * kRedGroups) {
                    uint32_t v[6];
#pragma unroll
                    for (uint32_t u = 0; u < 6; ++u) v[u] = p[min(c0 + u * kRedGroups, n_chunks - 1u) * stride];
#pragma unroll
                    for (uint32_t u = 0; u < 6; ++u)
                        if (c0 + u * kRedGroups < n_chunks) cnt += v[u];
                }
                acc = (double)cnt;
            }
        }
    }
    part[grp][el] = acc;
    __syncthreads();
    if (grp == 0 && e < total) {
        double r = part[0][el];
        for (uint32_t g = 1; g < kRedGroups; ++g) r += part[g][el];
        if (!skip) slab[e] = r;
    }
}

// one lane per centroid component.  RUN (device-driven run, vqhip_kmeans_run): a paused run is left alone, an active
// subspace with an empty cluster is flagged for k_run_decide, which ends the iteration behind this kernel.
template <bool RUN>
__global__ __launch_bounds__(256) void k_finalize(uint32_t m, uint32_t k, uint32_t sd,
                                                  const double *__restrict__ slab,
                                                  uint8_t *__restrict__ active,
                                                  float *__restrict__ centroids,
                                                  uint32_t *__restrict__ counts,
                                                  uint32_t *__restrict__ changed, int exact_div,
                                                  uint32_t *__restrict__ gate_halt, uint32_t *__restrict__ iters,
                                                  uint32_t *__restrict__ done_blocks) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const bool halted = gate_halt && *gate_halt;  // paused run: centroids, counts and flags keep the pausing iteration's values
    if (e < m * k * sd && !halted) {
        const uint32_t sj = e / sd, t = e - sj * sd, s = sj / k;
        const bool act = !active || active[s];
        const double *row = slab + (size_t)sj * (sd + 1);
        const double cnt = row[sd];
        if (t == 0 && counts) counts[sj] = act ? (uint32_t)cnt : 0u;
        // RUN: an active subspace with an empty cluster is flagged by the workgroup that sees it (done_blocks[1]; every
        // writer stores the same value) -- the last workgroup used to scan all m * k counts by itself: 85 us at m = 96
        if (RUN && t == 0 && act && !(cnt > 0.0)) done_blocks[1] = 1u;
        if (act && cnt > 0.0) {
            const float EPSILON = 1e-6f;  // vector.rs:439
            float nv;
            if (exact_div) nv = (float)row[t] / (float)cnt;  // row[t] holds an exact f32 value; vector.rs:373,382
            else nv = (float)(row[t] / cnt);
            const float diff = nv - centroids[e];
            // every writer stores the same value: no atomic needed (34816 contended atomicOr on m words
            // cost 370 us)
            if (!(fabsf(diff) < EPSILON)) changed[s] = 1u;  // vector.rs:232-240, 444-446
            centroids[e] = nv;
        }
    }
}

// Device-driven run on ONE rank (nothing is exchanged between the sums and the means): k_reduce_partials_pos and the
// arithmetic of k_finalize<true> in one launch.  A workgroup sums 32 centroid components over the row chunks exactly as
// k_reduce_partials_pos does (chunk groups strided, then groups ascending) -- and their clusters' counts beside them (every
// component's lanes re-add its cluster's count: a few hundred KB of L2 reads for not having to find it in another
// workgroup) -- and goes straight on to the mean, the |new - old| < 1e-6 test and the stores; the f64 slab is still
// written (vqhip_kmeans_partials).  `changed` of the iteration is gathered in a scratch word per subspace that k_run_decide
// copies out and clears: k_reduce_partials_pos cleared the flags in front of k_finalize, which in one launch would race
// with their setting.  No fence, no counter: the decisions wait behind the kernel boundary.
__global__ __launch_bounds__(256) void k_reduce_finalize_run(
    const float *__restrict__ partial_sums, const uint32_t *__restrict__ partial_counts, uint32_t n_chunks, uint32_t n_sub,
    const int32_t *__restrict__ sub_pos, uint32_t m, uint32_t k, uint32_t sd, double *__restrict__ slab, const uint8_t *__restrict__ active,
    float *__restrict__ centroids, uint32_t *__restrict__ counts, const uint32_t *__restrict__ gate_halt,
    uint32_t *__restrict__ done_blocks, uint32_t *__restrict__ chg_scratch) {
    __shared__ double part[kRedGroups][32];
    __shared__ unsigned long long partc[kRedGroups][32];
    if (*gate_halt) return;  // paused run: slab, centroids, counts and flags keep the pausing iteration's values
    const uint32_t total = m * k * sd;
    const uint32_t el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const uint32_t e = blockIdx.x * 32 + el;
    const bool in = e < total;
    double acc = 0.0;
    unsigned long long cnt = 0;
    uint32_t sj = 0, t = 0, s = 0;
    bool act = false;
    if (in) {
        sj = e / sd, t = e - sj * sd, s = sj / k;
        const uint32_t j = sj - s * k;
        const int32_t pos = sub_pos[s];
        act = active[s] != 0;  // a subspace that converged inside this run: slab, centroids stay as they are, counts read 0
        if (pos >= 0 && act) {
            const size_t stride = (size_t)n_sub * k * sd, cstride = (size_t)n_sub * k;
            const float *p = partial_sums + ((size_t)pos * k + j) * sd + t;
            const uint32_t *pc = partial_counts + (size_t)pos * k + j;
            for (uint32_t c0 = grp; c0 < n_chunks; c0 += 6 * kRedGroups) {  // six partials at a time, added in order (k_reduce_partials_pos)
                float v[6];
                uint32_t w[6];
#pragma unroll
                for (uint32_t u = 0; u < 6; ++u) {
                    const uint32_t c = min(c0 + u * kRedGroups, n_chunks - 1u);  // (clamped: nothing under a test)
                    v[u] = p[c * stride];
                    w[u] = pc[c * cstride];
                }
#pragma unroll
                for (uint32_t u = 0; u < 6; ++u)
                    if (c0 + u * kRedGroups < n_chunks) acc += (double)v[u], cnt += w[u];
            }
        }
    }
    part[grp][el] = acc;
    partc[grp][el] = cnt;
    __syncthreads();
    if (grp != 0 || !in) return;
    if (!act) {
        if (t == 0) counts[sj] = 0u;
        return;
    }
    double r = part[0][el];
    unsigned long long c = partc[0][el];
    for (uint32_t g = 1; g < kRedGroups; ++g) r += part[g][el], c += partc[g][el];
    const double cd = (double)c;
    slab[(size_t)sj * (sd + 1) + t] = r;
    if (t == 0) {
        slab[(size_t)sj * (sd + 1) + sd] = cd;
        counts[sj] = (uint32_t)cd;
        if (!(cd > 0.0)) done_blocks[1] = 1u;  // an active subspace with an empty cluster: k_run_decide pauses the run
    }
    if (cd > 0.0) {
        const float EPSILON = 1e-6f;  // vector.rs:439
        const float nv = (float)(r / cd);
        const float diff = nv - centroids[e];
        if (!(fabsf(diff) < EPSILON)) chg_scratch[s] = 1u;  // vector.rs:232-240, 444-446 (every writer stores the same value)
        centroids[e] = nv;
    }
}

// The end of one iteration of a device-driven run (src/core/vector.rs:440-457 without the host), behind k_finalize<true>: an
// active subspace with an empty cluster pauses the run (the caller reseeds: the draw is the host's); otherwise subspaces
// whose centroids did not move retire, the others count one more iteration.  A launch of its own: as the tail of the
// workgroup of k_finalize that finishes last it needed every wave's stores at agent scope first -- `__threadfence()` is
// `buffer_wbl2` + `buffer_inv` on this chip, an L2 write-back per wave -- plus a ticket on one word and the tail's cold
// loads: 20 us for the kernel where its arithmetic is one round trip; a kernel boundary orders the same stores for ~5.
__global__ __launch_bounds__(256) void k_run_decide(uint32_t m, uint8_t *__restrict__ active, uint32_t *__restrict__ changed,
                                                    uint32_t *__restrict__ gate_halt, uint32_t *__restrict__ iters,
                                                    uint32_t *__restrict__ done_blocks, uint32_t *__restrict__ chg_scratch) {
    // chg_scratch (behind k_reduce_finalize_run): the iteration's flags are there; they become `changed` here
    __shared__ int any_empty, halted;
    if (threadIdx.x == 0) {
        halted = *gate_halt != 0u;
        any_empty = done_blocks[1] != 0u;
        done_blocks[1] = 0u;  // ready for the next iteration
    }
    __syncthreads();
    if (halted) return;  // paused earlier: everything keeps the pausing iteration's values
    for (uint32_t s = threadIdx.x; s < m; s += 256) {
        uint32_t chg = changed[s];
        if (chg_scratch) {
            chg = chg_scratch[s];
            chg_scratch[s] = 0u;
            changed[s] = chg;
        }
        if (!active[s]) continue;
        iters[s] += 1u;
        if (!any_empty && !chg) active[s] = 0;  // converged (vector.rs:455-457); on a pause the host decides
    }
    if (threadIdx.x == 0 && any_empty) *gate_halt = 1u;
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

// sharded form: `rows` are GLOBAL ids; a rank contributes the bits of the rows it owns
// ([row_offset, row_offset + n_local)) and zero words for the rest, so that a u32 sum over the ranks
// carries every value bit for bit (a float sum would turn -0.0 into +0.0)
__global__ __launch_bounds__(256) void k_gather_rows_owned(const float *__restrict__ X, uint32_t d, uint32_t m,
                                                           uint32_t k, uint32_t sd, const uint64_t *__restrict__ rows,
                                                           uint64_t row_offset, uint64_t n_local,
                                                           uint32_t *__restrict__ out_bits) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * k * sd) return;
    const uint32_t t = e % sd, sj = e / sd, s = sj / k;
    const uint64_t r = rows[sj];
    const bool mine = r >= row_offset && r - row_offset < n_local;
    out_bits[e] = mine ? __float_as_uint(X[(r - row_offset) * d + (size_t)s * sd + t]) : 0u;
}

}  // namespace

// sub_dims with a wave-owned instantiation (KS = sd/4 lanes per row)
static bool owned_sub_dim(uint32_t sd) {
    switch (sd) {
    case 4: case 8: case 12: case 16: case 24: case 32: case 48: case 64: case 96: case 128: case 256: return true;
    case 20: case 28: case 36: case 40: case 44: case 52: case 56: case 60: return true;  // 16-byte parts
    case 26: case 30: return true;                                                         // 8-byte parts
    default: return sd >= 5 && sd <= 23;                                                   // 8-byte / 4-byte parts
    }
}

int plan_update(uint32_t m, uint32_t k, uint32_t sd, uint64_t n, UpdatePlan *p) {
    p->m = m;
    p->k = k;
    p->sd = sd;
    if (sd > 1024) return fail(VQHIP_ERR_UNSUPPORTED, "sub_dim=%u > 1024 in the LDS update kernel", sd);
    const uint64_t per_sub = (uint64_t)k * (sd + 2);  // padded sums + counts
    uint32_t spc = 1;
    p->k_range = k;
    if (per_sub > kLdsBudgetWords) {
        // a subspace's accumulators do not fit one CU's LDS: split its clusters into ranges, one workgroup
        // (and one more pass over the rows) per range
        p->k_range = kLdsBudgetWords / (sd + 2);
    } else {
        spc = (uint32_t)(kLdsBudgetWords / per_sub);
        if (spc > m) spc = m;
        while (spc > 1 && (uint64_t)spc * sd > 1024) --spc;  // one workgroup pass covers >= 1 row
    }
    p->n_k_ranges = (k + p->k_range - 1) / p->k_range;
    p->subs_per_chunk = spc;
    p->n_sub_chunks = (m + spc - 1) / spc;
    uint32_t target = (uint32_t)num_cus();
    uint32_t rc = target / (p->n_sub_chunks * p->n_k_ranges);
    if (rc < 1) rc = 1;
    // at least ~512 rows per chunk so that the slab write-out stays a small fraction
    uint64_t max_rc = (n + 511) / 512;
    if (max_rc < 1) max_rc = 1;
    if (rc > max_rc) rc = (uint32_t)max_rc;
    p->n_row_chunks = rc;
    // wave-owned path: sub_dim with an instantiation (owned_sub_dim); as many waves (= subspaces) per
    // workgroup as fit the LDS budget, at most 8
    p->owned_waves = 0;
    if (owned_sub_dim(sd)) {
        uint64_t per_wave = ((uint64_t)k * (sd + 1) + 3) & ~3ull;
        uint32_t w = (uint32_t)(kLdsBudgetWords / per_wave);
        if (w > 8) w = 8;
        const uint32_t w_fit = w;  // waves whose accumulators fit (<= 8)
        if (w > m) w = m;
        if (w > 0) w = (m + ((m + w - 1) / w) - 1) / ((m + w - 1) / w);  // same number of workgroups, evenly filled (m = 10: 5 + 5, not 8 + 2)
        // few subspaces: the other waves of the workgroup take further parts of the rows (own slab each)
        uint32_t split = (w > 0) ? w_fit / w : 1u;
        if (split < 1) split = 1;
        if (w * split < 2) w = 0;  // accumulators filling the LDS: one wave per workgroup, the atomic kernel's 16 waves win
        p->owned_row_split = split;
        p->owned_waves = w * split;
        if (w > 0) {
            uint32_t sub_groups = (m + w - 1) / w;
            // small accumulators (short sub-vectors, few clusters) leave LDS for 2-4 workgroups per CU: more row
            // chunks, so that the loads of several workgroups overlap
            uint32_t per_cu = (uint32_t)(kLdsBudgetWords / (per_wave * w * split));
            per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
            uint32_t rc2 = (uint32_t)num_cus() * per_cu / sub_groups;
            if (rc2 < 1) rc2 = 1;
            const uint64_t max_rc2 = (max_rc + split - 1) / split;  // ~512 rows per wave, not per workgroup
            if (rc2 > max_rc2) rc2 = (uint32_t)max_rc2;
            if (rc2 < 1) rc2 = 1;
            p->n_row_chunks = rc2 * split;  // slabs
        }
    }
    p->partial_floats = (size_t)m * k * sd;
    p->partial_counts = (size_t)m * k;
    return VQHIP_OK;
}

template <int KS, int VW = 4>
static int launch_owned(const UpdatePlan &p, const float *X, uint64_t n, uint32_t d,
                        const uint8_t *codes, const uint8_t *active, float *partial_sums,
                        uint32_t *partial_counts, uint32_t wpb, hipStream_t stream) {
    static PerDeviceOnce attr_set;
    if (attr_set.needed()) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate_owned<KS, VW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    const uint32_t split = p.owned_row_split, subs = wpb / split, chunks = p.n_row_chunks / split;  // n_row_chunks counts slabs
    const uint64_t rows_per_chunk = (n + chunks - 1) / chunks;
    const size_t lds_bytes = (size_t)wpb * ((p.k * (p.sd + 1) + 3u) & ~3u) * 4;
    dim3 grid(chunks, (p.m + subs - 1) / subs);
    hipLaunchKernelGGL((k_accumulate_owned<KS, VW>), grid, dim3(wpb * 64), lds_bytes, stream, X, n, d, p.m,
                       p.k, wpb, rows_per_chunk, codes, active, partial_sums, partial_counts, split);
    VQ_LAUNCH_CHECK("k_accumulate_owned");
    return VQHIP_OK;
}

int launch_accumulate(const UpdatePlan &p, const float *X, uint64_t n, uint32_t d,
                      const uint8_t *codes, const uint8_t *active, float *partial_sums,
                      uint32_t *partial_counts, hipStream_t stream) {
    const bool aligned = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    // narrower parts for the sub_dims of the zero-padded screen: 8-byte parts need d even (it is: m * even) and the
    // base 8-byte aligned, single floats need nothing
    if (p.owned_waves > 0 && (reinterpret_cast<uintptr_t>(X) & 7) == 0) {
#define VQ_OWNED(SDV, KSV, VWV) \
    case SDV: return launch_owned<KSV, VWV>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        switch (p.sd) {
            VQ_OWNED(6, 3, 2) VQ_OWNED(10, 5, 2) VQ_OWNED(14, 7, 2) VQ_OWNED(18, 9, 2) VQ_OWNED(22, 11, 2)
            VQ_OWNED(5, 5, 1) VQ_OWNED(7, 7, 1) VQ_OWNED(9, 9, 1) VQ_OWNED(11, 11, 1) VQ_OWNED(13, 13, 1)
            VQ_OWNED(15, 15, 1) VQ_OWNED(17, 17, 1) VQ_OWNED(19, 19, 1) VQ_OWNED(21, 21, 1) VQ_OWNED(23, 23, 1)
            VQ_OWNED(26, 13, 2) VQ_OWNED(30, 15, 2)  // wider rows leave < 4 rows per wave step: the atomic kernel is faster there
        default: break;
        }
#undef VQ_OWNED
    }
    // wave-owned accumulators: sub_dim a power of two in [4, 256], one subspace's [k][sd+1]
    // words per wave
    if (aligned && p.owned_waves > 0) {
        switch (p.sd) {
        case 4: return launch_owned<1>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 8: return launch_owned<2>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 12: return launch_owned<3>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 16: return launch_owned<4>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 20: return launch_owned<5>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 24: return launch_owned<6>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 28: return launch_owned<7>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 36: return launch_owned<9>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 40: return launch_owned<10>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 44: return launch_owned<11>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 52: return launch_owned<13>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 56: return launch_owned<14>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 60: return launch_owned<15>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 48: return launch_owned<12>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 96: return launch_owned<24>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 32: return launch_owned<8>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 64: return launch_owned<16>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 128: return launch_owned<32>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        case 256: return launch_owned<64>(p, X, n, d, codes, active, partial_sums, partial_counts, p.owned_waves, stream);
        default: break;
        }
    }
    const uint64_t rows_per_chunk = (n + p.n_row_chunks - 1) / p.n_row_chunks;
    if (rows_per_chunk * (uint64_t)p.subs_per_chunk * p.sd >= (1ull << 32))
        return fail(VQHIP_ERR_UNSUPPORTED, "row chunk too large for 32-bit item index");
    const size_t lds_bytes = ((size_t)p.subs_per_chunk * p.k_range * (p.sd + 2)) * 4;
    dim3 grid(p.n_row_chunks, p.n_sub_chunks, p.n_k_ranges);
    const bool vec4 = (p.sd % 4 == 0) && aligned;
    if (vec4) {
        static PerDeviceOnce attr_set4;
        if (attr_set4.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set4.done();
        }
        hipLaunchKernelGGL(k_accumulate<4>, grid, dim3(kAccBlock), lds_bytes, stream, X, n, d, p.m,
                           p.k, p.sd, p.subs_per_chunk, p.k_range, rows_per_chunk, codes, active, partial_sums,
                           partial_counts);
    } else {
        static PerDeviceOnce attr_set1;
        if (attr_set1.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set1.done();
        }
        hipLaunchKernelGGL(k_accumulate<1>, grid, dim3(kAccBlock), lds_bytes, stream, X, n, d, p.m,
                           p.k, p.sd, p.subs_per_chunk, p.k_range, rows_per_chunk, codes, active, partial_sums,
                           partial_counts);
    }
    VQ_LAUNCH_CHECK("k_accumulate");
    return VQHIP_OK;
}

int launch_reduce_partials(const UpdatePlan &p, const float *partial_sums,
                           const uint32_t *partial_counts, const uint8_t *active, double *slab,
                           hipStream_t stream) {
    const uint32_t total = p.m * p.k * (p.sd + 1);
    hipLaunchKernelGGL(k_reduce_partials, dim3((total + 31) / 32), dim3(256), 0, stream,
                       partial_sums, partial_counts, p.n_row_chunks, p.m, p.k, p.sd, active, slab);
    VQ_LAUNCH_CHECK("k_reduce_partials");
    return VQHIP_OK;
}

int launch_reduce_partials_pos(uint32_t m, uint32_t k, uint32_t sd, const float *partial_sums, const uint32_t *partial_counts,
                               uint32_t n_chunks, uint32_t n_sub, const int32_t *sub_pos, double *slab, hipStream_t stream,
                               const uint8_t *gate_active, const uint32_t *gate_halt, uint32_t *clear_changed) {
    const uint32_t total = m * k * (sd + 1);
    hipLaunchKernelGGL(k_reduce_partials_pos, dim3((total + 31) / 32), dim3(256), 0, stream, partial_sums, partial_counts,
                       n_chunks, n_sub, sub_pos, m, k, sd, slab, gate_active, gate_halt, clear_changed);
    VQ_LAUNCH_CHECK("k_reduce_partials_pos");
    return VQHIP_OK;
}

int launch_finalize(uint32_t m, uint32_t k, uint32_t sd, const double *slab, const uint8_t *active,
                    float *centroids, uint32_t *counts, uint32_t *changed, int exact_div,
                    hipStream_t stream) {
    VQ_HIP(hipMemsetAsync(changed, 0, (size_t)m * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_finalize<false>, dim3((m * k * sd + 255) / 256), dim3(256), 0, stream, m, k, sd, slab,
                       const_cast<uint8_t *>(active), centroids, counts, changed, exact_div, (uint32_t *)nullptr, (uint32_t *)nullptr,
                       (uint32_t *)nullptr);
    VQ_LAUNCH_CHECK("k_finalize");
    return VQHIP_OK;
}

// finalize + the end of one iteration of a device-driven run (`changed` was cleared by the gated k_reduce_partials_pos;
// run_state = {halt flag, iterations executed [m]}; done_blocks: a zeroed counter the kernel leaves zeroed)
int launch_finalize_run(uint32_t m, uint32_t k, uint32_t sd, const double *slab, uint8_t *active, float *centroids,
                        uint32_t *counts, uint32_t *changed, uint32_t *halt, uint32_t *iters, uint32_t *done_blocks,
                        hipStream_t stream) {
    hipLaunchKernelGGL(k_finalize<true>, dim3((m * k * sd + 255) / 256), dim3(256), 0, stream, m, k, sd, slab, active, centroids,
                       counts, changed, 0, halt, iters, done_blocks);
    hipLaunchKernelGGL(k_run_decide, dim3(1), dim3(256), 0, stream, m, active, changed, halt, iters, done_blocks, (uint32_t *)nullptr);
    VQ_LAUNCH_CHECK("k_finalize<run>");
    return VQHIP_OK;
}

// k_reduce_partials_pos + k_finalize<true> + k_run_decide of a one-rank device-driven run in two launches; chg_scratch: m zeroed
// words the decision kernel leaves zeroed
int launch_reduce_finalize_run(uint32_t m, uint32_t k, uint32_t sd, const float *partial_sums, const uint32_t *partial_counts,
                               uint32_t n_chunks, uint32_t n_sub, const int32_t *sub_pos, double *slab, uint8_t *active, float *centroids,
                               uint32_t *counts, uint32_t *changed, uint32_t *halt, uint32_t *iters, uint32_t *done_blocks,
                               uint32_t *chg_scratch, hipStream_t stream) {
    const uint32_t total = m * k * sd;
    hipLaunchKernelGGL(k_reduce_finalize_run, dim3((total + 31) / 32), dim3(256), 0, stream, partial_sums, partial_counts, n_chunks, n_sub,
                       sub_pos, m, k, sd, slab, active, centroids, counts, halt, done_blocks, chg_scratch);
    hipLaunchKernelGGL(k_run_decide, dim3(1), dim3(256), 0, stream, m, active, changed, halt, iters, done_blocks, chg_scratch);
    VQ_LAUNCH_CHECK("k_reduce_finalize_run");
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

int launch_accumulate_listed(uint32_t m, uint32_t k, uint32_t sd, const float *X, uint32_t d, const uint8_t *codes,
                             const uint32_t *sub_list, uint32_t n_sub, const uint32_t *wl_rows, uint64_t wl_stride,
                             const uint32_t *wl_seg, uint32_t n_seg, uint32_t first_chunk, uint32_t n_patch,
                             float *partial_sums, uint32_t *partial_counts, hipStream_t stream) {
    if (n_sub == 0 || n_patch == 0) return VQHIP_OK;
    const uint32_t spp = (n_seg + n_patch - 1) / n_patch;
    const size_t lds = (size_t)k * (sd + 1) * 4;
    const dim3 grid(n_patch, n_sub);
#define VQ_LISTED(KSV)                                                                                                  \
    hipLaunchKernelGGL(k_accumulate_listed<KSV>, grid, dim3(64), lds, stream, X, d, m, k, codes, sub_list, wl_rows, wl_stride, \
                       wl_seg, n_seg, spp, first_chunk, partial_sums, partial_counts)
    switch (sd) {
    case 8: VQ_LISTED(2); break;
    case 16: VQ_LISTED(4); break;
    case 24: VQ_LISTED(6); break;
    default: return fail(VQHIP_ERR_UNSUPPORTED, "listed accumulate: sub_dim=%u", sd);
    }
#undef VQ_LISTED
    VQ_LAUNCH_CHECK("k_accumulate_listed");
    return VQHIP_OK;
}

int launch_gather_rows_owned(const float *X, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const uint64_t *rows,
                             uint64_t row_offset, uint64_t n_local, uint32_t *out_bits, hipStream_t stream) {
    const uint32_t total = m * k * sd;
    hipLaunchKernelGGL(k_gather_rows_owned, dim3((total + 255) / 256), dim3(256), 0, stream, X, d, m, k, sd, rows,
                       row_offset, n_local, out_bits);
    VQ_LAUNCH_CHECK("k_gather_rows_owned");
    return VQHIP_OK;
}

// ---- exact (reference-order) cluster sums -------------------------------------------------
// mean_vector_by_indices (src/core/vector.rs:368-384) adds a cluster's member rows in ascending
// row order, one rounding per add.  To reproduce those bits the rows are first bucketed by
// code with a STABLE counting sort (per-chunk histograms -> offsets -> in-order scatter, the
// GPU form of `cluster_indices[c].push(i)`, vector.rs:432-435), then one lane per
// (subspace, cluster, dimension) walks its member list and adds sequentially.  The chain is
// latency-bound (N/k dependent adds per lane, 32 row gathers in flight), ~0.5 ms at C2.
namespace {

constexpr uint32_t kXsChunks = 1024;     // row chunks of the bucket passes
constexpr uint32_t kXsMaxK = 16384;      // hist / cursor words of one workgroup (64 KiB of LDS)
// fewer chunks for large m*k so that the [chunks][m][k] offset table stays <= 256 MiB
static uint32_t xs_chunk_cap(uint32_t m, uint32_t k) {
    const uint64_t cap = (64ull << 20) / std::max<uint64_t>(1, (uint64_t)m * k);
    return (uint32_t)std::min<uint64_t>(kXsChunks, std::max<uint64_t>(1, cap));
}

__global__ __launch_bounds__(256) void k_chunk_counts(const uint8_t *__restrict__ codes, uint64_t n,
                                                      uint32_t m, uint32_t k, uint64_t rows_per_chunk,
                                                      const uint8_t *__restrict__ active,
                                                      uint32_t *__restrict__ chunk_counts) {
    extern __shared__ uint32_t hist[];  // [k]
    const uint32_t s = blockIdx.y;
    for (uint32_t j = threadIdx.x; j < k; j += 256) hist[j] = 0;
    __syncthreads();
    if (!active || active[s]) {
        const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_chunk;
        uint64_t r1 = r0 + rows_per_chunk;
        if (r1 > n) r1 = n;
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += 256) atomicAdd(&hist[load_code(codes, r * m + s, k)], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < k; j += 256) chunk_counts[((size_t)blockIdx.x * m + s) * k + j] = hist[j];
}

// chunk_counts [chunks][m][k] -> in place: write offset of (chunk, s, j) inside members[s][..];
// start [m][k+1]: first member of cluster j (start[s][k] = rows of the subspace)
// Round 6: two kernels over (subspace, 32 clusters) workgroups instead of one workgroup per subspace whose threads each
// walked all n_chunks entries of their cluster twice, one after the other (8 workgroups on the chip at C2: 263 us of a
// 1.33 ms exact_update iteration).  A thread owns (cluster, one of 8 blocks of chunks): the walks are an eighth as long,
// 64 workgroups run them, and a wave's 32 + 32 lanes read two contiguous 128-byte runs per chunk.
constexpr uint32_t kBoClusters = 32, kBoBlocks = 8;  // 256 threads

// totals[s][j] = rows of cluster j; block sums [s][j][cb] for the second kernel
__global__ __launch_bounds__(256) void k_bucket_totals(const uint32_t *__restrict__ chunk_counts, uint32_t n_chunks, uint32_t m, uint32_t k,
                                                       uint32_t *__restrict__ totals, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t part[kBoBlocks][kBoClusters];
    const uint32_t s = blockIdx.x, jl = threadIdx.x % kBoClusters, cb = threadIdx.x / kBoClusters;
    const uint32_t j = blockIdx.y * kBoClusters + jl;
    const uint32_t per = (n_chunks + kBoBlocks - 1) / kBoBlocks, c0 = cb * per, c1 = min(n_chunks, c0 + per);
    uint32_t sum = 0;
    if (j < k)
        for (uint32_t c = c0; c < c1; ++c) sum += chunk_counts[((size_t)c * m + s) * k + j];
    part[cb][jl] = sum;
    __syncthreads();
    if (j < k) block_sums[((size_t)s * k + j) * kBoBlocks + cb] = sum;
    if (cb == 0 && j < k) {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < kBoBlocks; ++q) tot += part[q][jl];
        totals[(size_t)s * k + j] = tot;
    }
}

__global__ __launch_bounds__(256) void k_bucket_offsets(uint32_t *__restrict__ chunk_counts, uint32_t n_chunks,
                                                        uint32_t m, uint32_t k, uint32_t *__restrict__ start,
                                                        const uint32_t *__restrict__ totals, const uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t seg[256];
    __shared__ uint32_t first_of[kBoClusters];
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    // first member of this workgroup's clusters: the clusters in front of them (every thread adds a strided share), then the
    // 32 of its own in order
    {
        const uint32_t jb = blockIdx.y * kBoClusters;
        uint32_t acc = 0;
        for (uint32_t j = tid; j < jb; j += 256) acc += totals[(size_t)s * k + j];
        seg[tid] = acc;
        __syncthreads();
        for (uint32_t off = 128; off > 0; off >>= 1) {
            if (tid < off) seg[tid] += seg[tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            uint32_t run = seg[0];
            for (uint32_t q = 0; q < kBoClusters; ++q) {
                first_of[q] = run;
                if (jb + q < k) run += totals[(size_t)s * k + jb + q];
            }
            if (jb + kBoClusters >= k) start[s * (k + 1) + k] = run;  // (the last workgroup of the subspace: all rows)
        }
        __syncthreads();
    }
    const uint32_t jl = tid % kBoClusters, cb = tid / kBoClusters, j = blockIdx.y * kBoClusters + jl;
    if (j >= k) return;
    uint32_t run = first_of[jl];
    if (cb == 0) start[s * (k + 1) + j] = run;
    for (uint32_t q = 0; q < cb; ++q) run += block_sums[((size_t)s * k + j) * kBoBlocks + q];
    const uint32_t per = (n_chunks + kBoBlocks - 1) / kBoBlocks, c0 = cb * per, c1 = min(n_chunks, c0 + per);
    for (uint32_t c = c0; c < c1; ++c) {
        uint32_t *p = &chunk_counts[((size_t)c * m + s) * k + j];
        const uint32_t cnt = *p;
        *p = run;
        run += cnt;
    }
}

// in-order scatter of row ids: one wave per (chunk, subspace); 64 consecutive rows per step
__global__ __launch_bounds__(64) void k_bucket_scatter(const uint8_t *__restrict__ codes, uint64_t n, uint32_t m,
                                                       uint32_t k, uint64_t rows_per_chunk,
                                                       const uint8_t *__restrict__ active,
                                                       const uint32_t *__restrict__ offsets,
                                                       uint32_t *__restrict__ members, uint64_t members_stride) {
    extern __shared__ uint32_t cur[];  // [k]
    const uint32_t s = blockIdx.y, lane = threadIdx.x;
    if (active && !active[s]) return;
    for (uint32_t j = lane; j < k; j += 64) cur[j] = offsets[((size_t)blockIdx.x * m + s) * k + j];
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_chunk;
    uint64_t r1 = r0 + rows_per_chunk;
    if (r1 > n) r1 = n;
    uint32_t *dst = members + (size_t)s * members_stride;
    if (r0 >= r1) return;
    // The lanes that hold the same code as this one, bit by bit of the code (8 ballots at k = 256; comparing against every
    // lane through v_readlane was ~250 instructions per step), and the next step's codes requested before this step's are
    // used (a step was one exposed memory round trip): 155 -> 60 us at C2.
    const uint32_t nbits = (k > 1) ? 32u - (uint32_t)__builtin_clz(k - 1u) : 0u;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t code_next = load_code(codes, min(r0 + lane, r1 - 1) * m + s, k);
    for (uint64_t base = r0; base < r1; base += 64) {
        const uint64_t row = base + lane;
        const bool valid = row < r1;
        const uint32_t code = code_next;
        code_next = load_code(codes, min(row + 64, r1 - 1) * m + s, k);  // (clamped, never under a test)
        uint64_t same = __ballot(valid);
        for (uint32_t b = 0; b < nbits; ++b) {
            const bool bit = ((code >> b) & 1u) != 0u;
            const uint64_t mb = __ballot(bit);
            same &= bit ? mb : ~mb;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below);
        const bool later = ((same >> lane) >> 1) != 0ull;
        if (valid) {
            const uint32_t pos = cur[code] + rank;
            dst[pos] = (uint32_t)row;
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && !later) cur[code] += rank + 1;  // last row of its code in this step
        __builtin_amdgcn_wave_barrier();
    }
}

// one lane per (s, j, t): sequential f32 sum of the members in ascending row order.  The chain itself is ~20 K cycles per
// cluster (3900 dependent adds at C2); the kernel's time is the gathers' latency over what is in flight (512 waves on the
// whole chip).  Round 6: batches of 64 members (32: 2 % slower), the row ids two batches ahead and the gathers one batch ahead of the
// additions, every load unconditional with clamped indices (a load under a per-lane test is followed by vmcnt(0)):
// 563 -> ~200 us at C2.
// (workgroups of ONE wave: 512 of them at C2 reach every CU's load path; 128 workgroups of four left half the chip idle)
__global__ __launch_bounds__(64) void k_chain_sums(const float *__restrict__ X, uint32_t d, uint32_t m, uint32_t k,
                                                   uint32_t sd, const uint8_t *__restrict__ active,
                                                   const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ members, uint64_t members_stride,
                                                   double *__restrict__ slab) {
    const uint32_t e = blockIdx.x * 64 + threadIdx.x;
    if (e >= m * k * sd) return;
    const uint32_t sj = e / sd, t = e - sj * sd, s = sj / k, j = sj - s * k;
    double *row = slab + (size_t)sj * (sd + 1);
    if (active && !active[s]) {
        row[t] = 0.0;
        if (t == 0) row[sd] = 0.0;
        return;
    }
    const uint32_t a = start[s * (k + 1) + j], b = start[s * (k + 1) + j + 1];
    const uint32_t *mem = members + (size_t)s * members_stride;
    const float *px = X + (size_t)s * sd + t;
    float acc = 0.0f;  // vector.rs:374
    constexpr int U = 64;
    const uint32_t last = (b > a) ? b - 1 : a;  // (clamp target: a valid member slot whenever the cluster has one)
    if (b > a) {
        uint32_t ids[2][U];
        float v[2][U];
#pragma unroll
        for (int u = 0; u < U; ++u) ids[0][u] = mem[min(a + (uint32_t)u, last)];
#pragma unroll
        for (int u = 0; u < U; ++u) ids[1][u] = mem[min(a + (uint32_t)(U + u), last)];
#pragma unroll
        for (int u = 0; u < U; ++u) v[0][u] = px[(size_t)ids[0][u] * d];
        for (uint32_t i = a; i < b; i += 2 * U) {
            // batch [i, i + U): its values are in flight in v[0]; request batch i + U (ids in ids[1]), the ids of i + 2U, add v[0]
#pragma unroll
            for (int u = 0; u < U; ++u) v[1][u] = px[(size_t)ids[1][u] * d];
#pragma unroll
            for (int u = 0; u < U; ++u) ids[0][u] = mem[min(i + (uint32_t)(2 * U + u), last)];
#pragma unroll
            for (int u = 0; u < U; ++u) acc = (i + (uint32_t)u < b) ? acc + v[0][u] : acc;
            // batch [i + U, i + 2U): in v[1]; request batch i + 2U (ids[0]), the ids of i + 3U, add v[1]
#pragma unroll
            for (int u = 0; u < U; ++u) v[0][u] = px[(size_t)ids[0][u] * d];
#pragma unroll
            for (int u = 0; u < U; ++u) ids[1][u] = mem[min(i + (uint32_t)(3 * U + u), last)];
#pragma unroll
            for (int u = 0; u < U; ++u) acc = (i + (uint32_t)(U + u) < b) ? acc + v[1][u] : acc;
        }
    }
    row[t] = (double)acc;
    if (t == 0) row[sd] = (double)(b - a);
}

}  // namespace

size_t exact_sums_workspace_bytes(uint32_t m, uint32_t k, uint64_t n) {
    // members [m][n] | chunk offsets [chunks][m][k] | start [m][k + 1] | totals [m][k] | block sums [m][k][8]
    return ((size_t)m * n + (size_t)xs_chunk_cap(m, k) * m * k + (size_t)m * (k + 1) + (size_t)m * k * (1 + kBoBlocks)) * 4 + 256;
}

int launch_exact_sums(uint32_t m, uint32_t k, uint32_t sd, const float *X, uint64_t n, uint32_t d,
                      const uint8_t *codes, const uint8_t *active, void *workspace, size_t workspace_bytes,
                      double *slab, hipStream_t stream) {
    if (k > kXsMaxK) return fail(VQHIP_ERR_UNSUPPORTED, "exact update needs k <= %u", kXsMaxK);
    if (workspace_bytes < exact_sums_workspace_bytes(m, k, n)) return fail(VQHIP_ERR_FAILURE, "exact-update workspace too small");
    uint32_t *members = reinterpret_cast<uint32_t *>(workspace);
    uint32_t *chunk_counts = members + (size_t)m * n;
    uint32_t *start = chunk_counts + (size_t)xs_chunk_cap(m, k) * m * k;
    uint32_t n_chunks = (uint32_t)std::min<uint64_t>(xs_chunk_cap(m, k), (n + 255) / 256);
    if (n_chunks < 1) n_chunks = 1;
    const uint64_t rows_per_chunk = (n + n_chunks - 1) / n_chunks;
    hipLaunchKernelGGL(k_chunk_counts, dim3(n_chunks, m), dim3(256), (size_t)k * 4, stream, codes, n, m, k, rows_per_chunk, active,
                       chunk_counts);
    VQ_LAUNCH_CHECK("k_chunk_counts");
    uint32_t *totals = start + (size_t)m * (k + 1), *block_sums = totals + (size_t)m * k;
    const dim3 bo_grid(m, (k + kBoClusters - 1) / kBoClusters);
    hipLaunchKernelGGL(k_bucket_totals, bo_grid, dim3(256), 0, stream, chunk_counts, n_chunks, m, k, totals, block_sums);
    hipLaunchKernelGGL(k_bucket_offsets, bo_grid, dim3(256), 0, stream, chunk_counts, n_chunks, m, k, start, totals, block_sums);
    VQ_LAUNCH_CHECK("k_bucket_offsets");
    hipLaunchKernelGGL(k_bucket_scatter, dim3(n_chunks, m), dim3(64), (size_t)k * 4, stream, codes, n, m, k, rows_per_chunk, active,
                       chunk_counts, members, n);
    VQ_LAUNCH_CHECK("k_bucket_scatter");
    const uint32_t total = m * k * sd;
    hipLaunchKernelGGL(k_chain_sums, dim3((total + 63) / 64), dim3(64), 0, stream, X, d, m, k, sd, active, start,
                       members, n, slab);
    VQ_LAUNCH_CHECK("k_chain_sums");
    return VQHIP_OK;
}

}  // namespace vqhip
