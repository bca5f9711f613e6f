// k_lloyd_small.hip -- one Lloyd iteration of a SMALL problem in two launches (gfx950).
//
// The reference's own sizes are small (tests/integration_tests.rs:367-382, pyvq/tests/test_integrations.py:175-197:
// a few thousand rows, k <= 16; BASELINE configs[0]: 10k x 64, m = 4, k = 16).  On the general path such an iteration
// is seven dependent launches -- centre and pack the codebook for the bf16 screen, screen + fused update, exact
// re-check, list-driven update, f64 reduction, finalize -- each paying ~4.8 us of launch and drain for microseconds of
// work: 47 us per iteration at C1, 0.3 % of any roofline.  Here lbg_quantize's loop body (src/core/vector.rs:415-458) is
//   k_sm_assign : a workgroup owns 64 consecutive rows of one subspace -- rows and that subspace's codebook staged in LDS,
//                 four lanes per row scan the k centroids in the reference's arithmetic (sub, mul, add per dimension,
//                 sequential; strict `<`, first minimum wins: vector.rs:135-143, 352-363): no screen, no re-check, the
//                 codes are the reference's by construction; then lane (j, t) walks the 64 rows IN ROW ORDER and adds the
//                 members of cluster j (f32, sequential): a partial is the reference's sum of that row range;
//   k_sm_reduce : 16 lanes per (cluster, dimension) combine the partials of all row ranges in f64 in a fixed order, form
//                 the means and the `|new - old| < 1e-6` test (vector.rs:232-240, 439-446) -- deterministic, and within
//                 the reference's own rounding error of its n-term f32 sum.  SLAB form
//                 (vqhip_kmeans_accumulate, row-sharded training): the f64 sums and counts go to the slab the ranks
//                 all-reduce and k_finalize divides -- the same f64 values divided the same way: the same bits.
// A device-driven run (vqhip_kmeans_run) keeps the loop's decisions in per-subspace flags the NEXT iteration's kernels
// read (below).  One launch with a last-workgroup tail was measured first: 31 us per iteration (release fence + ticket
// 6-9 us, the tail's five cold batches of loads 9-19 us in the one workgroup everybody waits for) against the two
// launches here; both against 47 us.  What is left is latency: two launches (~4.8 us each) and four dependent memory
// round trips at an otherwise idle chip's clocks (2-3 us each).
// Shapes: k * sub_dim <= 1024, sub_dim <= 32, n <= 32768, n * m <= 2^20; everything else keeps the general path.
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr uint32_t kSmRows = 64;      // rows per workgroup of k_sm_assign
constexpr uint32_t kSmThreads = 256;  // four lanes per row in the assignment, one lane per (cluster, dimension) in the sums
constexpr uint32_t kSmLanes = 16;     // lanes per element in k_sm_reduce
constexpr uint32_t kSmElems = kSmThreads / kSmLanes;  // elements per workgroup of k_sm_reduce

// The state of a device-driven run, per subspace, in two sets used alternately (set i % 2 is written by iteration i's
// k_sm_reduce and read by iteration i + 1's kernels, which clear the other set for their own reduce):
//   moved[s] : some centroid of s moved (vector.rs:444-446) -- a subspace that executed and did not move retires
//   empty[s] : s executed and has a cluster without members -- the run pauses (every later launch is a no-op; the
//              reseed row is the caller's draw, vector.rs:448-452), nobody retires at that iteration
//   ran[s]   : s executed iteration i
struct SmFlags {
    uint32_t *moved, *empty, *ran;
};
__device__ __forceinline__ SmFlags sm_flags(uint32_t *base, uint32_t m, uint32_t set) {
    SmFlags f;
    f.moved = base + (size_t)set * 3 * m;
    f.empty = f.moved + m;
    f.ran = f.empty + m;
    return f;
}
// does subspace s execute this iteration, and has the run paused?  (wave-uniform; `it` = iterations queued before this one)
__device__ __forceinline__ bool sm_run_state(const uint8_t *__restrict__ active, uint32_t *flags, uint32_t m, uint32_t s, uint32_t it,
                                             bool *halted) {
    bool act = active[s] != 0;
    *halted = false;
    if (it > 0) {
        const SmFlags prev = sm_flags(flags, m, (it - 1) & 1u);
        uint32_t any = 0;
        for (uint32_t q = 0; q < m; ++q) any |= prev.empty[q];
        *halted = any != 0u;
        act = act && prev.ran[s] && prev.moved[s];
    }
    return act;
}

// SDP: sub_dim rounded up to 4, 8, 16 or 32 -- a row's share lives in registers and the loops over a sub-vector are
// unrolled; elements past sub_dim are never added.  RUN: inside vqhip_kmeans_run (flags as above); gate_halt: the slab
// form inside a device-driven sharded run (the general path's halt word).
template <bool RUN, int SDP>
__global__ __launch_bounds__(kSmThreads) void k_sm_assign(const float *__restrict__ X, uint32_t n, uint32_t d, uint32_t m, uint32_t k,
                                                          uint32_t sd, const float *__restrict__ cb, uint8_t *__restrict__ codes,
                                                          float *__restrict__ psum, uint32_t *__restrict__ pcnt,
                                                          const uint8_t *__restrict__ active, uint32_t *__restrict__ flags, uint32_t it,
                                                          const uint32_t *__restrict__ gate_halt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // lds: codebook [k][sd], rows [64][sd + 1], codes [64] (u32)
    float *cbs = lds;
    float *rows = cbs + k * sd;
    uint32_t *codes_l = reinterpret_cast<uint32_t *>(rows + kSmRows * (sd + 1));
    const uint32_t chunk = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const uint32_t r0 = chunk * kSmRows, rows_here = min(kSmRows, n - r0), ksd = k * sd, pitch = sd + 1;
    bool act, halted = false;
    if (RUN) {
        act = sm_run_state(active, flags, m, s, it, &halted);
        if (halted) {
            // a paused run: codes, centroids, counts and the pausing iteration's flag set keep their values; the pause is
            // handed on to this iteration's set (only `empty`), so that the launches queued behind see it too
            if (chunk == 0 && tid == 0 && sm_flags(flags, m, (it - 1) & 1u).empty[s]) sm_flags(flags, m, it & 1u).empty[s] = 1u;
            return;
        }
        if (chunk == 0 && tid == 0) {  // this iteration's reduce writes the other flag set: cleared here (its last readers were iteration it - 1's kernels)
            const SmFlags mine = sm_flags(flags, m, it & 1u);
            mine.moved[s] = 0u, mine.empty[s] = 0u, mine.ran[s] = 0u;
        }
    } else {
        if (gate_halt && *gate_halt) return;
        act = !active || active[s];
    }
    if (!act) return;
    // ---- stage: this subspace's codebook and the range's sub-vectors.  Loads UNCONDITIONAL with clamped indices and all in
    // flight before the first LDS store: a load under a per-lane condition is followed by the merge with the other branch's
    // value, i.e. by s_waitcnt vmcnt(0) -- one memory round trip per load
    {
        const float *cb_s = cb + (size_t)s * ksd;
        float cv[4], xv[SDP / 4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) cv[i] = cb_s[min(tid + i * kSmThreads, ksd - 1u)];
#pragma unroll
        for (uint32_t i = 0; i < SDP / 4; ++i) {  // 64 rows x sd <= 64 x SDP floats = SDP / 4 per thread
            const uint32_t e = min(tid + i * kSmThreads, rows_here * sd - 1u), r = e / sd, t = e - r * sd;
            xv[i] = X[(size_t)(r0 + r) * d + (size_t)s * sd + t];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i)
            if (tid + i * kSmThreads < ksd) cbs[tid + i * kSmThreads] = cv[i];
#pragma unroll
        for (uint32_t i = 0; i < SDP / 4; ++i) {
            const uint32_t e = tid + i * kSmThreads, r = e / sd, t = e - r * sd;
            if (e < rows_here * sd) rows[r * pitch + t] = xv[i];
        }
    }
    __syncthreads();
    // ---- assignment: find_nearest_centroid (vector.rs:352-363) over distance2 (vector.rs:135-143).  Four lanes per row, lane
    // q scans centroids q, q + 4, ...; merged with "smaller distance, then smaller index" -- the sequential scan's first
    // minimum.  The scan starts from centroid 0's distance whatever it is, so a NaN there wins (nothing is < NaN).
    {
        const uint32_t r = tid >> 2, q = tid & 3u;
        float x[SDP];
#pragma unroll
        for (int t = 0; t < SDP; ++t) x[t] = ((uint32_t)t < sd && r < rows_here) ? rows[r * pitch + t] : 0.0f;
        float best_dist = __builtin_inff();
        uint32_t best = 0xFFFFFFFFu;
        float d0 = 0.0f;
        for (uint32_t j = q; j < k; j += 4) {
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < SDP; ++t) {
                const float c = cbs[j * sd + min((uint32_t)t, sd - 1u)];
                const float diff = x[t] - c;
                const float sq = diff * diff;
                acc = ((uint32_t)t < sd) ? acc + sq : acc;
            }
            if (j == 0) d0 = acc;
            if (acc < best_dist) best_dist = acc, best = j;  // strict: the lane's first minimum; NaN never enters
        }
#pragma unroll
        for (int off = 1; off < 4; off <<= 1) {
            const float od = __shfl_xor(best_dist, off);
            const uint32_t oj = (uint32_t)__shfl_xor((int)best, off);
            if (od < best_dist || (od == best_dist && oj < best)) best_dist = od, best = oj;
        }
        d0 = __shfl(d0, (int)(tid & 60u));  // from the row's lane 0 (lane index inside the wave)
        if (best == 0xFFFFFFFFu || d0 != d0) best = 0;  // nothing below +inf, or a NaN at centroid 0: the scan never leaves 0
        if (q == 0 && r < rows_here) {
            codes_l[r] = best;
            store_code(codes, (size_t)(r0 + r) * m + s, best, k);
        }
    }
    __syncthreads();
    // ---- this range's per-cluster sums, members added in row order (mean_vector_by_indices, vector.rs:368-384) ----
    for (uint32_t e = tid; e < ksd; e += kSmThreads) {
        const uint32_t j = e / sd, t = e - j * sd;
        float acc = 0.0f;
        uint32_t cnt = 0;
        // (eight rows' reads in flight: the additions stay one after the other, in row order)
        for (uint32_t rb = 0; rb < rows_here; rb += 8) {
            uint32_t cj[8];
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t r = min(rb + u, rows_here - 1u);
                cj[u] = codes_l[r];
                v[u] = rows[r * pitch + t];
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const bool mine = (rb + u < rows_here) && cj[u] == j;
                acc = mine ? acc + v[u] : acc;
                cnt += mine ? 1u : 0u;
            }
        }
        psum[((size_t)chunk * m + s) * ksd + e] = acc;
        if (t == 0) pcnt[((size_t)chunk * m + s) * k + j] = cnt;
    }
}

// MODE 0: means + convergence test (vqhip_kmeans_step); 1: the same inside a device-driven run (flags); 2: f64 slab.
// grid (ceil(k * sd / 16), m): sixteen lanes per element, lane g sums the contiguous share g of the row ranges (at most
// 32 loads in flight: n <= 32768 is at most 512 ranges), the shares are added in range order -- a fixed tree of f64
// additions for a given shape, the same in every form.
template <int MODE>
__global__ __launch_bounds__(kSmThreads) void k_sm_reduce(uint32_t m, uint32_t k, uint32_t sd, uint32_t n_chunks,
                                                          const float *__restrict__ psum, const uint32_t *__restrict__ pcnt,
                                                          float *__restrict__ cb, uint32_t *__restrict__ counts,
                                                          uint32_t *__restrict__ changed, const uint8_t *__restrict__ active,
                                                          uint32_t *__restrict__ flags, uint32_t it, uint32_t *__restrict__ iters,
                                                          const uint32_t *__restrict__ gate_halt, double *__restrict__ slab) {
    const uint32_t s = blockIdx.y, tid = threadIdx.x, ksd = k * sd;
    bool act, halted = false;
    if (MODE == 1) {
        act = sm_run_state(active, flags, m, s, it, &halted);
        if (halted) return;
    } else {
        if (gate_halt && *gate_halt) return;
        act = !active || active[s];
    }
    const uint32_t e_raw = blockIdx.x * kSmElems + tid / kSmLanes, g = tid % kSmLanes;
    const bool live = e_raw < ksd;
    const uint32_t e = min(e_raw, ksd - 1u), j = e / sd, t = e - j * sd;  // (clamped: every lane's loads are valid and unconditional)
    if (MODE == 2 && gate_halt && blockIdx.x == 0 && tid == 0) changed[s] = 0u;  // a device-driven sharded run: k_finalize<true> only sets the flag
    if (!act) {  // a subspace that does not execute: counts read 0, nothing moved (include/vqhip.h)
        // ... of the last EXECUTED iteration: an iteration queued behind the retirement of every subspace executes
        // nothing and must leave the counts of the subspaces that ran the last executed one alone
        bool any_runs = true;
        if (MODE == 1 && it > 0) {
            const SmFlags prev = sm_flags(flags, m, (it - 1) & 1u);
            any_runs = false;
            for (uint32_t q = 0; q < m; ++q) any_runs = any_runs || (active[q] != 0 && prev.ran[q] && prev.moved[q]);
        }
        if (MODE != 2 && any_runs && live && g == 0 && t == 0) counts[(size_t)s * k + j] = 0u;
        return;
    }
    const uint32_t per = (n_chunks + kSmLanes - 1) / kSmLanes;  // <= 32
    const uint32_t q_lo = min(n_chunks, g * per), q_hi = min(n_chunks, q_lo + per);
    float v[32];
    uint32_t c[32];
#pragma unroll
    for (uint32_t u = 0; u < 32; ++u) {
        const uint32_t qq = min(q_lo + u, n_chunks - 1u);
        v[u] = psum[((size_t)qq * m + s) * ksd + e];
        c[u] = pcnt[((size_t)qq * m + s) * k + j];
    }
    // (pin the loads where they are: left alone, the compiler sinks each one into the conditional addition that uses it,
    // one memory round trip per range)
#pragma unroll
    for (uint32_t u = 0; u < 32; ++u) asm volatile("" : "+v"(v[u]), "+v"(c[u]));
    double acc = 0.0;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t u = 0; u < 32; ++u)
        if (q_lo + u < q_hi) acc += (double)v[u], cnt += c[u];
#pragma unroll
    for (uint32_t off = 1; off < kSmLanes; off <<= 1) {  // shares in range order: lane g takes lane g + off's behind its own
        const double oa = __shfl_down(acc, off);
        const uint32_t oc = (uint32_t)__shfl_down((int)cnt, off);
        if ((g & (2 * off - 1)) == 0) acc += oa, cnt += oc;
    }
    if (!live || g != 0) return;
    if (MODE == 2) {
        // the f64 slab [m][k][sd + 1] (last column: the count) of this rank's rows; k_finalize (behind the all-reduce, if
        // any) divides
        slab[((size_t)s * k + j) * (sd + 1) + t] = acc;
        if (t == 0) slab[((size_t)s * k + j) * (sd + 1) + sd] = (double)cnt;
        return;
    }
    if (t == 0) counts[(size_t)s * k + j] = cnt;
    bool moved = false;
    if (cnt != 0u) {  // an empty cluster keeps its centroid until the caller patches it
        float *cb_s = cb + (size_t)s * ksd;
        const float nv = (float)(acc / (double)cnt);
        const float diff = nv - cb_s[e];
        moved = !(fabsf(diff) < 1e-6f);  // approx_eq, vector.rs:232-240
        cb_s[e] = nv;
    }
    if (MODE == 0) {
        if (moved) changed[s] = 1u;  // (cleared by the launcher's memset; every writer stores the same value)
    } else {
        const SmFlags mine = sm_flags(flags, m, it & 1u);  // cleared by this iteration's k_sm_assign
        if (moved) mine.moved[s] = 1u;
        if (cnt == 0u) mine.empty[s] = 1u;
        if (e == 0) {
            mine.ran[s] = 1u;
            iters[s] += 1u;  // this subspace executed the iteration (vector.rs:415)
        }
    }
}

}  // namespace

bool lloyd_small_supported(uint64_t n, uint32_t m, uint32_t k, uint32_t sd) {
    return n >= 1 && n <= 32768 && n * m <= (1ull << 20) && (uint64_t)k * sd <= 1024 && sd >= 1 && sd <= 32 && m <= 65535;
}

// bytes of the f32 partial sums; *cnt_bytes of the partial counts, *flag_bytes of the run's two flag sets
size_t lloyd_small_workspace(uint64_t n, uint32_t m, uint32_t k, uint32_t sd, size_t *cnt_bytes, size_t *flag_bytes) {
    const size_t n_chunks = (size_t)((n + kSmRows - 1) / kSmRows);
    *cnt_bytes = n_chunks * m * k * 4;
    *flag_bytes = (size_t)2 * 3 * m * 4;
    return n_chunks * m * k * sd * 4;
}

template <bool RUN>
static int launch_assign(const float *X, uint32_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const float *cb, uint8_t *codes, float *psum,
                         uint32_t *pcnt, const uint8_t *active, uint32_t *flags, uint32_t it, const uint32_t *gate_halt, hipStream_t stream) {
    const dim3 grid((n + kSmRows - 1) / kSmRows, m);
    const size_t lds = ((size_t)k * sd + (size_t)kSmRows * (sd + 1) + kSmRows) * 4;
#define VQ_SM(SDPV) \
    hipLaunchKernelGGL((k_sm_assign<RUN, SDPV>), grid, dim3(kSmThreads), lds, stream, X, n, d, m, k, sd, cb, codes, psum, pcnt, active, flags, it, gate_halt)
    if (sd <= 4) VQ_SM(4);
    else if (sd <= 8) VQ_SM(8);
    else if (sd <= 16) VQ_SM(16);
    else VQ_SM(32);
#undef VQ_SM
    VQ_LAUNCH_CHECK("k_sm_assign");
    return VQHIP_OK;
}

// one iteration.  run_flags / run_iters non-null: iteration `it` of a device-driven run (`active` = the set the run started
// from, never null); else a host-driven step (`changed` cleared here).  The flags of a run: see lloyd_small_run_result.
int launch_lloyd_small(const float *X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, float *cb, uint8_t *codes,
                       float *psum, uint32_t *pcnt, uint32_t *counts, uint32_t *changed, const uint8_t *active, uint32_t *run_flags,
                       uint32_t *run_iters, uint32_t it, hipStream_t stream) {
    const uint32_t n_chunks = (uint32_t)((n + kSmRows - 1) / kSmRows);
    const dim3 rgrid((k * sd + kSmElems - 1) / kSmElems, m);
    if (run_flags) {
        VQ_TRY(launch_assign<true>(X, (uint32_t)n, d, m, k, sd, cb, codes, psum, pcnt, active, run_flags, it, nullptr, stream));
        hipLaunchKernelGGL(k_sm_reduce<1>, rgrid, dim3(kSmThreads), 0, stream, m, k, sd, n_chunks, psum, pcnt, cb, counts, changed, active, run_flags, it,
                           run_iters, (const uint32_t *)nullptr, (double *)nullptr);
    } else {
        VQ_HIP(hipMemsetAsync(changed, 0, (size_t)m * 4, stream));
        VQ_TRY(launch_assign<false>(X, (uint32_t)n, d, m, k, sd, cb, codes, psum, pcnt, active, nullptr, 0, nullptr, stream));
        hipLaunchKernelGGL(k_sm_reduce<0>, rgrid, dim3(kSmThreads), 0, stream, m, k, sd, n_chunks, psum, pcnt, cb, counts, changed, active,
                           (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, (const uint32_t *)nullptr, (double *)nullptr);
    }
    VQ_LAUNCH_CHECK("k_sm_reduce");
    return VQHIP_OK;
}

// What a device-driven run of `queued` iterations left, from the host copies of the start set, the two flag sets (as laid
// out above) and the iterations executed per subspace: the pause flag, the set still active, and the `changed` flags of
// the last executed iteration (vector.rs:448-457: an empty cluster pauses the run and nobody retires at that iteration;
// otherwise a subspace whose centroids did not move retires; one that stopped executing earlier retired then).
void lloyd_small_run_result(uint32_t m, const uint8_t *start_active, const uint32_t *flags, const uint32_t *iters, bool *paused,
                            uint8_t *active_out, uint32_t *changed_out) {
    uint32_t last = 0;
    for (uint32_t s = 0; s < m; ++s) last = iters[s] > last ? iters[s] : last;
    *paused = false;
    if (last == 0) {  // nothing executed
        for (uint32_t s = 0; s < m; ++s) active_out[s] = start_active[s], changed_out[s] = 0;
        return;
    }
    const uint32_t *set = flags + (size_t)((last - 1) & 1u) * 3 * m;  // the last executed iteration's: moved, empty, ran
    for (uint32_t s = 0; s < m; ++s) *paused = *paused || (set[m + s] != 0);
    for (uint32_t s = 0; s < m; ++s) {
        const bool ran_last = start_active[s] && iters[s] == last;
        changed_out[s] = (ran_last && set[s]) ? 1u : 0u;
        active_out[s] = (ran_last && (*paused || set[s])) ? 1 : 0;
    }
}

// assignment + the f64 slab of this rank's rows (vqhip_kmeans_accumulate and the row-sharded run): gate_halt non-null
// inside a device-driven run (the launches are no-ops once the run has paused; `active` is then the device's own set;
// `changed` is cleared for k_finalize<true>, which only sets it)
int launch_lloyd_small_slab(const float *X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const float *cb, uint8_t *codes,
                            float *psum, uint32_t *pcnt, const uint8_t *active, const uint32_t *gate_halt, uint32_t *changed, double *slab,
                            hipStream_t stream) {
    const uint32_t n_chunks = (uint32_t)((n + kSmRows - 1) / kSmRows);
    const dim3 rgrid((k * sd + kSmElems - 1) / kSmElems, m);
    VQ_TRY(launch_assign<false>(X, (uint32_t)n, d, m, k, sd, cb, codes, psum, pcnt, active, nullptr, 0, gate_halt, stream));
    hipLaunchKernelGGL(k_sm_reduce<2>, rgrid, dim3(kSmThreads), 0, stream, m, k, sd, n_chunks, psum, pcnt, (float *)nullptr, (uint32_t *)nullptr,
                       changed, active, (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, gate_halt, slab);
    VQ_LAUNCH_CHECK("k_sm_reduce<slab>");
    return VQHIP_OK;
}

}  // namespace vqhip
