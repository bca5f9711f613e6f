// k_screen_bf16.hip -- the MFMA screen on the bf16 matrix pipe (3-way operand split).
//
// Why: on gfx950 the fp32 MFMA (v_mfma_f32_16x16x4_f32) and ordinary VALU instructions do
// not overlap -- their times add, within a wave and across the waves of a SIMD
// (profiles/r1/ubench_mfma_valu_overlap.txt) -- so the fp32 screen (k_screen.hip) pays
// 64 MFMA x 32 cycles PLUS ~340 VALU x ~5 cycles per 16-row tile.  bf16 MFMAs run on the
// matrix pipe proper and hide completely behind VALU work.  This kernel therefore forms the
// same  s_j = |c_j|^2 - 2 x.c_j  from bf16 pieces:
//
//     x = x1 + x2 + x3,   a = -2c = a1 + a2 + a3      (exact: three 8-bit slices of the 24-bit
//                                                      significand, by truncation)
//     a.x ~= a1x1 + a2x1 + a1x2 + a3x1 + a2x2 + a1x3  (the three dropped cross terms are
//                                                      < 2^-21 |a||x| per dimension)
//
// i.e. 6*sub_dim bf16 products per (row, centroid) on v_mfma_f32_32x32x16_bf16 with f32
// accumulation, |c_j|^2 fused as the accumulator's initial value, an in-register
// min / second-min / argmin, a margin test and a work list for the exact re-check.
// Margin: DESIGN.md "screen soundness" (the per-MFMA accumulation bound eps_M comes from the
// bit-exact model of the instruction's adder, tests/mfma_model.py, validated on the device).
//
// The round-1 16x16x32 variants (registers / software-pipelined / A images in LDS with 2 and 4
// waves per SIMD; 0.60 / 0.87 / 0.80 ms at C2 against 0.49 here) were removed in round 2; they
// are in the history up to commit 65d0c95.
#include <cstdlib>

#include "kernels.hpp"

namespace vqhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * 64;
constexpr uint64_t kPipeMinSteps = 8;  // shortest row chunk (in 32-row steps) given to the pipelined X32 screen

// term pairs, big to small: (A part, X part), parts numbered 0..2
__host__ __device__ constexpr int pair_a(int p) { return p == 0 ? 0 : p == 1 ? 1 : p == 2 ? 0 : p == 3 ? 2 : p == 4 ? 1 : 0; }
__host__ __device__ constexpr int pair_x(int p) { return p == 0 ? 0 : p == 1 ? 0 : p == 2 ? 1 : p == 3 ? 0 : p == 4 ? 1 : 2; }

// three 8-bit slices of a float, each as the high half of an f32 word (= a bf16), exact
__host__ __device__ inline void split3(float v, uint32_t (&part)[3]) {
    uint32_t b0, b1, b2;
#if defined(__HIP_DEVICE_COMPILE__)
    b0 = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(b0);
    b1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b1);
    b2 = __float_as_uint(r2) & 0xFFFF0000u;
#else
    auto fu = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto uf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    b0 = fu(v) & 0xFFFF0000u;
    const float r1 = v - uf(b0);
    b1 = fu(r1) & 0xFFFF0000u;
    const float r2 = r1 - uf(b1);
    b2 = fu(r2) & 0xFFFF0000u;
#endif
    part[0] = b0;
    part[1] = b1;
    part[2] = b2;
}

// ---- variant X32: 32x32x16 MFMA tiles, 32 rows per wave step ------------------------------------
// Ablation of variant P showed the per-tile tail (two-level cross-lane merge, margin test, byte
// stores, work-list append) costing a third of the kernel.  With v_mfma_f32_32x32x16_bf16 a lane's
// column is one of 32 rows and its 16 accumulator registers are 16 centroids, so (a) a wave step
// covers 32 rows, (b) only lanes l and l+32 share a row -- ONE v_permlane32_swap level -- and (c)
// the tail is paid once per 32 rows.  The epilogue stays single-pass (index tag + pairwise med3 / min3 update,
// 2.5 VALU instructions per value, see reduce8) and is interleaved by hand with the next centroid tile's MFMA chain
// (a chain of 32x32x16 MFMAs on one accumulator issues back to back at full rate).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A image for the 32x32x16 form: [m][NT32][NMF][4 dwords][64 lanes]; lane (h = l>>5, c = l&31),
// k-slot j (0..7) of MFMA f: flat = 8f + j, pair = flat / DPH, dim = h*DPH + flat % DPH, with DPH = sd/2
// (sd = 8: two pairs per MFMA; 16: one; 24: nine MFMAs carry the 6 x 12 (pair, dim) slots of a lane half)
// Centred copy of a codebook for the X32 squared-L2 screen.  Distances do not change under a
// common translation, but the margin T is proportional to (|x - mu| + max|c - mu|)^2, so the
// screen works on x - mu and c - mu with mu = the mean centroid of the subspace: Uniform[0,1)
// sub-vectors of 16 dimensions shrink from |x| ~ 2.3 to ~ 1.15 (T / 4), data with a large common
// offset by far more.  cen[s] = {mu[sd], max|c - mu|, margin coefficient, -, -}; cn32 = |c - mu|^2
// (sequential f32 like k_prepare_codebook), padded with a large finite value.
// sdp >= sd: sub_dim of the screen kernel that serves this codebook (x32_padded_sd); the copy and mu are
// written sdp wide with zeros in the padding, where the screen's operands are zero too.
__global__ __launch_bounds__(256) void k_center_codebook_x32(const float *__restrict__ cb, uint32_t m, uint32_t k,
                                                             uint32_t sd, uint32_t sdp, uint32_t cn_stride, uint32_t nmf,
                                                             float *__restrict__ cbc, float *__restrict__ cen,
                                                             float *__restrict__ cn32) {
    __shared__ float s_mu[256];
    __shared__ float s_max[256];
    __shared__ int s_bad[256];
    const uint32_t s = blockIdx.x;
    const float *cbs = cb + (size_t)s * k * sd;
    float *cen_s = cen + (size_t)s * (sdp + 4);
    if (threadIdx.x >= sd && threadIdx.x < sdp) cen_s[threadIdx.x] = 0.0f;
    // column means: 256 threads = (256/sd) row groups x sd columns, partial sums through LDS
    // (a single thread per column walking k dependent loads cost 25 us per launch)
    {
        __shared__ double s_part[256];
        const uint32_t col = threadIdx.x % sd, grp = threadIdx.x / sd, n_grp = 256 / sd;
        double acc = 0.0;
        if (grp < n_grp)
            for (uint32_t j = grp; j < k; j += n_grp) acc += (double)cbs[(size_t)j * sd + col];
        s_part[threadIdx.x] = (grp < n_grp) ? acc : 0.0;
        __syncthreads();
        if (threadIdx.x < sd) {
            double tot = 0.0;
            for (uint32_t g2 = 0; g2 < n_grp; ++g2) tot += s_part[g2 * sd + threadIdx.x];
            const float mu = (float)(tot / (double)k);
            s_mu[threadIdx.x] = mu;
            cen_s[threadIdx.x] = mu;
        }
    }
    __syncthreads();
    float lmax = 0.0f;
    int bad = 0;
    for (uint32_t j = threadIdx.x; j < cn_stride; j += 256) {
        if (j < k) {
            float acc = -0.0f;
            for (uint32_t t = 0; t < sd; ++t) {
                const float v = cbs[(size_t)j * sd + t] - s_mu[t];
                cbc[((size_t)s * k + j) * sdp + t] = v;
                const float p = v * v;
                acc = acc + p;
                if (!(fabsf(v) <= 3.0e38f)) bad = 1;
            }
            for (uint32_t t = sd; t < sdp; ++t) cbc[((size_t)s * k + j) * sdp + t] = 0.0f;
            if (!(acc <= 3.0e38f)) bad = 1;
            cn32[(size_t)s * cn_stride + j] = (acc <= 3.0e38f) ? acc + 0.0f : 3.0e38f;
            lmax = fmaxf(lmax, acc);
        } else {
            cn32[(size_t)s * cn_stride + j] = 3.0e38f;  // padding never wins; finite (index packing)
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
    if (threadIdx.x == 0) {
        // DESIGN.md "screen soundness": bf16 coefficient + 128 (index packing) + 16 (rounding of x - mu, c - mu)
        const float u = 5.9604644775390625e-08f;
        const float coef = (8.0f * (float)sdp + 16.0f + 2.0f * kBf16AssumedUlps * (float)nmf + 16.0f + 128.0f + 16.0f) * u;
        cen_s[sdp] = sqrtf(s_max[0]) * 1.0000005f + 1e-30f;
        cen_s[sdp + 1] = s_bad[0] ? __builtin_inff() : coef;
        cen_s[sdp + 2] = 0.0f;
        cen_s[sdp + 3] = 0.0f;
    }
}

// cosine != 0: the image holds -c/|c| (zero for |c| < 1e-10, whose distance is the constant 1.0,
// src/core/distance.rs:113-115) so that the screen forms s_j = -|x| cos(x, c_j)
// sd_src: row length of `cb` (the codebook as is for cosine, already sd wide for the centred copy); dimensions
// from sd_src up to the kernel's sd are zero padding
// src_stride: floats between consecutive centroids of `cb` (a 64-dimension chunk of a wider sub-vector is prepared
// with cb advanced to the chunk and sd_src = the chunk's live dimensions)
__global__ __launch_bounds__(256) void k_prepare_bf16_x32(const float *__restrict__ cb, uint32_t m, uint32_t k,
                                                          uint32_t sd_src, uint32_t src_stride, uint32_t sd, uint32_t nt32,
                                                          uint32_t nmf, int cosine,
                                                          const float *__restrict__ cnsqrt,
                                                          uint32_t *__restrict__ prepA32) {
    const uint32_t s = blockIdx.x;
    const uint32_t dph = sd / 2;
    const float *cbs = cb + (size_t)s * k * src_stride;
    const uint32_t total = nt32 * nmf * 4 * 64;
    for (uint32_t e = blockIdx.y * blockDim.x + threadIdx.x; e < total; e += gridDim.y * blockDim.x) {
        const uint32_t lane = e & 63, w = (e >> 6) & 3, f = (e >> 8) % nmf, i = (e >> 8) / nmf;
        const uint32_t h = lane >> 5, c = lane & 31, j = 32 * i + c;
        uint32_t half[2] = {0u, 0u};
        for (uint32_t hh = 0; hh < 2; ++hh) {
            const uint32_t flat = 8 * f + 2 * w + hh;  // k-slot of this lane half over all MFMAs
            const uint32_t pair = flat / dph, dd = flat - pair * dph;
            if (pair < 6 && j < k && h * dph + dd < sd_src) {
                uint32_t parts[3];
                const float c = cbs[(size_t)j * src_stride + h * dph + dd];
                float av = -2.0f * c;
                if (cosine) {
                    const float nb = cnsqrt[(size_t)s * k + j];
                    av = (nb < 1e-10f) ? 0.0f : -(c / nb);
                }
                split3(av, parts);
                half[hh] = parts[pair_a((int)pair)] >> 16;
            }
        }
        prepA32[(size_t)s * total + e] = half[0] | (half[1] << 16);
    }
}

// G > 1 ("centroid groups", sub_dim 32 / 48 at k up to 256): the A image of all k centroids does not
// fit the register file, so G waves share a (row chunk, subspace), each holding NT32 tiles = one group
// of centroids; a wave then stops after merging its two lane halves and writes {min, second min,
// argmin, |x - mu|^2} per row to `part`; k_merge_partials_x32 merges the groups, applies the margin
// test and feeds the (unsegmented) re-check list.  G = 1 is the single-pass kernel.
// PVW > 0 ("padded"): the data's sub_dim is the run-time sdr < SD, the kernel works on SD dimensions whose last
// SD - sdr are zero in both operands (k_center_codebook_x32 writes the copy SD wide): a sub_dim between two
// instantiated ones rides on the next one up instead of the exact engine.  PVW = floats per load part, as the
// sub-vectors' alignment allows: 4 (sdr % 4 == 0), 2 (even sdr) or 1.
// copies of the fused update's LDS sums per wave: 2 while four (or eight) waves' accumulators still fit a CU
// A image of NT32 tiles at sub_dim SD: NT32 x ceil(3 SD / 8) MFMA operands of 4 registers.  Up to 96 registers two
// waves fit a SIMD (k <= 128 at sub_dim 16: 0.30 vs 0.37 ms at C2 / k = 128; k <= 256 at sub_dim 8 -- C3 and C5)
__host__ __device__ constexpr bool x32_two_waves(int sd, int nt32) { return nt32 * ((6 * (sd / 2) + 7) / 8) * 4 <= 96; }
__host__ __device__ constexpr uint32_t x32_acc_copies(int sd, int nt32) {
    const uint32_t waves = x32_two_waves(sd, nt32) ? 8u : 4u;
    return (waves * (uint32_t)nt32 * 32u * (2u * (uint32_t)sd + 1u) * 4u <= 140u * 1024u) ? 2u : 1u;
}

// ACC ("fused update", training only): the wave also OWNS the per-cluster sums and counts of its (row chunk,
// subspace) in LDS and adds every row it has just PROVEN -- the row's sub-vector is still in its registers -- so a
// Lloyd iteration reads X once (src/core/vector.rs:432-447, 368-384).  Rows that go to the exact
// re-check are left out here and added by k_accumulate_listed (k_update.hip) once their code is known.  Rows of one
// 32-row step that share a cluster are serialised by rank (earlier rows first): rank = ticket - count-before-the-step,
// the ticket from a returning LDS add on the cluster's counter -- which is the count the update needs anyway.
template <int SD, int NT32, int G = 1, int PVW = 0, bool ACC = false>
__global__ __launch_bounds__(kBlock, x32_two_waves(SD, NT32) ? 2 : 1) void k_assign_screen_bf16_x32(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m,
    const uint32_t *__restrict__ prepA32, const float *__restrict__ prepCn, uint32_t cn_stride,
    const float *__restrict__ meta, const uint32_t *__restrict__ sub_list, uint32_t n_sub,
    uint8_t *__restrict__ codes, uint32_t *__restrict__ wl_rows, uint32_t *__restrict__ wl_seg,
    uint32_t n_seg, uint64_t wl_stride, int cosine, uint32_t k_real, const float *__restrict__ cen,
    uint4 *__restrict__ part, uint32_t groups_rt, uint32_t sdr, float *__restrict__ acc_sums,
    uint32_t *__restrict__ acc_counts, const uint8_t *__restrict__ gate_active, const uint32_t *__restrict__ gate_halt,
    uint8_t *__restrict__ codes_t, uint64_t codes_t_pitch) {
    // device-side gates of vqhip_kmeans_run (iterations queued ahead of the host): a paused run or a subspace that
    // has converged meanwhile does nothing but publish an EMPTY work-list segment (the re-check then finds nothing)
    const bool halted = gate_halt && *gate_halt;
    static_assert(!ACC || (G == 1 && PVW == 0 && SD % 8 == 0), "fused update: single-pass kernels, lane halves of whole 16-byte parts");
    // G > 0: compile-time group count (k <= 256); G == 0: k > 256, the count comes in groups_rt
    const uint32_t groups = (G > 0) ? (uint32_t)G : groups_rt;
    constexpr int DPH = SD / 2;            // dims owned by a lane half
    constexpr int NMF = (6 * DPH + 7) / 8;  // MFMAs per 32x32 tile: 6 term pairs x DPH dims per lane half
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t h = lane >> 5, p = lane & 31;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;
    const uint32_t total_waves = gridDim.x * kWavesPerBlock;
    const uint32_t n_virt = n_sub * groups;  // (subspace, centroid group) pairs
    const uint32_t n_chunks = (G == 1) ? n_seg : total_waves / n_virt;  // (G == 1: the launcher's count, see the pipelined kernel)
    if (gw >= n_chunks * n_virt) return;
    const uint32_t vv = gw % n_virt;
    const uint32_t s = sub_list[vv / groups];
    const uint32_t grp = vv % groups;
    const uint32_t chunk = gw / n_virt;
    // every (listed subspace, chunk) has exactly one owner wave, and every owner writes its segment header -- also the
    // ones that do nothing: no memset of the headers in front of the launch
    auto empty_segment = [&]() {
        if (G == 1 && lane == 0) {
            uint32_t *sg = wl_seg + ((size_t)s * n_seg + chunk) * 2;
            sg[0] = 0u;
            sg[1] = 0u;
        }
    };
    if (halted || (gate_active && !gate_active[s])) {
        empty_segment();
        return;
    }
    const uint64_t n_steps = (n + 31) / 32;
    const uint64_t steps_per_chunk = (n_steps + n_chunks - 1) / n_chunks;
    const uint64_t st0 = (uint64_t)chunk * steps_per_chunk;
    uint64_t st1 = st0 + steps_per_chunk;
    if (st1 > n_steps) st1 = n_steps;
    // fused update: this wave's [NT32*32][SD] sums + counts in LDS (it is their only writer) -> one partial slab
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];
    float *sums = nullptr;
    uint32_t *cnts = nullptr;
    // R = 2 copies of the sums where the LDS allows: the first two rows of a step that share a cluster go to different
    // copies, so their read-modify-writes are independent (one LDS round trip instead of two); added up at the end
    constexpr uint32_t R = x32_acc_copies(SD, NT32);
    constexpr uint32_t kCopy = NT32 * 32 * SD;
    if constexpr (ACC) {
        constexpr uint32_t kPerWave = NT32 * 32 * (R * SD + 1);
        sums = acc_lds + (size_t)wave * kPerWave;
        cnts = reinterpret_cast<uint32_t *>(sums + R * kCopy);
        for (uint32_t e = lane; e < R * kCopy / 4; e += 64) reinterpret_cast<float4 *>(sums)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t e = lane; e < NT32 * 32; e += 64) cnts[e] = 0u;
    }
    auto write_partial = [&]() {
        if constexpr (ACC) {
            // slab of (row chunk, position of the subspace in the active list): [chunk][n_sub][k][SD]
            float4 *ps = reinterpret_cast<float4 *>(acc_sums + ((size_t)chunk * n_sub + vv) * k_real * SD);
            for (uint32_t e = lane; e < k_real * SD / 4; e += 64) {
                float4 t = reinterpret_cast<const float4 *>(sums)[e];
                if constexpr (R == 2) {
                    const float4 u = reinterpret_cast<const float4 *>(sums + kCopy)[e];
                    t.x = t.x + u.x;
                    t.y = t.y + u.y;
                    t.z = t.z + u.z;
                    t.w = t.w + u.w;
                }
                ps[e] = t;
            }
            uint32_t *pc = acc_counts + ((size_t)chunk * n_sub + vv) * k_real;
            for (uint32_t e = lane; e < k_real; e += 64) pc[e] = cnts[e];
        }
    };
    // The update of step st runs INSIDE step st + 1 (an in-order wave cannot hide an LDS round trip behind its own
    // dependent code): the ticket (count before + returning add) is issued at the end of st, the reads of the
    // accumulator rows after the operand split of st + 1, the adds and writes after its first MFMA block.
    bool pend_mine = false;
    uint32_t pend_j = 0, pend_before = 0, pend_seq = 0, pend_rank = 0xFFFFFFFFu;
    float pend_x[ACC ? DPH : 1];
    float4 pend_t[ACC ? DPH / 4 : 1];
    float4 *pend_slot = nullptr;
#ifndef VQ_ACC_EXP
#define VQ_ACC_EXP 0
#endif
    auto acc_issue = [&]() {   // ranks of the pending step's rows; start reading the rows of ranks 0 (and 1)
        if constexpr (ACC && VQ_ACC_EXP < 1) {
            uint32_t rank = (pend_mine && h == 0) ? pend_seq - pend_before : 0xFFFFFFFFu;
            rank = __builtin_amdgcn_permlane32_swap(rank, rank, false, false)[0];  // the row's other half takes the same turn
            pend_rank = rank;
            const uint32_t copy = (R == 2 && rank == 1u) ? kCopy : 0u;
            pend_slot = reinterpret_cast<float4 *>(sums + copy + (size_t)(pend_mine ? pend_j : 0u) * SD + DPH * h);
            if (rank < R) {
#pragma unroll
                for (int q = 0; q < DPH / 4; ++q) pend_t[q] = pend_slot[q];
            }
        }
    };
    auto acc_commit = [&]() {  // add and write back; rows of rank >= R (three of a cluster in one step) take turns after
        if constexpr (ACC && VQ_ACC_EXP < 1) {
            if (pend_rank < R) {
#pragma unroll
                for (int q = 0; q < DPH / 4; ++q) {
                    float4 t = pend_t[q];
                    t.x = t.x + pend_x[4 * q + 0];
                    t.y = t.y + pend_x[4 * q + 1];
                    t.z = t.z + pend_x[4 * q + 2];
                    t.w = t.w + pend_x[4 * q + 3];
                    pend_slot[q] = t;
                }
            }
            for (uint32_t r = R;; ++r) {
                if (!__any(pend_rank != 0xFFFFFFFFu && pend_rank >= r)) break;
                if (pend_rank == r) {
#pragma unroll
                    for (int q = 0; q < DPH / 4; ++q) {
                        float4 t = pend_slot[q];
                        t.x = t.x + pend_x[4 * q + 0];
                        t.y = t.y + pend_x[4 * q + 1];
                        t.z = t.z + pend_x[4 * q + 2];
                        t.w = t.w + pend_x[4 * q + 3];
                        pend_slot[q] = t;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            pend_mine = false;
            pend_rank = 0xFFFFFFFFu;
        }
    };
    if (st0 >= st1) {
        write_partial();  // an empty chunk still owns a (zero) slab
        empty_segment();
        return;
    }
    // wave-private work-list segment: slots [seg_first, seg_first + rows of the chunk) of the
    // subspace's list, filled in row order without atomics (a returning global atomic stalls a
    // lone wave for ~3000 cycles); {first, count} is published once at the end
    const uint32_t seg_first = (uint32_t)(st0 * 32);
    uint32_t seg_count = 0;

    __shared__ __attribute__((aligned(16))) float lds_cn[kWavesPerBlock][NT32 * 32];
    bf16x8 a[NT32][NMF];
    {
        const uint32_t *base = prepA32 + ((size_t)s * (NT32 * groups) + (size_t)grp * NT32) * NMF * 4 * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT32; ++i)
#pragma unroll
            for (int f = 0; f < NMF; ++f) {
                u32x4 v;
#pragma unroll
                for (int w = 0; w < 4; ++w) v[w] = base[((i * NMF + f) * 4 + w) * 64];
                a[i][f] = __builtin_bit_cast(bf16x8, v);
            }
        const uint32_t c_first = grp * NT32 * 32;           // first centroid of this wave's group
        const float *pc = prepCn + (size_t)s * cn_stride + c_first;  // padded up to cn_stride >= G*NT32*32
        // padding (and +inf norms) as a large FINITE value: a packed +inf would turn into a NaN
        for (uint32_t e = lane; e < NT32 * 32; e += 64) {
            float v = (c_first + e < cn_stride) ? pc[e] : 3.0e38f;
            if (cosine) v = (c_first + e < k_real) ? 0.0f : 3.0e38f;  // s_j = -x.c_j/|c_j| has no norm term
            lds_cn[wave][e] = (v < 3.0e38f) ? v : 3.0e38f;
        }
    }
    // |c|^2 in C/D layout: register r of tile i <- centroid 32i + (r&3) + 8(r>>2) + 4h
    const f32x4 *cnp = reinterpret_cast<const f32x4 *>(&lds_cn[wave][4 * h]);  // + 8*i*... see init_acc
    float pinf = __builtin_inff(), ninf = -__builtin_inff();
    uint32_t idx_mask = 0xFFFFFFC0u;  // low 6 mantissa bits carry the value index
    asm volatile("" : "+s"(idx_mask));
    asm volatile("" : "+s"(pinf), "+s"(ninf));
    // squared-L2 / Euclidean: centred operands (k_center_codebook_x32); cosine: as is
    float cmax, tcoef;
    float mu[DPH];
    if (cosine) {
        cmax = 0.0f;
        tcoef = meta[s * 4 + 3];  // only its finiteness matters here
        if (tcoef <= 3.0e38f)     // DESIGN.md "screen soundness", cosine: (6*sd + 2.5*eps_M*NMF + 200) * 2^-24 * |x|
            tcoef = (6.0f * SD + 2.5f * kBf16AssumedUlps * NMF + 200.0f) * 5.9604644775390625e-08f;
#pragma unroll
        for (int q = 0; q < DPH; ++q) mu[q] = 0.0f;
    } else {
        const float *cs = cen + (size_t)s * (SD + 4);
        cmax = cs[SD];
        tcoef = cs[SD + 1];
#pragma unroll
        for (int q = 0; q < DPH; ++q) mu[q] = cs[DPH * h + q];
    }

    const size_t col0 = (size_t)s * (PVW > 0 ? sdr : (uint32_t)SD) + (size_t)DPH * h;
    const size_t sub0 = (size_t)s * (PVW > 0 ? sdr : (uint32_t)SD);  // first float of the sub-vector: always a valid part
    auto load_x = [&](uint64_t row, float (&x)[DPH]) {
        if (row >= n) row = n - 1;
        const float *ptr = X + row * d + col0;
        if constexpr (PVW > 0) {
            // parts that run past the sub-vector re-read its first part (a valid address) and are zeroed
            constexpr int VW = (PVW == 4 && DPH % 4 != 0) ? 2 : PVW;
            const float *first = X + row * d + sub0;
#pragma unroll
            for (int q = 0; q < DPH; q += VW) {
                const bool live = (uint32_t)(DPH * h + q) < sdr;
                const float *pq = live ? ptr + q : first;
                if constexpr (VW == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(pq);
                    x[q + 0] = live ? t.x : 0.0f;
                    x[q + 1] = live ? t.y : 0.0f;
                    x[q + 2] = live ? t.z : 0.0f;
                    x[q + 3] = live ? t.w : 0.0f;
                } else if constexpr (VW == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(pq);
                    x[q + 0] = live ? t.x : 0.0f;
                    x[q + 1] = live ? t.y : 0.0f;
                } else {
                    const float t = *pq;
                    x[q] = live ? t : 0.0f;
                }
            }
        } else if constexpr (DPH % 4 == 0) {
#pragma unroll
            for (int q = 0; q < DPH; q += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(ptr + q);
                x[q + 0] = t.x;
                x[q + 1] = t.y;
                x[q + 2] = t.z;
                x[q + 3] = t.w;
            }
        } else {  // sub_dim 12: the lane half's 6 floats start on an 8-byte boundary only
#pragma unroll
            for (int q = 0; q < DPH; q += 2) {
                const float2 t = *reinterpret_cast<const float2 *>(ptr + q);
                x[q + 0] = t.x;
                x[q + 1] = t.y;
            }
        }
    };
    auto init_acc = [&](f32x16 &acc, int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 c4 = cnp[(32 * i + 8 * q) / 4];  // floats 32i + 8q + 4h .. +3
            acc[4 * q + 0] = c4[0];
            acc[4 * q + 1] = c4[1];
            acc[4 * q + 2] = c4[2];
            acc[4 * q + 3] = c4[3];
        }
    };

    float xn_[DPH];
    load_x(st0 * 32 + p, xn_);
    for (uint64_t st = st0; st < st1; ++st) {
        float x[DPH];
        float xo[ACC ? DPH : 1];  // the row's own values (the screen works on x - mu)
#pragma unroll
        for (int q = 0; q < DPH; ++q) {
            x[q] = xn_[q] - mu[q];
            if constexpr (ACC) xo[q] = xn_[q];
        }
        if (st + 1 < st1) load_x((st + 1) * 32 + p, xn_);

        // three bf16 slices of the lane's DPH components, packed two per dword
        uint32_t xp[3][DPH];
        float xs = 0.0f;
#pragma unroll
        for (int q = 0; q < DPH; ++q) {
            uint32_t parts[3];
            split3(x[q], parts);
            xp[0][q] = parts[0];
            xp[1][q] = parts[1];
            xp[2][q] = parts[2];
            xs = fmaf(x[q], x[q], xs);
        }
        bf16x8 b[NMF];
#pragma unroll
        for (int f = 0; f < NMF; ++f) {
            u32x4 v;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t hw[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int flat = 8 * f + 2 * w + hh, pair = flat / DPH, dd = flat % DPH;
                    hw[hh] = (pair < 6) ? xp[pair_x(pair)][dd] : 0u;
                }
                v[w] = (hw[0] >> 16) | (hw[1] & 0xFFFF0000u);
            }
            b[f] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int f = 0; f < NMF; ++f) asm volatile("" ::"v"(b[f]));
        if constexpr (ACC) {
            __builtin_amdgcn_sched_barrier(0);
            acc_issue();  // previous step's update: tickets are back, start the accumulator reads
            __builtin_amdgcn_sched_barrier(0);
        }

        float q1[4] = {pinf, pinf, pinf, pinf}, q2[4] = {pinf, pinf, pinf, pinf};
        // Four rotating accumulator tiles: in phase i the wave reduces tile i (finished a whole
        // phase ago -- no MFMA->VALU wait states needed), issues the MFMA chain of tile i+2 between
        // the reduce groups, and starts the |c|^2 reads that initialise tile i+3 (they land during
        // the phase -- no LDS wait in front of an MFMA).
        f32x16 acc[4];
        // reduce 16 values of a finished tile, interleaved with the NMF MFMAs of the next tile.
        // A lone wave issues one VALU instruction per ~4.4 cycles and the epilogue is most of the step's instructions,
        // so it is cut to 2.5 per value, none of them a compare / select (no VCC traffic): the value's index (6 bits,
        // an inline constant) replaces the 6 low mantissa bits (v_and_or_b32), then TWO packed values a, b enter a
        // chain's (q1 <= q2 = the two smallest so far) in three instructions:
        //     t = med3(q1, a, b);  q1 = min3(q1, a, b);  q2 = min(q2, t)
        // (q1 <= q2 and min(q1, a, b) <= q1, so the smallest of {q1, q2, a, b} is min3(q1, a, b) and the next one is
        // the smaller of q2 and the median of {q1, a, b}) -- the same two values the one-at-a-time form
        // q2 = med3(q1, q2, v); q1 = min(q1, v) leaves, at 1.5 instead of 2 instructions per value.  The winner's
        // index is read back from q1's low bits.  The perturbation (< 2^-18 relative) is paid for in the margin
        // (+128 * 2^-24).  Values 64..127 reuse the 6-bit codes; which half won is recovered from a snapshot of q1
        // taken after the first 64 values.
        auto reduce8 = [&](const f32x16 &fin, int ifin, int g8) {  // values 8 g8 .. 8 g8 + 7 of the tile: two per chain
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t ca = (uint32_t)((16 * ifin + 8 * g8 + r) & 63), cb = (uint32_t)((16 * ifin + 8 * g8 + 4 + r) & 63);
                const float pa = __uint_as_float((__float_as_uint(fin[8 * g8 + r]) & idx_mask) | ca);
                const float pb = __uint_as_float((__float_as_uint(fin[8 * g8 + 4 + r]) & idx_mask) | cb);
                const float t = __builtin_amdgcn_fmed3f(q1[r], pa, pb);
                q1[r] = __builtin_fminf(__builtin_fminf(q1[r], pa), pb);  // v_min3_f32
                q2[r] = __builtin_fminf(q2[r], t);
            }
        };
        // A operands are named as AGPRs while they fit (256); the image of sub_dim 24 at k = 256 is 288
        // registers, its last tile stays in VGPRs (an "a" operand beyond the file would be copied in
        // front of every use, behind the compiler's back as far as MFMA hazards go)
        auto mfma = [&](f32x16 &accv, int ti, int f) {
            if ((ti * NMF + f + 1) * 4 <= 256)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : "a"(a[ti][f]), "v"(b[f]));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : "v"(a[ti][f]), "v"(b[f]));
        };
        float snap[4] = {pinf, pinf, pinf, pinf};
        init_acc(acc[0], 0);
        if (NT32 > 1) init_acc(acc[1], 1);
        if (NT32 > 2) init_acc(acc[2], 2);
        asm volatile("s_nop 1");
#pragma unroll
        for (int f = 0; f < NMF; ++f)
            mfma(acc[0], 0, f);
        if (NT32 > 1) {
#pragma unroll
            for (int f = 0; f < NMF; ++f)
                mfma(acc[1], 1, f);
        } else {
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        }
        if constexpr (ACC) {
            __builtin_amdgcn_sched_barrier(0);
            acc_commit();  // previous step's update: add, write back (in the shadow of the MFMAs just issued)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NT32; ++i) {
            if (i + 3 < NT32) init_acc(acc[(i + 3) & 3], i + 3);
            if (i == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) snap[r] = q1[r];
            }
            constexpr int kGroups = 4;
            if constexpr (NMF <= 6) {
#pragma unroll
                for (int f = 0; f < (NMF > kGroups ? NMF : kGroups); ++f) {
                    if (i + 2 < NT32 && f < NMF) mfma(acc[(i + 2) & 3], (i + 2 < NT32) ? i + 2 : 0, f);
                    if (f < kGroups && (f & 1)) reduce8(acc[i & 3], i, f >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {  // longer chains (sub_dim 24): reduce group g follows MFMA ceil((g+1) NMF / 4) - 1
#pragma unroll
                for (int f = 0; f < NMF; ++f) {
                    if (i + 2 < NT32) mfma(acc[(i + 2) & 3], (i + 2 < NT32) ? i + 2 : 0, f);
#pragma unroll
                    for (int g8 = 0; g8 < kGroups / 2; ++g8)
                        if (f == ((g8 + 1) * NMF + kGroups / 2 - 1) / (kGroups / 2) - 1) reduce8(acc[i & 3], i, g8);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (i + 2 >= NT32 && i + 1 < NT32) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        }

        // ---- tail: merge chains, merge the two lane halves, margin test, outputs ----
        auto merge2 = [&](float a1, float a2, uint32_t ai, float b1, float b2, uint32_t bi, float &o1, float &o2,
                          uint32_t &oi) {
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            o2 = __builtin_amdgcn_fmed3f(hi, lo2, ninf);
            const bool tb = b1 < a1;
            o1 = tb ? b1 : a1;
            oi = tb ? bi : ai;
        };
        float u1, u2, w1, w2, m1, m2;
        uint32_t ui, wi, vidx;
        // 7-bit value index of each chain's minimum: low 6 bits of q1 + 64 if it moved after the snapshot
        uint32_t vi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            vi[r] = (__float_as_uint(q1[r]) & 63u) + ((NT32 > 4 && q1[r] < snap[r]) ? 64u : 0u);
        merge2(q1[0], q2[0], vi[0], q1[1], q2[1], vi[1], u1, u2, ui);
        merge2(q1[2], q2[2], vi[2], q1[3], q2[3], vi[3], w1, w2, wi);
        merge2(u1, u2, ui, w1, w2, wi, m1, m2, vidx);
        // value index -> centroid: tile = vidx>>4, r = vidx&15: 32*tile + (r&3) + 8*(r>>2) + 4*h
        uint32_t j = ((vidx >> 4) << 5) + (vidx & 3u) + (((vidx >> 2) & 3u) << 3) + 4 * h + grp * NT32 * 32;
        {
            const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m2), __float_as_uint(m2), false, false);
            const auto rj = __builtin_amdgcn_permlane32_swap(j, j, false, false);
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(xs), __float_as_uint(xs), false, false);
            const float a1 = __uint_as_float(r1[0]), b1 = __uint_as_float(r1[1]);
            const float a2 = __uint_as_float(r2[0]), b2 = __uint_as_float(r2[1]);
            xs = __uint_as_float(rx[0]) + __uint_as_float(rx[1]);
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            m2 = __builtin_amdgcn_fmed3f(lo2, hi, ninf);
            const bool take = (b1 < a1) || (b1 == a1 && rj[1] < rj[0]);
            j = take ? rj[1] : rj[0];
            m1 = take ? b1 : a1;
        }
        if constexpr (G != 1) {  // this group's verdict per row; the margin test happens after the groups are merged
            const uint64_t prow = st * 32 + p;
            if (h == 0 && prow < n)
                part[((size_t)s * groups + grp) * n + prow] =
                    make_uint4(__float_as_uint(m1), __float_as_uint(m2), j, __float_as_uint(xs));
            continue;
        }
        // squared-L2 / Euclid: T = coef * (|x| + max|c|)^2.  Cosine: s = -|x| cos, T = coef * |x|, and the
        // row is also re-checked when the best cosine is not clearly positive (all distances may
        // clamp to 1.0, src/core/distance.rs:118) or |x| is near the 1e-10 cut-off (distance.rs:113)
        const float xnorm = __builtin_sqrtf(xs) * 1.000001f;
        const float xn = xnorm + cmax;
        const float bnd = cosine ? xnorm : xn * xn;
        const float T = tcoef * bnd + 1e-35f * xn + 1e-37f;
        const float gap = m2 - m1;
        bool proven = (gap > T) && (fabsf(m1) <= 3.0e38f) && (T <= 3.0e38f);
        if (cosine) proven = proven && (m1 < -T) && (xnorm > 4e-10f);
        const uint64_t row = st * 32 + p;
        const bool writer = (h == 0) && (row < n);
        if (writer) {
            // [m][pitch] scratch when the caller transposes behind this kernel: the wave's 32 codes of a step are 32
            // contiguous bytes there, one byte in each of 32 rows (m bytes apart) in the final layout
            if (codes_t) codes_t[(size_t)s * codes_t_pitch + row] = (uint8_t)j;
            else codes[row * m + s] = (uint8_t)j;
        }
        const bool recheck = writer && !proven;
        const unsigned long long mask = __ballot(recheck);
        if (mask != 0ull) {
            if (recheck) {
                const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                wl_rows[(size_t)s * wl_stride + seg_first + seg_count + rank] = (uint32_t)row;
            }
            seg_count += (uint32_t)__popcll(mask);
        }
        if constexpr (ACC) {
            // this step's ticket: issued now, used (and the sums updated) inside the next step
            pend_mine = proven && (row < n);  // both lane halves of the row agree
            pend_j = j;
#pragma unroll
            for (int q = 0; q < DPH; ++q) pend_x[q] = xo[q];
            if (VQ_ACC_EXP < 2 && pend_mine && h == 0) {
                pend_before = cnts[j];              // every lane reads before any lane adds (one wave, in order)
                pend_seq = atomicAdd(&cnts[j], 1u);  // ds_add_rtn_u32: the rows of one cluster get before, before + 1, ...
            }
        }
    }
    if constexpr (ACC) {  // the last step's update
        acc_issue();
        acc_commit();
    }
    write_partial();
    if (G == 1 && lane == 0) {
        uint32_t *sg = wl_seg + ((size_t)s * n_seg + chunk) * 2;
        sg[0] = seg_first;
        sg[1] = seg_count;
    }
}

// ---- fused update of the pipelined screen, round 6: LDS f64 atomics ---------------------------------------------------
// profiles/ubench/valu_waves.hip (P10-P16): on gfx950 `ds_add_f64` / `ds_add_u32` WITHOUT return cost ~18 / ~11 ns per wave
// instruction beside the screen's own instruction mix at two waves per SIMD -- `ds_add_f32` 250 ns (!), a returning add
// plus its wait 25 ns.  So the wave's per-cluster sums live in LDS as DOUBLES, dimension-major, and a proven row is added
// by one fire-and-forget atomic per dimension: no ticket (count before + returning add), no rank, no second copy of the
// sums, no read-modify-write round trips and no turns for three rows of one cluster -- ~25 of the 48 instructions the
// update added to a sub_dim-8 step, and every wait it had.  Rows of one instruction that share a cluster are
// serialised by the LDS in a fixed lane order and a wave's LDS operations execute in program order: the same bits run to
// run.  The partial slab a wave leaves is still [k][SD] f32 (the f64 sum rounded once: closer to the exact sum than the
// sequential f32 chain it replaces), so k_accumulate_listed / k_reduce_* see the format they always did.
//   plane t (one dimension of all NT32*32 clusters) = NT32*32 doubles; planes of the upper lane half start 64 bytes
//   (16 banks) later, so the two halves of an instruction use different banks; counts: u32 [NT32*32] + 64 dummies (the
//   upper half's lanes add there: one address form, no exec masking)
#ifndef VQ_ACC_F64
#define VQ_ACC_F64 1
#endif
#ifndef VQ_ACC_RELOAD_PAR
#define VQ_ACC_RELOAD_PAR 0  // the trip's step that re-reads the pair's rows: 0 = a whole step ahead of the tail that uses them
#endif
__host__ __device__ constexpr uint32_t x32p_acc_plane_bytes(int nt32) { return (uint32_t)nt32 * 32u * 8u; }
__host__ __device__ constexpr uint32_t x32p_acc_bytes_per_wave(int sd, int nt32) {
    return (uint32_t)sd * x32p_acc_plane_bytes(nt32) + 64u + ((uint32_t)nt32 * 32u + 64u) * 4u;
}

// ---- variant X32P: the X32 kernel software-pipelined across steps -------------------------------------------
// A lone wave pays an issue slot of ~5 cycles for EVERY instruction, VALU or not (profiles/ubench/valu_issue.hip), so
// this kernel's time is its instruction count -- and in the X32 kernel the 12 MFMAs of a step's first two tiles issue
// back to back with nothing between them, the operand split (~100 instructions) and the tail (~70) run with the matrix
// pipe idle.  Here the accumulator ring simply runs on across steps -- phase i of step st reduces tile i, issues the
// MFMA chain of tile i + 2 (tiles 8 and 9 are tiles 0 and 1 of step st + 1) and starts the |c|^2 reads of tile i + 3
// (the tiles whose image is not kept in VGPRs) -- and the rest of the work rides in the gaps between those MFMAs, one
// piece per gap:
//     the operand split of step st + 1 (its rows were loaded two steps ago; the load of step st + 3 is issued as soon
//     as they are consumed) and the tail of step st - 1 (merge of the chains and lane halves, margin test, code,
//     work list), all within phases 0..5 -- the operands must be complete when phase 6 issues tile 0 of step st + 1.
// Two operand sets and two sets of chains alternate (the loop body is unrolled twice; a chunk of an odd number of
// steps runs one dummy step whose rows are clamped and whose tail writes nothing).  The arithmetic is the X32 kernel's
// and the codes are the same bits; since round 4 a lane runs two chains under other tags (reduce_hg) and the encode form
// shares one tail between two steps (tail_pair_piece), so a row whose gap sits within the tags' 64 ulps of the margin may
// land on the other side of the test -- in the re-check list or out of it.  launch_one_x32 picks this variant for chunks of at least kPipeMinSteps steps, 8 tiles (k in 225..256),
// sub_dim 8 or 16, one centroid group.
template <int SD, int NT32, bool ACC = false>
__global__ __launch_bounds__(kBlock, x32_two_waves(SD, NT32) ? 2 : 1) void k_assign_screen_bf16_x32p(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m, const uint32_t *__restrict__ prepA32,
    const float *__restrict__ prepCn, uint32_t cn_stride, const float *__restrict__ meta,
    const uint32_t *__restrict__ sub_list, uint32_t n_sub, uint8_t *__restrict__ codes, uint32_t *__restrict__ wl_rows,
    uint32_t *__restrict__ wl_seg, uint32_t n_seg, uint64_t wl_stride, int cosine, uint32_t k_real,
    const float *__restrict__ cen, const uint8_t *__restrict__ gate_active, const uint32_t *__restrict__ gate_halt,
    uint8_t *__restrict__ codes_t, uint64_t codes_t_pitch, float *__restrict__ acc_sums, uint32_t *__restrict__ acc_counts) {
    static_assert(NT32 == 8 && (SD == 8 || SD == 16), "pipelined screen: 8 tiles, sub_dim 8 or 16");
    constexpr int DPH = SD / 2;
    constexpr int NMF = (6 * DPH + 7) / 8;
    static_assert(6 * DPH == 8 * NMF, "every MFMA carries one term pair (sub_dim 16) or two (sub_dim 8), no padding slots");
    const bool halted = gate_halt && *gate_halt;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t h = lane >> 5, p = lane & 31;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;
    // the row chunks are the LAUNCHER's (n_seg of them: launch_one_x32 derives the count from ALL m subspaces in training,
    // so a fit's partial sums group the same rows whether retired subspaces are gated or dropped from the list)
    const uint32_t n_chunks = n_seg;
    if (gw >= n_chunks * n_sub) return;
    const uint32_t vv = gw % n_sub;
    const uint32_t s = sub_list[vv];
    const uint32_t chunk = gw / n_sub;
    uint32_t *const seg_hdr = wl_seg + ((size_t)s * n_seg + chunk) * 2;
    // rows and steps in 32 bits (the work list holds 32-bit rows; launch_one_x32 sends n >= 2^32 - 64 to the X32 kernel)
    const uint32_t n32 = (uint32_t)n;
    const uint32_t n_steps = (n32 + 31) / 32;
    const uint32_t steps_per_chunk = (n_steps + n_chunks - 1) / n_chunks;
    const uint32_t st0 = chunk * steps_per_chunk;
    uint32_t st1 = st0 + steps_per_chunk;
    if (st1 > n_steps) st1 = n_steps;
    if (halted || (gate_active && !gate_active[s])) {
        if (lane == 0) seg_hdr[0] = 0u, seg_hdr[1] = 0u;
        return;
    }
    // fused update (ACC): this wave's sums + counts in LDS, one partial slab per (row chunk, subspace)
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];
    constexpr uint32_t R = x32_acc_copies(SD, NT32);
    constexpr uint32_t kCopy = NT32 * 32 * SD;
    float *sums = nullptr;
    uint32_t *cnts = nullptr;
#if VQ_ACC_F64
    // (layout: x32p_acc_bytes_per_wave) sums64: plane t at byte t * kPlane (+ 64 for t >= DPH); counts behind them
    constexpr uint32_t kPlane = x32p_acc_plane_bytes(NT32);
    char *acc_wave = nullptr;
    if constexpr (ACC) {
        acc_wave = reinterpret_cast<char *>(acc_lds) + (size_t)wave * x32p_acc_bytes_per_wave(SD, NT32);
        cnts = reinterpret_cast<uint32_t *>(acc_wave + SD * kPlane + 64);
        for (uint32_t e = lane; e < (SD * kPlane + 64) / 16; e += 64) reinterpret_cast<float4 *>(acc_wave)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t e = lane; e < NT32 * 32 + 64; e += 64) cnts[e] = 0u;
    }
    auto write_partial = [&]() {
        if constexpr (ACC) {
            float *ps = acc_sums + ((size_t)chunk * n_sub + vv) * k_real * SD;
            for (uint32_t e = lane; e < k_real * SD; e += 64) {  // slab element (cluster j, dimension t): the f64 sum rounded once
                const uint32_t j = e / SD, t = e % SD;
                ps[e] = (float)*reinterpret_cast<const double *>(acc_wave + t * kPlane + (t >= DPH ? 64u : 0u) + j * 8u);
            }
            uint32_t *pc = acc_counts + ((size_t)chunk * n_sub + vv) * k_real;
            for (uint32_t e = lane; e < k_real; e += 64) pc[e] = cnts[e];
        }
    };
#else
    if constexpr (ACC) {
        constexpr uint32_t kPerWave = NT32 * 32 * (R * SD + 1);
        sums = acc_lds + (size_t)wave * kPerWave;
        cnts = reinterpret_cast<uint32_t *>(sums + R * kCopy);
        for (uint32_t e = lane; e < R * kCopy / 4; e += 64) reinterpret_cast<float4 *>(sums)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t e = lane; e < NT32 * 32; e += 64) cnts[e] = 0u;
    }
    auto write_partial = [&]() {
        if constexpr (ACC) {
            float4 *ps = reinterpret_cast<float4 *>(acc_sums + ((size_t)chunk * n_sub + vv) * k_real * SD);
            for (uint32_t e = lane; e < k_real * SD / 4; e += 64) {
                float4 t = reinterpret_cast<const float4 *>(sums)[e];
                if constexpr (R == 2) {
                    const float4 u = reinterpret_cast<const float4 *>(sums + kCopy)[e];
                    t.x = t.x + u.x;
                    t.y = t.y + u.y;
                    t.z = t.z + u.z;
                    t.w = t.w + u.w;
                }
                ps[e] = t;
            }
            uint32_t *pc = acc_counts + ((size_t)chunk * n_sub + vv) * k_real;
            for (uint32_t e = lane; e < k_real; e += 64) pc[e] = cnts[e];
        }
    };
#endif
    if (st0 >= st1) {
        write_partial();  // an empty chunk still owns a (zero) slab
        if (lane == 0) seg_hdr[0] = 0u, seg_hdr[1] = 0u;
        return;
    }
    const uint32_t nst = st1 - st0;
    const uint32_t seg_first = st0 * 32;
    uint32_t seg_count = 0;

    __shared__ __attribute__((aligned(16))) float lds_cn[kWavesPerBlock][NT32 * 32];
    bf16x8 a[NT32][NMF];
    {
        const uint32_t *base = prepA32 + (size_t)s * NT32 * NMF * 4 * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT32; ++i)
#pragma unroll
            for (int f = 0; f < NMF; ++f) {
                u32x4 v;
#pragma unroll
                for (int w = 0; w < 4; ++w) v[w] = base[((i * NMF + f) * 4 + w) * 64];
                a[i][f] = __builtin_bit_cast(bf16x8, v);
            }
        const float *pc = prepCn + (size_t)s * cn_stride;
        for (uint32_t e = lane; e < NT32 * 32; e += 64) {
            float v = (e < cn_stride) ? pc[e] : 3.0e38f;
            if (cosine) v = (e < k_real) ? 0.0f : 3.0e38f;
            lds_cn[wave][e] = (v < 3.0e38f) ? v : 3.0e38f;
        }
    }
    const f32x4 *cnp = reinterpret_cast<const f32x4 *>(&lds_cn[wave][4 * h]);
    float pinf = __builtin_inff(), ninf = -__builtin_inff();
    uint32_t idx_mask = 0xFFFFFFC0u;
    asm volatile("" : "+s"(idx_mask));
    asm volatile("" : "+s"(pinf), "+s"(ninf));
    float cmax, tcoef;
    float mu[DPH];
    if (cosine) {
        cmax = 0.0f;
        tcoef = meta[s * 4 + 3];
        if (tcoef <= 3.0e38f) tcoef = (6.0f * SD + 2.5f * kBf16AssumedUlps * NMF + 200.0f) * 5.9604644775390625e-08f;
#pragma unroll
        for (int q = 0; q < DPH; ++q) mu[q] = 0.0f;
    } else {
        const float *cs = cen + (size_t)s * (SD + 4);
        cmax = cs[SD];
        tcoef = cs[SD + 1];
#pragma unroll
        for (int q = 0; q < DPH; ++q) mu[q] = cs[DPH * h + q];
    }
    f32x4 mu4[DPH / 4];
#pragma unroll
    for (int q = 0; q < DPH; ++q) mu4[q / 4][q % 4] = mu[q];
    const char *const x_base = reinterpret_cast<const char *>(X + (size_t)s * SD + (size_t)DPH * h);
    const uint32_t x_pitch = d * 4;
    // codes: [row][m] bytes, or the [m][pitch] scratch of the transposing caller -- one address form for both
    uint8_t *const code_base = codes_t ? codes_t + (size_t)s * codes_t_pitch : codes + s;
    const uint32_t code_stride = codes_t ? 1u : m;
    // rows in flight, TWO steps deep (1024 waves x 2 KB per step in flight is ~2 MB, what 1.1 TB/s needs at ~2 us of
    // latency; one step deep the wave waited 19 % of its time): at the top of step st, xn_[par ^ 1] holds step st + 1
    // (about to be consumed, then reloaded with st + 1 + kDeep) and xn_[par] step st + 2
    // (two waves per SIMD: one step deep -- the partner wave covers the latency and the registers are short)
    constexpr int kDeep = x32_two_waves(SD, NT32) ? 1 : 2;
    // kept as the 16-byte vectors the loads return and consumed as such (x - mu on whole vectors): handed on as scalars,
    // the packed subtraction paired lanes 0 / 3 and 1 / 2 of a vector and the register copies that pairing needs were
    // placed right behind the LOAD -- with its wait: the sub_dim-8 kernel waited for every row load a quarter of a step
    // (or less) after issuing it
    f32x4 xn_[kDeep][DPH / 4];
    auto load_x = [&](uint32_t row, int buf) {
        row = row < n32 ? row : n32 - 1;
        const float *ptr = reinterpret_cast<const float *>(x_base + (uint64_t)row * x_pitch);
#pragma unroll
        for (int q = 0; q < DPH / 4; ++q) xn_[buf][q] = *reinterpret_cast<const f32x4 *>(ptr + 4 * q);
    };
    auto init_acc = [&](f32x16 &acc, int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 c4 = cnp[(32 * i + 8 * q) / 4];
            acc[4 * q + 0] = c4[0];
            acc[4 * q + 1] = c4[1];
            acc[4 * q + 2] = c4[2];
            acc[4 * q + 3] = c4[3];
        }
    };
    // One wave per SIMD (sub_dim 16) has VGPRs to spare: the |c|^2 image of the first kCnV tiles (16 values per lane and
    // tile) stays in registers and enters as the C operand of the tile's first MFMA -- no LDS re-read of the
    // accumulator's initial value for them (4 ds_read_b128 and a wait per tile; every instruction of a lone wave costs
    // an issue slot of ~5 cycles, profiles/ubench/valu_issue.hip).  (C and D of an MFMA share their register class, so
    // the 64 spare AGPRs cannot hold the other tiles' images.)  Two waves per SIMD (sub_dim 8): LDS for all tiles.
    constexpr int kCnV = x32_two_waves(SD, NT32) ? 0 : (ACC ? 4 : 6);
    f32x16 cnr[kCnV > 0 ? kCnV : 1];
#pragma unroll
    for (int i = 0; i < kCnV; ++i) init_acc(cnr[i], i);
    auto mfma = [&](f32x16 &accv, int ti, int f, const bf16x8 &bv) {
        if (f == 0 && ti < kCnV) {  // D = A B + |c|^2 (register image), the chain then runs in place
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(accv) : "a"(a[ti][f]), "v"(bv), "v"(cnr[ti]));
            return;
        }
        // two waves per SIMD: a kernel that names AGPRs gets the register file split 128 : 128, and this one needs ~165
        // VGPRs next to the 96 of the A image -- all of it in VGPRs (256) instead of spilling through v_accvgpr moves
        if constexpr (x32_two_waves(SD, NT32)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : "v"(a[ti][f]), "v"(bv));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : "a"(a[ti][f]), "v"(bv));
    };

    bf16x8 b[2][NMF];
    float q1[2][2], q2[2][2];  // [step parity][chain]: the two smallest tagged values so far (see reduce_hg)
    float xc[DPH];
    uint32_t xp[3][DPH];
    float xsT = 0.0f, xsM = 0.0f, xsN = 0.0f;  // |x - mu|^2 (this lane half) of steps st - 1, st, st + 1
    f32x16 acc[4];

    // operand split of one step in 1 + DPH / 4 + NMF pieces (four dimensions to a split piece: two independent packed
    // chains, no wait states between a v_pk_add_f32 and its consumer); piece 0 consumes the loaded rows (buffer nb) and issues the
    // load that refills the buffer
    constexpr int kSplitPieces = 1 + DPH / 4 + NMF;
    // (anchor: the loads of the peeled first trip are consumed by the loop, and left alone the compiler sinks them out
    // of the trip to the loop's door -- the operations in flight at the loop's first wait are then again not the steady
    // state's.  A use the compiler cannot rule out, in a block of the trip itself, keeps them where they are written;
    // launch_one_x32 never sends n = 2^32 - 1 here, the branch is never taken and its wait never paid.)
    auto split_piece = [&](int k, int nb, uint32_t next_step, bool anchor = false) {
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < DPH / 4; ++q) {
                if constexpr (x32_two_waves(SD, NT32) && (!ACC || VQ_ACC_F64)) {
                    f32x4 dv = xn_[nb % kDeep][q] - mu4[q];
                    asm volatile("" : "+v"(dv));  // consumed HERE (see reduce_hg), the registers are free for the next load
                    xc[4 * q + 0] = dv[0], xc[4 * q + 1] = dv[1], xc[4 * q + 2] = dv[2], xc[4 * q + 3] = dv[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xc[4 * q + e] = xn_[nb % kDeep][q][e] - mu[4 * q + e];
                        asm volatile("" ::"v"(xc[4 * q + e]));
                    }
                }
            }
            xsN = 0.0f;
            load_x(next_step * 32 + p, nb % kDeep);
            if (anchor && n32 == 0xFFFFFFFFu) {
#pragma unroll
                for (int q = 0; q < DPH / 4; ++q) asm volatile("" ::"v"(xn_[nb % kDeep][q]));
            }
        } else if (k <= DPH / 4) {
#pragma unroll
            for (int q = 4 * (k - 1); q < 4 * k; ++q) {
                uint32_t parts[3];
                split3(xc[q], parts);
                xp[0][q] = parts[0];
                xp[1][q] = parts[1];
                xp[2][q] = parts[2];
                xsN = fmaf(xc[q], xc[q], xsN);
            }
        } else {
            const int f = k - 1 - DPH / 4;
            u32x4 v;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t hw[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int flat = 8 * f + 2 * w + hh, pair = flat / DPH, dd = flat % DPH;
                    hw[hh] = xp[pair_x(pair)][dd];
                }
                v[w] = (hw[0] >> 16) | (hw[1] & 0xFFFF0000u);
            }
            b[nb][f] = __builtin_bit_cast(bf16x8, v);
            asm volatile("" ::"v"(b[nb][f]));
        }
    };
    // Values 8 (hg >> 1) + 2 (hg & 1) + {0, 1, 4, 5} of a finished tile enter the lane's TWO chains, two values per chain
    // (reduce8 above has the three-instruction update).  Element e of tile i goes to chain e & 1 under the 6-bit tag
    // [5:3] = i, [2:1] = e >> 2, [0] = bit 1 of e: a chain sees its 64 values of a step under 64 different tags, so the
    // winner's index is read off its low bits and its chain -- the X32 kernel's four chains re-use their tags after
    // four tiles and carry a snapshot through three merges to tell the halves apart (~23 instructions of every step's
    // tail).  The tags sit in bits the margin already gives away, and (minimum, second minimum) of a multiset do not
    // depend on the order of arrival: the codes are the same; a row whose gap is within 64 ulps of T may change sides
    // of the test (it is then settled by the exact re-check, or was).  The chain's second minimum takes both pairs of
    // a half-tile in one v_min3 (odd hg); the results are pinned to their gap (an empty asm that reads them): without a
    // use here the compiler sinks the whole reduction below the tail's branches.
    float ta[2];
    auto min3 = [](float x, float y, float z) { return __builtin_fminf(__builtin_fminf(x, y), z); };  // v_min3_f32
    auto reduce_hg = [&](const f32x16 &fin, int ifin, int hg, int par) {
        const int g8 = hg >> 1;
        const uint32_t ca = (uint32_t)(8 * ifin + 4 * g8 + (hg & 1)), cb = ca + 2u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int r = 2 * (hg & 1) + c;
            const float pa = __uint_as_float((__float_as_uint(fin[8 * g8 + r]) & idx_mask) | ca);
            const float pb = __uint_as_float((__float_as_uint(fin[8 * g8 + 4 + r]) & idx_mask) | cb);
            const float t = __builtin_amdgcn_fmed3f(q1[par][c], pa, pb);
            q1[par][c] = min3(q1[par][c], pa, pb);
            if ((hg & 1) == 0) {
                ta[c] = t;
                asm volatile("" ::"v"(q1[par][c]), "v"(ta[c]));
            } else {
                q2[par][c] = min3(q2[par][c], ta[c], t);
                asm volatile("" ::"v"(q1[par][c]), "v"(q2[par][c]));
            }
        }
    };
    float t_m1 = 0.0f, t_m2 = 0.0f, t_xs = 0.0f;
    uint32_t t_j = 0;
    bool t_proven = false;
    // Fused update of the rows a tail has just proven, spread over the step the tail runs in: the row's own values are
    // RE-READ (they were consumed two steps ago; an L2 hit, issued in gap 0), the ticket (count before + returning add
    // on the cluster's counter) is taken at the end of the tail, the accumulator rows are read in phase 6 and written
    // back in phase 7 -- each LDS round trip hides behind the gaps in between.
    bool pend_mine = false;
    uint32_t pend_j = 0, pend_before = 0, pend_seq = 0, pend_rank = 0xFFFFFFFFu;
    float pend_x[ACC ? DPH : 1];
    f32x4 pend_t[ACC ? DPH / 4 : 1];
    float4 *pend_slot = nullptr;
    auto acc_reload = [&](uint32_t tst) {  // the rows of step tst, as loaded (the screen worked on x - mu)
        if constexpr (ACC) {
            uint32_t row = tst * 32 + p;
            row = row < n32 ? row : n32 - 1;
            const float *ptr = reinterpret_cast<const float *>(x_base + (uint64_t)row * x_pitch);
#pragma unroll
            for (int q = 0; q < DPH; q += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(ptr + q);
                pend_x[q + 0] = t.x;
                pend_x[q + 1] = t.y;
                pend_x[q + 2] = t.z;
                pend_x[q + 3] = t.w;
            }
        }
    };
    auto acc_ticket = [&](bool mine) {
        if constexpr (ACC) {
#if VQ_ACC_F64
            // the whole update: one atomic per dimension of this lane half + the cluster's count (lower half; the upper half
            // adds to a dummy word of its own), all without return -- nothing to wait for, nothing left for later gaps
            if (mine) {
                const uint32_t a0 = (uint32_t)(uintptr_t)(acc_wave) + (DPH * kPlane + 64u) * h + t_j * 8u;
                const uint32_t ac = (uint32_t)(uintptr_t)(cnts) + (h ? (NT32 * 32u + p) * 4u : t_j * 4u);
                const uint32_t one = 1u;
#pragma unroll
                for (int q = 0; q < DPH; ++q) {
                    const double v = (double)pend_x[q];
                    asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(a0), "v"(v), "n"(q * (int)kPlane) : "memory");
                }
                asm volatile("ds_add_u32 %0, %1" ::"v"(ac), "v"(one) : "memory");
            }
            (void)pend_mine, (void)pend_j, (void)pend_before, (void)pend_seq;
#else
            pend_mine = mine;  // both lane halves of the row agree
            pend_j = t_j;
            if (pend_mine && h == 0) {
                pend_before = cnts[pend_j];              // every lane reads before any lane adds (one wave, in order)
                pend_seq = atomicAdd(&cnts[pend_j], 1u);  // ds_add_rtn_u32: the rows of one cluster get before, before + 1, ...
            }
#endif
        }
    };
    auto acc_issue = [&]() {
        if constexpr (ACC && !VQ_ACC_F64) {
            uint32_t rank = (pend_mine && h == 0) ? pend_seq - pend_before : 0xFFFFFFFFu;
            rank = __builtin_amdgcn_permlane32_swap(rank, rank, false, false)[0];  // the row's other half takes the same turn
            pend_rank = rank;
            const uint32_t copy = (R == 2 && rank == 1u) ? kCopy : 0u;
            pend_slot = reinterpret_cast<float4 *>(__builtin_assume_aligned(sums + copy + (size_t)(pend_mine ? pend_j : 0u) * SD + DPH * h, 16));
            // read by every lane (a lane without a turn reads a valid record and drops it), as ext_vector loads: the
            // float4 STRUCT loads came out of the compiler as ds_read_b96 + ds_read2_b32 + ds_read_b32 pieces (twice the LDS
            // instructions, 2.5x the bank-conflict cycles)
#pragma unroll
            for (int q = 0; q < DPH / 4; ++q) pend_t[q] = reinterpret_cast<const f32x4 *>(pend_slot)[q];
        }
    };
    auto acc_commit = [&]() {
        if constexpr (ACC && !VQ_ACC_F64) {
            if (pend_rank < R) {
#pragma unroll
                for (int q = 0; q < DPH / 4; ++q) {
                    f32x4 t = pend_t[q];
                    t[0] = t[0] + pend_x[4 * q + 0];
                    t[1] = t[1] + pend_x[4 * q + 1];
                    t[2] = t[2] + pend_x[4 * q + 2];
                    t[3] = t[3] + pend_x[4 * q + 3];
                    reinterpret_cast<f32x4 *>(pend_slot)[q] = t;
                }
            }
            for (uint32_t r = R;; ++r) {  // three or more rows of one cluster in a step take turns
                if (!__any(pend_rank != 0xFFFFFFFFu && pend_rank >= r)) break;
                if (pend_rank == r) {
#pragma unroll
                    for (int q = 0; q < DPH / 4; ++q) {
                        float4 t = pend_slot[q];
                        t.x = t.x + pend_x[4 * q + 0];
                        t.y = t.y + pend_x[4 * q + 1];
                        t.z = t.z + pend_x[4 * q + 2];
                        t.w = t.w + pend_x[4 * q + 3];
                        pend_slot[q] = t;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            pend_mine = false;
            pend_rank = 0xFFFFFFFFu;
        }
    };
    // ACC, f64 form, in the loop: the update rides on the PAIRED tail (one tail for two steps, as in the encode form) -- there a
    // lane owns a whole ROW's verdict (lanes 0..31 row p of step S - 1, lanes 32..63 row p of step S), so it re-reads the
    // row's SD values (gap 0 of the trip's second step; an L2 hit) and adds all of them: SD atomics + the count per lane and
    // two steps, the same number per step as the half-row form, and no second tail.  (The tail behind the loop is the one-step
    // form above.)
    constexpr bool kAccPair = ACC && VQ_ACC_F64;
    float pend_xr[kAccPair ? SD : 1];
    // The upper lane half walks the dimensions rotated by DPH (its registers hold x[DPH..SD), x[0..DPH)): in every atomic
    // instruction the two halves then address different planes, 16 banks apart -- 32 rows per 16 bank pairs instead of 64.
    auto acc_reload_pair = [&](uint32_t S) {  // rows of steps S - 1 (lower lanes) and S (upper lanes), as loaded
        if constexpr (kAccPair) {
            uint32_t row = (S - 1 + h) * 32 + p;
            row = row < n32 ? row : n32 - 1;
            const float *ptr = reinterpret_cast<const float *>(reinterpret_cast<const char *>(X + (size_t)s * SD) + (uint64_t)row * x_pitch);
            const float *pa = ptr + DPH * h, *pb = ptr - DPH * h;
#pragma unroll
            for (int q = 0; q < SD; q += 4) {
                const float4 t = *reinterpret_cast<const float4 *>((q < DPH ? pa : pb) + q);
                pend_xr[q + 0] = t.x;
                pend_xr[q + 1] = t.y;
                pend_xr[q + 2] = t.z;
                pend_xr[q + 3] = t.w;
            }
        }
    };
    // the update in kAccPieces pieces (spread over the step's free gaps: nine atomics in one gap hold the LDS pipe for
    // ~250 ns, in front of the other waves' |c|^2 reads); piece kAccPieces is the count
    constexpr int kAccPieces = SD / 4;
    bool upd_mine = false;
    uint32_t upd_j = 0;
    auto acc_update_begin = [&](bool mine) {
        if constexpr (kAccPair) upd_mine = mine, upd_j = t_j;
    };
    auto acc_update_piece = [&](int k) {
#if VQ_ACC_F64
        if constexpr (kAccPair) {
#ifdef VQ_ACC_ABL
            if (VQ_ACC_ABL == 1) return;  // (ablation builds only: timing without the atomics)
#endif
            if (upd_mine) {
                const uint32_t base = (uint32_t)(uintptr_t)(acc_wave) + upd_j * 8u;
                if (k < kAccPieces) {
                    const uint32_t a0 = base + ((4 * k < DPH) ? (DPH * kPlane + 64u) * h : (DPH * kPlane + 64u) * (1u - h));
#pragma unroll
                    for (int q = 4 * k; q < 4 * k + 4; ++q) {
                        const double v = (double)pend_xr[q];
                        asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(a0), "v"(v), "n"((q < DPH ? q : q - DPH) * (int)kPlane) : "memory");
                    }
                } else {
                    const uint32_t ac = (uint32_t)(uintptr_t)(cnts) + upd_j * 4u;
                    const uint32_t one = 1u;
                    asm volatile("ds_add_u32 %0, %1" ::"v"(ac), "v"(one) : "memory");
                }
            }
        }
#endif
    };
    // tail of a finished step in 4 pieces (the X32 kernel's tail, same order of operations)
    auto tail_piece = [&](int k, int tp, uint32_t tst, bool in_loop, bool no_step = false) {
        if (k == 0) {
            // the lane's two chains -> (minimum, second minimum, centroid): element e = 4 tag[2:1] + 2 tag[0] + chain of
            // tile tag[5:3] is centroid 32 tile + 8 (e >> 2) + 4 h + (e & 3)
            const float a1 = q1[tp][0], a2 = q2[tp][0], b1 = q1[tp][1], b2 = q2[tp][1];
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            t_m2 = __builtin_amdgcn_fmed3f(hi, lo2, ninf);
            const bool tb = b1 < a1;
            t_m1 = tb ? b1 : a1;
            const uint32_t tag = __float_as_uint(t_m1) & 63u;
            t_j = ((tag & 62u) << 2) + ((tag & 1u) << 1) + (tb ? 1u : 0u) + 4 * h;
#pragma unroll
            for (int c = 0; c < 2; ++c) q1[tp][c] = pinf, q2[tp][c] = pinf;  // the chains of step tst + 2 start here
        } else if (k == 1) {
            const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t_m1), __float_as_uint(t_m1), false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t_m2), __float_as_uint(t_m2), false, false);
            const auto rj = __builtin_amdgcn_permlane32_swap(t_j, t_j, false, false);
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(xsT), __float_as_uint(xsT), false, false);
            const float a1 = __uint_as_float(r1[0]), b1 = __uint_as_float(r1[1]);
            const float a2 = __uint_as_float(r2[0]), b2 = __uint_as_float(r2[1]);
            t_xs = __uint_as_float(rx[0]) + __uint_as_float(rx[1]);
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            t_m2 = __builtin_amdgcn_fmed3f(lo2, hi, ninf);
            const bool take = (b1 < a1) || (b1 == a1 && rj[1] < rj[0]);
            t_j = take ? rj[1] : rj[0];
            t_m1 = take ? b1 : a1;
        } else if (k == 2) {
            const float xnorm = __builtin_sqrtf(t_xs) * 1.000001f;
            const float xn = xnorm + cmax;
            const float bnd = cosine ? xnorm : xn * xn;
            const float T = tcoef * bnd + 1e-35f * xn + 1e-37f;
            const float gap = t_m2 - t_m1;
            bool proven = (gap > T) && (fabsf(t_m1) <= 3.0e38f) && (T <= 3.0e38f);
            if (cosine) proven = proven && (t_m1 < -T) && (xnorm > 4e-10f);
            t_proven = proven;
        } else {
            // In the loop the only tail without a step behind it is the first one (no_step: tst = st0 - 1, chains still +inf): it is
            // aimed at the rows of step st0, which the real tail of st0 overwrites a step later; rows past n repeat row
            // n - 1 (they were loaded from it) and both lane halves hold the same verdict -- so the code is stored
            // by every lane, without a branch.  The tail after the loop may belong to a dummy step: it asks.
            const bool valid = in_loop ? !no_step : (tst < st1);
            const uint32_t row = (in_loop && !valid ? st0 : tst) * 32 + p;
            const uint32_t rowc = row < n32 ? row : n32 - 1;
            if (in_loop || valid) code_base[(uint64_t)rowc * code_stride] = (uint8_t)t_j;
            const bool writer = (h == 0) && (row < n32) && valid;
            const bool recheck = writer && !t_proven;
            const unsigned long long mask = __ballot(recheck);
            if (mask != 0ull) {
                if (recheck) {
                    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    wl_rows[(size_t)s * wl_stride + seg_first + seg_count + rank] = (uint32_t)row;
                }
                seg_count += (uint32_t)__popcll(mask);
            }
            acc_ticket(t_proven && valid && (row < n32));
        }
    };

    // ACC = false: TWO steps share the part of the tail behind the lane's own merge.  Piece 0 of step S - 1 (run inside
    // step S, the first of a loop trip) parks the lane's (minimum, second minimum, centroid, |x|^2 half) in held_*;
    // a step later piece 0 of step S leaves its own in t_*, and ONE v_permlane32_swap per quantity hands lanes 0..31
    // both halves of row p of step S - 1 and lanes 32..63 both halves of row p of step S (the one-step form swaps a
    // register with itself and both halves of the wave then do the same work).  Merge, margin test, code store and
    // work-list append run once for the two steps: 64 rows in 64 lanes, the same values in the same order of rows.
    // The first trip has no step S - 1: its lanes 0..31 are no writers.
    float held_m1 = 0.0f, held_m2 = 0.0f, held_xs = 0.0f;
    uint32_t held_j = 0;
    auto tail_hold = [&]() { held_m1 = t_m1, held_m2 = t_m2, held_j = t_j, held_xs = xsT; };
    auto tail_pair_piece = [&](int k, uint32_t S, bool first) {
        if (k == 1) {
            const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(held_m1), __float_as_uint(t_m1), false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(held_m2), __float_as_uint(t_m2), false, false);
            const auto rj = __builtin_amdgcn_permlane32_swap(held_j, t_j, false, false);
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(held_xs), __float_as_uint(xsT), false, false);
            const float a1 = __uint_as_float(r1[0]), b1 = __uint_as_float(r1[1]);
            const float a2 = __uint_as_float(r2[0]), b2 = __uint_as_float(r2[1]);
            t_xs = __uint_as_float(rx[0]) + __uint_as_float(rx[1]);
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            t_m2 = __builtin_amdgcn_fmed3f(lo2, hi, ninf);
            const bool take = (b1 < a1) || (b1 == a1 && rj[1] < rj[0]);
            t_j = take ? rj[1] : rj[0];
            t_m1 = take ? b1 : a1;
        } else {  // k == 3 (piece 2, the margin test, is tail_piece's)
            const bool valid = (h != 0) || !first;
            const uint32_t row = (S - 1 + h) * 32 + p;
            // The store is unconditional in every trip (a predicated one is an operation the compiler cannot count on
            // when it sizes the vmcnt wait in front of the row loads).  First trip: lanes 0..31 have no row of their own
            // and repeat the store of their partner lane, same address, same byte.
            uint32_t srow = row, sj = t_j;
            if (first) {
                sj = __builtin_amdgcn_permlane32_swap(t_j, t_j, false, false)[1];
                srow = S * 32 + p;
            }
            const uint32_t rowc = srow < n32 ? srow : n32 - 1;
            code_base[(uint64_t)rowc * code_stride] = (uint8_t)sj;
            const bool recheck = valid && (row < n32) && !t_proven;
            const unsigned long long mask = __ballot(recheck);
            if (mask != 0ull) {
                if (recheck) {
                    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    wl_rows[(size_t)s * wl_stride + seg_first + seg_count + rank] = (uint32_t)row;
                }
                seg_count += (uint32_t)__popcll(mask);
            }
            acc_update_begin(valid && (row < n32) && t_proven);
        }
    };

    // ---- prologue: operands of step st0, its first two MFMA chains, the rows of st0 + 1 on their way ----
    load_x(st0 * 32 + p, 0);
    if (kDeep == 2) load_x((st0 + 1) * 32 + p, 1);
#pragma unroll
    for (int k = 0; k < kSplitPieces; ++k) split_piece(k, 0, st0 + kDeep);  // buffer 0: step st0 now, st0 + kDeep next
    xsM = xsN;
#pragma unroll
    for (int c = 0; c < 2; ++c) q1[0][c] = q2[0][c] = q1[1][c] = q2[1][c] = pinf;
    if (0 >= kCnV) init_acc(acc[0], 0);
    if (1 >= kCnV) init_acc(acc[1], 1);
    if (2 >= kCnV) init_acc(acc[2], 2);
    asm volatile("s_nop 1");
#pragma unroll
    for (int f = 0; f < NMF; ++f) mfma(acc[0], 0, f, b[0][f]);
#pragma unroll
    for (int f = 0; f < NMF; ++f) mfma(acc[1], 1, f, b[0][f]);
    __builtin_amdgcn_sched_barrier(0);

    // always an even number of steps: a step at or past st1 is a dummy (rows clamped, nothing written), so the loop body
    // has no exit in the middle; the tail of the very last step follows the loop.
    // The first trip is peeled (the outer two-trip loop is unrolled), for the compiler's wait counts: vmcnt is ONE in-order
    // counter of loads and stores, and the count in front of a consumed row load is the smallest number of younger
    // operations over all paths to that point.  Entered from the prologue, only the other buffer's two loads are
    // younger, so both waits of the loop came out as vmcnt(2) -- which in the steady state also waits for the loads
    // issued ONE step ago (a tail's store follows them): the rows were in fact prefetched one step deep, not two.  With
    // the first trip outside, every path into the loop carries the steady state's operations.  (`first` is a constant
    // in both copies.)
#pragma clang loop unroll(full)
    for (int peel = 0; peel < 2; ++peel) {
      const bool first = peel == 0;
      for (uint32_t it0 = first ? 0u : 2u; it0 < (first ? 1u : nst); it0 += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const uint32_t st = st0 + it0 + par;
#pragma unroll
            for (int i = 0; i < NT32; ++i) {
                if ((i + 3) % NT32 >= kCnV) init_acc(acc[(i + 3) & 3], (i + 3) % NT32);
#pragma unroll
                for (int f = 0; f < NMF; ++f) {
                    if (i + 2 < NT32) mfma(acc[(i + 2) & 3], i + 2, f, b[par][f]);
                    else mfma(acc[(i + 2) & 3], i + 2 - NT32, f, b[par ^ 1][f]);
                    // the gap's share of the phase: 4 reduce half-groups (9 VALU instructions each) and the fillers.
                    // Order of the fillers: the rows of step st + 1 are consumed (and the load of st + 2 issued) FIRST,
                    // the tail's stores follow -- the wait in front of the next consumption covers every earlier
                    // memory operation, so the stores get most of a step to retire before it.
                    if constexpr (NMF == 6) {
                        if (f == 0) reduce_hg(acc[i & 3], i, 0, par);
                        if (f == 1) reduce_hg(acc[i & 3], i, 1, par);
                        if (f == 3) reduce_hg(acc[i & 3], i, 2, par);
                        if (f == 4) reduce_hg(acc[i & 3], i, 3, par);
                        if (f == 2 || f == 5) {
                            const int slot = 2 * i + (f == 5);  // 0..15
                            if (slot == 0) {
                                if constexpr (kAccPair) { if (par == VQ_ACC_RELOAD_PAR) acc_reload_pair(st - VQ_ACC_RELOAD_PAR); }
                                else acc_reload(st - 1);
                                split_piece(0, par ^ 1, st + 1 + kDeep, first);
                            } else if (slot <= 4) {
                                if constexpr (ACC && !kAccPair) {
                                    tail_piece(slot - 1, par ^ 1, st - 1, true, first && par == 0);
                                } else if (par == 0) {  // step st - 1: the lane's own merge, parked for the partner step
                                    if (slot == 1) tail_piece(0, 1, st - 1, true), tail_hold();
                                } else {                // steps st - 2 and st - 1 together
                                    if (slot == 1) tail_piece(0, 0, st - 1, true);
                                    else if (slot == 3) tail_piece(2, 0, st - 1, true);
                                    else tail_pair_piece(slot - 1, st - 1, first);
                                }
                            }
                            else if (slot <= 10) split_piece(slot - 4, par ^ 1, st + 1 + kDeep);      // 2 splits, packs 0..3
                            else if (slot == 11) split_piece(7, par ^ 1, st + 1 + kDeep), split_piece(8, par ^ 1, st + 1 + kDeep);
                            else if (slot == 12) { if constexpr (!kAccPair) acc_issue(); }
                            else if (slot == 14) { if constexpr (!kAccPair) acc_commit(); }
                            if constexpr (kAccPair) {  // the pair's update: four dimensions per gap behind the tail, the count last
                                if (par == 1 && slot >= 5 && slot < 5 + kAccPieces) acc_update_piece(slot - 5);
                                if (par == 1 && slot == 5 + kAccPieces) acc_update_piece(kAccPieces);
                            }
                        }
                    } else {  // NMF == 3: one filler gap per phase
                        if (f == 0) reduce_hg(acc[i & 3], i, 0, par), reduce_hg(acc[i & 3], i, 1, par);
                        if (f == 1) reduce_hg(acc[i & 3], i, 2, par), reduce_hg(acc[i & 3], i, 3, par);
                        if (f == 2) {
                            if (i == 0) {
                                if constexpr (kAccPair) { if (par == VQ_ACC_RELOAD_PAR) acc_reload_pair(st - VQ_ACC_RELOAD_PAR); }
                                else acc_reload(st - 1);
                                split_piece(0, par ^ 1, st + 1 + kDeep, first);
                                tail_piece(0, par ^ 1, st - 1, true);
                                if constexpr (ACC && !kAccPair) tail_piece(1, par ^ 1, st - 1, true, first && par == 0);
                                else if (par == 0) tail_hold();
                                else tail_pair_piece(1, st - 1, first);
                            } else if (i == 1) {
                                if constexpr (ACC && !kAccPair) tail_piece(2, par ^ 1, st - 1, true), tail_piece(3, par ^ 1, st - 1, true, first && par == 0);
                                else if (par == 1) tail_piece(2, 0, st - 1, true), tail_pair_piece(3, st - 1, first);
                                split_piece(1, par ^ 1, st + 1 + kDeep);
                            } else if (i < kSplitPieces) {
                                split_piece(i, par ^ 1, st + 1 + kDeep);
                            } else if (i == 6) {
                                if constexpr (!kAccPair) acc_issue();
                            } else if (i == 7) {
                                if constexpr (!kAccPair) acc_commit();
                            }
                            if constexpr (kAccPair) {  // the pair's update: four dimensions per gap behind the tail, the count last
                                if (par == 1 && i >= 5 && i < 5 + kAccPieces) acc_update_piece(i - 5);
                                if (par == 1 && i == 5 + kAccPieces) acc_update_piece(kAccPieces);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            xsT = xsM;
            xsM = xsN;
        }
      }
    }
    {
        const uint32_t last = st0 + ((nst + 1) & ~1u) - 1;  // parity 1
        acc_reload(last);
#pragma unroll
        for (int k = 0; k < 4; ++k) tail_piece(k, 1, last, false);
        acc_issue();
        acc_commit();
        write_partial();
    }
    if (lane == 0) seg_hdr[0] = seg_first, seg_hdr[1] = seg_count;
}

// merges the G group verdicts of every (row, subspace), applies the margin test of the single-pass
// kernel and either writes the code or appends the row to the subspace's re-check list
template <int G>
__global__ __launch_bounds__(256) void k_merge_partials_x32(const uint4 *__restrict__ part, uint64_t n, uint32_t m,
                                                            uint32_t sd, const uint32_t *__restrict__ sub_list,
                                                            const float *__restrict__ cen, const float *__restrict__ meta,
                                                            int cosine, uint8_t *__restrict__ codes,
                                                            uint32_t *__restrict__ wl_rows, uint32_t *__restrict__ wl_count,
                                                            uint64_t wl_stride, uint32_t k, uint32_t groups_rt) {
    const uint32_t groups = (G > 0) ? (uint32_t)G : groups_rt;
    const uint32_t s = sub_list[blockIdx.y];
    const float *cs = cen + (size_t)s * (sd + 4);
    float cmax = cs[sd], tcoef = cs[sd + 1];
    if (cosine) {  // same margin as the single-pass kernel: (6 sd + 2.5 eps_M NMF + 200) 2^-24 |x|, inf for a non-finite codebook
        const float nmf = (float)((6 * (sd / 2) + 7) / 8);
        cmax = 0.0f;
        tcoef = (meta[s * 4 + 3] <= 3.0e38f) ? (6.0f * (float)sd + 2.5f * kBf16AssumedUlps * nmf + 200.0f) * 5.9604644775390625e-08f
                                             : __builtin_inff();
    }
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t row0 = (uint64_t)blockIdx.x * 256; row0 < n; row0 += (uint64_t)gridDim.x * 256) {
        const uint64_t row = row0 + threadIdx.x;
        bool recheck = false;
        if (row < n) {
            const uint4 p0 = part[((size_t)s * groups) * n + row];
            float m1 = __uint_as_float(p0.x), m2 = __uint_as_float(p0.y);
            uint32_t j = p0.z;
            const float xs = __uint_as_float(p0.w);
#pragma unroll
            for (uint32_t g = 1; g < groups; ++g) {
                const uint4 pg = part[((size_t)s * groups + g) * n + row];
                const float b1 = __uint_as_float(pg.x), b2 = __uint_as_float(pg.y);
                const float hi = fmaxf(m1, b1), lo2 = fminf(m2, b2);
                const bool take = (b1 < m1) || (b1 == m1 && pg.z < j);
                m2 = fminf(hi, lo2);
                j = take ? pg.z : j;
                m1 = take ? b1 : m1;
            }
            const float xnorm = __builtin_sqrtf(xs) * 1.000001f;
            const float xn = xnorm + cmax;
            const float T = tcoef * (cosine ? xnorm : xn * xn) + 1e-35f * xn + 1e-37f;
            bool proven = (m2 - m1 > T) && (fabsf(m1) <= 3.0e38f) && (T <= 3.0e38f);
            if (cosine) proven = proven && (m1 < -T) && (xnorm > 4e-10f);
            store_code(codes, row * m + s, j, k);
            recheck = !proven;
        }
        const unsigned long long mask = __ballot(recheck);
        if (mask != 0ull) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&wl_count[s], (uint32_t)__popcll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (recheck) wl_rows[(size_t)s * wl_stride + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)row;
        }
    }
}

// ---- wide sub-vectors (64 < sub_dim <= 64 NCH) -----------------------------------------------------
// The operands of a 96- or 128-dimensional sub-vector do not fit a wave next to its accumulators, so they are
// built and consumed 64 dimensions at a time: the NCH chunk products accumulate into ONE 32-centroid tile per
// wave (A images of the tile's chunks resident: 96 NCH registers), a (row chunk, subspace) is shared by
// ceil(k/32) waves, and the per-row partial verdicts go through k_merge_partials_x32 like the grouped screen's.
// Plain MFMA builtins and a compare/select epilogue (16 values per lane): this kernel is for reach -- 10x over
// the exact engine -- not tuned like the sub_dim <= 64 one.  sdr = the data's sub_dim (a multiple of 4).
template <int NCH>
__global__ __launch_bounds__(kBlock, 1) void k_assign_screen_bf16_wide(
    const float *__restrict__ X, uint64_t n, uint32_t d, const uint32_t *__restrict__ prepA, size_t chunk_stride,
    const float *__restrict__ prepCn, uint32_t cn_stride, const uint32_t *__restrict__ sub_list, uint32_t n_sub,
    int cosine, uint32_t k_real, const float *__restrict__ cen, uint4 *__restrict__ part, uint32_t groups,
    uint32_t sdr) {
    constexpr int DPH = 32, NMF = 24, SDP = 64 * NCH;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t h = lane >> 5, p = lane & 31;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;
    const uint32_t total_waves = gridDim.x * kWavesPerBlock;
    const uint32_t n_virt = n_sub * groups;
    const uint32_t n_chunks = total_waves / n_virt;
    if (gw >= n_chunks * n_virt) return;
    const uint32_t vv = gw % n_virt;
    const uint32_t s = sub_list[vv / groups];
    const uint32_t grp = vv % groups;
    const uint32_t chunk = gw / n_virt;
    const uint64_t n_steps = (n + 31) / 32;
    const uint64_t steps_per_chunk = (n_steps + n_chunks - 1) / n_chunks;
    const uint64_t st0 = (uint64_t)chunk * steps_per_chunk;
    uint64_t st1 = st0 + steps_per_chunk;
    if (st1 > n_steps) st1 = n_steps;
    if (st0 >= st1) return;

    bf16x8 a[NCH][NMF];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t *base = prepA + (size_t)c * chunk_stride + ((size_t)s * groups + grp) * NMF * 4 * 64 + lane;
#pragma unroll
        for (int f = 0; f < NMF; ++f) {
            u32x4 v;
#pragma unroll
            for (int w = 0; w < 4; ++w) v[w] = base[(f * 4 + w) * 64];
            a[c][f] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // |c - mu|^2 (cosine: 0) of this lane's 16 centroids: register r <- centroid 32 grp + (r&3) + 8(r>>2) + 4h
    float cnv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t idx = grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = (idx < cn_stride) ? prepCn[(size_t)s * cn_stride + idx] : 3.0e38f;
        if (cosine) v = (idx < k_real) ? 0.0f : 3.0e38f;
        cnv[r] = (v < 3.0e38f) ? v : 3.0e38f;
    }
    const float *cs = cen + (size_t)s * (SDP + 4);  // mu (zeros for cosine are not stored: skipped below)
    const float pinf = __builtin_inff(), ninf = -__builtin_inff();
    const size_t sub0 = (size_t)s * sdr;

    for (uint64_t st = st0; st < st1; ++st) {
        uint64_t row = st * 32 + p;
        if (row >= n) row = n - 1;
        const float *xrow = X + row * d + sub0;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = cnv[r];
        float xs = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float x[DPH];
#pragma unroll
            for (int q = 0; q < DPH; q += 4) {
                const uint32_t dim = 64 * c + DPH * h + q;
                const bool live = dim < sdr;  // parts behind the sub-vector re-read its first part and count as zeros
                const float4 t = *reinterpret_cast<const float4 *>(xrow + (live ? dim : 0u));
                float4 mu4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!cosine) mu4 = *reinterpret_cast<const float4 *>(cs + dim);  // zero in the padding
                x[q + 0] = live ? t.x - mu4.x : 0.0f;
                x[q + 1] = live ? t.y - mu4.y : 0.0f;
                x[q + 2] = live ? t.z - mu4.z : 0.0f;
                x[q + 3] = live ? t.w - mu4.w : 0.0f;
            }
            uint32_t xp[3][DPH];
#pragma unroll
            for (int q = 0; q < DPH; ++q) {
                uint32_t parts[3];
                split3(x[q], parts);
                xp[0][q] = parts[0];
                xp[1][q] = parts[1];
                xp[2][q] = parts[2];
                xs = fmaf(x[q], x[q], xs);
            }
#pragma unroll
            for (int f = 0; f < NMF; ++f) {
                u32x4 v;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t hw[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int flat = 8 * f + 2 * w + hh, pair = flat / DPH, dd = flat % DPH;  // 6 pairs x 32 dims = 24 x 8
                        hw[hh] = xp[pair_x(pair)][dd];
                    }
                    v[w] = (hw[0] >> 16) | (hw[1] & 0xFFFF0000u);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][f], __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
            }
        }
        // the lane's 16 values -> (min, second min, argmin); equal values keep the lower index and a zero gap
        float m1 = pinf, m2 = pinf;
        uint32_t ji = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r];
            const uint32_t idx = (r & 3) + 8 * (r >> 2) + 4 * h;
            m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
            const bool take = (v < m1) || (v == m1 && idx < ji);
            ji = take ? idx : ji;
            m1 = take ? v : m1;
        }
        uint32_t j = grp * 32 + ji;
        {
            const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m2), __float_as_uint(m2), false, false);
            const auto rj = __builtin_amdgcn_permlane32_swap(j, j, false, false);
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(xs), __float_as_uint(xs), false, false);
            const float a1 = __uint_as_float(r1[0]), b1 = __uint_as_float(r1[1]);
            const float a2 = __uint_as_float(r2[0]), b2 = __uint_as_float(r2[1]);
            xs = __uint_as_float(rx[0]) + __uint_as_float(rx[1]);
            const float hi = __builtin_amdgcn_fmed3f(a1, b1, pinf);
            const float lo2 = __builtin_amdgcn_fmed3f(a2, b2, ninf);
            m2 = __builtin_amdgcn_fmed3f(lo2, hi, ninf);
            const bool take = (b1 < a1) || (b1 == a1 && rj[1] < rj[0]);
            j = take ? rj[1] : rj[0];
            m1 = take ? b1 : a1;
        }
        const uint64_t prow = st * 32 + p;
        if (h == 0 && prow < n)
            part[((size_t)s * groups + grp) * n + prow] = make_uint4(__float_as_uint(m1), __float_as_uint(m2), j, __float_as_uint(xs));
    }
}

// codes_t [m][pitch] (a subspace's codes contiguous) -> codes [n][m], 256 rows per workgroup through LDS.  Subspaces the
// screen did not process (not in sub_list, or retired on the device inside vqhip_kmeans_run) keep the bytes they have.
__global__ __launch_bounds__(256) void k_codes_transpose(const uint8_t *__restrict__ ct, uint64_t pitch, uint8_t *__restrict__ codes,
                                                         uint64_t n, uint32_t m, const uint32_t *__restrict__ sub_list,
                                                         uint32_t n_sub, const uint8_t *__restrict__ gate_active,
                                                         const uint32_t *__restrict__ gate_halt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tr_lds[];  // [256][pitch] codes, then [m] activity flags
    if (gate_halt && *gate_halt) return;  // a paused run: the screen wrote nothing
    const uint32_t tp = codes_transpose_pitch(m);  // row pitch in LDS: a multiple of 4 that is not a multiple of 128
    uint8_t *tile = tr_lds, *act = tr_lds + 256 * tp;
    const uint64_t row0 = (uint64_t)blockIdx.x * 256;
    for (uint32_t s = threadIdx.x; s < m; s += 256) act[s] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_sub; i += 256) {
        const uint32_t s = sub_list[i];
        act[s] = gate_active ? gate_active[s] : (uint8_t)1;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lane l: rows 4 l .. 4 l + 3 of subspace s (pitch and row0 are multiples of 256).  Eight subspaces' loads are issued
    // before the first is used: one load per trip left every trip waiting for its own memory round trip (24 in a row at
    // m = 96: 68 us for the 96 MB of C3)
    for (uint32_t s0 = wave; s0 < m; s0 += 32) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            // unconditional (index clamped; the scratch holds m x pitch bytes): a load under an `if` is waited for at once
            const uint32_t s = min(s0 + 4 * u, m - 1u);
            v[u] = *reinterpret_cast<const uint32_t *>(ct + (size_t)s * pitch + row0 + 4 * lane);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t s = s0 + 4 * u;
            if (s >= m) break;
            uint32_t w = v[u];
            if (!act[s]) {
                w = 0;
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    const uint64_t r = row0 + 4 * lane + i;
                    if (r < n) w |= (uint32_t)codes[r * m + s] << (8 * i);
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) tile[(4 * lane + i) * tp + s] = (uint8_t)(w >> (8 * i));
        }
    }
    __syncthreads();
    const uint32_t rows_here = (uint32_t)min((uint64_t)256, n - row0), qpr = m / 4;  // dwords per row
    uint32_t *out = reinterpret_cast<uint32_t *>(codes + row0 * m);
    for (uint32_t idx = threadIdx.x; idx < rows_here * qpr; idx += 256) {
        const uint32_t r = idx / qpr, q = idx - r * qpr;
        out[idx] = *reinterpret_cast<const uint32_t *>(tile + r * tp + 4 * q);
    }
}

template <int SD, int NT32, int G = 1, int PVW = 0, bool ACC = false>
int launch_one_x32(const CodebookView &cb, const AssignArgs &a, hipStream_t stream, uint32_t groups_rt = 0) {
    const uint32_t groups = (G > 0) ? (uint32_t)G : groups_rt;  // G == 0: run-time group count (k > 256)
    const uint64_t n_steps = (a.n + 31) / 32;
    const uint32_t waves_per_simd = x32_two_waves(SD, NT32) ? 2 : 1;  // small A images leave room for two
    const uint32_t n_virt = a.n_sub * groups;
    // Training: the chunk count comes from ALL m subspaces, listed or not.  A host-driven fit drops retired subspaces
    // from the list where the device-driven run gates them; with the geometry of the list the survivors' rows were
    // regrouped into other partial sums (each rounded to f32 once), and the two fits could part by an iteration on data
    // whose convergence test hangs on the last bit (tests/test_gpu_fuzz.py seed 117 at VQ_FUZZ_SCALE=40).  The waves
    // a dropped subspace would have had stay unused, as they do behind a gate.
    const uint32_t n_virt_geom = (ACC && G == 1) ? cb.m * groups : n_virt;
    uint64_t want_waves = (uint64_t)num_cus() * kWavesPerBlock * waves_per_simd;
    // training (fused update): every row chunk costs a partial slab that k_reduce_partials_pos reads back, so a chunk
    // gets at least 8 steps (10k rows: 39 chunks instead of 313; the reduction 18 -> 5 us of a 60 us iteration)
    const uint64_t max_useful = (ACC ? std::max<uint64_t>(1, n_steps / 8) : n_steps) * n_virt_geom;
    if (want_waves > max_useful) want_waves = max_useful;
    if (want_waves < n_virt_geom) want_waves = n_virt_geom;
    uint32_t blocks = (uint32_t)((want_waves + kWavesPerBlock - 1) / kWavesPerBlock);
    while ((uint64_t)blocks * kWavesPerBlock < n_virt_geom) ++blocks;
    const uint32_t n_chunks = (blocks * kWavesPerBlock) / n_virt_geom;
    if (n_virt_geom != n_virt) blocks = (uint32_t)(((uint64_t)n_chunks * n_virt + kWavesPerBlock - 1) / kWavesPerBlock);
    if (G == 1) {
        if (!a.wl_seg || n_chunks > a.wl_seg_cap)
            return fail(VQHIP_ERR_FAILURE, "segmented work list missing or too small (%u > %u)", n_chunks, a.wl_seg_cap);
        // the kernel writes the header of every (listed subspace, chunk); the headers of subspaces that are not listed
        // (retired ones of a host-driven fit) are only read by the statistics: zeroed when there are any
        if (a.n_sub < cb.m) VQ_HIP(hipMemsetAsync(a.wl_seg, 0, (size_t)cb.m * n_chunks * 8, stream));
        a.n_seg = n_chunks;
    } else {
        if (!a.part) return fail(VQHIP_ERR_FAILURE, "grouped screen without a partial-result buffer");
        a.n_seg = 0;  // the merge kernel appends to the unsegmented list
    }
    size_t dyn_lds = 0;
    if constexpr (ACC) {
        // one partial slab per (row chunk, subspace): the chunk count is this launch's wave geometry
        if (n_chunks > a.acc_chunk_cap)
            return fail(VQHIP_ERR_FAILURE, "fused update: %u row chunks exceed the partial-slab capacity %u", n_chunks, a.acc_chunk_cap);
        a.acc_chunks = n_chunks;
        dyn_lds = (size_t)kWavesPerBlock * NT32 * 32 * (x32_acc_copies(SD, NT32) * SD + 1) * 4;
        static PerDeviceOnce attr_set;
        if (attr_set.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_assign_screen_bf16_x32<SD, NT32, G, PVW, ACC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
            attr_set.done();
        }
    }
    bool piped = false;
    if constexpr (G == 1 && PVW == 0 && NT32 == 8 && (SD == 16 || SD == 8)) {
        // chunks of at least kPipeMinSteps steps: the software-pipelined variant (same results)
        static const bool pipe_on = [] {
            const char *e = std::getenv("VQHIP_SCREEN_PIPE");
            return !(e && e[0] == '0');
        }();
        const uint64_t steps_per_chunk = (n_steps + n_chunks - 1) / n_chunks;
        if (pipe_on && steps_per_chunk >= kPipeMinSteps && a.n < 0xFFFFFFC0ull) {
            piped = true;
            if constexpr (ACC) {
                static PerDeviceOnce attr_set_p;
                if (attr_set_p.needed()) {
                    VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_assign_screen_bf16_x32p<SD, NT32, ACC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
                    attr_set_p.done();
                }
            }
            const size_t dyn_lds_p = (ACC && VQ_ACC_F64) ? (size_t)kWavesPerBlock * x32p_acc_bytes_per_wave(SD, NT32) : dyn_lds;
            hipLaunchKernelGGL((k_assign_screen_bf16_x32p<SD, NT32, ACC>), dim3(blocks), dim3(kBlock), dyn_lds_p, stream, a.X, a.n, a.d,
                               cb.m, cb.prepA32, cb.cn32, NT32 * 32, cb.meta, a.sub_list, a.n_sub, a.codes, a.wl_rows, a.wl_seg,
                               n_chunks, a.wl_stride, a.metric == VQHIP_COSINE ? 1 : 0, cb.k, cb.cen, a.gate_active,
                               a.gate_halt, a.codes_t, a.codes_t_pitch, ACC ? a.acc_sums : nullptr, ACC ? a.acc_counts : nullptr);
        }
    }
    if (!piped)
        hipLaunchKernelGGL((k_assign_screen_bf16_x32<SD, NT32, G, PVW, ACC>), dim3(blocks), dim3(kBlock), dyn_lds, stream, a.X, a.n, a.d,
                           cb.m, cb.prepA32, cb.cn32, NT32 * groups * 32, cb.meta, a.sub_list, a.n_sub, a.codes, a.wl_rows,
                           a.wl_seg, n_chunks, a.wl_stride, a.metric == VQHIP_COSINE ? 1 : 0, cb.k, cb.cen,
                           reinterpret_cast<uint4 *>(a.part), groups, cb.sd, ACC ? a.acc_sums : nullptr, ACC ? a.acc_counts : nullptr,
                           a.gate_active, a.gate_halt, (G == 1) ? a.codes_t : nullptr, a.codes_t_pitch);
    VQ_LAUNCH_CHECK("k_assign_screen_bf16_x32");
    if (G == 1 && a.codes_t) {
        hipLaunchKernelGGL(k_codes_transpose, dim3((uint32_t)((a.n + 255) / 256)), dim3(256), (size_t)256 * codes_transpose_pitch(cb.m) + cb.m, stream,
                           a.codes_t, a.codes_t_pitch, a.codes, a.n, cb.m, a.sub_list, a.n_sub, a.gate_active, a.gate_halt);
        VQ_LAUNCH_CHECK("k_codes_transpose");
    }
    if (G != 1) {
        uint64_t mblocks = (a.n + 255) / 256;
        if (mblocks > (uint64_t)num_cus() * 8) mblocks = (uint64_t)num_cus() * 8;
        hipLaunchKernelGGL((k_merge_partials_x32<G>), dim3((uint32_t)mblocks, a.n_sub), dim3(256), 0, stream,
                           reinterpret_cast<const uint4 *>(a.part), a.n, cb.m, (uint32_t)SD, a.sub_list, cb.cen, cb.meta,
                           a.metric == VQHIP_COSINE ? 1 : 0, a.codes, a.wl_rows, a.wl_count, a.wl_stride, cb.k, groups);
        VQ_LAUNCH_CHECK("k_merge_partials_x32");
    }
    return VQHIP_OK;
}

template <int NCH>
int launch_wide(const CodebookView &cb, const AssignArgs &a, hipStream_t stream, uint32_t groups) {
    if (!a.part) return fail(VQHIP_ERR_FAILURE, "wide screen without a partial-result buffer");
    if (cb.sd % 4 != 0) return fail(VQHIP_ERR_UNSUPPORTED, "wide screen: sub_dim=%u", cb.sd);
    const uint64_t n_steps = (a.n + 31) / 32;
    const uint32_t n_virt = a.n_sub * groups;
    uint64_t want_waves = (uint64_t)num_cus() * kWavesPerBlock;
    const uint64_t max_useful = n_steps * n_virt;
    if (want_waves > max_useful) want_waves = max_useful;
    if (want_waves < n_virt) want_waves = n_virt;
    uint32_t blocks = (uint32_t)((want_waves + kWavesPerBlock - 1) / kWavesPerBlock);
    while ((uint64_t)blocks * kWavesPerBlock < n_virt) ++blocks;
    a.n_seg = 0;  // the merge kernel appends to the unsegmented list
    const size_t chunk_stride = (size_t)cb.m * groups * 24 * 4 * 64;
    hipLaunchKernelGGL((k_assign_screen_bf16_wide<NCH>), dim3(blocks), dim3(kBlock), 0, stream, a.X, a.n, a.d, cb.prepA32,
                       chunk_stride, cb.cn32, groups * 32, a.sub_list, a.n_sub, a.metric == VQHIP_COSINE ? 1 : 0, cb.k, cb.cen,
                       reinterpret_cast<uint4 *>(a.part), groups, cb.sd);
    VQ_LAUNCH_CHECK("k_assign_screen_bf16_wide");
    uint64_t mblocks = (a.n + 255) / 256;
    if (mblocks > (uint64_t)num_cus() * 8) mblocks = (uint64_t)num_cus() * 8;
    hipLaunchKernelGGL((k_merge_partials_x32<0>), dim3((uint32_t)mblocks, a.n_sub), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(a.part), a.n, cb.m, (uint32_t)(64 * NCH), a.sub_list, cb.cen, cb.meta,
                       a.metric == VQHIP_COSINE ? 1 : 0, a.codes, a.wl_rows, a.wl_count, a.wl_stride, cb.k, groups);
    VQ_LAUNCH_CHECK("k_merge_partials_x32");
    return VQHIP_OK;
}

}  // namespace

// tiles of 32 centroids per wave and centroid groups for a shape (0 = no X32 form)
// sub_dim of the X32 kernel that serves `sd`: itself when instantiated, the next one up for the even sub_dims in
// between (zero padding, see the kernel's SDR parameter), 0 when there is none
uint32_t x32_padded_sd(uint32_t sd) {
    switch (sd) {
    case 8: case 12: case 16: case 24: case 32: case 48: case 64: return sd;
    case 5: case 6: case 7: return 8;
    case 9: case 10: case 11: return 12;
    case 13: case 14: case 15: return 16;
    case 17: case 18: case 19: case 20: case 21: case 22: case 23: return 24;
    default: break;
    }
    if (sd >= 25 && sd <= 31) return 32;
    if (sd >= 33 && sd <= 47) return 48;
    if (sd >= 49 && sd <= 63) return 64;
    // two or three chunks of 64 dimensions (k_assign_screen_bf16_wide): any multiple of 4 (16-byte row parts)
    if (sd > 64 && sd <= 128 && sd % 4 == 0) return 128;
    if (sd > 128 && sd <= 192 && sd % 4 == 0) return 192;
    return 0;
}

void screen_bf16_x32_tiling(uint32_t sd_real, uint32_t k, uint32_t *nt32_per_group, uint32_t *groups) {
    *nt32_per_group = *groups = 0;
    if (k == 0 || k > kMaxCentroids) return;
    const uint32_t sd = x32_padded_sd(sd_real);
    if (sd == 0) return;
    if (sd > 64) {  // wide kernel: one 32-centroid tile per wave, any k up to 16 groups
        const uint32_t g = (k + 31) / 32;
        if (g > kX32MaxGroups) return;
        *nt32_per_group = 1;
        *groups = g;
        return;
    }
    if (sd != sd_real) {
        // padded variants exist for images of 8 and of 4 tiles: smaller codebooks are filled up with never-winning
        // centroids as long as that costs at most twice the useful work
        if (k <= 64 || k > 256) return;
        if (k <= 128) {  // 4 tiles in all
            *nt32_per_group = (sd <= 32) ? 4 : 2;
            *groups = 4 / *nt32_per_group;
            return;
        }
        *nt32_per_group = (sd <= 24) ? 8 : (sd == 32) ? 4 : 2;  // 8 tiles in all: 1, 2 or 4 centroid groups
        *groups = 8 / *nt32_per_group;
        return;
    }
    const uint32_t nt = (k + 31) / 32;
    uint32_t cap;  // tiles whose A image fits next to the working set: NMF * cap * 4 registers
    switch (sd) {
    case 8: case 12: case 16: case 24: cap = 8; break;
    case 32: cap = 4; break;   // 12 MFMAs per tile: 192 registers
    case 48: cap = 2; break;   // 18 MFMAs per tile: 144 registers
    case 64: cap = 2; break;   // 24 MFMAs per tile: 192 registers
    default: return;
    }
    const uint32_t per = nt < cap ? nt : cap;
    const uint32_t g = (nt + per - 1) / per;
    // k > 256 (two-byte codes): more groups of the same kernels, up to kX32MaxGroups; each group costs a
    // screen pass and a 16-byte partial verdict per row, beyond that the exact scan is the better engine
    if (k > 256 && g > kX32MaxGroups) return;
    *nt32_per_group = per;
    *groups = g;
}

// shapes whose training assignment can carry the fused update (launch_assign_screen_bf16 with acc_sums set)
bool screen_bf16_fused_update_supported(uint32_t sd, uint32_t k) {
    uint32_t per, groups;
    screen_bf16_x32_tiling(sd, k, &per, &groups);
    return per != 0 && groups == 1 && k <= 256 && (sd == 8 || sd == 16 || sd == 24) && x32_padded_sd(sd) == sd;
}

bool screen_bf16_x32_supported(uint32_t sd, uint32_t k) {
    uint32_t per, groups;
    screen_bf16_x32_tiling(sd, k, &per, &groups);
    return per != 0;
}
uint32_t screen_bf16_x32_mfmas(uint32_t sd_real) {
    const uint32_t sd = x32_padded_sd(sd_real) ? x32_padded_sd(sd_real) : sd_real;
    return (6 * (sd / 2) + 7) / 8;
}
// X32 images of a codebook.  Squared-L2 / Euclidean: centred copy (cbc), its norms (cn32) and {mu, max|c-mu|,
// coefficient} (cen), then the bf16 slices of -2(c - mu); cosine: bf16 slices of -c/|c| from the codebook as is.
int launch_prepare_bf16_x32(const CodebookView &v, uint32_t *prepA32, int cosine, float *cbc, float *cen,
                            float *cn32, hipStream_t stream) {
    if (v.m == 0) return VQHIP_OK;
    uint32_t per = 0, groups = 0;
    screen_bf16_x32_tiling(v.sd, v.k, &per, &groups);
    const uint32_t nt32 = per * groups;  // image padded to whole groups (zero operands, never-winning norms)
    const float *src = v.cb;
    const uint32_t sdp = x32_padded_sd(v.sd);
    if (!cosine) {
        hipLaunchKernelGGL(k_center_codebook_x32, dim3(v.m), dim3(256), 0, stream, v.cb, v.m, v.k, v.sd, sdp, nt32 * 32,
                           screen_bf16_x32_mfmas(v.sd), cbc, cen, cn32);
        VQ_LAUNCH_CHECK("k_center_codebook_x32");
        src = cbc;  // sdp wide
    }
    if (sdp > 64) {  // wide sub-vectors: one image per 64-dimension chunk, chunk-major
        const uint32_t stride = cosine ? v.sd : sdp;
        const size_t chunk_stride = (size_t)v.m * nt32 * 24 * 4 * 64;
        for (uint32_t c = 0; c < sdp / 64; ++c) {
            const uint32_t live = v.sd > 64 * c ? std::min(64u, v.sd - 64 * c) : 0u;
            hipLaunchKernelGGL(k_prepare_bf16_x32, dim3(v.m, 16), dim3(256), 0, stream, src + 64 * c, v.m, v.k, live, stride, 64u,
                               nt32, 24u, cosine, v.cnsqrt, prepA32 + c * chunk_stride);
        }
        VQ_LAUNCH_CHECK("k_prepare_bf16_x32");
        return VQHIP_OK;
    }
    hipLaunchKernelGGL(k_prepare_bf16_x32, dim3(v.m, 16), dim3(256), 0, stream, src, v.m, v.k, cosine ? v.sd : sdp,
                       cosine ? v.sd : sdp, sdp, nt32, screen_bf16_x32_mfmas(v.sd), cosine, v.cnsqrt, prepA32);
    VQ_LAUNCH_CHECK("k_prepare_bf16_x32");
    return VQHIP_OK;
}

int launch_assign_screen_bf16(const CodebookView &cb, const AssignArgs &a, hipStream_t stream) {
    if (a.n == 0 || a.n_sub == 0) return VQHIP_OK;
    if (!(cb.prepA32 && screen_bf16_x32_supported(cb.sd, cb.k)))
        return fail(VQHIP_ERR_UNSUPPORTED, "no bf16 MFMA screen for sub_dim=%u k=%u", cb.sd, cb.k);
    {
        uint32_t nt32 = 0, groups = 0;
        screen_bf16_x32_tiling(cb.sd, cb.k, &nt32, &groups);
        // fused update (training): the same kernels with the cluster sums of the proven rows kept in LDS
        if (a.acc_sums && groups == 1 && a.metric != VQHIP_COSINE && cb.k <= 256) {
#define VQ_X32A(SDV, NTV) \
    if (cb.sd == SDV && nt32 == NTV) return launch_one_x32<SDV, NTV, 1, 0, true>(cb, a, stream);
            VQ_X32A(16, 1) VQ_X32A(16, 2) VQ_X32A(16, 3) VQ_X32A(16, 4) VQ_X32A(16, 5) VQ_X32A(16, 6) VQ_X32A(16, 7) VQ_X32A(16, 8)
            VQ_X32A(8, 1) VQ_X32A(8, 2) VQ_X32A(8, 3) VQ_X32A(8, 4) VQ_X32A(8, 5) VQ_X32A(8, 6) VQ_X32A(8, 7) VQ_X32A(8, 8)
            VQ_X32A(24, 1) VQ_X32A(24, 2) VQ_X32A(24, 3) VQ_X32A(24, 4) VQ_X32A(24, 5) VQ_X32A(24, 6) VQ_X32A(24, 7) VQ_X32A(24, 8)
#undef VQ_X32A
            return fail(VQHIP_ERR_FAILURE, "fused update requested for a shape without a fused screen (sub_dim=%u)", cb.sd);
        }
#define VQ_X32(SDV, NTV) \
    if (cb.sd == SDV && nt32 == NTV && groups == 1) return launch_one_x32<SDV, NTV>(cb, a, stream);
#define VQ_X32G(SDV, NTV, GV) \
    if (cb.sd == SDV && nt32 == NTV && groups == GV) return launch_one_x32<SDV, NTV, GV>(cb, a, stream);
        VQ_X32(16, 1) VQ_X32(16, 2) VQ_X32(16, 3) VQ_X32(16, 4) VQ_X32(16, 5) VQ_X32(16, 6) VQ_X32(16, 7) VQ_X32(16, 8)
        VQ_X32(8, 1) VQ_X32(8, 2) VQ_X32(8, 3) VQ_X32(8, 4) VQ_X32(8, 5) VQ_X32(8, 6) VQ_X32(8, 7) VQ_X32(8, 8)
        VQ_X32(12, 1) VQ_X32(12, 2) VQ_X32(12, 3) VQ_X32(12, 4) VQ_X32(12, 5) VQ_X32(12, 6) VQ_X32(12, 7) VQ_X32(12, 8)
        VQ_X32(24, 1) VQ_X32(24, 2) VQ_X32(24, 3) VQ_X32(24, 4) VQ_X32(24, 5) VQ_X32(24, 6) VQ_X32(24, 7) VQ_X32(24, 8)
        VQ_X32(32, 1) VQ_X32(32, 2) VQ_X32(32, 3) VQ_X32(32, 4) VQ_X32G(32, 4, 2)
        VQ_X32(48, 1) VQ_X32(48, 2) VQ_X32G(48, 2, 2) VQ_X32G(48, 2, 3) VQ_X32G(48, 2, 4)
        VQ_X32(64, 1) VQ_X32(64, 2) VQ_X32G(64, 2, 2) VQ_X32G(64, 2, 3) VQ_X32G(64, 2, 4)
        if (x32_padded_sd(cb.sd) == 128) return launch_wide<2>(cb, a, stream, groups);
        if (x32_padded_sd(cb.sd) == 192) return launch_wide<3>(cb, a, stream, groups);
        // padded sub_dims (narrower than the kernel's SD), full image of 8 tiles; load parts by alignment
        {
            const uint32_t sdp = x32_padded_sd(cb.sd);
            if (sdp != cb.sd) {
                const int pvw = (cb.sd % 4 == 0) ? 4 : (cb.sd % 2 == 0) ? 2 : 1;
#define VQ_X32P(SDV, NTV, GV)                                                                        \
    if (sdp == SDV && nt32 == NTV && groups == GV) {                                                 \
        if (pvw == 4) return launch_one_x32<SDV, NTV, GV, 4>(cb, a, stream);                         \
        if (pvw == 2) return launch_one_x32<SDV, NTV, GV, 2>(cb, a, stream);                         \
        return launch_one_x32<SDV, NTV, GV, 1>(cb, a, stream);                                       \
    }
                VQ_X32P(8, 8, 1) VQ_X32P(12, 8, 1) VQ_X32P(16, 8, 1) VQ_X32P(24, 8, 1)
                VQ_X32P(32, 4, 2) VQ_X32P(48, 2, 4) VQ_X32P(64, 2, 4)
                VQ_X32P(8, 4, 1) VQ_X32P(12, 4, 1) VQ_X32P(16, 4, 1) VQ_X32P(24, 4, 1)  // 64 < k <= 128
                VQ_X32P(32, 4, 1) VQ_X32P(48, 2, 2) VQ_X32P(64, 2, 2)
#undef VQ_X32P
                return fail(VQHIP_ERR_UNSUPPORTED, "no padded bf16 screen for sub_dim=%u tiles=%u groups=%u", cb.sd, nt32, groups);
            }
        }
        // k > 256: full groups, run-time count
#define VQ_X32R(SDV, NTV) \
    if (cb.sd == SDV && nt32 == NTV && cb.k > 256) return launch_one_x32<SDV, NTV, 0>(cb, a, stream, groups);
        VQ_X32R(8, 8) VQ_X32R(12, 8) VQ_X32R(16, 8) VQ_X32R(24, 8) VQ_X32R(32, 4) VQ_X32R(48, 2) VQ_X32R(64, 2)
#undef VQ_X32R
#undef VQ_X32
#undef VQ_X32G
    }
    return fail(VQHIP_ERR_UNSUPPORTED, "no bf16 screen instantiation for sub_dim=%u tiles=%u", cb.sd, cb.nt);
}

}  // namespace vqhip
