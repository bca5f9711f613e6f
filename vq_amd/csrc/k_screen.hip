// k_screen.hip -- MFMA screen for the nearest-centroid search (gfx950, wave64).
//
// For one subspace the N x K squared-L2 matrix is  |x|^2 + (|c_j|^2 - 2 x.c_j).  The row
// constant does not move the argmin, so the screen forms  s_j = |c_j|^2 - 2 x.c_j  as a
// dense contraction on v_mfma_f32_16x16x4_f32 with |c_j|^2 as the accumulator's initial
// value (fused squared norm), and reduces it to (min, second min, argmin) in registers.
//
// An MFMA evaluates an fmaf chain, whose roundings differ from the reference's
// sequential  sum((x-c)*(x-c))  (src/core/vector.rs:135-143), so the screen does not
// decide near-ties: a row's provisional winner is final only if the second-best screened
// value exceeds the best by more than a margin T that bounds (a) the screen's own error,
// (b) the reference's rounding error and (c) the sqrt tie-collapse of Distance::Euclidean
// (src/core/distance.rs:58).  All other rows are appended to a per-subspace work list and
// re-decided by the exact kernel (k_exact.hip) over ALL centroids in the reference's
// arithmetic.  Derivation of T: DESIGN.md "screen soundness".  With i.i.d. Uniform[0,1)
// data, sub_dim 16, k 256 about 0.2 % of (row, subspace) pairs take the re-check.
//
// Work decomposition (MI355X-first, not a GEMM library tiling):
//   * one wave owns one subspace for its whole life: the subspace's codebook sits in
//     registers as MFMA A operands (K/16 tiles x sub_dim/4 k-steps VGPRs, 64 at K=256,
//     sub_dim=16) together with |c|^2 in C/D layout, so the main loop touches neither LDS
//     nor the codebook again;
//   * the wave streams 16-row tiles of its row chunk: each lane loads its B operand
//     (sub_dim/4 consecutive floats of one row) straight from HBM/L2 -- 16 B per lane for
//     sub_dim 16; the 8 waves of a workgroup walk the same rows for 8 adjacent subspaces,
//     so every 128-B line is fetched once;
//   * D[centroid][row]: a lane's column is its row (lane&15) and its 4 accumulator
//     registers are 4 centroids, so min/second-min/argmin over the 16 tiles are pure
//     per-lane VALU work on the accumulators; lanes l, l+16, l+32, l+48 are merged once per
//     tile with two cross-lane steps (wavefront-64 argmin reduction).
//
// Roofline: fp32 MFMA, 157.3 TFLOP/s.  Algorithmic work 2*K*D flop per row.
#include "kernels.hpp"

namespace vqhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kScreenWavesPerBlock = 8;
constexpr int kScreenBlock = kScreenWavesPerBlock * 64;

template <int KS>
struct BFrag {
    float v[KS];
};

template <int KS>
__device__ __forceinline__ BFrag<KS> load_b(const float *p) {
    BFrag<KS> b;
    if constexpr (KS == 1) {
        b.v[0] = p[0];
    } else if constexpr (KS == 2) {
        float2 t = *reinterpret_cast<const float2 *>(p);
        b.v[0] = t.x;
        b.v[1] = t.y;
    } else {
#pragma unroll
        for (int q = 0; q < KS; q += 4) {
            float4 t = *reinterpret_cast<const float4 *>(p + q);
            b.v[q + 0] = t.x;
            b.v[q + 1] = t.y;
            b.v[q + 2] = t.z;
            b.v[q + 3] = t.w;
        }
    }
    return b;
}

// SD = sub_dim (multiple of 4), NT = number of 16-centroid tiles (K padded to 16*NT)
template <int SD, int NT>
__global__ __launch_bounds__(kScreenBlock, 2) void k_assign_screen(
    const float *__restrict__ X, uint64_t n, uint32_t d, uint32_t m,
    const float *__restrict__ prepA, const float *__restrict__ prepCn,
    const float *__restrict__ meta, const uint32_t *__restrict__ sub_list, uint32_t n_sub,
    uint8_t *__restrict__ codes, uint32_t *__restrict__ wl_rows, uint32_t *__restrict__ wl_count,
    uint64_t wl_stride) {
    constexpr int KS = SD / 4;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t g = lane >> 4;   // k group of the MFMA operands / centroid quad of C/D
    const uint32_t p = lane & 15;   // row within the tile (column of D)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kScreenWavesPerBlock + wave;
    const uint32_t total_waves = gridDim.x * kScreenWavesPerBlock;
    const uint32_t n_chunks = total_waves / n_sub;
    if (gw >= n_chunks * n_sub) return;
    const uint32_t s = sub_list[gw % n_sub];
    const uint32_t chunk = gw / n_sub;

    const uint64_t n_tiles = (n + 15) / 16;
    const uint64_t tiles_per_chunk = (n_tiles + n_chunks - 1) / n_chunks;
    const uint64_t t0 = (uint64_t)chunk * tiles_per_chunk;
    uint64_t t1 = t0 + tiles_per_chunk;
    if (t1 > n_tiles) t1 = n_tiles;
    if (t0 >= t1) return;

    // codebook of subspace s -> registers
    float a[NT][KS];
    f32x4 cn[NT];
    {
        const float *pa = prepA + (size_t)s * NT * KS * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < KS; ++q) a[i][q] = pa[(i * KS + q) * 64];
        const float *pc = prepCn + (size_t)s * NT * 16 + 4 * g;
#pragma unroll
        for (int i = 0; i < NT; ++i) cn[i] = *reinterpret_cast<const f32x4 *>(pc + 16 * i);
    }
    // +-inf kept opaque (SGPRs): LLVM folds med3(a, b, +-inf) back into fmin/fmax otherwise
    float pinf = __builtin_inff(), ninf = -__builtin_inff();
    asm volatile("" : "+s"(pinf), "+s"(ninf));
    const float cmax = meta[s * 4 + 0];
    const float tcoef = meta[s * 4 + 1];

    const size_t col0 = (size_t)s * SD + (size_t)KS * g;
    auto row_ptr = [&](uint64_t tile) {
        uint64_t row = tile * 16 + p;
        if (row >= n) row = n - 1;  // tail: duplicate a valid row, masked at the store
        return X + row * d + col0;
    };

    BFrag<KS> b_next = load_b<KS>(row_ptr(t0));
    for (uint64_t tile = t0; tile < t1; ++tile) {
        const BFrag<KS> b = b_next;
        if (tile + 1 < t1) b_next = load_b<KS>(row_ptr(tile + 1));

        // s_j = |c_j|^2 - 2 x.c_j  for 16*NT centroids x 16 rows
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], b.v[0], cn[i], 0, 0, 0);
#pragma unroll
        for (int q = 1; q < KS; ++q)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b.v[q], acc[i], 0, 0, 0);

        // per-lane min / second min over this lane's 4*NT centroids.  min/max are written
        // as v_med3_f32 against -inf/+inf: fminf() on MFMA results makes hipcc insert a
        // canonicalising v_max_f32 x,x in front of every v_min_f32.
        float m1 = pinf, m2 = pinf;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[i][r];
                m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
                m1 = __builtin_amdgcn_fmed3f(m1, v, ninf);
            }
        // which register held the minimum (unique whenever the row is not re-checked).
        // The register number 4*i+r stays within the inline-constant range 0..63.
        // Four independent select chains (one per accumulator register r) so that compares and
        // selects of different chains can interleave (a v_cmp -> v_cndmask pair on one VCC needs
        // wait states between them).
        uint32_t cr[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = NT - 1; i >= 0; --i)
#pragma unroll
            for (int r = 3; r >= 0; --r) cr[r] = (acc[i][r] == m1) ? (uint32_t)(4 * i + r) : cr[r];
        const uint32_t c01 = cr[0] > cr[1] ? cr[0] : cr[1];
        const uint32_t c23 = cr[2] > cr[3] ? cr[2] : cr[3];
        const uint32_t creg = c01 > c23 ? c01 : c23;  // at most one chain matched (else re-checked)
        uint32_t j = ((creg & ~3u) << 2) + (creg & 3u) + 4 * g;  // 16*i + 4*g + r

        // |x|^2 over the subspace: this lane holds KS of the SD components
        float xs = 0.0f;
#pragma unroll
        for (int q = 0; q < KS; ++q) xs = fmaf(b.v[q], b.v[q], xs);

        // merge the 4 lanes that share a row: l, l^16, l^32, l^48
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float om1 = __shfl_xor(m1, off);
            const float om2 = __shfl_xor(m2, off);
            const uint32_t oj = (uint32_t)__shfl_xor((int)j, off);
            xs += __shfl_xor(xs, off);
            const float hi = __builtin_amdgcn_fmed3f(m1, om1, pinf);   // max
            const float lo2 = __builtin_amdgcn_fmed3f(m2, om2, ninf);  // min
            m2 = __builtin_amdgcn_fmed3f(lo2, hi, ninf);
            const bool take = (om1 < m1) || (om1 == m1 && oj < j);
            j = take ? oj : j;
            m1 = __builtin_amdgcn_fmed3f(m1, om1, ninf);
        }

        // margin test.  bnd >= (|x| + max|c|)^2 ; T = coef * bnd (+ denormal slack)
        const float xn = __builtin_sqrtf(xs) * 1.000001f + cmax;
        const float bnd = xn * xn;
        const float T = tcoef * bnd + 1e-37f;
        const float gap = m2 - m1;
        const bool proven = (gap > T) && (fabsf(m1) <= 3.0e38f) && (T <= 3.0e38f);

        const uint64_t row = tile * 16 + p;
        const bool writer = (g == 0) && (row < n);
        if (writer) codes[row * m + s] = (uint8_t)j;
        const bool recheck = writer && !proven;
        const unsigned long long mask = __ballot(recheck);
        if (mask != 0ull) {
            const uint32_t cnt = (uint32_t)__popcll(mask);
            uint32_t base = 0;
            if (lane == (uint32_t)(__ffsll((long long)mask) - 1)) base = atomicAdd(&wl_count[s], cnt);
            base = (uint32_t)__shfl((int)base, __ffsll((long long)mask) - 1);
            if (recheck) {
                const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                wl_rows[(size_t)s * wl_stride + base + rank] = (uint32_t)row;
            }
        }
    }
}

struct ScreenShape {
    uint32_t sd, nt;
};

template <int SD, int NT>
int launch_one(const CodebookView &cb, const AssignArgs &a, hipStream_t stream) {
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_assign_screen<SD, NT>,
                                                                kScreenBlock, 0);
    if (e != hipSuccess || occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    const uint64_t n_tiles = (a.n + 15) / 16;
    // enough waves that every subspace gets >= 1 chunk; no more chunks than tiles
    uint64_t want_waves = (uint64_t)num_cus() * occ * kScreenWavesPerBlock;
    uint64_t max_useful = n_tiles * a.n_sub;
    if (want_waves > max_useful) want_waves = max_useful;
    if (want_waves < a.n_sub) want_waves = a.n_sub;
    uint32_t blocks = (uint32_t)((want_waves + kScreenWavesPerBlock - 1) / kScreenWavesPerBlock);
    // total waves must be >= n_sub for n_chunks >= 1
    while ((uint64_t)blocks * kScreenWavesPerBlock < a.n_sub) ++blocks;
    hipLaunchKernelGGL((k_assign_screen<SD, NT>), dim3(blocks), dim3(kScreenBlock), 0, stream, a.X,
                       a.n, a.d, cb.m, cb.prepA, cb.prepCn, cb.meta, a.sub_list, a.n_sub, a.codes,
                       a.wl_rows, a.wl_count, a.wl_stride);
    VQ_LAUNCH_CHECK("k_assign_screen");
    return VQHIP_OK;
}

}  // namespace

void screen_tiling(uint32_t sd, uint32_t k, uint32_t *nt, uint32_t *ks) {
    uint32_t t = (k + 15) / 16, p = 1;
    while (p < t) p <<= 1;
    *nt = p;
    *ks = sd / 4;
}

bool screen_supported(uint32_t sd, uint32_t k) {
    if (k == 0 || k > 256) return false;
    if (!(sd == 4 || sd == 8 || sd == 16 || sd == 32)) return false;
    uint32_t nt, ks;
    screen_tiling(sd, k, &nt, &ks);
    if (sd == 32 && nt > 8) return false;  // A operands would not fit 2 waves/SIMD
    return true;
}

int launch_assign_screen(const CodebookView &cb, const AssignArgs &a, hipStream_t stream) {
    if (a.n == 0 || a.n_sub == 0) return VQHIP_OK;
    if (!screen_supported(cb.sd, cb.k) || !cb.prepA)
        return fail(VQHIP_ERR_UNSUPPORTED, "no MFMA screen for sub_dim=%u k=%u", cb.sd, cb.k);
    if (a.n >= (1ull << 32))
        return fail(VQHIP_ERR_UNSUPPORTED, "more than 2^32-1 rows per device");
    if ((a.d % 4) != 0 && cb.sd >= 4) {
        // vector loads of the B operand need 16-byte (sd>=16), 8-byte (sd=8) alignment
    }
#define VQ_SCREEN_CASE(SDV, NTV) \
    if (cb.sd == SDV && cb.nt == NTV) return launch_one<SDV, NTV>(cb, a, stream);
    VQ_SCREEN_CASE(4, 1)
    VQ_SCREEN_CASE(4, 2)
    VQ_SCREEN_CASE(4, 4)
    VQ_SCREEN_CASE(4, 8)
    VQ_SCREEN_CASE(4, 16)
    VQ_SCREEN_CASE(8, 1)
    VQ_SCREEN_CASE(8, 2)
    VQ_SCREEN_CASE(8, 4)
    VQ_SCREEN_CASE(8, 8)
    VQ_SCREEN_CASE(8, 16)
    VQ_SCREEN_CASE(16, 1)
    VQ_SCREEN_CASE(16, 2)
    VQ_SCREEN_CASE(16, 4)
    VQ_SCREEN_CASE(16, 8)
    VQ_SCREEN_CASE(16, 16)
    VQ_SCREEN_CASE(32, 1)
    VQ_SCREEN_CASE(32, 2)
    VQ_SCREEN_CASE(32, 4)
    VQ_SCREEN_CASE(32, 8)
#undef VQ_SCREEN_CASE
    return fail(VQHIP_ERR_UNSUPPORTED, "no MFMA screen instantiation for sub_dim=%u tiles=%u",
                cb.sd, cb.nt);
}

}  // namespace vqhip
