// TSVQ encode, squared-L2 / Euclidean: screened descent + exact continuation.
//
// Reference: TSVQNode::find_leaf, src/tsvq.rs:117-132 -- at every node with two children the
// row goes left iff fl(dist(x, c_l)) <= fl(dist(x, c_r)), both distances sequential un-fused
// f32 sums over all D dimensions (src/core/distance.rs:76-82).  Done literally (k_tsvq.hip,
// one lane per row) that is 2*3*D dependent VALU ops per level and a divergent centroid gather.
//
// The screen decides the same comparison from ONE dot product per level:
//     delta = d_l - d_r = (|a_l|^2 - |a_r|^2) - 2 y.(c_l - c_r),   y = x - mu, a = c - mu
// (mu = root centroid, so |y| and |a| are small and the error bound is tight), with
// w = c_l - c_r, b = |a_l|^2 - |a_r|^2 prepared per node on the host in f64.  A row continues
// while |delta^| > T(row, node) (DESIGN.md 4.4 "descent soundness"); otherwise (row, node) goes to
// a work list and k_tsvq_continue finishes it from that node in the reference's arithmetic.
//
// Mapping: 8 lanes per row (a row's 128-B pieces are read whole), each lane keeps D/8 values
// of y in registers for the whole descent; the tree's w vectors live in LDS (130 KB at depth
// 8, D = 128); lane groups 2,3 of every 32 walk the 128-B chunks in swapped order so that a
// ds_read_b128 lane group touches all 64 banks when its four rows sit in different nodes.
#include "common.hpp"
#include "kernels.hpp"

namespace vqhip {
namespace {

constexpr int kWaves = 16;        // waves per workgroup (one workgroup per CU: the LDS holds the tree)
constexpr int kWlBuf = 64;        // work-list entries buffered per wave between flushes

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
    return v + o;
}
// sum over the 8 lanes of a row group; every lane gets the same bits (each step adds a symmetric pair)
__device__ __forceinline__ float allreduce8(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    return v;
}

template <int DPL, int U>
__global__ __launch_bounds__(kWaves * 64) void k_tsvq_screen_descend(
    const float *__restrict__ X, uint64_t n, const float *__restrict__ w_g, const int4 *__restrict__ info_g,
    const float *__restrict__ mu_g, uint32_t n_nodes, uint32_t n_int, float R, float coef_a, float coef_b,
    int32_t *__restrict__ leaf_out, uint2 *__restrict__ wl, uint32_t *__restrict__ wl_count) {
    constexpr int D = DPL * 8;
    constexpr int NCH = DPL / 4;  // 128-B chunks (32 floats) per row
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_w = lds;                                                       // [n_int][D]
    int4 *lds_info = reinterpret_cast<int4 *>(lds + (size_t)n_int * D);        // [n_nodes]
    uint2 *lds_wl = reinterpret_cast<uint2 *>(lds_info + n_nodes);             // [kWaves][kWlBuf]
    for (uint32_t e = threadIdx.x; e < n_int * (D / 4); e += kWaves * 64)
        reinterpret_cast<float4 *>(lds_w)[e] = reinterpret_cast<const float4 *>(w_g)[e];
    for (uint32_t e = threadIdx.x; e < n_nodes; e += kWaves * 64) lds_info[e] = info_g[e];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t j = lane & 7, g = lane >> 3;
    const uint32_t swap = (NCH >= 2) ? ((lane >> 4) & 1u) : 0u;
    uint32_t off[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) off[q] = ((uint32_t)q ^ swap) * 32 + 4 * j;
    float4 mu[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) mu[q] = *reinterpret_cast<const float4 *>(mu_g + off[q]);
    uint2 *my_wl = lds_wl + (size_t)wave * kWlBuf;
    uint32_t wl_n = 0;  // wave-uniform

    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(wl_count, wl_n);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < wl_n) wl[base + lane] = my_wl[lane];
        wl_n = 0;
    };

    const uint64_t n_tiles = (n + 8 * U - 1) / (8 * U);
    for (uint64_t tile = (uint64_t)blockIdx.x * kWaves + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * kWaves) {
        float4 y[U][NCH];
        float base[U];
        int32_t node[U];
        uint32_t state[U];  // 0 active, 1 leaf reached, 2 undecided at `node`, 3 no row
        uint64_t row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = tile * (8 * U) + (uint64_t)u * 8 + g;
            const bool valid = row[u] < n;
            const float *px = X + (valid ? row[u] : 0) * D;
            float ysq = 0.0f;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const float4 xv = *reinterpret_cast<const float4 *>(px + off[q]);
                y[u][q] = make_float4(xv.x - mu[q].x, xv.y - mu[q].y, xv.z - mu[q].z, xv.w - mu[q].w);
                ysq = fmaf(y[u][q].x, y[u][q].x, ysq);
                ysq = fmaf(y[u][q].y, y[u][q].y, ysq);
                ysq = fmaf(y[u][q].z, y[u][q].z, ysq);
                ysq = fmaf(y[u][q].w, y[u][q].w, ysq);
            }
            ysq = allreduce8(ysq);
            base[u] = (__builtin_sqrtf(ysq) + R) * 1.0001f;
            node[u] = 0;
            state[u] = valid ? 0u : 3u;
        }
        for (;;) {
            bool any_active = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int4 inf = lds_info[node[u]];
                const int32_t l = (inf.x & 0xFFFF) - 1, r = (int32_t)((uint32_t)inf.x >> 16) - 1;
                const bool both = (l >= 0) && (r >= 0);
                const float *wp = lds_w + (size_t)(both ? inf.y : 0) * D;
                float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const float4 wv = *reinterpret_cast<const float4 *>(wp + off[q]);
                    acc0 = fmaf(y[u][q].x, wv.x, acc0);
                    acc1 = fmaf(y[u][q].y, wv.y, acc1);
                    acc0 = fmaf(y[u][q].z, wv.z, acc0);
                    acc1 = fmaf(y[u][q].w, wv.w, acc1);
                }
                const float acc = allreduce8(acc0 + acc1);
                const float delta = fmaf(-2.0f, acc, __int_as_float(inf.z));
                const float T = 5.9604644775390625e-08f * base[u] * fmaf(coef_b, __int_as_float(inf.w), coef_a * base[u]) +
                                1e-36f;
                const bool pass = fabsf(delta) > T;  // false for NaN / inf thresholds
                const bool active = state[u] == 0u;
                const bool leaf = (l < 0) && (r < 0);
                if (active) {
                    if (leaf) state[u] = 1u;
                    else if (both && !pass) state[u] = 2u;
                    else node[u] = both ? (delta < 0.0f ? l : r) : (l >= 0 ? l : r);
                }
                any_active = any_active || (state[u] == 0u);
            }
            if (!__any(any_active)) break;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j == 0 && state[u] == 1u) leaf_out[row[u]] = node[u];
            const bool push = (j == 0) && (state[u] == 2u);
            const uint64_t mask = __ballot(push);
            if (mask) {
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (push) my_wl[wl_n + before] = make_uint2((uint32_t)row[u], (uint32_t)node[u]);
                wl_n += (uint32_t)__popcll(mask);
            }
        }
        if (wl_n > kWlBuf - 8 * U) flush();
    }
    if (wl_n) flush();
}

// Finishes the undecided rows in the reference's arithmetic: entry = (row, node to resume from).
// 16 lanes per entry: lane j keeps the V-float pieces [q*16V + jV, +V) of the row, computes its
// (x-c)^2 terms for both children in parallel, and the two running sums travel lane 0 -> 15 (DPP
// row rotate) chunk after chunk, i.e. the additions happen in the reference's order t = 0..D-1
// (src/core/distance.rs:76-82) while four entries per wave progress side by side.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int V>
__device__ __forceinline__ void load_piece(const float *__restrict__ p, float *out) {
    if (V == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        out[0] = v.x, out[1] = v.y, out[2] = v.z, out[3] = v.w;
    } else {
        const float2 v = *reinterpret_cast<const float2 *>(p);
        out[0] = v.x, out[1] = v.y;
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_tsvq_continue(const float *__restrict__ X,
                                                       const float *__restrict__ centroids,
                                                       const int32_t *__restrict__ left,
                                                       const int32_t *__restrict__ right, int euclid,
                                                       const uint2 *__restrict__ wl,
                                                       const uint32_t *__restrict__ wl_count,
                                                       int32_t *__restrict__ leaf_out) {
    constexpr int NQ = (D >= 64) ? D / 64 : 1;  // chunks of 16 lanes x V floats
    constexpr int V = D / NQ / 16;              // 2 (D = 32) or 4
    const uint32_t count = *wl_count;
    const uint32_t lane = threadIdx.x & 63, j = lane & 15;
    const uint32_t slot = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const uint32_t n_slots = (gridDim.x * 256) >> 4;
    for (uint32_t e0 = 0; e0 < count; e0 += n_slots) {  // wave-uniform trip count
        const uint32_t e = e0 + slot;
        const bool valid = e < count;
        const uint2 ent = valid ? wl[e] : make_uint2(0u, 0u);
        const float *px = X + (size_t)ent.x * D + j * V;
        float x[NQ][V];
#pragma unroll
        for (int q = 0; q < NQ; ++q) load_piece<V>(px + q * 16 * V, x[q]);
        int32_t node = (int32_t)ent.y;
        bool walking = valid;
        while (__any(walking)) {
            const int32_t l = left[node], r = right[node];
            const bool both = (l >= 0) && (r >= 0);
            const float *pl = centroids + (size_t)(both ? l : 0) * D + j * V;
            const float *pr = centroids + (size_t)(both ? r : 0) * D + j * V;
            float s1[NQ][V], s2[NQ][V];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float cl[V], cr[V];
                load_piece<V>(pl + q * 16 * V, cl);
                load_piece<V>(pr + q * 16 * V, cr);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float d1 = x[q][v] - cl[v], d2 = x[q][v] - cr[v];
                    s1[q][v] = d1 * d1;
                    s2[q][v] = d2 * d2;
                }
            }
            float al = -0.0f, ar = -0.0f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll 1
                for (int hop = 0; hop < 16; ++hop) {
                    // lane `hop` receives the running sums from lane hop-1 (lane 15 -> 0 between chunks),
                    // adds its V terms in order; the other lanes compute values nobody reads
                    float tl = dpp_move<0x121>(al), tr = dpp_move<0x121>(ar);  // row_ror:1
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        tl = tl + s1[q][v];
                        tr = tr + s2[q][v];
                    }
                    if (j == (uint32_t)hop) {
                        al = tl;
                        ar = tr;
                    }
                }
            }
            // lane 15 holds both distances
            float dl = dpp_move<0x121>(al), dr = dpp_move<0x121>(ar);  // now in lane 0
            if (euclid) {
                dl = sqrtf(dl);
                dr = sqrtf(dr);
            }
            int go_left = (dl <= dr) ? 1 : 0;  // left on ties, tsvq.rs:122
            // broadcast lane 0's verdict to its 16 lanes (row_shr chain would cost more than one readlane set)
            go_left = __builtin_amdgcn_ds_bpermute((int)((lane & 48u) << 2), go_left);
            if (walking) {
                if (both) node = go_left ? l : r;
                else if (l >= 0) node = l;
                else if (r >= 0) node = r;
                else walking = false;
            }
        }
        if (valid && j == 0) leaf_out[ent.x] = node;
    }
}

template <int D>
static int launch_continue(const float *X, const float *centroids, const int32_t *left, const int32_t *right,
                           int euclid, const TsvqScreen &s, int32_t *leaf, hipStream_t stream) {
    hipLaunchKernelGGL(k_tsvq_continue<D>, dim3(1024), dim3(256), 0, stream, X, centroids, left, right, euclid, s.wl,
                       s.wl_count, leaf);
    VQ_LAUNCH_CHECK("k_tsvq_continue");
    return VQHIP_OK;
}

template <int DPL>
int launch_screen(const float *X, uint64_t n, const TsvqScreen &s, hipStream_t stream, int32_t *leaf) {
    constexpr int U = 2;
    const size_t lds_bytes = tsvq_screen_lds_bytes(s.n_int, s.n_nodes, DPL * 8);
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tsvq_screen_descend<DPL, U>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const uint64_t n_tiles = (n + 8 * U - 1) / (8 * U);
    uint64_t grid = (n_tiles + kWaves - 1) / kWaves;
    if (grid > (uint64_t)num_cus()) grid = (uint64_t)num_cus();
    hipLaunchKernelGGL((k_tsvq_screen_descend<DPL, U>), dim3((uint32_t)grid), dim3(kWaves * 64), lds_bytes, stream, X,
                       n, s.w, s.info, s.mu, s.n_nodes, s.n_int, s.R, s.coef_a, s.coef_b, leaf, s.wl, s.wl_count);
    VQ_LAUNCH_CHECK("k_tsvq_screen_descend");
    return VQHIP_OK;
}

}  // namespace

size_t tsvq_screen_lds_bytes(uint32_t n_int, uint32_t n_nodes, uint32_t d) {
    return (size_t)n_int * d * 4 + (size_t)n_nodes * 16 + (size_t)kWaves * kWlBuf * 8;
}

bool tsvq_screen_supported(uint32_t n_int, uint32_t n_nodes, uint32_t d, int metric) {
    if (metric != VQHIP_SQUARED_EUCLIDEAN && metric != VQHIP_EUCLIDEAN) return false;
    if (!(d == 32 || d == 64 || d == 128 || d == 256)) return false;
    if (n_nodes >= 65535 || n_int == 0) return false;
    return tsvq_screen_lds_bytes(n_int, n_nodes, d) <= 160 * 1024;
}

int launch_tsvq_screen_encode(const float *X, uint64_t n, uint32_t d, const float *centroids, const int32_t *left,
                              const int32_t *right, int metric, const TsvqScreen &s, int32_t *leaf,
                              hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    if (n > 0xFFFFFFFFull) return fail(VQHIP_ERR_UNSUPPORTED, "screened TSVQ descent takes < 2^32 rows per call");
    VQ_HIP(hipMemsetAsync(s.wl_count, 0, 4, stream));
    const int euclid = metric == VQHIP_EUCLIDEAN ? 1 : 0;
    switch (d) {
    case 32:
        VQ_TRY(launch_screen<4>(X, n, s, stream, leaf));
        VQ_TRY(launch_continue<32>(X, centroids, left, right, euclid, s, leaf, stream));
        break;
    case 64:
        VQ_TRY(launch_screen<8>(X, n, s, stream, leaf));
        VQ_TRY(launch_continue<64>(X, centroids, left, right, euclid, s, leaf, stream));
        break;
    case 128:
        VQ_TRY(launch_screen<16>(X, n, s, stream, leaf));
        VQ_TRY(launch_continue<128>(X, centroids, left, right, euclid, s, leaf, stream));
        break;
    case 256:
        VQ_TRY(launch_screen<32>(X, n, s, stream, leaf));
        VQ_TRY(launch_continue<256>(X, centroids, left, right, euclid, s, leaf, stream));
        break;
    default: return fail(VQHIP_ERR_UNSUPPORTED, "screened TSVQ descent: d=%u", d);
    }
    return VQHIP_OK;
}

}  // namespace vqhip
