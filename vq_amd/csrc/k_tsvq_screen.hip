// TSVQ encode, all four metrics: screened descent + exact continuation.
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
// Mapping: 8 lanes per row, each lane keeps its D/8 values of y in registers for the whole
// descent; the tree's w vectors live in LDS (130 KB at depth 8, D = 128).
#include <cstdint>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace vqhip {
namespace {

constexpr int kWaves = 16;        // waves per workgroup (one workgroup per CU: the LDS holds the tree)
constexpr int kWlBuf = 64;        // work-list entries buffered per wave between flushes
constexpr int kScrL2 = 0, kScrCos = 1, kScrMan = 2;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
    return v + o;
}

// State of a row = the SLOT of the two-child node it stands at (slots number those nodes; chains
// of one-child nodes are collapsed on the host).  info[slot] = {code_l, code_r, bits(b), bits(|w|)},
// code >= 0: the child's slot, code < 0: the child ends in leaf -1-code.  `cur` >= 0: deciding at
// slot cur; cur in [-n_nodes, -1]: arrived at leaf -1-cur; cur < kFlagBase/2: undecided at slot
// cur - kFlagBase.  The w vector and the info record of a slot are fetched together (one LDS round
// trip per level) and the verdict is select-only: ~46 instructions per level for 64/LPR rows.
//
// LPR = 8 lanes share a row, so every global load instruction asks for whole 128-B lines (with 4
// lanes per row the two 64-B halves of a line arrive as separate L2 misses: measured FETCH_SIZE
// 1.7x the algorithmic bytes and 176 us instead of 137 us at C4, profiles/r1).  In step c a lane
// reads chunk (c + rot) of its row, rot chosen so that the rows inside one ds_read_b128 lane group
// ({0-3,12-15,20-27}, ... MI355X_MICROARCH.md LDS) hit different bank halves when they stand in
// different nodes.
constexpr int32_t kFlagBase = INT32_MIN;
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// reference's cosine distance from its three sequential sums (src/core/distance.rs:95-118; cnorm = sqrt(sum c^2))
__device__ __forceinline__ float cosine_from_sums(float dot, float na, float nb) {
    if (na < 1e-10f || nb < 1e-10f) return 1.0f;
    const float den = na * nb;
    const float qq = dot / den;
    const float v = 1.0f - qq;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);  // f32::clamp: NaN stays NaN
}

// COS: the two running sums are the dot products x.c_l, x.c_r (same order), the row's own squared norm is summed
// once before the walk through the same lane chain
// what the continuation reads besides the screen's own arguments (round 6: the descent kernel finishes its own undecided
// rows, see k_tsvq_screen_descend's FOLD)
struct TsvqCont {
    const float *centroids, *cnorm;
    const int32_t *left, *right, *slot_node, *node_slot;
    int euclid;
    uint32_t *clear_next;
};

// entries wl[slot], wl[slot + n_slots], ... (< count), one per 16 lanes: `wl` may point into LDS (the descent kernel's own
// list) or to the global list; every lane of the wave calls it with the same count
template <int D, int MODE>
__device__ __forceinline__ void tsvq_continue_entries(const uint2 *wl, const uint32_t count, const uint32_t slot, const uint32_t n_slots,
                                                      const uint32_t lane, const float *__restrict__ X,
                                                      const float *__restrict__ centroids, const float *__restrict__ cnorm,
                                                      const int32_t *__restrict__ left, const int32_t *__restrict__ right, int euclid,
                                                      const int32_t *__restrict__ slot_node, uint32_t d_real,
                                                      int32_t *__restrict__ leaf_out, const uint4 *__restrict__ table16,
                                                      uint4 *__restrict__ f16_out, const float *__restrict__ w_g,
                                                      const int4 *__restrict__ info_g, const int32_t *__restrict__ node_slot,
                                                      const float *__restrict__ mu_g, float R, float coef_a, float coef_b) {
    // d_real <= D: rows and centroids are d_real floats long; the pieces behind it count as zeros (a zero term
    // leaves a running sum that already holds a real term unchanged, so the reference's bits are kept)
    // w_g / info_g / node_slot: below the
    // node an entry was flagged at, every level is first put to the SAME screen test as in k_tsvq_screen_descend
    // (two sums over the entry's 16 lanes, two accumulators per lane: the summation depth D/32 + 5 the margin was
    // proven for) and only an undecided level pays the sequential chain -- 1.1 exact levels per entry instead of ~3
    constexpr int NQ = (D >= 64) ? D / 64 : 1;  // chunks of 16 lanes x V floats
    constexpr int V = D / NQ / 16;              // 2 (D = 32) or 4
    constexpr bool COS = MODE == kScrCos, MAN = MODE == kScrMan;
    const uint32_t j = lane & 15;
    for (uint32_t e0 = 0; e0 < count; e0 += n_slots) {  // wave-uniform trip count
        const uint32_t e = e0 + slot;
        const bool valid = e < count;
        const uint2 ent = valid ? wl[e] : make_uint2(0u, 0u);
        const float *px = X + (size_t)ent.x * d_real;
        float x[NQ][V];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t off = q * 16 * V + j * V;
            load_piece<V>(px + (off < d_real ? off : 0u), x[q]);
            if (off >= d_real) {
#pragma unroll
                for (int v = 0; v < V; ++v) x[q][v] = 0.0f;
            }
        }
        int32_t node = valid ? slot_node[ent.y] : 0;
        bool walking = valid;
        float na = 0.0f;
        if (COS) {
            float sa = -0.0f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll 1
                for (int hop = 0; hop < 16; ++hop) {
                    float ta = dpp_move<0x121>(sa);  // row_ror:1
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float p = x[q][v] * x[q][v];
                        ta = ta + p;
                    }
                    if (j == (uint32_t)hop) sa = ta;
                }
            }
            na = sqrtf(dpp_move<0x121>(sa));  // lane 15 -> lane 0
        }
        // screen thresholds of the row (cosine: |x|^ = 1.0001 x the reference's own f32 norm, in all 16 lanes;
        // squared L2 / Euclidean: y = x - mu and T's two row terms exactly as in k_tsvq_screen_descend)
        float t_a = 0.0f, t_b = 0.0f;
        float y[(!COS && !MAN) ? NQ : 1][V];
        if (!COS && !MAN && w_g) {
            float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float mu[V];
                load_piece<V>(mu_g + q * 16 * V + j * V, mu);  // D wide, zeros behind d_real
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    y[q][v] = x[q][v] - mu[v];
                    if (((q * V + v) & 1) == 0) q0 = fmaf(y[q][v], y[q][v], q0);
                    else q1 = fmaf(y[q][v], y[q][v], q1);
                }
            }
            float ys = q0 + q1;
            ys = dpp_add<0xB1>(ys);
            ys = dpp_add<0x4E>(ys);
            ys = dpp_add<0x141>(ys);
            ys = dpp_add<0x140>(ys);
            const float base = (__builtin_sqrtf(ys) + R) * 1.0001f;
            t_a = fmaf(5.9604644775390625e-08f * coef_a * base, base, 1e-36f);
            t_b = 5.9604644775390625e-08f * coef_b * base;
        }
        if (COS && w_g) {
            const float nb = __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane & 48u) << 2), __float_as_int(na)));
            t_a = (nb >= 1e-9f && nb <= 1e18f) ? nb * 1.0001f : __builtin_nanf("");
            t_b = 0.999f * t_a;
        }
        bool fresh = true;  // the flagged node itself: the screen has already failed there
        while (__any(walking)) {
            const int32_t l = left[node], r = right[node];
            const bool both = (l >= 0) && (r >= 0);
            int verdict = -1;  // 1 / 0: the screen proves left / right at this level
            if (!COS && !MAN && w_g) {
                const bool try_screen = walking && both && !fresh;
                if (__any(try_screen)) {
                    const int32_t sl = try_screen ? node_slot[node] : 0;
                    const float *wp = w_g + (size_t)sl * D;
                    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float wv[V];
                        load_piece<V>(wp + q * 16 * V + j * V, wv);
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            if (((q * V + v) & 1) == 0) a0 = fmaf(y[q][v], wv[v], a0);
                            else a1 = fmaf(y[q][v], wv[v], a1);
                        }
                    }
                    float acc = a0 + a1;
                    acc = dpp_add<0xB1>(acc);
                    acc = dpp_add<0x4E>(acc);
                    acc = dpp_add<0x141>(acc);
                    acc = dpp_add<0x140>(acc);
                    const int4 inf = info_g[sl];
                    const float delta = fmaf(-2.0f, acc, __int_as_float(inf.z));
                    const float T = fmaf(t_b, __int_as_float(inf.w), t_a);  // NaN / inf thresholds never pass
                    if (try_screen && fabsf(delta) > T) verdict = (delta < 0.0f) ? 1 : 0;
                }
            }
            if ((COS || MAN) && w_g) {
                const bool try_screen = walking && both && !fresh;
                if (__any(try_screen)) {
                    const int32_t sl = try_screen ? node_slot[node] : 0;
                    const float *wl_ = w_g + (size_t)sl * 2 * D, *wr_ = wl_ + D;
                    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float cl[V], cr[V];
                        const uint32_t off = q * 16 * V + j * V;  // w rows are D wide, zeros behind d_real
                        load_piece<V>(wl_ + off, cl);
                        load_piece<V>(wr_ + off, cr);
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            if (COS) {
                                if (((q * V + v) & 1) == 0) {
                                    a0 = fmaf(x[q][v], cl[v], a0);
                                    b0 = fmaf(x[q][v], cr[v], b0);
                                } else {
                                    a1 = fmaf(x[q][v], cl[v], a1);
                                    b1 = fmaf(x[q][v], cr[v], b1);
                                }
                            } else {
                                if (((q * V + v) & 1) == 0) {
                                    a0 = a0 + fabsf(x[q][v] - cl[v]);
                                    b0 = b0 + fabsf(x[q][v] - cr[v]);
                                } else {
                                    a1 = a1 + fabsf(x[q][v] - cl[v]);
                                    b1 = b1 + fabsf(x[q][v] - cr[v]);
                                }
                            }
                        }
                    }
                    auto reduce16 = [](float v) {
                        v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
                        v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
                        v = dpp_add<0x141>(v);  // row_half_mirror
                        v = dpp_add<0x140>(v);  // row_mirror
                        return v;
                    };
                    const float P_l = reduce16(a0 + a1), P_r = reduce16(b0 + b1);
                    const float mrg = __int_as_float(info_g[sl].z);  // NaN: exact-only slot
                    bool go_l, go_r;
                    if (COS) {
                        const float T = t_a * mrg, dlt = P_l - P_r;
                        go_l = (dlt > T) || (P_r < -T);
                        go_r = (-dlt > T) && (P_r > T) && (P_l < t_b);
                    } else {
                        const bool fin = (P_l < 1e37f) && (P_r < 1e37f);
                        go_l = fin && (fmaf(P_l, mrg, P_l) <= P_r * 0.99999988f);
                        go_r = fin && (fmaf(P_r, mrg, P_r) < P_l * 0.99999988f);
                    }
                    if (try_screen && (go_l || go_r)) verdict = go_l ? 1 : 0;
                }
            }
            fresh = false;
            const bool need_exact = walking && both && verdict < 0;
            if (!__any(need_exact)) {  // every entry of the wave was decided by the screen (or passes through)
                if (walking) {
                    if (both) node = verdict ? l : r;
                    else if (l >= 0) node = l;
                    else if (r >= 0) node = r;
                    else walking = false;
                }
                continue;
            }
            const float *pl = centroids + (size_t)(both ? l : 0) * d_real;
            const float *pr = centroids + (size_t)(both ? r : 0) * d_real;
            float s1[NQ][V], s2[NQ][V];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float cl[V], cr[V];
                const uint32_t off = q * 16 * V + j * V;
                const bool live = off < d_real;
                load_piece<V>(pl + (live ? off : 0u), cl);
                load_piece<V>(pr + (live ? off : 0u), cr);
                if (!live) {
#pragma unroll
                    for (int v = 0; v < V; ++v) cl[v] = cr[v] = 0.0f;
                }
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (COS) {
                        s1[q][v] = x[q][v] * cl[v];
                        s2[q][v] = x[q][v] * cr[v];
                    } else if (MAN) {
                        s1[q][v] = fabsf(x[q][v] - cl[v]);
                        s2[q][v] = fabsf(x[q][v] - cr[v]);
                    } else {
                        const float d1 = x[q][v] - cl[v], d2 = x[q][v] - cr[v];
                        s1[q][v] = d1 * d1;
                        s2[q][v] = d2 * d2;
                    }
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
            if (COS) {
                dl = cosine_from_sums(dl, na, cnorm[both ? l : 0]);
                dr = cosine_from_sums(dr, na, cnorm[both ? r : 0]);
            } else if (euclid) {
                dl = sqrtf(dl);
                dr = sqrtf(dr);
            }
            int go_left = (dl <= dr) ? 1 : 0;  // left on ties, tsvq.rs:122
            // broadcast lane 0's verdict to its 16 lanes (row_shr chain would cost more than one readlane set)
            go_left = __builtin_amdgcn_ds_bpermute((int)((lane & 48u) << 2), go_left);
            if (verdict >= 0) go_left = verdict;
            if (walking) {
                if (both) node = go_left ? l : r;
                else if (l >= 0) node = l;
                else if (r >= 0) node = r;
                else walking = false;
            }
        }
        if (valid && j == 0) leaf_out[ent.x] = node;
        if (valid && f16_out) {  // the entry's 16 lanes copy the leaf's f16 row
            const uint32_t pieces = d_real / 8;
            const uint4 *src = table16 + (size_t)node * pieces;
            uint4 *dst = f16_out + (size_t)ent.x * pieces;
            for (uint32_t q = j; q < pieces; q += 16) dst[q] = src[q];
        }
    }
}

// COS: the two running sums are the dot products x.c_l, x.c_r (same order), the row's own squared norm is summed
// once before the walk through the same lane chain
template <int D, int MODE>
__global__ __launch_bounds__(256) void k_tsvq_continue(const float *__restrict__ X,
                                                       const float *__restrict__ centroids,
                                                       const float *__restrict__ cnorm,
                                                       const int32_t *__restrict__ left,
                                                       const int32_t *__restrict__ right, int euclid,
                                                       const int32_t *__restrict__ slot_node,
                                                       const uint2 *__restrict__ wl,
                                                       const uint32_t *__restrict__ wl_count, uint32_t d_real,
                                                       int32_t *__restrict__ leaf_out, const uint4 *__restrict__ table16,
                                                       uint4 *__restrict__ f16_out, const float *__restrict__ w_g,
                                                       const int4 *__restrict__ info_g, const int32_t *__restrict__ node_slot,
                                                       const float *__restrict__ mu_g, float R, float coef_a, float coef_b,
                                                       uint32_t *__restrict__ clear_next) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *clear_next = 0u;  // the next call's counter (nobody reads or appends to it in this call)
    tsvq_continue_entries<D, MODE>(wl, *wl_count, (blockIdx.x * 256 + threadIdx.x) >> 4, (gridDim.x * 256) >> 4, threadIdx.x & 63, X,
                                   centroids, cnorm, left, right, euclid, slot_node, d_real, leaf_out, table16, f16_out, w_g, info_g,
                                   node_slot, mu_g, R, coef_a, coef_b);
}

// WAVES per workgroup: 16 while a row's registers (3 x D/8 floats) fit 128 VGPRs, 8 or 4 for the long rows
//
// COS (Distance::CosineDistance, src/core/distance.rs:95-118): a slot holds the two UNIT vectors c_l/|c_l|, c_r/|c_r|
// (c over the reference's own f32 norm of c; f64 on the host, rounded once) instead of w; per level P_l = x.c^_l and
// P_r = x.c^_r, and with T = M(slot) |x|^
//     left   iff  P_l - P_r > T  or  P_r < -T      (q_l >= q_r: 1-q and the clamp are monotone;  q_r < 0: d_r = 1 >= d_l)
//     right  iff  P_r - P_l > T and  P_r >  T and P_l < 0.999 |x|^
//                                                  (q_r - q_l > 2^-24 survives the rounding of 1-q; q_r > 0: d_r < 1; q_l < 1)
// anything else is undecided (DESIGN.md 4.4 "cosine descent").  Rows whose norm is outside [1e-9, 1e18] (the
// reference's EPSILON rule, overflow of the squared norm) or not finite make T NaN; slots with such a child carry a
// NaN margin in info.z.
//
// MAN (Distance::Manhattan, src/core/distance.rs:84-93): a slot holds the two children's centroids; both L1 distances
// are summed per level by the row's 8 lanes (the terms fl(|x - c|) are the reference's own, only the order of the
// additions differs).  All terms are non-negative, so both sums are within a RELATIVE gamma of the same exact sum:
//     left   iff  S_l (1 + m) <= S_r,      right  iff  S_r (1 + m) < S_l,      m = 2.01 (d + D/32 + 6) u
// and anything else -- or a sum that is not finite and below 1e37 -- is undecided.
// DEEP: the tree has more two-child nodes than LDS holds and the deeper ones are read from L2.  Its own instantiation:
// a global load anywhere in the descent loop makes the loop wait on vmcnt(0) at every level -- the counter the NEXT
// tile's rows (requested before the descent, to travel during it) are counted on, so every tile waited out a full HBM
// round trip before its first level.  Without the path (the usual case: depth <= 8 at d = 128) the rows travel while the
// descent runs.
// FOLD (round 6): a wave finishes its OWN undecided rows -- tsvq_continue_entries over its list in LDS, four entries at a
// time -- when its tiles are done (or the list is full), instead of flushing them to a global list for k_tsvq_continue: the
// pass is one kernel (the continuation's launch, its ramp over 3000 entries and the gap in front of it were 18 of
// 181 us at C4).  A wave holds ~0.7 entries on uniform rows; the loads of the continuation sit behind the tile loop
// (a break out of it when the list is full), so the descent's own waits are the ones it always had.
template <int D, int LPR, int WAVES, int MODE, bool DEEP, bool FOLD = false>
__global__ __launch_bounds__(WAVES * 64) void k_tsvq_screen_descend(
    const float *__restrict__ X, uint64_t n, uint32_t d_real, const float *__restrict__ w_g,
    const int4 *__restrict__ info_g, const float *__restrict__ mu_g, uint32_t n_int, int32_t start_slot, float R,
    float coef_a, float coef_b, int32_t *__restrict__ leaf_out, uint2 *__restrict__ wl,
    uint32_t *__restrict__ wl_count, const uint4 *__restrict__ table16, uint4 *__restrict__ f16_out, TsvqCont ct) {
    if (FOLD && blockIdx.x == 0 && threadIdx.x == 0) *ct.clear_next = 0u;  // the next call's counter (this call only adds to its own)
    // table16 / f16_out (optional, d_real % 8 == 0): the f16 node table [n_nodes][d_real] and the reconstruction
    // [n][d_real]; a row that reaches its leaf here is written here (its 8 lanes copy consecutive 16-byte pieces), so
    // the leaf ids do not travel through HBM to a gather kernel and back
    // d_real <= D (a multiple of 4): rows are d_real floats apart; w, mu are D wide with zeros behind d_real, and the
    // 16-byte parts of a row behind d_real are read as zeros (from a valid address), so any such d rides on the
    // next instantiated width
    // n_int: slots resident in LDS (the levels nearest the root, breadth-first); w_g / info_g hold EVERY slot of the
    // tree: a row standing at a deeper slot takes the same verdict from L2 instead (one more round trip per level)
    constexpr int CH = LPR * 4;       // floats per chunk (LPR lanes x float4)
    constexpr int NCH = D / CH;       // chunks per row = float4 values per lane
    constexpr int RPW = 64 / LPR;     // rows per wave step
    constexpr bool COS = MODE == kScrCos, MAN = MODE == kScrMan;
    constexpr int NV = (COS || MAN) ? 2 : 1;   // vectors per slot
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_w = lds;                                                       // [n_int][NV][D]
    int4 *lds_info = reinterpret_cast<int4 *>(lds + (size_t)n_int * NV * D);   // [n_int]
    uint2 *lds_wl = reinterpret_cast<uint2 *>(lds_info + n_int);               // [kWaves][kWlBuf]
    {
        constexpr uint32_t T = WAVES * 64;
        const uint32_t total = n_int * NV * (D / 4);
        for (uint32_t e0 = 0; e0 < total; e0 += 4 * T) {  // 4 loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t e = e0 + i * T + threadIdx.x;
                v[i] = reinterpret_cast<const float4 *>(w_g)[e < total ? e : 0];
            }
            // (pinned: left alone, the compiler sinks each load into the conditional LDS store below and waits for it there:
            // one memory round trip per 16 KB of the 130 KB image, ~16 us in front of every workgroup's first row)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t e = e0 + i * T + threadIdx.x;
                if (e < total) reinterpret_cast<float4 *>(lds_w)[e] = v[i];
            }
        }
        for (uint32_t e = threadIdx.x; e < n_int; e += T) lds_info[e] = info_g[e];
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t j = lane % LPR, g = lane / LPR;
    const uint32_t rot = (LPR == 4) ? ((lane >> 3) & 3u) : ((lane >> 4) & 1u);
    uint32_t off[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) off[c] = (((uint32_t)c + rot) % NCH) * CH + 4 * j;
    float4 mu[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) mu[c] = *reinterpret_cast<const float4 *>(mu_g + off[c]);
    uint2 *my_wl = lds_wl + (size_t)wave * kWlBuf;
    uint32_t wl_n = 0;  // wave-uniform

    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(wl_count, wl_n);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < wl_n) wl[base + lane] = my_wl[lane];
        wl_n = 0;
    };
    auto allreduce = [&](float v) {
        v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
        v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
        if (LPR == 8) v = dpp_add<0x141>(v);  // row_half_mirror
        return v;
    };

    const uint64_t n_tiles = (n + RPW - 1) / RPW;
    const uint64_t tile_stride = (uint64_t)gridDim.x * WAVES;
    auto load_tile = [&](uint64_t tile, float4 (&xv)[NCH]) {
        const uint64_t r = tile * RPW + g;
        const float *px = X + ((r < n) ? r : (n - 1)) * d_real;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool live = off[c] < d_real;
            const float4 v = *reinterpret_cast<const float4 *>(px + (live ? off[c] : 0u));
            xv[c] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    uint64_t tile = (uint64_t)blockIdx.x * WAVES + wave;
    float4 xn[NCH];
    if (tile < n_tiles) load_tile(tile, xn);
    for (;;) {  // (FOLD: one more trip per full list; otherwise a single trip)
    for (; tile < n_tiles; tile += tile_stride) {
        const uint64_t row = tile * RPW + g;
        float4 y[NCH];
        float ysq0 = 0.0f, ysq1 = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            y[c] = make_float4(xn[c].x - mu[c].x, xn[c].y - mu[c].y, xn[c].z - mu[c].z, xn[c].w - mu[c].w);
            ysq0 = fmaf(y[c].x, y[c].x, ysq0);
            ysq1 = fmaf(y[c].y, y[c].y, ysq1);
            ysq0 = fmaf(y[c].z, y[c].z, ysq0);
            ysq1 = fmaf(y[c].w, y[c].w, ysq1);
        }
        const float ynorm = __builtin_sqrtf(allreduce(ysq0 + ysq1));
        const float base = (ynorm + R) * 1.0001f;
        // L2: T = u * base * (coef_a * base + coef_b * |w|) + 1e-36;   cosine: T = u * coef_a * |x|^ (+ the slot's NaN flag)
        float t_a = fmaf(5.9604644775390625e-08f * coef_a * base, base, 1e-36f);
        float t_b = 5.9604644775390625e-08f * coef_b * base;
        if (COS) {  // t_a = |x|^ (>= the row's norm and the reference's f32 norm), NaN outside the screen's range
            t_a = (ynorm >= 1e-9f && ynorm <= 1e18f) ? base : __builtin_nanf("");
            t_b = 0.999f * t_a;
        }
        int32_t cur = (row < n) ? start_slot : -1;
        if (tile + tile_stride < n_tiles) load_tile(tile + tile_stride, xn);  // in flight during the descent
        for (;;) {
            const int32_t a = cur > 0 ? cur : 0;
            const bool deep = DEEP && a >= (int32_t)n_int;
            int4 inf;
            float P[NV];
            const bool any_deep = DEEP && __any(deep);
            if (!any_deep || !deep) inf = lds_info[a];
            else inf = info_g[a];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 wv[NCH];
                if (!any_deep || !deep) {  // the usual case: whole wave inside the LDS-resident levels
                    const float *wp = lds_w + ((size_t)a * NV + v) * D;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) wv[c] = *reinterpret_cast<const float4 *>(wp + off[c]);
                } else {
                    const float *wp = w_g + ((size_t)a * NV + v) * D;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) wv[c] = *reinterpret_cast<const float4 *>(wp + off[c]);
                }
                if (MAN) {
                    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;  // four chains per lane: depth NCH + 5 with the reductions
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        // the differences two at a time (v_pk_add_f32), |.| as a source modifier of the additions
                        const f32x2 d01 = f32x2{y[c].x, y[c].y} - f32x2{wv[c].x, wv[c].y};
                        const f32x2 d23 = f32x2{y[c].z, y[c].w} - f32x2{wv[c].z, wv[c].w};
                        a0 = a0 + fabsf(d01.x);
                        a1 = a1 + fabsf(d01.y);
                        a2 = a2 + fabsf(d23.x);
                        a3 = a3 + fabsf(d23.y);
                    }
                    P[v] = allreduce((a0 + a1) + (a2 + a3));
                    continue;
                }
                // register-adjacent pairs -> v_pk_fma_f32 without operand shuffles
                f32x2 acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    acc01 = __builtin_elementwise_fma(f32x2{y[c].x, y[c].y}, f32x2{wv[c].x, wv[c].y}, acc01);
                    acc23 = __builtin_elementwise_fma(f32x2{y[c].z, y[c].w}, f32x2{wv[c].z, wv[c].w}, acc23);
                }
                acc01 = acc01 + acc23;
                P[v] = allreduce(acc01.x + acc01.y);
            }
            // NaN / inf thresholds never pass
            const float T = COS ? t_a * __int_as_float(inf.z) : fmaf(t_b, __int_as_float(inf.w), t_a);
            int32_t next;
            if (MAN) {
                const float sl = P[0], sr = P[NV - 1], mrg = __int_as_float(inf.z);  // NaN margin: exact-only slot
                const bool fin = (sl < 1e37f) && (sr < 1e37f);
                const bool go_l = fin && (fmaf(sl, mrg, sl) <= sr * 0.99999988f);   // one rounding each, absorbed by the 1 % on m
                const bool go_r = fin && (fmaf(sr, mrg, sr) < sl * 0.99999988f);
                next = go_l ? inf.x : (go_r ? inf.y : (kFlagBase + a));
            } else if (COS) {
                const float dlt = P[0] - P[NV - 1];
                const bool go_l = (dlt > T) || (P[NV - 1] < -T);
                const bool go_r = (-dlt > T) && (P[NV - 1] > T) && (P[0] < t_b);  // q_l < 1: 1 - q_l stays positive
                next = go_l ? inf.x : (go_r ? inf.y : (kFlagBase + a));
            } else {
                const float delta = fmaf(-2.0f, P[0], __int_as_float(inf.z));
                const int32_t code = (delta < 0.0f) ? inf.x : inf.y;
                next = (fabsf(delta) > T) ? code : (kFlagBase + a);
            }
            cur = (cur >= 0) ? next : cur;
            if (!__any(cur >= 0)) break;
        }
        if (j == 0 && cur < 0 && cur > kFlagBase / 2 && row < n) leaf_out[row] = -1 - cur;
        if (f16_out && cur < 0 && cur > kFlagBase / 2 && row < n) {
            // the leaf's f16 row: this lane's pieces j, j + LPR, ... all requested before the first is stored (a load next to
            // its store waits for the table AND for the store in front of it: both sit on vmcnt)
            const uint32_t pieces = d_real / 8;
            const uint4 *src = table16 + (size_t)(-1 - cur) * pieces;
            uint4 *dst = f16_out + row * pieces;
            constexpr int NP = (D / 8 + LPR - 1) / LPR;
            uint4 tmp[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) tmp[i] = src[min(j + (uint32_t)i * LPR, pieces - 1u)];
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(tmp[i].x), "+v"(tmp[i].y), "+v"(tmp[i].z), "+v"(tmp[i].w));
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (j + (uint32_t)i * LPR < pieces) dst[j + (uint32_t)i * LPR] = tmp[i];
        }
        const bool push = (j == 0) && (cur <= kFlagBase / 2);
        const uint64_t mask = __ballot(push);
        if (mask) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (push) my_wl[wl_n + before] = make_uint2((uint32_t)row, (uint32_t)(cur - kFlagBase));
            wl_n += (uint32_t)__popcll(mask);
        }
        if (wl_n > kWlBuf - RPW) {
            if constexpr (FOLD) {
                tile += tile_stride;
                break;  // finish the list below, then come back for the remaining tiles
            } else {
                flush();
            }
        }
    }
    if constexpr (FOLD) {
        if (wl_n) {
            if (lane == 0) atomicAdd(wl_count, wl_n);  // statistics only (vqhip_tsvq_last_stats)
            tsvq_continue_entries<D, MODE>(my_wl, wl_n, lane >> 4, 4u, lane, X, ct.centroids, ct.cnorm, ct.left, ct.right, ct.euclid,
                                           ct.slot_node, d_real, leaf_out, table16, f16_out, ct.node_slot ? w_g : nullptr,
                                           ct.node_slot ? info_g : nullptr, ct.node_slot, mu_g, R, coef_a, coef_b);
            wl_n = 0;
        }
        if (tile >= n_tiles) break;
    } else {
        if (wl_n) flush();
        break;
    }
    }
}

// the same continuation for a dimension without a k_tsvq_continue instantiation (padded screen widths): one lane per
// entry, run-time-length loops in the reference's order; the entries are few
__global__ __launch_bounds__(256) void k_tsvq_continue_any(const float *__restrict__ X, uint32_t d,
                                                           const float *__restrict__ centroids,
                                                           const float *__restrict__ cnorm,
                                                           const int32_t *__restrict__ left,
                                                           const int32_t *__restrict__ right, int euclid, int mode,
                                                           const int32_t *__restrict__ slot_node,
                                                           const uint2 *__restrict__ wl,
                                                           const uint32_t *__restrict__ wl_count,
                                                           int32_t *__restrict__ leaf_out, const uint4 *__restrict__ table16,
                                                           uint4 *__restrict__ f16_out, uint32_t *__restrict__ clear_next) {
    if (clear_next && blockIdx.x == 0 && threadIdx.x == 0) *clear_next = 0u;
    const uint32_t count = *wl_count;
    const bool cosine = mode == kScrCos, manh = mode == kScrMan;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < count; e += gridDim.x * 256) {
        const uint2 ent = wl[e];
        const float *x = X + (size_t)ent.x * d;
        int32_t node = slot_node[ent.y];
        float na = 0.0f;
        if (cosine) {
            float sa = -0.0f;
            for (uint32_t t = 0; t < d; ++t) {
                const float p = x[t] * x[t];
                sa = sa + p;
            }
            na = sqrtf(sa);
        }
        for (;;) {
            const int32_t l = left[node], r = right[node];
            if (l >= 0 && r >= 0) {
                const float *cl = centroids + (size_t)l * d, *cr = centroids + (size_t)r * d;
                float al = -0.0f, ar = -0.0f;
                for (uint32_t t = 0; t < d; ++t) {
                    const float v = x[t];
                    float s1, s2;
                    if (cosine) {
                        s1 = v * cl[t];
                        s2 = v * cr[t];
                    } else if (manh) {
                        s1 = fabsf(v - cl[t]);
                        s2 = fabsf(v - cr[t]);
                    } else {
                        const float d1 = v - cl[t], d2 = v - cr[t];
                        s1 = d1 * d1;
                        s2 = d2 * d2;
                    }
                    al = al + s1;
                    ar = ar + s2;
                }
                const float dl = cosine ? cosine_from_sums(al, na, cnorm[l]) : (euclid ? sqrtf(al) : al);
                const float dr = cosine ? cosine_from_sums(ar, na, cnorm[r]) : (euclid ? sqrtf(ar) : ar);
                node = (dl <= dr) ? l : r;  // left on ties, tsvq.rs:122
            } else if (l >= 0) {
                node = l;
            } else if (r >= 0) {
                node = r;
            } else {
                break;
            }
        }
        leaf_out[ent.x] = node;
        if (f16_out) {
            const uint32_t pieces = d / 8;
            for (uint32_t q = 0; q < pieces; ++q) f16_out[(size_t)ent.x * pieces + q] = table16[(size_t)node * pieces + q];
        }
    }
}

template <int D, int MODE>
static int launch_continue(const float *X, const float *centroids, const float *cnorm, const int32_t *left,
                           const int32_t *right, int euclid, const TsvqScreen &s, int32_t *leaf, hipStream_t stream,
                           uint32_t d_real, const uint4 *table16, uint4 *f16_out, uint32_t *count, uint32_t *clear_next) {
    const bool rescreen = s.node_slot != nullptr;
    hipLaunchKernelGGL((k_tsvq_continue<D, MODE>), dim3(1024), dim3(256), 0, stream, X, centroids, cnorm, left, right, euclid,
                       s.slot_node, s.wl, count, d_real, leaf, table16, f16_out, rescreen ? s.w : nullptr,
                       rescreen ? s.info : nullptr, rescreen ? s.node_slot : nullptr, s.mu, s.R, s.coef_a, s.coef_b, clear_next);
    VQ_LAUNCH_CHECK("k_tsvq_continue");
    return VQHIP_OK;
}

template <int D, int LPR, int MODE, bool FOLD>
int launch_screen(const float *X, uint64_t n, uint32_t d_real, const TsvqScreen &s, hipStream_t stream, int32_t *leaf,
                  const uint4 *table16, uint4 *f16_out, uint32_t *count, const TsvqCont &ct) {
    constexpr int RPW = 64 / LPR;
    constexpr int WAVES = (D >= 512) ? 4 : (D >= 256) ? 8 : kWaves;  // 512 / 256 / 128 VGPRs per lane
    const size_t lds_bytes = tsvq_screen_lds_bytes(s.n_int, MODE == kScrL2 ? 1 : 2, D);
    static PerDeviceOnce attr_set;
    if (attr_set.needed()) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tsvq_screen_descend<D, LPR, WAVES, MODE, false, FOLD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tsvq_screen_descend<D, LPR, WAVES, MODE, true, FOLD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    const uint64_t n_tiles = (n + RPW - 1) / RPW;
    uint64_t grid = (n_tiles + WAVES - 1) / WAVES;
    if (grid > (uint64_t)num_cus()) grid = (uint64_t)num_cus();
    if (s.n_slots > s.n_int)
        hipLaunchKernelGGL((k_tsvq_screen_descend<D, LPR, WAVES, MODE, true, FOLD>), dim3((uint32_t)grid), dim3(WAVES * 64), lds_bytes, stream, X, n,
                           d_real, s.w, s.info, s.mu, s.n_int, s.start_slot, s.R, s.coef_a, s.coef_b, leaf, s.wl, count, table16, f16_out, ct);
    else
        hipLaunchKernelGGL((k_tsvq_screen_descend<D, LPR, WAVES, MODE, false, FOLD>), dim3((uint32_t)grid), dim3(WAVES * 64), lds_bytes, stream, X, n,
                           d_real, s.w, s.info, s.mu, s.n_int, s.start_slot, s.R, s.coef_a, s.coef_b, leaf, s.wl, count, table16, f16_out, ct);
    VQ_LAUNCH_CHECK("k_tsvq_screen_descend");
    return VQHIP_OK;
}

}  // namespace

// nv: vectors per slot (1: w = c_l - c_r for the L2 family; 2: the children's unit vectors for cosine)
size_t tsvq_screen_lds_bytes(uint32_t n_int, uint32_t nv, uint32_t d) {
    return (size_t)n_int * nv * d * 4 + (size_t)n_int * 16 + (size_t)kWaves * kWlBuf * 8;
}

// instantiated width that serves dimension d: d itself, or the next width up for other multiples of 4 (zero padding)
uint32_t tsvq_screen_width(uint32_t d) {
    if (d == 0 || d % 4 != 0) return 0;
    for (uint32_t w : {32u, 64u, 128u, 192u, 256u, 384u, 512u, 768u, 1024u})
        if (d <= w) return w;
    return 0;
}

bool tsvq_screen_supported(uint32_t n_int, uint32_t n_nodes, uint32_t d, int metric) {
    (void)n_nodes;
    if (metric != VQHIP_SQUARED_EUCLIDEAN && metric != VQHIP_EUCLIDEAN && metric != VQHIP_COSINE && metric != VQHIP_MANHATTAN)
        return false;
    const uint32_t dp = tsvq_screen_width(d);
    if (dp == 0 || n_int == 0) return false;
    return tsvq_screen_lds_bytes(n_int, (metric == VQHIP_COSINE || metric == VQHIP_MANHATTAN) ? 2 : 1, dp) <= 160 * 1024;
}

int launch_tsvq_screen_encode(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                              const int32_t *left, const int32_t *right, int metric, const TsvqScreen &s, int32_t *leaf,
                              hipStream_t stream, const uint16_t *table16_h, uint16_t *f16_out_h) {
    if (n == 0) return VQHIP_OK;
    // optional fused reconstruction (f16 node table -> f16 rows): needs whole 16-byte pieces
    const uint4 *table16 = reinterpret_cast<const uint4 *>(table16_h);
    uint4 *f16_out = reinterpret_cast<uint4 *>(f16_out_h);
    if (f16_out && (!table16 || d % 8 != 0 || (reinterpret_cast<uintptr_t>(f16_out_h) & 15) != 0))
        return fail(VQHIP_ERR_INVALID_INPUT, "fused f16 output needs d %% 8 == 0 and a 16-byte aligned buffer");
    if (n > 0xFFFFFFFFull) return fail(VQHIP_ERR_UNSUPPORTED, "screened TSVQ descent takes < 2^32 rows per call");
    // this call's counter (zero: cleared by the previous call's continuation kernel, or at creation) and the next call's
    uint32_t *const count = s.wl_count + s.turn, *const clear_next = s.wl_count + (s.turn ^ 1u);
    s.turn ^= 1u;
    // a launch that fails below leaves the continuation -- the kernel that clears the next call's counter -- out: both
    // counters are then cleared on the stream, so that the next call does not append behind a stale count
    const int rc = [&]() -> int {
    const int euclid = metric == VQHIP_EUCLIDEAN ? 1 : 0;
    const int mode = metric == VQHIP_COSINE ? kScrCos : metric == VQHIP_MANHATTAN ? kScrMan : kScrL2;
    const uint32_t dp = tsvq_screen_width(d);  // instantiated width serving d (d itself, or the next one up: zero padding)
    // one kernel: the descent finishes its own undecided rows (FOLD) wherever k_tsvq_continue has an instantiation for the
    // width (VQHIP_TSVQ_FOLD=0: the two-kernel form, for A/B)
    static const char *fold_env = getenv("VQHIP_TSVQ_FOLD");
    const bool fold_on = !(fold_env && fold_env[0] == '0');
    const TsvqCont ct{centroids, cnorm, left, right, s.slot_node, s.node_slot, euclid, clear_next};
#define VQ_TSVQ_DM(DV, MV)                                                                         \
    if ((DV >= 64 || d == DV) && fold_on) {                                                        \
        VQ_TRY((launch_screen<DV, 8, MV, true>(X, n, d, s, stream, leaf, table16, f16_out, count, ct)));   \
    } else {                                                                                       \
        VQ_TRY((launch_screen<DV, 8, MV, false>(X, n, d, s, stream, leaf, table16, f16_out, count, ct)));  \
        if (DV >= 64 || d == DV)                                                                   \
            VQ_TRY((launch_continue<DV, MV>(X, centroids, cnorm, left, right, euclid, s, leaf, stream, d, table16, f16_out, count, clear_next))); \
    }
#define VQ_TSVQ_D(DV)                                                                              \
    case DV:                                                                                       \
        if (mode == kScrCos) {                                                                     \
            VQ_TSVQ_DM(DV, kScrCos)                                                                \
        } else if (mode == kScrMan) {                                                              \
            VQ_TSVQ_DM(DV, kScrMan)                                                                \
        } else {                                                                                   \
            VQ_TSVQ_DM(DV, kScrL2)                                                                 \
        }                                                                                          \
        break;
    switch (dp) {
        VQ_TSVQ_D(32) VQ_TSVQ_D(64) VQ_TSVQ_D(128) VQ_TSVQ_D(192) VQ_TSVQ_D(256)
        VQ_TSVQ_D(384) VQ_TSVQ_D(512) VQ_TSVQ_D(768) VQ_TSVQ_D(1024)  // embedding widths (the reference's eval: 384)
    default: return fail(VQHIP_ERR_UNSUPPORTED, "screened TSVQ descent: d=%u", d);
    }
#undef VQ_TSVQ_D
#undef VQ_TSVQ_DM
    if (d != dp && dp < 64) {  // d < 32 (8-byte pieces in k_tsvq_continue): the run-time-length kernel
        hipLaunchKernelGGL(k_tsvq_continue_any, dim3(256), dim3(256), 0, stream, X, d, centroids, cnorm, left, right, euclid,
                           mode, s.slot_node, s.wl, count, leaf, table16, f16_out, clear_next);
        VQ_LAUNCH_CHECK("k_tsvq_continue_any");
    }
    return VQHIP_OK;
    }();
    if (rc != VQHIP_OK) {
        (void)hipMemsetAsync(s.wl_count, 0, 2 * sizeof(uint32_t), stream);
        s.turn = 0;
    }
    return rc;
}

}  // namespace vqhip
