// k_tsvq.hip -- TSVQ tree build and tree descent on gfx950.
//
// Build replaces TSVQNode::build (src/tsvq.rs:31-115): per node the MEAN of its rows, the
// per-dimension un-normalised variance, the split on the dimension of maximum variance at the
// exact median, and a stable partition (`x <= median` left, NaN right).  The reference's
// arithmetic is sequential f32 in row order (src/core/vector.rs:332-348, src/tsvq.rs:46-57),
// and the split dimension is an argmax over sums that differ by O(1/sqrt(n)) on isotropic
// data, so a different summation order would pick other dimensions and change the whole
// tree.  The column sums here therefore keep the reference order exactly: one lane owns one
// dimension of one node and adds its rows in ascending order (the stable partition keeps a
// node's rows in their original relative order).  Throughput comes from the level: all nodes
// of a level and all dimension groups run concurrently, rows are staged through LDS by 15
// loader waves per workgroup so that the single consumer wave's dependent add chain (the
// true critical path, ~5 cycles per row) never waits for HBM.
//
// Median: exact order statistics by 4-round radix select on order-preserving keys
// (f32::total_cmp order, src/tsvq.rs:75), two ranks at once for even counts (tsvq.rs:77-81).
// Partition: flags + exclusive scan + scatter (stable), src/tsvq.rs:84-85.
//
// Encode replaces find_leaf (src/tsvq.rs:117-132): one lane per row walks the tree, the two
// child distances in the reference's sequential arithmetic, left on `<=`.
//
// Rooflines: build = dependent-add latency at the top levels (N adds per dimension), HBM at
// the deep ones (2 passes x 4*N*D bytes per level); encode = HBM, 4*D in + 2*D out per row.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <type_traits>
#include <utility>
#include <vector>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr uint32_t kInactive = 0xFFFFFFFFu;
constexpr uint32_t kCsBlock = 12;  // rows the consumer of the column-sum kernel stages per register block (5 x ds_read_b128)
constexpr uint32_t kCsAhead = 3;   // tiles a loader wave keeps in flight in registers

// geometry of the plain column-sum kernel for DGT columns per workgroup (one consumer lane per column):
// a loader wave covers 64 / DGT groups of 4 consecutive rows per step
template <uint32_t DGT>
struct CsGeom {
    static constexpr uint32_t kGroups = 64 / DGT;                 // row groups per loader wave
    static constexpr uint32_t kSteps = 1;                         // load steps per wave and tile
    static constexpr uint32_t kWaveRows = 4 * kGroups * kSteps;   // rows one loader wave stages per tile
    static constexpr uint32_t kTileRows = 15 * kWaveRows;         // 120 (DGT 32) / 240 (DGT 16)
    static constexpr uint32_t kPitch = kTileRows + 4;             // 124 / 244 floats: the columns of 16 consecutive lanes start in distinct 16-byte bank groups
    static constexpr uint32_t kBlocks = kTileRows / kCsBlock;     // 6 / 12: even, so the consumer's two register blocks alternate without a copy
    static_assert(kTileRows % kCsBlock == 0 && kBlocks % 2 == 0, "consumer blocks");
};

// One per tree level, in device memory: the level loop runs without the host (the planning kernels below fill it,
// every other kernel of the level reads its counts from it and is launched over an upper bound).
struct LevelInfo {
    uint32_t first, count;    // the level's nodes are ids [first, first + count)  (children are numbered in order)
    uint32_t n_split;         // nodes that split (more than one row, depth left): lvl_split[0 .. n_split)
    uint32_t n_fast, n_slow;  // nodes by column-sum path (tile-parallel emulation / plain chain)
    uint32_t n_tiles;         // tiles of the fast nodes
    uint32_t error;           // != 0: a split dimension whose values are all NaN (the reference panics, src/tsvq.rs:77-78)
    uint32_t pad;             // batches of 64 tiles over the fast nodes (k_fs_prep's grid bound)
};

struct NodeArrays {
    uint32_t *seg_start, *seg_len, *split_dim, *nv, *nleft;
    float *median;
    uint32_t *sel_prefix, *sel_rank;  // [cap][2]
    uint32_t *child_local;            // [cap][2] level-local index of children in the NEXT level
    float *centroid;                  // [cap][d]
    float *var;                       // [cap][d]
    uint32_t fs_seg_stride;           // k_fs_*: segment summaries are [column][segment slot], this many slots per column
    float2 *sel_bin;                  // [cap] {lo, scale}: the linear bins of the median selection's first round (k_pick_split)
};

// order-preserving map f32 -> u32 (total order: -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN)
__device__ __forceinline__ uint32_t order_key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

// MODE 0: out = (sequential sum of x) / n  -> centroid;  MODE 1: out = sequential sum (x-mu)^2
//
// Workgroup = 16 waves for one (node, DGT-column group).  Wave 0 is the consumer: lane c owns column c and adds the
// node's rows in order -- a chain of dependent v_add_f32, one per row, which is the floor of this kernel (about five
// cycles per row with the staging reads).  Waves 1..15 are loaders: they gather the rows through `perm` into a
// three-tile LDS ring, column-major, so that the consumer reads four consecutive rows of its column per ds_read_b128.
//
//   * a loader lane reads FOUR CONSECUTIVE ROWS of one column (four global_load_dword, each a set of coalesced row
//     slices) and stores them with ONE ds_write_b128 -- the loaders share the consumer's SIMDs, so their VALU work is
//     kept small;
//   * kCsAhead tiles of loads are in flight per loader wave (the tile loop is unrolled by kCsAhead, so the register ring
//     needs no moves) and the row indices are read one tile earlier still: an HBM gather takes several tile times
//     (a tile of 120 rows is ~600 consumer cycles);
//   * the ring holds tile ti (being added), ti+1 (complete) and ti+2 (being written): the consumer reads the first
//     block of tile ti+1 before the barrier that ends tile ti, and within a tile it reads block b+1 before adding
//     block b, so its add chain never waits for LDS.
//   * rows past the node's end are staged as +0.0: after the first real row the running sum is never -0.0 (MODE 0
//     starts from +0.0; MODE 1 adds squares), so adding +0.0 changes no bit and every tile is whole.
template <int MODE, uint32_t DGT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_seg_colsum(const float *__restrict__ X, uint32_t d,
                                                     const uint32_t *__restrict__ perm,
                                                     const uint32_t *__restrict__ lvl_node,
                                                     const LevelInfo *__restrict__ lv, NodeArrays na) {
    using G = CsGeom<DGT>;
    constexpr uint32_t S = G::kSteps, A = kCsAhead;
    static_assert(A % 3 == 0, "the unrolled tile loop keeps the ring index static");
    __shared__ __attribute__((aligned(16))) float tile[3][DGT][G::kPitch];
    if (blockIdx.x >= lv->n_slow) return;  // launched over an upper bound
    const uint32_t node = lvl_node[blockIdx.x];
    const uint32_t a = na.seg_start[node], n = na.seg_len[node];
    if (MODE == 1 && n <= 1) return;       // a leaf: no variance pass (src/tsvq.rs:38-44)
    const uint32_t t0 = blockIdx.y * DGT;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t g = lane / DGT, c = lane % DGT;
    const uint32_t n_tiles = (n + G::kTileRows - 1) / G::kTileRows;
    const uint32_t col = min(t0 + c, d - 1);  // columns past d are computed and dropped
    float acc = (MODE == 0) ? 0.0f : -0.0f;
    if (wave == 0) __builtin_amdgcn_s_setprio(3);  // the consumer's add chain is the critical path

    // ---- loader state ----
    // variance pass: the loader lanes form (x - mean)^2 (same two roundings as tsvq.rs:47-55), the consumer only
    // carries the ordered additions
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + col] : 0.0f;
    const uint32_t wrow = (wave - 1) * G::kWaveRows + 4 * g;  // this lane's first row inside a tile (step 0)
    const float *Xc = X + col;
    uint32_t pn[S][4];        // row ids of the next tile to be requested
    float4 ring[A][S];        // tiles in flight: tile t sits in ring[t % A]
    auto load_ids = [&](uint32_t t) {
#pragma unroll
        for (uint32_t s = 0; s < S; ++s)
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t idx = min(t * G::kTileRows + wrow + s * 4 * G::kGroups + i, n - 1);  // clamped: staged as zero
                pn[s][i] = perm[a + idx];
            }
    };
    auto request = [&](float4 (&v)[S]) {
#pragma unroll
        for (uint32_t s = 0; s < S; ++s) {
            v[s].x = Xc[(size_t)pn[s][0] * d];
            v[s].y = Xc[(size_t)pn[s][1] * d];
            v[s].z = Xc[(size_t)pn[s][2] * d];
            v[s].w = Xc[(size_t)pn[s][3] * d];
        }
    };
    auto stage = [&](uint32_t buf, uint32_t t, const float4 (&v)[S]) {
        const bool whole = (t + 1) * G::kTileRows <= n;  // uniform
#pragma unroll
        for (uint32_t s = 0; s < S; ++s) {
            const uint32_t row = wrow + s * 4 * G::kGroups;
            float e[4] = {v[s].x, v[s].y, v[s].z, v[s].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 1) {
                    const float diff = e[i] - mu;
                    e[i] = diff * diff;
                }
                if (!whole && t * G::kTileRows + row + i >= n) e[i] = 0.0f;
            }
            *reinterpret_cast<float4 *>(&tile[buf][c][row]) = make_float4(e[0], e[1], e[2], e[3]);
        }
    };

    // ---- consumer state: two register blocks of kCsBlock rows ----
    float4 va[kCsBlock / 4], vb[kCsBlock / 4];
    auto read_block = [&](uint32_t buf, uint32_t r, float4 (&v)[kCsBlock / 4]) {
        const float *src = &tile[buf][c][r];
#pragma unroll
        for (uint32_t u = 0; u < kCsBlock / 4; ++u) v[u] = *reinterpret_cast<const float4 *>(src + 4 * u);
    };
    auto add_block = [&](const float4 (&v)[kCsBlock / 4]) {
#pragma unroll
        for (uint32_t u = 0; u < kCsBlock / 4; ++u) {
            acc = acc + v[u].x;
            acc = acc + v[u].y;
            acc = acc + v[u].z;
            acc = acc + v[u].w;
        }
    };

    if (wave != 0) {
        // tiles 0 .. A-1 requested, tiles 0 and 1 staged, tiles A and A+1 requested into the freed slots
        load_ids(0);
#pragma unroll
        for (uint32_t t = 0; t < A; ++t) {
            request(ring[t]);
            load_ids(t + 1);
        }
#pragma unroll
        for (uint32_t t = 0; t < 2; ++t) {
            stage(t, t, ring[t]);
            request(ring[t]);
            load_ids(A + t + 1);
        }
    }
    __syncthreads();
    if (wave == 0) read_block(0, 0, va);
    for (uint32_t ti0 = 0; ti0 < n_tiles; ti0 += A) {
#pragma unroll
        for (uint32_t j = 0; j < A; ++j) {
            const uint32_t ti = ti0 + j;  // every wave runs all A steps: the barriers stay matched
            if (wave != 0) {
                const uint32_t slot = (j + 2) % A, buf = (j + 2) % 3;
                stage(buf, ti + 2, ring[slot]);  // tile ti+2, requested A tiles ago
                request(ring[slot]);             // tile ti+2+A
                load_ids(ti + 3 + A);
            } else if (ti < n_tiles) {
                const uint32_t buf = j % 3, nbuf = (j + 1) % 3;
#pragma unroll
                for (uint32_t b = 0; b < G::kBlocks; b += 2) {
                    read_block(buf, (b + 1) * kCsBlock, vb);
                    add_block(va);
                    if (b + 2 < G::kBlocks) read_block(buf, (b + 2) * kCsBlock, va);
                    else read_block(nbuf, 0, va);  // tile ti+1 is complete since the last barrier
                    add_block(vb);
                }
            }
            __syncthreads();
        }
    }
    if (wave == 0 && t0 + c < d && g == 0) {
        if (MODE == 0) na.centroid[(size_t)node * d + t0 + c] = acc / (float)n;  // T::from_usize(n)
        else na.var[(size_t)node * d + t0 + c] = acc;
    }
}

// split dimension: NaN filtered, LAST maximum wins (Iterator::max_by), none -> 0 (tsvq.rs:59-66)
// one wave per node: lane l scans dimensions l, l + 64, ... in ascending order (later wins a tie), the lanes are
// merged with "larger value, then larger dimension" -- the same winner as the sequential scan
__global__ __launch_bounds__(256) void k_pick_split(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv, uint32_t d, NodeArrays na) {
    const uint32_t li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (li >= lv->n_split) return;
    const uint32_t node = lvl_node[li];
    const float *v = na.var + (size_t)node * d;
    uint32_t best_t = 0;
    int have = 0;
    float best = 0.0f;
    for (uint32_t t = lane; t < d; t += 64) {
        const float x = v[t];
        if (x != x) continue;
        if (!have || !(x < best)) {
            best = x;
            best_t = t;
            have = 1;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const uint32_t ot = (uint32_t)__shfl_xor((int)best_t, off);
        const int oh = __shfl_xor(have, off);
        const bool take = oh && (!have || ob > best || (ob == best && ot > best_t));
        if (take) {
            best = ob;
            best_t = ot;
            have = 1;
        }
    }
    if (lane == 0) {
        na.split_dim[node] = have ? best_t : 0u;
        na.nv[node] = na.seg_len[node];  // non-NaN count: k_gather_vals subtracts the NaNs
        na.sel_prefix[2 * node] = na.sel_prefix[2 * node + 1] = 0;
        // The median lies within one standard deviation of the mean (|mean - median| <= sigma, whatever the distribution):
        // the selection's first round cuts [mean - 1.01 sigma, mean + 1.01 sigma] into 2046 equal bins (+ one on either
        // side).  Any monotone map of the values is a valid first digit; this one leaves ~n / 1000 candidates where the top
        // eleven bits of a float -- sign, exponent, two mantissa bits -- leave an eighth of the rows.
        const float mean = na.centroid[(size_t)node * d + (have ? best_t : 0u)];
        const float sigma = sqrtf(best / (float)na.seg_len[node]);
        const bool ok = have && sigma > 0.0f && sigma < 3.0e38f && mean == mean && fabsf(mean) < 3.0e38f;
        na.sel_bin[node] = ok ? make_float2(mean - 1.01f * sigma, 2046.0f / (2.02f * sigma)) : make_float2(0.0f, 0.0f);
    }
}
// first digit of the median selection: 0 below the window, 1 .. 2046 inside, 2047 above (monotone in x; scale 0: one bin)
__device__ __forceinline__ uint32_t sel_bin_of(float x, float2 par) {
    if (!(par.y > 0.0f)) return 1u;
    const float t = (x - par.x) * par.y;
    return t < 0.0f ? 0u : (t >= 2046.0f ? 2047u : 1u + (uint32_t)t);
}

// one radix-select round: histogram of the byte at `shift` among keys matching the prefix.
// Positions are grouped by node, so a workgroup's contiguous chunk touches very few nodes:
// histograms are privatised in LDS (8 slots keyed by the level-local node index, claimed with
// a CAS) and flushed once; the rare slot collision falls back to a global atomic.  Unprivatised,
// the root level is 1M atomics on 512 words (0.64 ms per round).
constexpr uint32_t kHistSlots = 8;    // 8-bit rounds (four of them): eight privatised node histograms per workgroup
constexpr uint32_t kHistSlots11 = 2;  // 11-bit rounds (three: 11 + 11 + 10 bits): two (2 x 2 x 2048 words = 32 KB of LDS)
// GATHER: the first round (shift 24).  The values are not there yet: vals[i] = X[perm[i]][split_dim(node of i)] is formed
// here (and written for the later rounds, the median test and the partition), NaNs are taken off nv[node] (k_pick_split
// set it to the segment length); no prefix has been chosen yet, so every key counts for both ranks.
// BITS: 8 (rounds at shift 24 / 16 / 8 / 0) or 11 (shift 21 / 10 / 0, the last one 10 bits wide: its top bins stay empty)
template <bool GATHER, int BITS, bool LIN = false>
__global__ __launch_bounds__(256) void k_select_hist(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                     float *__restrict__ vals, uint32_t n, uint32_t chunk,
                                                     const uint32_t *__restrict__ node_of,
                                                     const uint32_t *__restrict__ remap,
                                                     const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                     uint32_t shift, uint32_t *__restrict__ hist, uint32_t width) {
    // width: key bits of THIS round (<= BITS: the histograms keep 2^BITS bins, a narrower round leaves the top ones empty)
    constexpr uint32_t kBins = 1u << BITS, kSlots = BITS == 8 ? kHistSlots : kHistSlots11;
    __shared__ uint32_t tags[kSlots];
    __shared__ uint32_t lh[kSlots][2][kBins];
    for (uint32_t e = threadIdx.x; e < kSlots * 2 * kBins; e += 256) (&lh[0][0][0])[e] = 0u;
    if (threadIdx.x < kSlots) tags[threadIdx.x] = kInactive;
    __syncthreads();
    const uint32_t hi_mask = (shift + width >= 32) ? 0u : (0xFFFFFFFFu << (shift + width)), bin_mask = (1u << width) - 1u;
    const uint32_t i0 = blockIdx.x * chunk;
    const uint32_t i1 = min(n, i0 + chunk);
    // Eight rows per thread and trip, every stage of the chain row -> node -> split dimension -> value as eight independent
    // loads (indices clamped, nothing under an `if`): one row at a time the kernel was five dependent memory round trips
    // per row, eight rows in series per thread -- 41 us for a pass that moves 130 MB.
    constexpr uint32_t kPer = 8;
    for (uint32_t r0 = i0; r0 < i1; r0 += 256 * kPer) {
        uint32_t ic[kPer], li[kPer], node[kPer];
        bool live[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = r0 + threadIdx.x + 256 * u;
            live[u] = i < i1;
            ic[u] = min(i, i1 - 1u);
            li[u] = node_of[ic[u]];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t r = remap[li[u] == kInactive ? 0u : li[u]];  // level-local node index -> index among the level's split nodes
            li[u] = (li[u] == kInactive) ? kInactive : r;
            live[u] = live[u] && li[u] != kInactive;
        }
        // (a row outside every split node reads entry 0 of the tables -- words nobody may have written -- and then stands at
        // node 0, dimension 0: valid addresses for the loads behind; its value is dropped)
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t nd = lvl_node[live[u] ? li[u] : 0u];
            node[u] = live[u] ? nd : 0u;
        }
        float x[kPer];
        if (GATHER) {
            uint32_t dim[kPer], pr[kPer];
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) {
                const uint32_t dm = na.split_dim[node[u]];
                dim[u] = live[u] ? dm : 0u, pr[u] = perm[ic[u]];
            }
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) x[u] = X[(size_t)pr[u] * d + dim[u]];
        } else {
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) x[u] = vals[ic[u]];
        }
        float2 par[kPer];
        uint32_t pf0[kPer], pf1[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            if (LIN) par[u] = na.sel_bin[node[u]];
            if (!GATHER) pf0[u] = na.sel_prefix[2 * node[u]], pf1[u] = na.sel_prefix[2 * node[u] + 1];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            if (!live[u]) continue;
            if (GATHER) {
                vals[ic[u]] = x[u];
                if (x[u] != x[u]) atomicSub(&na.nv[node[u]], 1u);  // nv starts at seg_len (k_pick_split); NaNs are rare
            }
            if (x[u] != x[u]) continue;
            const uint32_t key = order_key(x[u]);
            const uint32_t slot = li[u] & (kSlots - 1);
            uint32_t owner = tags[slot];
            if (owner == kInactive) {
                const uint32_t old = atomicCAS(&tags[slot], kInactive, li[u]);
                owner = (old == kInactive) ? li[u] : old;
            }
            const uint32_t bin = LIN ? sel_bin_of(x[u], par[u]) : (key >> shift) & bin_mask;  // LIN: k_pick_split's linear bins
#pragma unroll
            for (uint32_t sel = 0; sel < 2u; ++sel)
                if (GATHER || (key & hi_mask) == (sel ? pf1[u] : pf0[u])) {
                    if (owner == li[u]) atomicAdd(&lh[slot][sel][bin], 1u);
                    else atomicAdd(&hist[((size_t)li[u] * 2 + sel) * kBins + bin], 1u);
                }
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < kSlots * 2 * kBins; e += 256) {
        const uint32_t slot = e / (2 * kBins), rest = e % (2 * kBins);
        const uint32_t c = (&lh[0][0][0])[e];
        const uint32_t li = tags[slot];
        if (c != 0 && li != kInactive) atomicAdd(&hist[(size_t)li * 2 * kBins + rest], c);
    }
}

// one wave per (node, which of the two ranks): lane l owns bins 4l..4l+3, a wave prefix scan finds
// the bin holding the rank (the serial walk over 256 dependent loads cost 25 us per launch)
// FIRST (shift 24): the ranks of the two order statistics the median needs come from nv (tsvq.rs:77-81).
template <bool FIRST, int BITS>
__global__ __launch_bounds__(64) void k_select_pick(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv,
                                                    NodeArrays na, uint32_t shift, uint32_t *__restrict__ hist) {
    constexpr uint32_t kBins = 1u << BITS, kPer = kBins / 64, kV = kPer / 4;  // bins per lane (4 or 32), as 16-byte vectors
    const uint32_t idx = blockIdx.x, lane = threadIdx.x;
    if (idx >= lv->n_split * 2) return;
    const uint32_t li = idx >> 1, sel = idx & 1;
    const uint32_t node = lvl_node[li];
    uint4 *h4 = reinterpret_cast<uint4 *>(hist + ((size_t)li * 2 + sel) * kBins) + (size_t)lane * kV;
    uint4 c[kV];
#pragma unroll
    for (uint32_t v = 0; v < kV; ++v) c[v] = h4[v];
#pragma unroll
    for (uint32_t v = 0; v < kV; ++v) h4[v] = make_uint4(0u, 0u, 0u, 0u);  // ready for the next round
    uint32_t rank;
    if (FIRST) {
        const uint32_t nv = na.nv[node], h = nv / 2;
        rank = (nv == 0) ? 0u : (sel == 0 ? ((nv % 2 == 0) ? h - 1 : h) : h);
    } else {
        rank = na.sel_rank[2 * node + sel];
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t v = 0; v < kV; ++v) mine += c[v].x + c[v].y + c[v].z + c[v].w;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += up;
    }
    const uint32_t excl = incl - mine;
    // first bin b with rank < cumulative(b); the serial rule stops at the last bin if none does
    const bool here = (rank >= excl) && (rank < incl);
    const uint64_t m = __ballot(here);
    uint32_t b = kBins - 1, new_rank;
    if (na.nv[node] == 0) {
        b = 0;
        new_rank = rank;
    } else if (m) {
        const int src = __builtin_ctzll(m);
        uint32_t r = rank - excl, bb = kPer * lane;
        bool found = false;
#pragma unroll
        for (uint32_t v = 0; v < kV; ++v) {
            const uint32_t w4[4] = {c[v].x, c[v].y, c[v].z, c[v].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool last = (v == kV - 1) && (q == 3);
                if (!found && (r < w4[q] || last)) found = true;
                else if (!found) r -= w4[q], ++bb;
            }
        }
        b = (uint32_t)__shfl((int)bb, src);
        new_rank = (uint32_t)__shfl((int)r, src);
    } else {  // rank beyond every bin: the walk ends at the last bin with the counts of the bins in front removed
        const uint32_t total = (uint32_t)__shfl((int)incl, 63), last = (uint32_t)__shfl((int)c[kV - 1].w, 63);
        new_rank = rank - (total - last);
    }
    if (lane == 0) {
        na.sel_rank[2 * node + sel] = new_rank;
        if (FIRST) na.sel_prefix[2 * node + sel] = b << shift;
        else na.sel_prefix[2 * node + sel] |= b << shift;
    }
}

// Behind the first round (k_pick_split's 2048 linear bins around the mean) the keys that can still be the node's two
// middle order statistics are the few in the chosen bin: instead of two more histogram rounds over all rows (hist +
// pick twice: four launches per level) they are COLLECTED, per (node, rank), into the node's own stretch of a scratch
// array (k_select_collect: positions from LDS counters per workgroup, one global atomic per workgroup and list to reserve
// a range) and k_select_final picks the rank among them with four 8-bit rounds inside one workgroup (the candidates staged
// in LDS when they fit).  Same two keys as the radix rounds give -- the k-th smallest is the k-th smallest; values that
// pile up in one bin (a constant column, a lattice, half the rows at zero) only make the lists long: every row of the
// node at worst, which is what they are sized for (one workgroup then streams the node's keys four times).
__global__ __launch_bounds__(256) void k_select_collect(const float *__restrict__ vals, uint32_t n, uint32_t chunk,
                                                        const uint32_t *__restrict__ node_of, const uint32_t *__restrict__ remap,
                                                        const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                        uint32_t *__restrict__ cand0, uint32_t *__restrict__ cand1,
                                                        uint32_t *__restrict__ cnt) {
    constexpr uint32_t kSlots = 2, kPer = 8;  // privatised counters: two nodes per workgroup (a third: straight to the global counter)
    __shared__ uint32_t tags[kSlots], lcount[kSlots][2], lbase[kSlots][2];
    if (threadIdx.x < kSlots) tags[threadIdx.x] = kInactive, lcount[threadIdx.x][0] = lcount[threadIdx.x][1] = 0u;
    __syncthreads();
    const uint32_t i0 = blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    for (uint32_t r0 = i0; r0 < i1; r0 += 256 * kPer) {  // (one trip: the chunks are ~2000 rows)
        uint32_t key[kPer], slot_of[kPer], lpos[kPer][2], a_of[kPer], mt[kPer];  // mt: bit sel = in rank sel's bin; bit 2 = privatised
        // (the chain row -> node -> bins as stages of eight independent loads, indices clamped: k_select_hist)
        uint32_t ic[kPer], li[kPer], node[kPer];
        bool live[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = r0 + threadIdx.x + 256 * u;
            live[u] = i < i1;
            ic[u] = min(i, i1 - 1u);
            li[u] = node_of[ic[u]];
        }
        float x[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t r = remap[li[u] == kInactive ? 0u : li[u]];
            li[u] = (li[u] == kInactive) ? kInactive : r;
            live[u] = live[u] && li[u] != kInactive;
            x[u] = vals[ic[u]];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t nd = lvl_node[live[u] ? li[u] : 0u];
            node[u] = live[u] ? nd : 0u;  // (k_select_hist: node 0 for a row outside every split node)
        }
        float2 par[kPer];
        uint32_t pf0[kPer], pf1[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            par[u] = na.sel_bin[node[u]], pf0[u] = na.sel_prefix[2 * node[u]], pf1[u] = na.sel_prefix[2 * node[u] + 1], a_of[u] = na.seg_start[node[u]];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            mt[u] = 0u, key[u] = 0u, slot_of[u] = 0u, lpos[u][0] = lpos[u][1] = 0u;
            if (!live[u] || x[u] != x[u]) continue;
            const uint32_t bin = sel_bin_of(x[u], par[u]) << 21;  // (k_select_pick<true, 11> left the chosen bins at bit 21)
            const uint32_t m = ((bin == pf0[u]) ? 1u : 0u) | ((bin == pf1[u]) ? 2u : 0u);
            if (!m) continue;
            const uint32_t k = order_key(x[u]), slot = li[u] & (kSlots - 1);
            uint32_t owner = tags[slot];
            if (owner == kInactive) {
                const uint32_t old = atomicCAS(&tags[slot], kInactive, li[u]);
                owner = (old == kInactive) ? li[u] : old;
            }
            key[u] = k;
            if (owner == li[u]) {
                mt[u] = m | 4u;
                slot_of[u] = slot;
                if (m & 1u) lpos[u][0] = atomicAdd(&lcount[slot][0], 1u);
                if (m & 2u) lpos[u][1] = atomicAdd(&lcount[slot][1], 1u);
            } else {  // a third node in this workgroup's chunk: its own global atomics
                if (m & 1u) cand0[a_of[u] + atomicAdd(&cnt[2 * li[u]], 1u)] = k;
                if (m & 2u) cand1[a_of[u] + atomicAdd(&cnt[2 * li[u] + 1], 1u)] = k;
            }
        }
        __syncthreads();
        if (threadIdx.x < kSlots * 2) {
            const uint32_t slot = threadIdx.x >> 1, sel = threadIdx.x & 1u, c = lcount[slot][sel];
            lbase[slot][sel] = c ? atomicAdd(&cnt[2 * tags[slot] + sel], c) : 0u;
            lcount[slot][sel] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (mt[u] & 4u) {
                if (mt[u] & 1u) cand0[a_of[u] + lbase[slot_of[u]][0] + lpos[u][0]] = key[u];
                if (mt[u] & 2u) cand1[a_of[u] + lbase[slot_of[u]][1] + lpos[u][1]] = key[u];
            }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_select_final(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv, NodeArrays na,
                                                       const uint32_t *__restrict__ cand0, const uint32_t *__restrict__ cand1,
                                                       uint32_t *__restrict__ cnt) {
    constexpr uint32_t kStage = 8192;
    __shared__ uint32_t keys[kStage], h[256], pick_bin, pick_rank;
    const uint32_t idx = blockIdx.x;
    if (idx >= lv->n_split * 2) return;
    const uint32_t li = idx >> 1, sel = idx & 1, node = lvl_node[li];
    const uint32_t c = cnt[2 * li + sel];
    __syncthreads();
    if (threadIdx.x == 0) cnt[2 * li + sel] = 0u;  // ready for the next level
    if (na.nv[node] == 0) {  // (every value NaN: the keys stay 0, as the radix rounds leave them)
        if (threadIdx.x == 0) na.sel_prefix[2 * node + sel] = 0u;
        return;
    }
    const uint32_t *src = (sel ? cand1 : cand0) + na.seg_start[node];
    const bool staged = c <= kStage;
    if (staged)
        for (uint32_t e = threadIdx.x; e < c; e += 1024) keys[e] = src[e];
    uint32_t prefix = 0, rank = na.sel_rank[2 * node + sel];  // key bits chosen so far; rank among the candidates that carry them
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) h[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t above = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t e = threadIdx.x; e < c; e += 1024) {
            const uint32_t k = staged ? keys[e] : src[e];
            if ((k & above) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave, four bins per lane; the serial rule: first bin whose cumulative count passes the rank, else the last
            const uint32_t l = threadIdx.x;
            const uint32_t cb[4] = {h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]};
            const uint32_t mine = cb[0] + cb[1] + cb[2] + cb[3];
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
                if ((int)l >= off) incl += up;
            }
            const uint32_t excl = incl - mine;
            const bool here = rank >= excl && rank < incl;
            const uint64_t m = __ballot(here);
            if (m) {
                if (here) {
                    uint32_t r = rank - excl, q = 0;
                    while (q < 3u && r >= cb[q]) r -= cb[q], ++q;
                    pick_bin = 4 * l + q;
                    pick_rank = r;
                }
            } else if (l == 63) {
                pick_bin = 255u;
                pick_rank = rank - (incl - cb[3]);
            }
        }
        __syncthreads();
        prefix |= pick_bin << shift;
        rank = pick_rank;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        na.sel_rank[2 * node + sel] = rank;
        na.sel_prefix[2 * node + sel] = prefix;
    }
}

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, off);
        if ((int)lane >= off) v += u;
    }
    return v;
}
// partition flags (left = value <= the node's median, NaN -> right; tsvq.rs:84-85) and their exclusive scan inside
// blocks of 1024 positions; block totals for k_scan_sums.  (The median of a node is formed here, by every thread that
// needs it, from the two selected keys: one launch instead of k_median + k_flags + k_scan_blocks.)
__global__ __launch_bounds__(256) void k_flags_scan(const float *__restrict__ vals, uint32_t n,
                                                    const uint32_t *__restrict__ node_of,
                                                    const uint32_t *__restrict__ remap,
                                                    const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                    uint32_t *__restrict__ flags, uint32_t *__restrict__ out,
                                                    uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t sh[8];
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], s = 0;
    // the chain position -> node -> (count, the two selected keys, segment start) as stages of four independent loads,
    // clamped and never under a test (k_select_hist); a position outside every split node stands at node 0 and gets flag 0
    uint32_t ic[4], li[4], node[4];
    bool in[4], live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        in[q] = base + q < n;
        ic[q] = min(base + (uint32_t)q, n - 1u);
        li[q] = node_of[ic[q]];
    }
    float x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t r = remap[li[q] == kInactive ? 0u : li[q]];
        li[q] = (li[q] == kInactive) ? kInactive : r;
        live[q] = in[q] && li[q] != kInactive;
        x[q] = vals[ic[q]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t nd = lvl_node[live[q] ? li[q] : 0u];
        node[q] = live[q] ? nd : 0u;
    }
    uint32_t nvv[4], k0[4], k1[4], st[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        nvv[q] = na.nv[node[q]], k0[q] = na.sel_prefix[2 * node[q]], k1[q] = na.sel_prefix[2 * node[q] + 1], st[q] = na.seg_start[node[q]];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t f = 0;
        if (live[q]) {
            float med = 0.0f;  // tsvq.rs:77-81 from the two selected keys
            if (nvv[q] != 0u) {
                const float lo = key_to_float(k0[q]), hi = key_to_float(k1[q]);
                if (nvv[q] % 2 == 0) {
                    const float s2 = lo + hi;
                    med = s2 / 2.0f;
                } else {
                    med = hi;
                }
            }
            f = (x[q] <= med) ? 1u : 0u;  // NaN -> right
            if (base + (uint32_t)q == st[q]) na.median[node[q]] = med;
        }
        if (in[q]) flags[base + q] = f;
        v[q] = f;
        s += f;
    }
    // exclusive scan of the thread totals over the workgroup's four waves
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(s, lane);
    if (lane == 63u) sh[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, wtot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t t = sh[w];
        wbase += (w < wave) ? t : 0u;
        wtot += t;
    }
    uint32_t excl = wbase + incl - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (base + q < n) out[base + q] = excl;
        excl += v[q];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = wtot;
}
// (body shared with the fused per-level planner k_plan_fused: one workgroup of 1024 threads, `sh` its 4 KB of LDS)
// exclusive scan of one value per thread over the workgroup (1024 threads); returns the total.  Wave scans through
// ds_bpermute plus one scan of the 16 wave totals: three barriers (the 10-step LDS scan it replaces had thirty, and the
// planner runs five such scans per level: 23 us per level for what is a few hundred additions)
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *sh, uint32_t *total) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63u) sh[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const uint32_t w = lane < 16u ? sh[lane] : 0u;
        const uint32_t wi = wave_incl_scan_u32(w, lane);
        if (lane < 16u) sh[32 + lane] = wi - w;  // exclusive offset of wave `lane`
        if (lane == 15u) sh[48] = wi;
    }
    __syncthreads();
    const uint32_t base = sh[32 + wave];
    *total = sh[48];
    __syncthreads();  // sh is free again for the caller's next scan
    return base + incl - v;
}

// What one phase of the single-workgroup planning hands the next stays in LDS when it fits (k_plan_fused): a value read
// back from memory costs a fence (an L2 write-back on this chip) and a round trip per phase.
constexpr uint32_t kPlanSb = 4096, kPlanNext = 2048, kPlanFast = 1024;
struct PlanLds {
    uint32_t *sb;              // [kPlanSb] scanned block totals (nb <= kPlanSb)
    uint32_t *nstart, *nlen;   // [kPlanNext] the next level's segments by level-local index (count <= kPlanNext)
    uint32_t *fnode, *fstart, *flen, *ftile, *fbat;  // [kPlanFast] the level's long nodes by fast index: id, segment, first tile, first batch
};
__device__ __forceinline__ uint32_t *sh_tb(const PlanLds *pl) { return pl->ftile; }
__device__ inline void scan_sums_body(uint32_t *__restrict__ block_sums, uint32_t nb, uint32_t *sh, uint32_t *sb = nullptr) {
    // single workgroup; nb <= a few thousand: serial chunks of 1024
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sh, &tot);
        if (i < nb) {
            block_sums[i] = carry + ex;
            if (sb && i < kPlanSb) sb[i] = carry + ex;
        }
        carry += tot;
    }
}
// stable partition of every split node's segment (tsvq.rs:84-85)
__global__ __launch_bounds__(256) void k_scatter(uint32_t n, const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ node_of,
                                                 const uint32_t *__restrict__ remap,
                                                 const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                 const uint32_t *__restrict__ Pb, const uint32_t *__restrict__ bsums,
                                                 const uint32_t *__restrict__ flags,
                                                 uint32_t *__restrict__ perm2, uint32_t *__restrict__ node_of2) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t li = node_of[i];
    if (li != kInactive) li = remap[li];
    if (li == kInactive) {
        perm2[i] = perm[i];
        node_of2[i] = kInactive;
        return;
    }
    const uint32_t node = lvl_node[li];
    const uint32_t a = na.seg_start[node];
    // exclusive scan of the flags = the in-block scan + the scanned block totals (k_flags_scan, k_scan_sums)
    const uint32_t lr = (Pb[i] + bsums[i >> 10]) - (Pb[a] + bsums[a >> 10]);
    const uint32_t f = flags[i];
    const uint32_t pos = f ? (a + lr) : (a + na.nleft[node] + (i - a - lr));
    perm2[pos] = perm[i];
    node_of2[pos] = na.child_local[2 * node + (f ? 0 : 1)];
}

// The whole initial state of a build in ONE launch (it was k_iota + five memsets + three 4-byte uploads from the host's
// stack and the synchronisation they needed: ~70 us in front of the first useful kernel): identity permutation, every row in
// node 0; no children anywhere; the level table {first 0, count 1} then zeros; the root's segment; diagnostics and
// radix-select histograms at zero.  Grid-stride over the largest of the ranges.
__global__ __launch_bounds__(256) void k_build_init(uint32_t *__restrict__ perm, uint32_t *__restrict__ node_of, uint32_t n,
                                                    int32_t *__restrict__ left, int32_t *__restrict__ right, uint32_t dcap,
                                                    uint32_t *__restrict__ lv_words, uint32_t n_lv_words, uint32_t *__restrict__ seg_start,
                                                    uint32_t *__restrict__ seg_len, uint32_t *__restrict__ fb, uint32_t n_fb,
                                                    uint32_t *__restrict__ hist, uint32_t n_hist) {
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) perm[i] = i, node_of[i] = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < dcap; i += stride) left[i] = -1, right[i] = -1;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_lv_words; i += stride) lv_words[i] = (i == 1u) ? 1u : 0u;  // LevelInfo[0] = {first 0, count 1, ...}
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_fb; i += stride) fb[i] = 0u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_hist; i += stride) hist[i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) seg_start[0] = 0u, seg_len[0] = n;
}

// ---- encode ------------------------------------------------------------------------------
__device__ float dist_rt(int metric, const float *__restrict__ a, const float *__restrict__ b, uint32_t n) {
    if (metric == VQHIP_SQUARED_EUCLIDEAN || metric == VQHIP_EUCLIDEAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            const float diff = a[t] - b[t];
            const float sq = diff * diff;
            acc = acc + sq;
        }
        return metric == VQHIP_EUCLIDEAN ? sqrtf(acc) : acc;
    }
    if (metric == VQHIP_MANHATTAN) {
        float acc = -0.0f;
        for (uint32_t t = 0; t < n; ++t) {
            const float diff = a[t] - b[t];
            acc = acc + fabsf(diff);
        }
        return acc;
    }
    float dot = -0.0f, sa = -0.0f, sb = -0.0f;
    for (uint32_t t = 0; t < n; ++t) {
        const float p = a[t] * b[t];
        dot = dot + p;
    }
    for (uint32_t t = 0; t < n; ++t) {
        const float p = a[t] * a[t];
        sa = sa + p;
    }
    for (uint32_t t = 0; t < n; ++t) {
        const float p = b[t] * b[t];
        sb = sb + p;
    }
    const float na = sqrtf(sa), nb = sqrtf(sb);
    return vq_cosine_finish(metric, dot, na, nb);
}

__global__ __launch_bounds__(256) void k_tsvq_descend(const float *__restrict__ X, uint64_t n, uint32_t d,
                                                      const float *__restrict__ centroids,
                                                      const int32_t *__restrict__ left,
                                                      const int32_t *__restrict__ right, int metric,
                                                      int32_t *__restrict__ leaf_out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *x = X + i * d;
    int32_t node = 0;
    for (;;) {
        const int32_t l = left[node], r = right[node];
        if (l >= 0 && r >= 0) {
            const float dl = dist_rt(metric, x, centroids + (size_t)l * d, d);
            const float dr = dist_rt(metric, x, centroids + (size_t)r * d, d);
            node = (dl <= dr) ? l : r;  // left on ties, tsvq.rs:122
        } else if (l >= 0) {
            node = l;
        } else if (r >= 0) {
            node = r;
        } else {
            break;
        }
    }
    leaf_out[i] = node;
}

// Fast descent: a workgroup stages RB rows transposed in LDS (xs[t][i], conflict-free for the
// per-lane sequential walks over t), then every lane walks the tree for its row.  Both child
// distances are evaluated in one pass over t (two independent dependent-add chains), the
// child centroids come from L1/L2 as float4 (the whole tree is <= 261 KB at depth 8).
template <int METRIC, int RB>
__global__ __launch_bounds__(RB) void k_tsvq_descend_lds(const float *__restrict__ X, uint64_t n, uint32_t d,
                                                         const float *__restrict__ centroids,
                                                         const float *__restrict__ cnorm,
                                                         const int32_t *__restrict__ left,
                                                         const int32_t *__restrict__ right,
                                                         int32_t *__restrict__ leaf_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [d][RB]
    const uint64_t row0 = (uint64_t)blockIdx.x * RB;
    const uint32_t rows = (uint32_t)min((uint64_t)RB, n - row0);
    // coalesced stage: consecutive threads read consecutive floats of the row block
    for (uint32_t e = threadIdx.x; e < rows * d; e += RB) {
        const uint32_t i = e / d, t = e - i * d;
        xs[t * RB + i] = X[row0 * d + e];
    }
    __syncthreads();
    const uint32_t i = threadIdx.x;
    if (i >= rows) return;
    const float *x = xs + i;
    float na = 0.0f;
    if (vq_is_cos(METRIC)) {
        float sa = -0.0f;
        for (uint32_t t = 0; t < d; ++t) {
            const float v = x[t * RB];
            const float p = v * v;
            sa = sa + p;
        }
        na = sqrtf(sa);
    }
    int32_t node = 0;
    for (;;) {
        const int32_t l = left[node], r = right[node];
        if (l >= 0 && r >= 0) {
            const float4 *cl = reinterpret_cast<const float4 *>(centroids + (size_t)l * d);
            const float4 *cr = reinterpret_cast<const float4 *>(centroids + (size_t)r * d);
            float al = -0.0f, ar = -0.0f;
            for (uint32_t t4 = 0; t4 < d / 4; ++t4) {
                const float4 a4 = cl[t4], b4 = cr[t4];
                const float ca[4] = {a4.x, a4.y, a4.z, a4.w}, cb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = x[(t4 * 4 + u) * RB];
                    if (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
                        const float d1 = v - ca[u], d2 = v - cb[u];
                        const float s1 = d1 * d1, s2 = d2 * d2;
                        al = al + s1;
                        ar = ar + s2;
                    } else if (METRIC == VQHIP_MANHATTAN) {
                        const float d1 = v - ca[u], d2 = v - cb[u];
                        al = al + fabsf(d1);
                        ar = ar + fabsf(d2);
                    } else {
                        const float p1 = v * ca[u], p2 = v * cb[u];
                        al = al + p1;
                        ar = ar + p2;
                    }
                }
            }
            float dl, dr;
            if (METRIC == VQHIP_EUCLIDEAN) {
                dl = sqrtf(al);
                dr = sqrtf(ar);
            } else if (vq_is_cos(METRIC)) {
                dl = vq_cosine_finish(METRIC, al, na, cnorm[l]);
                dr = vq_cosine_finish(METRIC, ar, na, cnorm[r]);
            } else {
                dl = al;
                dr = ar;
            }
            node = (dl <= dr) ? l : r;
        } else if (l >= 0) {
            node = l;
        } else if (r >= 0) {
            node = r;
        } else {
            break;
        }
    }
    leaf_out[row0 + i] = node;
}

// The same walk with the row in REGISTERS (d = 32, 64, 96, 128): the LDS version holds 320 rows per CU at d = 128,
// one wave per SIMD, and every level starts with a dependent chain of global loads (children, their centroids, for
// cosine their norms) that nothing hides; with 128 + ~40 VGPRs three waves share a SIMD.  A lane reads its own row
// (consecutive 16-byte parts of a line are asked for by consecutive instructions, so the L1 serves 7 of 8).
template <int METRIC, int D>
__global__ __launch_bounds__(256) void k_tsvq_descend_reg(const float *__restrict__ X, uint64_t n,
                                                          const float *__restrict__ centroids,
                                                          const float *__restrict__ cnorm,
                                                          const int32_t *__restrict__ left,
                                                          const int32_t *__restrict__ right,
                                                          int32_t *__restrict__ leaf_out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x[D];
    {
        const float4 *px = reinterpret_cast<const float4 *>(X + i * D);
#pragma unroll
        for (int t4 = 0; t4 < D / 4; ++t4) {
            const float4 v = px[t4];
            x[4 * t4] = v.x, x[4 * t4 + 1] = v.y, x[4 * t4 + 2] = v.z, x[4 * t4 + 3] = v.w;
        }
    }
    float na = 0.0f;
    if (vq_is_cos(METRIC)) {
        float sa = -0.0f;
#pragma unroll
        for (int t = 0; t < D; ++t) {
            const float p = x[t] * x[t];
            sa = sa + p;
        }
        na = sqrtf(sa);
    }
    int32_t node = 0;
    for (;;) {
        const int32_t l = left[node], r = right[node];
        if (l >= 0 && r >= 0) {
            const float4 *cl = reinterpret_cast<const float4 *>(centroids + (size_t)l * D);
            const float4 *cr = reinterpret_cast<const float4 *>(centroids + (size_t)r * D);
            float al = -0.0f, ar = -0.0f;
#pragma unroll
            for (int t4 = 0; t4 < D / 4; ++t4) {
                const float4 a4 = cl[t4], b4 = cr[t4];
                const float ca[4] = {a4.x, a4.y, a4.z, a4.w}, cb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = x[4 * t4 + u];
                    if (METRIC == VQHIP_SQUARED_EUCLIDEAN || METRIC == VQHIP_EUCLIDEAN) {
                        const float d1 = v - ca[u], d2 = v - cb[u];
                        const float s1 = d1 * d1, s2 = d2 * d2;
                        al = al + s1;
                        ar = ar + s2;
                    } else if (METRIC == VQHIP_MANHATTAN) {
                        const float d1 = v - ca[u], d2 = v - cb[u];
                        al = al + fabsf(d1);
                        ar = ar + fabsf(d2);
                    } else {
                        const float p1 = v * ca[u], p2 = v * cb[u];
                        al = al + p1;
                        ar = ar + p2;
                    }
                }
            }
            float dl, dr;
            if (METRIC == VQHIP_EUCLIDEAN) {
                dl = sqrtf(al);
                dr = sqrtf(ar);
            } else if (vq_is_cos(METRIC)) {
                dl = vq_cosine_finish(METRIC, al, na, cnorm[l]);
                dr = vq_cosine_finish(METRIC, ar, na, cnorm[r]);
            } else {
                dl = al;
                dr = ar;
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
    leaf_out[i] = node;
}

// sqrt(sum c^2) per node: cosine's norm_b depends on the node only (src/core/distance.rs:109)
__global__ void k_tsvq_node_norms(const float *__restrict__ centroids, uint32_t n_nodes, uint32_t d,
                                  float *__restrict__ cnorm) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_nodes) return;
    float sb = -0.0f;
    for (uint32_t t = 0; t < d; ++t) {
        const float v = centroids[(size_t)j * d + t];
        const float p = v * v;
        sb = sb + p;
    }
    cnorm[j] = sqrtf(sb);
}

__global__ __launch_bounds__(256) void k_tsvq_gather_f16(const float *__restrict__ centroids, uint32_t d,
                                                         const int32_t *__restrict__ leaf, uint64_t n,
                                                         uint16_t *__restrict__ out) {
    const uint64_t total = n * d;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint64_t row = e / d;
        const uint32_t t = (uint32_t)(e - row * d);
        out[e] = __half_as_ushort(__float2half_rn(centroids[(size_t)leaf[row] * d + t]));
    }
}

// ---- tile-parallel EXACT emulation of the reference's sequential f32 column sums -------------
// acc = acc + v over the rows of a node, in row order, is one chain of len dependent rounded
// additions (mean: src/core/vector.rs:332-348, variance: src/tsvq.rs:47-55); at the top levels
// of a 1M-row build that chain (10 cycles per row) is the whole cost.  While the running sum s
// stays inside one binade [2^e, 2^(e+1)), ulp(s) = 2^(e-23) is constant and, with S = s/ulp an
// integer and x/ulp = a + f (a = floor, 0 <= f < 1),
//     fl(s + x) / ulp = S + a + (f > 1/2 ? 1 : f == 1/2 ? (S + a) odd : 0)       (round to nearest even)
// -- a transducer whose only state is the parity of S.  Such maps compose associatively
// ((delta for even S, delta for odd S), plus the min/max prefix needed to know whether the sum
// really stayed in the binade), so the chain is cut into tiles of 512 rows that are summarised
// IN PARALLEL under a guessed binade (from an f64 prefix of plain tile sums), and a short
// sequential pass per column then walks the tile summaries with the exact S, checks the guess
// (exponent of s, S + min > 2^23 strictly, S + max < 2^24) and re-adds a tile row by row whenever the
// check fails (binade crossings, the first tiles, cancellation, NaN/inf).  Every result is the
// reference's bit pattern; only the schedule differs.
constexpr uint32_t kFsTile = 512;        // rows per tile: the unit of the f64 binade guess
constexpr uint32_t kFsSeg = 64;                          // rows per segment summary = rows of one re-addition
constexpr uint32_t kFsBlk = 128;                         // rows per wave item of k_fs_fold (two segments)
constexpr uint32_t kFsSegsPerTile = kFsTile / kFsSeg;    // 8
constexpr uint32_t kFsBlksPerTile = kFsTile / kFsBlk;    // 4
constexpr uint32_t kFsCols = 32;         // columns per workgroup (one 128-byte line per row)
constexpr uint32_t kFsMinRows = 16384;   // shorter nodes always keep the plain sequential kernel (most tiles of a short node sit on a binade crossing and are re-added)

struct FsTile {
    uint32_t node, t;      // node id, tile index inside the node
    uint32_t start, rows;  // position of the tile's first row in perm, rows in the tile (<= kFsTile)
};

template <int MODE>
__device__ __forceinline__ float fs_value(float x, float mu) {
    if (MODE == 0) return x;
    const float diff = x - mu;
    return diff * diff;
}

// plain f64 sums per (tile, column): only used to guess the binade of the running sum at a tile.
// samp > 1: only every samp-th group of 32 rows is read and the sum scaled up -- 1/samp of the traffic for a guess whose
// relative error (~ sigma/mu / sqrt(rows read so far)) only moves the tiles next to a binade crossing into the
// re-addition path of k_fs_chain; the sums themselves stay exact whatever the guess.
// Is a guess from every r-th row good enough?  Its error is ~ sigma sqrt(r N) against a sum of ~ |mean| N: fine for
// columns with |mean| >= sigma (non-negative features), useless for zero-mean columns, whose running sum is a random
// walk no larger than that error (measured on N(0,1) data: 1.14 M re-added tiles and 31 ms per build with sampling,
// 0.52 M and 20.7 ms without).  One workgroup per block of 32 columns looks at 2048 evenly spaced rows once per build
// and allows sampling for the block iff every column of it has |mean| >= sigma.
__global__ __launch_bounds__(1024) void k_fs_policy(const float *__restrict__ X, uint32_t n, uint32_t d, uint32_t *__restrict__ policy) {
    __shared__ double p1[32][kFsCols], p2[32][kFsCols];
    __shared__ int all_ok;
    const uint32_t cl = threadIdx.x & 31u, c = blockIdx.x * kFsCols + cl, part = threadIdx.x >> 5;
    const uint32_t n_s = min(n, 2048u);
    double s1 = 0.0, s2 = 0.0;
    if (c < d) {
        for (uint32_t j0 = part; j0 < n_s; j0 += 32 * 8) {  // 8 loads in flight per thread
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t j = j0 + 32 * u;
                v[u] = (j < n_s) ? X[(size_t)(((uint64_t)j * n) / n_s) * d + c] : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                s1 += (double)v[u];
                s2 += (double)v[u] * (double)v[u];
            }
        }
    }
    p1[part][cl] = s1;
    p2[part][cl] = s2;
    if (threadIdx.x == 0) all_ok = 1;
    __syncthreads();
    if (threadIdx.x < kFsCols && c < d) {
        double a = 0.0, b = 0.0;
        for (int g = 0; g < 32; ++g) {
            a += p1[g][threadIdx.x];
            b += p2[g][threadIdx.x];
        }
        const double m = a / n_s, var = b / n_s - m * m;
        if (!(m * m >= var)) atomicAnd(&all_ok, 0);  // NaN: no sampling
    }
    __syncthreads();
    if (threadIdx.x == 0) policy[blockIdx.x] = (uint32_t)all_ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_fs_tile_sums(const float *__restrict__ X, uint32_t d,
                                                      const uint32_t *__restrict__ perm,
                                                      const FsTile *__restrict__ tiles, NodeArrays na,
                                                      double *__restrict__ tile_sum, uint32_t samp_arg,
                                                      const uint32_t *__restrict__ policy,
                                                      const LevelInfo *__restrict__ lv) {
    __shared__ double part[32][kFsCols + 1];
    if (blockIdx.x >= lv->n_tiles * ((d + kFsCols - 1) / kFsCols)) return;  // launched over an upper bound
    // 1-D grid, column block fastest: the d/32 workgroups that share a tile's rows (and DRAM pages) run together
    // (d is a multiple of 4; the last column block may be short: its missing 16-byte parts are skipped)
    const uint32_t ncb = (d + kFsCols - 1) / kFsCols, tile_id = blockIdx.x / ncb, cblk = blockIdx.x - tile_id * ncb;
    const FsTile tl = tiles[tile_id];
    const uint32_t samp = (policy && !policy[cblk]) ? 1u : samp_arg;  // k_fs_policy: every row for walk-like columns
    const uint32_t c0 = cblk * kFsCols, q = threadIdx.x & 7, rr = threadIdx.x >> 3;
    const bool col_ok = c0 + 4 * q < d;
    float mu[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && col_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mu[i] = na.centroid[(size_t)tl.node * d + c0 + 4 * q + i];
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // (row ids, then rows, as batches of eight independent loads -- clamped, nothing under an `if`: with the loads inside
    // the test every row was two dependent round trips waited for in turn)
    const uint32_t cq = col_ok ? c0 + 4 * q : c0, last = tl.rows - 1u;
    for (uint32_t i0 = 0; i0 < kFsTile / 32; i0 += 8 * samp) {
        uint32_t pr[8];
        float4 v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) pr[u] = perm[tl.start + min(rr + 32 * (i0 + u * samp), last)];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(X + (size_t)pr[u] * d + cq);
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u * samp;
            if (i < kFsTile / 32 && rr + 32 * i < tl.rows && col_ok) {
                acc[0] += (double)fs_value<MODE>(v[u].x, mu[0]);
                acc[1] += (double)fs_value<MODE>(v[u].y, mu[1]);
                acc[2] += (double)fs_value<MODE>(v[u].z, mu[2]);
                acc[3] += (double)fs_value<MODE>(v[u].w, mu[3]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[rr][4 * q + i] = acc[i];
    __syncthreads();
    if (threadIdx.x < kFsCols && c0 + threadIdx.x < d) {
        double s = 0.0;
        for (int r = 0; r < 32; ++r) s += part[r][threadIdx.x];
        if (samp > 1) {  // rows read: whole groups of 32 except possibly the last one of the tile
            uint32_t read = 0;
            for (uint32_t i = 0; i < kFsTile / 32; i += samp)
                if (32 * i < tl.rows) read += min(32u, tl.rows - 32 * i);
            s *= (double)tl.rows / (double)read;
        }
        tile_sum[(size_t)tile_id * d + c0 + threadIdx.x] = s;
    }
}

// exclusive prefix over the tiles of a node, per column (f64 sums in, f32 prefix out): 32 chunk lanes x 32 columns per
// workgroup (chunk sums -> LDS -> offsets -> prefix); only a guess is needed, so f64 order is free.
// VAR: the variance pass's guess needs no pass over the rows -- the mean pass left S1 = sum x and S2 = sum x^2 per (tile,
// column), and sum over the tile of (x - mu)^2 = S2 - 2 mu S1 + rows mu^2, in f64 (the f32 block sums behind S1 / S2 make
// it good to ~1e-7 (1 + mu^2 / sigma^2) relative: ample for a binade guess unless the offset dwarfs the spread, where
// it merely costs re-additions).
constexpr uint32_t kFsPrefCols = 8, kFsPrefChunks = 1024 / kFsPrefCols;
template <bool VAR>
__global__ __launch_bounds__(1024) void k_fs_prefix(uint32_t d, const uint32_t *__restrict__ fast_nodes, const uint32_t *__restrict__ tile_base,
                                                    const uint32_t *__restrict__ n_tiles_of, NodeArrays na, const double *__restrict__ tile_sum,
                                                    const double2 *__restrict__ tile_mom,
                                                    float *__restrict__ tile_pref, const LevelInfo *__restrict__ lv,
                                                    uint32_t *__restrict__ side_count) {
    // 8 columns x 128 chunks of tiles per workgroup (it was 32 x 32: at the root that is d / 32 = 4 workgroups walking
    // 61 tiles per thread twice -- 45 us of memory round trips on an otherwise idle chip)
    __shared__ double part[kFsPrefChunks][kFsPrefCols + 1];
    // the side buffer's slot counter of this pass (k_fs_fold hands slots out; the previous pass's chain is done)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *side_count = 0u;
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t cl = threadIdx.x % kFsPrefCols, c = blockIdx.y * kFsPrefCols + cl, lt = threadIdx.x / kFsPrefCols;
    const uint32_t base = tile_base[blockIdx.x], nt = n_tiles_of[blockIdx.x];
    const uint32_t chunk = (nt + kFsPrefChunks - 1) / kFsPrefChunks, t0 = min(nt, lt * chunk), t1 = min(nt, t0 + chunk);
    const bool col_ok = c < d;
    const uint32_t node = fast_nodes[blockIdx.x], len = na.seg_len[node];
    const double mu = (VAR && col_ok) ? (double)na.centroid[(size_t)node * d + c] : 0.0;
    auto tile_v = [&](uint32_t t) -> double {
        if (!VAR) return tile_sum[(size_t)(base + t) * d + c];
        const double2 m = tile_mom[(size_t)(base + t) * d + c];
        const double rows = (double)min(kFsTile, len - t * kFsTile);
        return m.y - 2.0 * mu * m.x + rows * mu * mu;
    };
    // (eight independent loads at a time: the loop is a chain of memory round trips otherwise -- 36 us at the root of 1M rows)
    double local = 0.0;
    if (col_ok)
        for (uint32_t t = t0; t < t1; t += 8) {
            double v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) v[u] = (t + u < t1) ? tile_v(t + u) : 0.0;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) local += v[u];
        }
    part[lt][cl] = local;
    __syncthreads();
    if (!col_ok) return;
    double run = 0.0;
    for (uint32_t q = 0; q < lt; ++q) run += part[q][cl];
    for (uint32_t t = t0; t < t1; t += 8) {
        double v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) v[u] = (t + u < t1) ? tile_v(t + u) : 0.0;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            if (t + u < t1) tile_pref[(size_t)(base + t + u) * d + c] = (float)run;  // only the binade is wanted
            run += v[u];
        }
    }
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// LDS exchange inside ONE wave (k_fs_fold and k_fs_chain run one-wave workgroups): the LDS serves a wave's instructions
// in order, so all that is needed is that the compiler keeps the order and waits for the LDS counter.  __syncthreads()
// would also wait for every global load in flight (s_waitcnt vmcnt(0)) -- the rows of the next block, the next batches
// of summaries, the parked addends: exactly the loads these kernels issue early to hide their latency.
// a * b + c on 24-bit signed factors in ONE full-rate instruction (v_mad_i32_i24)
__device__ __forceinline__ int32_t fs_mad24(int32_t a, int32_t b, int32_t c) {
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a value the compiler must keep in a vector register (it cannot prove it uniform any more)
__device__ __forceinline__ uint32_t fs_vgpr(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void fs_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}

// ---- 64-row segment summaries (round 3) -----------------------------------------------------------------------------
// A tile of 512 rows stays the unit of the f64 binade guess (k_fs_tile_sums / k_fs_prefix*), but the transducer
// summaries are kept per SEGMENT of 64 rows, each under its own guess, and the chain re-adds 64 rows -- not 512 -- when
// a summary does not hold.  The failing fraction goes with the square root of the rows a summary spans and the cost
// of a re-addition with the rows themselves: zero-mean columns, whose running sum is a random walk that keeps
// re-crossing binade edges (45-63 % of the 512-row tiles failed, 2.4 us each: 20.5 ms per build at 1M x 128), fail
// ~10 % of their segments at 0.3 us each.
//
// A summary is 16 bytes, [segment][column] (a wave writes whole 512-byte runs): {d, lo, hi, ef} of the stream for an even
// incoming S; ef = bit 0 unusable, bit 1 "the odd stream differs" (a run with an exact tie: that stream sits in the
// second array, same index), bits 2-9 the binade + 128, bits 10.. the side-buffer slot + 1 of a parked segment.
struct FsS {
    int32_t d, lo, hi, ef;
};
__device__ __forceinline__ int32_t fs_ef(int bad, int two, int e, int slot) {
    return (bad ? 1 : 0) | (two ? 2 : 0) | (((e + 128) & 0xFF) << 2) | ((slot + 1) << 10);
}
__device__ __forceinline__ int fs_ef_e(int32_t ef) { return ((ef >> 2) & 0xFF) - 128; }
__device__ __forceinline__ int fs_ef_slot(int32_t ef) { return (int)(((uint32_t)ef >> 10) & 0x1FFFFFu) - 1; }
// (bit 31: the sign of the guess -- with the binade, the raw-bit range the segment's incoming sums lie in; k_fs_prep)
__device__ __forceinline__ int32_t fs_ef_key(int32_t ef) { return ((ef >> 2) & 0xFF) | (int32_t)(((uint32_t)ef >> 31) << 8); }

struct FsT {  // parity transducer of a run of rows: stream 0 for an even incoming S, stream 1 for an odd one
    int32_t d0, d1, lo0, lo1, hi0, hi1;
};
// f first, then g (scalars throughout: an array indexed by the parity becomes a scratch access)
__device__ __forceinline__ FsT fs_compose(const FsT &f, const FsT &g) {
    FsT h;
    const bool o0 = (f.d0 & 1) != 0, o1 = ((1 + f.d1) & 1) != 0;
    const int32_t gd0 = o0 ? g.d1 : g.d0, gl0 = o0 ? g.lo1 : g.lo0, gh0 = o0 ? g.hi1 : g.hi0;
    const int32_t gd1 = o1 ? g.d1 : g.d0, gl1 = o1 ? g.lo1 : g.lo0, gh1 = o1 ? g.hi1 : g.hi0;
    h.d0 = f.d0 + gd0;
    h.lo0 = min(f.lo0, f.d0 + gl0);
    h.hi0 = max(f.hi0, f.d0 + gh0);
    h.d1 = f.d1 + gd1;
    h.lo1 = min(f.lo1, f.d1 + gl1);
    h.hi1 = max(f.hi1, f.d1 + gh1);
    return h;
}

// Segment summaries of all rows of the level's long nodes, in one streamed read of the rows (through `perm`), with
// nothing but 8 KB of partial summaries in LDS: one wave per task = one 512-row tile x 32 columns (one 128-byte line
// per row), walked in four blocks of 128 rows.
//   * lane (g, q) = (lane / 8, lane % 8) loads the 16-byte part q of the 16 CONSECUTIVE rows 16 g .. 16 g + 15 of a block
//     (the eight lanes of a row ask for one whole line), keeps them in registers and folds each of its 4 columns over
//     its 16 rows into the parity transducer; the four lanes that hold a segment's rows of a column leave their runs in
//     LDS (one array per field: 16-byte stores, no bank conflicts) and lane (segment, column) composes them in row order;
//   * the binade guess of a segment: the tile's f64 prefix plus the EXACT f32 sum of the tile's rows in front of the
//     segment -- the wave carries it from block to block, and a cross-lane reduction supplies the block's first half;
//   * the rows of block i + 1 travel while block i is folded, the row indices of block i + 2 behind them; waves are
//     independent (one-wave workgroups, no workgroup barrier), eight per CU with 16 KB in flight each.
// The 512 x 32 tile staged column-major in LDS (round 2) ran two workgroups per CU at 2.4-3.0 TB/s.
template <int MODE>
__global__ __launch_bounds__(64, 2) void k_fs_fold(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                   const FsTile *__restrict__ tiles, const LevelInfo *__restrict__ lv, NodeArrays na,
                                                   const float *__restrict__ tile_pref, FsS *__restrict__ summ,
                                                   FsS *__restrict__ summ_odd, float *__restrict__ side, uint32_t side_cap,
                                                   uint32_t *__restrict__ side_count, double2 *__restrict__ tile_mom,
                                                   float park_rel_arg, const uint32_t *__restrict__ policy,
                                                   uint4 *__restrict__ side_meta, uint32_t pass_tag) {
    __shared__ __attribute__((aligned(16))) int32_t p_d0[8][kFsCols], p_d1[8][kFsCols], p_lo0[8][kFsCols], p_lo1[8][kFsCols],
        p_hi0[8][kFsCols], p_hi1[8][kFsCols], p_bad[8][kFsCols], p_gb[8][kFsCols];
    __shared__ __attribute__((aligned(16))) float p_s1[8][kFsCols], p_s2[8][kFsCols];
    __shared__ int pslot[2][kFsCols];
    // the tile's summaries [segment of the tile][column] (rows padded to 33 x 16 bytes: the write-out reads eight rows at
    // once), written out per column when the tile is done: 128 contiguous bytes each (summaries are [column][segment])
    __shared__ __attribute__((aligned(16))) FsS t_sum[kFsSegsPerTile][kFsCols + 1];
    const uint32_t ncb = (d + kFsCols - 1) / kFsCols;  // the last column block may be short (d % 4 == 0)
    const uint32_t n_tasks = lv->n_tiles * ncb;
    const uint32_t lane = threadIdx.x, q = lane & 7u, g = lane >> 3;  // load / fold role
    const uint32_t cl = lane & 31u, sg = lane >> 5;                   // compose role: column cl of segment sg
    if (blockIdx.x >= n_tasks) return;
    // this wave's tasks are blockIdx.x + k gridDim.x (column block fastest: the waves that share rows run together);
    // local item i = 4 k + b is block b of its k-th task
    const uint32_t my_tasks = (n_tasks - blockIdx.x + gridDim.x - 1) / gridDim.x, n_local = my_tasks * kFsBlksPerTile;
    struct Where {
        FsTile tl;
        uint32_t tile_id, b, c0, cq;
    };
    auto where = [&](uint32_t i) {
        Where w;
        const uint32_t task = blockIdx.x + (i / kFsBlksPerTile) * gridDim.x;
        w.b = i % kFsBlksPerTile;
        w.tile_id = task / ncb;
        w.c0 = (task - w.tile_id * ncb) * kFsCols;
        w.tl = tiles[w.tile_id];
        // a short last column block: the missing parts read the block's first part instead (valid addresses, values
        // never used: their summaries are not written)
        w.cq = (w.c0 + 4 * q < d) ? w.c0 + 4 * q : w.c0;
        return w;
    };
    uint32_t prow[16];
    float4 vn[16];  // rows in flight
    float4 mu_n = make_float4(0.f, 0.f, 0.f, 0.f), pref_n = mu_n;  // one 16-byte load each, used as loaded (a conversion or a
                                                                   // register shuffle behind a load waits for it -- and for the row loads in front of it)
    auto issue_perm = [&](const Where &w) {  // row indices of the block (clamped: rows past the node's end fold as +0)
        const uint32_t first = min(kFsBlk * w.b, w.tl.rows - 1u);
#pragma unroll
        for (int i = 0; i < 16; ++i) prow[i] = perm[w.tl.start + min(first + 16 * g + (uint32_t)i, w.tl.rows - 1u)];
    };
    auto issue_rows = [&](const Where &w) {  // the rows prow names and the block's per-column data
#pragma unroll
        for (int i = 0; i < 16; ++i) vn[i] = *reinterpret_cast<const float4 *>(X + (size_t)prow[i] * d + w.cq);
        pref_n = *reinterpret_cast<const float4 *>(tile_pref + (size_t)w.tile_id * d + w.cq);
        if (MODE == 1) mu_n = *reinterpret_cast<const float4 *>(na.centroid + (size_t)w.tl.node * d + w.cq);
    };
    uint32_t item = 0;
    Where cur = where(0);
    issue_perm(cur);
    issue_rows(cur);
    Where nxt = where(n_local > 1 ? 1u : 0u);
    if (n_local > 1) issue_perm(nxt);
    float run[4] = {0.f, 0.f, 0.f, 0.f};  // exact f32 sum of the tile's rows in front of this block, per column
    uint32_t chunk_base = 0, chunk_left = 0;  // this wave's unused side-buffer slots
    double mom1 = 0.0, mom2 = 0.0;            // mean pass, lanes of segment 0: sum x and sum x^2 of the tile's column cl so far
    for (;;) {
        // ---- this block's rows out of the load registers (waits for them), the next block's requested ----
        const uint32_t blk_first = kFsBlk * cur.b;
        const bool live_item = blk_first < cur.tl.rows;  // uniform; a node's last tile may end before this block
        const uint32_t rows_blk = live_item ? min(kFsBlk, cur.tl.rows - blk_first) : 0u;
        float w[16][4], mu[4], pref[4];
        float ps[4] = {0.f, 0.f, 0.f, 0.f};  // plain sums: the guesses of the segments behind
        float sq[4] = {0.f, 0.f, 0.f, 0.f};  // mean pass: sums of squares (the variance pass's guess, k_fs_prefix_var)
#pragma unroll
        for (int j = 0; j < 4; ++j) mu[j] = (&mu_n.x)[j], pref[j] = (&pref_n.x)[j];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool live = 16 * g + (uint32_t)i < rows_blk;
            const float x[4] = {vn[i].x, vn[i].y, vn[i].z, vn[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[i][j] = live ? fs_value<MODE>(x[j], mu[j]) : 0.0f;
                ps[j] += w[i][j];
                if (MODE == 0) sq[j] = __builtin_fmaf(w[i][j], w[i][j], sq[j]);
            }
        }
        const bool has_next = item + 1 < n_local;  // uniform
        Where nxt2 = where(item + 2 < n_local ? item + 2 : item);
        if (has_next) {
            issue_rows(nxt);
            if (item + 2 < n_local) issue_perm(nxt2);
        }
        if (cur.b == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) run[j] = 0.0f;
        }
        if (live_item) {
            if (MODE == 0) {  // (stored at once: the registers are free again before the fold)
                *reinterpret_cast<float4 *>(&p_s1[g][4 * q]) = make_float4(ps[0], ps[1], ps[2], ps[3]);
                *reinterpret_cast<float4 *>(&p_s2[g][4 * q]) = make_float4(sq[0], sq[1], sq[2], sq[3]);
            }
            // The four columns of a lane are independent: everything below is written column-innermost so that the four
            // dependent chains (cross-lane sums, the running delta and its min / max) interleave in one basic block --
            // a lone pair of waves per SIMD hides nothing else (column after column, with the tie test in between,
            // the block ran at ~10 cycles per instruction).
            float scale[4];
            int32_t gbad[4], gbits[4];
            {
                // segment sums: lanes 8 apart (row_ror:8 of a 16-lane row), then the two rows of a segment
                // (v_permlane16_swap), then the block's other segment (v_permlane32_swap): no LDS round trips
                float t[4], other[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = ps[j] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ps[j]), 0x128, 0xF, 0xF, false));  // row_ror:8
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[j]), __float_as_uint(t[j]), false, false);
                    t[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // rows (0,1) and (2,3) of the wave: the sum of this lane's segment
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[j]), __float_as_uint(t[j]), false, false);
                    other[j] = (g >= 4) ? __uint_as_float(sw[0]) : __uint_as_float(sw[1]);  // ... of the block's other segment
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t0 = (g >= 4) ? other[j] : t[j];          // segment 0's sum on every lane
                    const float guess = pref[j] + run[j] + (g >= 4 ? t0 : 0.0f);
                    run[j] += t0 + ((g >= 4) ? t[j] : other[j]);          // the same value on every lane that holds the column
                    const uint32_t gb = __float_as_uint(guess), ex = (gb >> 23) & 0xFFu;
                    const int e = (int)ex - 127;
                    // scale = 2^(23-e): needs a normal guess and a representable power of two
                    const bool bad = (ex == 0u) || (ex == 255u) || (23 - e > 126) || (23 - e < -126);
                    scale[j] = bad ? 1.0f : __uint_as_float((uint32_t)(23 - e + 127) << 23);
                    gbad[j] = bad ? 1 : 0, gbits[j] = (int32_t)gb;
                }
            }
            // ---- fold: 16 rows x 4 columns per lane -----------------------------------------------------------
            // Fast fold, ONE stream: with q = x / ulp(s) and S = s / ulp(s) an integer, fl(s + x) / ulp = S + rne(q)
            // whenever q is not exactly half-way between two integers -- whatever the parity of S (the test is exact:
            // q - rne(q) = +-1/2); a run that does hold a tie is folded again by the two-stream code.
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {  // two columns at a time, their chains interleaved; results to LDS at once
                // rne(q) through the adder: t = fl(q + 1.5 * 2^23) in ONE rounding (fma; q = x * scale is exact) is 1.5 * 2^23 +
                // rne(q) for |q| < 2^22 (ties to even: the constant is even), so rne(q) as an integer is bits(t) - bits(1.5 * 2^23),
                // as a float t - 1.5 * 2^23 (exact), and q - rne(q) one more fma (exact): six instructions per element
                // where v_mul / v_rndne / v_cvt / the range test on bits(q) took eight.  |rne(q)| >= 2^22 (an addend a
                // quarter of the sum: only at a chain's start) and inf end in `rmax`, a NaN in the plain sum ps.
                constexpr float kRnd = 12582912.0f;
                int32_t d1[2] = {0, 0}, lo1[2] = {0, 0}, hi1[2] = {0, 0};
                float tmax[2] = {0.f, 0.f}, rmax[2] = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float t = __builtin_fmaf(w[i][jp + jj], scale[jp + jj], kRnd);
                        const float r = t - kRnd;
                        tmax[jj] = fmaxf(tmax[jj], fabsf(__builtin_fmaf(w[i][jp + jj], scale[jp + jj], -r)));  // |q - rne(q)|
                        rmax[jj] = fmaxf(rmax[jj], fabsf(r));
                        d1[jj] += (int32_t)(__float_as_uint(t) - 0x4B400000u);
                        lo1[jj] = min(lo1[jj], d1[jj]);
                        hi1[jj] = max(hi1[jj], d1[jj]);
                    }
                }
                int32_t od0[2], od1[2], olo0[2], olo1[2], ohi0[2], ohi1[2], obad[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp + jj;
                    obad[jj] = gbad[j] | ((rmax[jj] >= 4194304.0f || ps[j] != ps[j]) ? 1 : 0);  // |q| >= 2^22, inf, NaN: re-added
                    od0[jj] = od1[jj] = d1[jj], olo0[jj] = olo1[jj] = lo1[jj], ohi0[jj] = ohi1[jj] = hi1[jj];
                    if (!obad[jj] && tmax[jj] == 0.5f) {
                        // two streams (even / odd incoming S), exact tie handling.  q = a + f, a = floor(q), 0 <= f < 1,
                        // classified EXACTLY from 2q (exact: |q| < 2^24): with i2 = floor(2q), a = i2 >> 1 and f is above
                        // / at / below one half as (i2 odd, 2q not an integer) / (i2 odd, 2q an integer) / (i2 even).
                        const float scale2 = scale[j] + scale[j];
                        int32_t dA = 0, dB = 0, loA = 0, loB = 0, hiA = 0, hiB = 0;  // A: even incoming S, B: odd
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float qq = w[i][j] * scale2;
                            const float fl = floorf(qq);
                            const int32_t i2 = (int32_t)fl, sticky = qq != fl ? 1 : 0;
                            const int32_t ai = i2 >> 1, half = i2 & 1;
                            const int32_t up = half & sticky, tie = half & (sticky ^ 1);
                            const int32_t bA = dA + ai, bB = dB + ai;
                            dA = bA + (up | (tie & bA));        // tie: round to the even S
                            dB = bB + (up | (tie & (1 + bB)));
                            loA = min(loA, dA), hiA = max(hiA, dA);
                            loB = min(loB, dB), hiB = max(hiB, dB);
                        }
                        od0[jj] = dA, od1[jj] = dB, olo0[jj] = loA, olo1[jj] = loB, ohi0[jj] = hiA, ohi1[jj] = hiB;
                    }
                }
                *reinterpret_cast<int2 *>(&p_d0[g][4 * q + jp]) = make_int2(od0[0], od0[1]);
                *reinterpret_cast<int2 *>(&p_d1[g][4 * q + jp]) = make_int2(od1[0], od1[1]);
                *reinterpret_cast<int2 *>(&p_lo0[g][4 * q + jp]) = make_int2(olo0[0], olo0[1]);
                *reinterpret_cast<int2 *>(&p_lo1[g][4 * q + jp]) = make_int2(olo1[0], olo1[1]);
                *reinterpret_cast<int2 *>(&p_hi0[g][4 * q + jp]) = make_int2(ohi0[0], ohi0[1]);
                *reinterpret_cast<int2 *>(&p_hi1[g][4 * q + jp]) = make_int2(ohi1[0], ohi1[1]);
                *reinterpret_cast<int2 *>(&p_bad[g][4 * q + jp]) = make_int2(obad[0], obad[1]);
                *reinterpret_cast<int2 *>(&p_gb[g][4 * q + jp]) = make_int2(gbits[jp], gbits[jp + 1]);
            }
            fs_wave_lds_sync();
            // ---- compose the segment's four runs in row order; predict; write ---------------------------------
            const uint32_t c = cur.c0 + cl;
            const bool seg_live = kFsSeg * sg < rows_blk;
            int slot = -1;
            {
                FsT f;
                f.d0 = p_d0[4 * sg][cl], f.d1 = p_d1[4 * sg][cl], f.lo0 = p_lo0[4 * sg][cl], f.lo1 = p_lo1[4 * sg][cl];
                f.hi0 = p_hi0[4 * sg][cl], f.hi1 = p_hi1[4 * sg][cl];
                int anybad = p_bad[4 * sg][cl];
                const uint32_t gb = (uint32_t)p_gb[4 * sg][cl];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    FsT gk;
                    gk.d0 = p_d0[4 * sg + k][cl], gk.d1 = p_d1[4 * sg + k][cl], gk.lo0 = p_lo0[4 * sg + k][cl];
                    gk.lo1 = p_lo1[4 * sg + k][cl], gk.hi0 = p_hi0[4 * sg + k][cl], gk.hi1 = p_hi1[4 * sg + k][cl];
                    anybad |= p_bad[4 * sg + k][cl] | ((uint32_t)p_gb[4 * sg + k][cl] != gb ? 1 : 0);
                    f = fs_compose(f, gk);
                }
                const int32_t lim = 1 << 28;
                if (f.hi0 > lim || f.hi1 > lim || f.lo0 < -lim || f.lo1 < -lim) anybad = 1;
                // Will the chain have to re-add this segment?  Predict it from the guessed S and, if so, park the
                // segment's addends contiguously in the side buffer: the chain fetches 256 bytes ahead of time instead
                // of gathering 4 bytes from each of 64 rows through `perm` when it gets there.
                bool leaves = false;
                if (seg_live && c < d && side_cap) {
                    leaves = anybad != 0;
                    if (!anybad) {
                        const int32_t mag = (int32_t)((gb & 0x7FFFFFu) | 0x800000u);
                        const int32_t Sg = (gb >> 31) ? -mag : mag;
                        const int32_t lo2 = min(f.lo0, f.lo1), hi2 = max(f.hi0, f.hi1);
                        // 0.2 % of the binade where the guess is good to ~1e-5 (exact tile sums, the variance pass's
                        // moments); a guess from every r-th row is off by ~sigma sqrt(r N): three sigmas of that
                        int32_t margin = 1 << 14;
                        const float park_rel = (policy && !policy[cur.c0 / kFsCols]) ? 0.0f : park_rel_arg;
                        if (park_rel > 0.0f) {
                            const float m2 = park_rel * __builtin_amdgcn_rsqf((float)(cur.tl.t + 1u)) * (float)mag;
                            margin = m2 > (float)margin ? (m2 < 4194304.0f ? (int32_t)m2 : (1 << 22)) : margin;
                        }
                        leaves = (Sg > 0) ? (Sg + lo2 < (1 << 23) + margin || Sg + hi2 > (1 << 24) - margin)
                                          : (Sg + hi2 > -(1 << 23) - margin || Sg + lo2 < -(1 << 24) + margin);
                    }
                }
                // slots come out of a wave-private chunk of 64 (one atomic per chunk: a counter bumped per segment, or
                // even per block, serialised the waves on zero-mean data; the rest of a wave's last chunk stays unused)
                const uint64_t pm = __ballot(leaves);
                if (pm) {
                    const uint32_t want = (uint32_t)__builtin_popcountll(pm);
                    if (want > chunk_left) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(side_count, 64u);
                        chunk_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        chunk_left = 64u;
                    }
                    const uint32_t mine = chunk_base + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull));
                    if (leaves && mine < side_cap) slot = (int)mine;
                    chunk_base += want;
                    chunk_left -= want;
                }
                if (seg_live && c < d) {
                    const bool two = f.d0 != f.d1 || f.lo0 != f.lo1 || f.hi0 != f.hi1;
                    const size_t at = (size_t)c * na.fs_seg_stride + (size_t)cur.tile_id * kFsSegsPerTile + 2 * cur.b + sg;
                    // for k_fs_prep's tables: the guess the segment was folded under, this pass's tag (tells its slots from what
                    // an earlier pass left in the unused rest of a chunk), where its summary is, and whether the guess is a sampled one
                    // (second record: what the tables' test wants of the summary -- its prefix range over both streams, unusable or
                    // not -- so that k_fs_prep does not chase the summary behind `at`: one dependent round trip per table less)
                    if (slot >= 0 && side_meta) {
                        side_meta[2 * slot] = make_uint4(gb, pass_tag, (uint32_t)at, (policy && policy[cur.c0 / kFsCols] && park_rel_arg > 0.0f) ? 1u : 0u);
                        side_meta[2 * slot + 1] = make_uint4((uint32_t)min(f.lo0, f.lo1), (uint32_t)max(f.hi0, f.hi1), anybad ? 1u : 0u, 0u);
                    }
                    FsS o;
                    o.d = f.d0, o.lo = f.lo0, o.hi = f.hi0;
                    o.ef = fs_ef(anybad, two, (int)((gb >> 23) & 0xFFu) - 127, slot) | (int32_t)(gb & 0x80000000u);
                    t_sum[2 * cur.b + sg][cl] = o;
                    if (two) {
                        FsS o2;
                        o2.d = f.d1, o2.lo = f.lo1, o2.hi = f.hi1, o2.ef = 0;
                        summ_odd[at] = o2;
                    }
                }
            }
            pslot[sg][cl] = slot;
            if (MODE == 0 && sg == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    mom1 += (double)p_s1[k][cl];
                    mom2 += (double)p_s2[k][cl];
                }
            }
            fs_wave_lds_sync();
            // ---- park the flagged segments (rare): this lane's 16 rows of the column, 64 bytes --------------------
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = pslot[g >> 2][4 * q + j];
                if (sl >= 0) {
                    float4 *dst = reinterpret_cast<float4 *>(side + (size_t)sl * kFsSeg + 16 * (g & 3u));
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) dst[i4] = make_float4(w[4 * i4][j], w[4 * i4 + 1][j], w[4 * i4 + 2][j], w[4 * i4 + 3][j]);
                }
            }
        }
        if (cur.b == kFsBlksPerTile - 1) {  // the tile's last block: its summaries out, eight lanes per column (one whole line)
            fs_wave_lds_sync();
            const uint32_t piece = lane & 7u;
#pragma unroll
            for (uint32_t i = 0; i < kFsCols / 8; ++i) {
                const uint32_t col = 8u * i + (lane >> 3);
                if (cur.c0 + col < d)  // (slots past the node's last segment receive what an earlier tile left: nobody reads them)
                    summ[(size_t)(cur.c0 + col) * na.fs_seg_stride + (size_t)cur.tile_id * kFsSegsPerTile + piece] = t_sum[piece][col];
            }
            fs_wave_lds_sync();  // (the next tile's first block writes t_sum again)
        }
        if (MODE == 0 && tile_mom && (cur.b == kFsBlksPerTile - 1)) {  // the tile's last block: its sums out, reset
            if (sg == 0 && cur.c0 + cl < d) tile_mom[(size_t)cur.tile_id * d + cur.c0 + cl] = make_double2(mom1, mom2);
            mom1 = mom2 = 0.0;
        }
        if (!has_next) break;
        ++item;
        cur = nxt;
        nxt = nxt2;
    }
}

// the exact chain: one wave per (node, column).  The segment summaries are themselves parity transducers: every lane
// composes kFsSpl consecutive ones (they must share a binade), a wave scan composes the lanes, lane l learns the exact
// S entering its first segment (as if everything before it in the batch held) and checks its run (same binade as the
// guesses, every prefix strictly inside it).  The first lane that fails marks where the scan stops: the lanes before
// it are applied in one step, that lane's segments are walked one by one -- applied, or their 64 rows re-added in the
// reference's order -- and the scan resumes behind it over the SAME registers (lanes already consumed scan as the
// identity): no summary is loaded twice.  The next batch of 256 summaries is requested before the current one is
// scanned, and so are the addends of the first kFsAhead segments the fold kernel predicted to fail (it parked them
// contiguously): a re-addition costs its 64 additions, not a memory round trip.
constexpr int kFsSpl = 8;     // segments per lane and batch
constexpr int kFsAhead = 8;   // parked segments whose addends travel together (two such groups: one complete, one in flight)

// The chain's scans only place the runs: what they need of a composition is where it leaves S (d0 / d1 by the parity
// of the S entering) -- whether a run holds is tested against its OWN prefix bounds.  Two fields instead of six: 2 DPP
// moves and ~6 instructions per step instead of 6 and ~14.
struct FsD {
    int32_t d0, d1;
};
__device__ __forceinline__ FsD fs_compose_d(const FsD &f, const FsD &g) {
    FsD h;
    h.d0 = f.d0 + ((f.d0 & 1) ? g.d1 : g.d0);
    h.d1 = f.d1 + (((1 + f.d1) & 1) ? g.d1 : g.d0);
    return h;
}
#define VQ_FS_DPP(X, CTRL) __builtin_amdgcn_update_dpp(0, X, CTRL, 0xF, 0xF, true)
#define VQ_FS_STEP_D(CTRL, COND)                                   \
    {                                                              \
        FsD p;                                                     \
        p.d0 = VQ_FS_DPP(v.d0, CTRL), p.d1 = VQ_FS_DPP(v.d1, CTRL); \
        if (COND) v = fs_compose_d(p, v);                          \
    }
// in-row steps by DPP row_shr (lane i <- lane i - off of its 16-lane row), then the totals of rows 0 / 2 into rows 1 / 3
// (row_bcast:15) and of lane 31 into rows 2 and 3 (row_bcast:31): six steps, no LDS round trips
__device__ __forceinline__ void fs_scan_incl_d(FsD &v, uint32_t lane) {
    VQ_FS_STEP_D(0x111, (lane & 15u) >= 1u)
    VQ_FS_STEP_D(0x112, (lane & 15u) >= 2u)
    VQ_FS_STEP_D(0x114, (lane & 15u) >= 4u)
    VQ_FS_STEP_D(0x118, (lane & 15u) >= 8u)
    VQ_FS_STEP_D(0x142, (lane & 16u) != 0u)
    VQ_FS_STEP_D(0x143, lane >= 32u)
}
__device__ __forceinline__ void fs_scan_incl8_d(FsD &v, uint32_t jl) {  // every group of eight lanes
    VQ_FS_STEP_D(0x111, jl >= 1u)
    VQ_FS_STEP_D(0x112, jl >= 2u)
    VQ_FS_STEP_D(0x114, jl >= 4u)
}
#undef VQ_FS_STEP_D
#undef VQ_FS_DPP

// does a run with prefixes lo .. hi (relative to S) stay strictly inside the binade of S = +-[2^23, 2^24)?  Strictly on
// the zero side: a sum that rounds to exactly +-2^23 on this grid may have had a smaller magnitude, which the finer
// grid below represents differently (it may also be exact -- then the segment is merely re-added)
__device__ __forceinline__ bool fs_inside(int32_t S, int32_t lo, int32_t hi) {
    return (S > 0) ? (S + lo > (1 << 23) && S + hi <= (1 << 24) - 1) : (S + hi < -(1 << 23) && S + lo >= -((1 << 24) - 1));
}

// DBG (VQHIP_TSVQ_DEBUG): the instantiation with the counters; the production one carries none of it
template <int MODE, bool DBG>
__global__ __launch_bounds__(64) void k_fs_chain(const float *__restrict__ X, uint32_t d,
                                                 const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ fast_nodes,
                                                 const uint32_t *__restrict__ tile_base, NodeArrays na,
                                                 const FsS *__restrict__ summ, const FsS *__restrict__ summ_odd,
                                                 const float *__restrict__ side, uint32_t *__restrict__ n_fallback,
                                                 const LevelInfo *__restrict__ lv, uint32_t *__restrict__ dbg_arg,
                                                 const uint32_t *__restrict__ only_sampled) {
    uint32_t *const dbg = DBG ? dbg_arg : nullptr;  // folds every `if (dbg)` below away when !DBG
    // (round 5: the mean pass's columns with an exact guess -- zero-mean columns -- go through k_fs_prep / k_fs_chain3)
    if (only_sampled && only_sampled[blockIdx.y / kFsCols] == 0u) return;
    // dbg: 8 counters of this (level, pass): chains, re-added segments, most in one chain, and the first reason the
    // re-added segment failed: unusable summary / other binade than guessed / prefix leaves the binade / sum not normal
    __shared__ int plist[64 * kFsSpl];                            // side slots of the batch's parked segments, in order
    __shared__ __attribute__((aligned(16))) FsS lsum[64 * kFsSpl];  // the batch's summaries [lane][j], staged at its first failing lane
    __shared__ __attribute__((aligned(16))) FsS lsum2[64 * kFsSpl];  // ... and their odd streams (batches with an exact tie)
    __shared__ __attribute__((aligned(16))) float ladd[64];         // the addends of a gathered (not parked) segment being re-added
    __shared__ __attribute__((aligned(16))) float lpark[kFsAhead * 64];  // group A: the addends of kFsAhead parked segments
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t node = fast_nodes[blockIdx.x], c = blockIdx.y, lane = threadIdx.x;
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nseg = (len + kFsSeg - 1) / kFsSeg;
    const size_t seg0 = (size_t)tile_base[blockIdx.x] * kFsSegsPerTile;
    const FsS *sp = summ + (size_t)c * na.fs_seg_stride + seg0, *sp2 = summ_odd + (size_t)c * na.fs_seg_stride + seg0;
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    uint32_t fallbacks = 0;
    // dbg: core cycles (s_memtime) of the whole chain, inside the failing lanes' walks, inside the re-additions, in the
    // wave-wide scan passes of two-stream batches (one stream: the fetch of the staged summaries); failing lanes
    unsigned long long cyc_all = dbg ? clock64() : 0ull, cyc_walk = 0ull, cyc_readd = 0ull, cyc_lds = 0ull;
    uint32_t n_walks = 0;
    unsigned long long cy_fetch = 0ull, cy_scan = 0ull, cy_apply = 0ull, cy_pre = 0ull, cy_add = 0ull;  // dbg, chain (0, 0): parts of a failing lane's walk
    uint32_t n_iter = 0;
    // UNCONDITIONAL loads (index clamped): a load under `if (t < nseg)` is followed by the merge with the other branch's
    // value, i.e. by s_waitcnt vmcnt(0) right behind the load -- every batch then cost four serial memory round trips
    // (~3 us).  What lies past the node's end is marked unusable when the registers are consumed.
    auto load4 = [&](uint32_t tstart, FsS (&m)[kFsSpl]) {
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) m[j] = sp[min(tstart + kFsSpl * lane + (uint32_t)j, nseg - 1u)];
    };
    // summaries of the next batches in flight behind the one being scanned (a batch's scan is shorter than a memory round
    // trip): a set is consumed two iterations after its loads were issued
    constexpr uint32_t kBatch = 64 * kFsSpl;
    FsS cur[kFsSpl], nx1[kFsSpl], nx2[kFsSpl];
    load4(0, cur);
    load4(kBatch, nx1);
    load4(2 * kBatch, nx2);
    for (uint32_t t0 = 0; t0 < nseg; t0 += kBatch) {
        const uint32_t cnt = min(kBatch, nseg - t0), nl = (cnt + kFsSpl - 1) / kFsSpl;  // segments / lanes of this batch
        FsS m[kFsSpl];
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) {
            m[j] = cur[j], cur[j] = nx1[j], nx1[j] = nx2[j];
            if (t0 + kFsSpl * lane + (uint32_t)j >= nseg) m[j].d = m[j].lo = m[j].hi = 0, m[j].ef = 1;  // past the node's end: unusable
        }
        load4(t0 + 3 * kBatch, nx2);  // (into the set the shift has just freed: no register move waits for this load)
        // the odd streams (runs with an exact tie: rare), else a copy of the even ones
        FsS m2[kFsSpl];
        bool two_l = false;
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) {
            m2[j] = m[j];
            two_l = two_l || ((m[j].ef & 3) == 2);
        }
        const bool two = __ballot(two_l) != 0ull;
        if (two) {
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j)
                if ((m[j].ef & 3) == 2) {
                    const FsS o = sp2[t0 + kFsSpl * lane + (uint32_t)j];
                    m2[j].d = o.d, m2[j].lo = o.lo, m2[j].hi = o.hi;
                }
        }
        // ---- parked segments of the batch, in segment order: slots into LDS, the first kFsAhead requested ----
        uint32_t pk = 0;  // this lane's parked segments, bit j
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) pk |= (fs_ef_slot(m[j].ef) >= 0) ? (1u << j) : 0u;
        const uint32_t pc = (uint32_t)__builtin_popcount(pk);
        uint32_t pincl = pc;  // inclusive prefix of pc over the lanes
        {
#define VQ_FS_ADD(CTRL, COND) { const int32_t t = __builtin_amdgcn_update_dpp(0, (int32_t)pincl, CTRL, 0xF, 0xF, true); if (COND) pincl += (uint32_t)t; }
            VQ_FS_ADD(0x111, (lane & 15u) >= 1u)
            VQ_FS_ADD(0x112, (lane & 15u) >= 2u)
            VQ_FS_ADD(0x114, (lane & 15u) >= 4u)
            VQ_FS_ADD(0x118, (lane & 15u) >= 8u)
            VQ_FS_ADD(0x142, (lane & 16u) != 0u)
            VQ_FS_ADD(0x143, lane >= 32u)
#undef VQ_FS_ADD
        }
        const uint32_t pbefore = pincl - pc, ptotal = (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);
        if (dbg && lane == 0) atomicAdd(dbg + 14, ptotal);
        if (ptotal) {  // uniform
            fs_wave_lds_sync();  // the previous batch's reads of plist are done
            uint32_t r = pbefore;
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j)
                if (pk & (1u << j)) plist[r++] = fs_ef_slot(m[j].ef);
            fs_wave_lds_sync();
        }
        // addends of the parked segments, kFsAhead at a time: group A is complete (plain registers: reading it waits for
        // nothing), group B in flight behind it; when the walk passes A's last rank B becomes A and the next group is
        // requested.  The batch starts with its first group in B.
        float pb[kFsAhead];
        int32_t pa_base = -kFsAhead;  // rank of group A's first segment (lpark[0 .. 63])
        auto request_b = [&](uint32_t base) {
#pragma unroll
            for (int k = 0; k < kFsAhead; ++k) {
                pb[k] = 0.0f;
                if (base + (uint32_t)k < ptotal) pb[k] = side[(size_t)plist[base + (uint32_t)k] * kFsSeg + lane];
            }
        };
        request_b(0);
        // ---- this lane's run: its segments composed, usable iff all are and share a binade ----
        FsT mine;
        mine.d0 = m[0].d, mine.lo0 = m[0].lo, mine.hi0 = m[0].hi, mine.d1 = m2[0].d, mine.lo1 = m2[0].lo, mine.hi1 = m2[0].hi;
        int lane_bad = m[0].ef & 1;
        const int lane_e = fs_ef_e(m[0].ef);
#pragma unroll
        for (int j = 1; j < kFsSpl; ++j) {
            FsT gj;
            gj.d0 = m[j].d, gj.lo0 = m[j].lo, gj.hi0 = m[j].hi, gj.d1 = m2[j].d, gj.lo1 = m2[j].lo, gj.hi1 = m2[j].hi;
            lane_bad |= (m[j].ef & 1) | (fs_ef_e(m[j].ef) != lane_e ? 1 : 0);
            if (two) {
                mine = fs_compose(mine, gj);
            } else {
                mine.lo0 = min(mine.lo0, mine.d0 + gj.lo0);
                mine.hi0 = max(mine.hi0, mine.d0 + gj.hi0);
                mine.d0 = mine.d0 + gj.d0;
                mine.d1 = mine.d0, mine.lo1 = mine.lo0, mine.hi1 = mine.hi0;
            }
        }
        // the 64 rows of segment `seg` re-added in the reference's order (uniform: every lane carries the same s)
        // The 64 additions in row order, the addends in LDS: sixteen BROADCAST reads (every lane reads the same 16 bytes:
        // four addends per ds_read_b128, no bank conflicts), all in flight ahead of the additions that use them -- the
        // chain is 64 dependent v_add_f32 on VGPR operands at the add's own ~4 cycles.  (Round 3 handed the addends from
        // lane to SGPR with 64 v_readlane, sixteen ahead: 32 SGPRs of a kernel that spills 45 already, and ~1300 cycles
        // per segment measured in place -- 20 per addition.)
        auto add64 = [&](const float *lds64) {
            const f32x4_t *l4 = reinterpret_cast<const f32x4_t *>(lds64);
            f32x4_t rq[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) rq[g] = l4[g];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                s = s + rq[g][0];
                s = s + rq[g][1];
                s = s + rq[g][2];
                s = s + rq[g][3];
            }
        };
        // the 64 rows of segment `seg` re-added in the reference's order (uniform: every lane carries the same s).  Parked
        // segments: the group of kFsAhead that has arrived sits in LDS (written when it became group A: the fold parked
        // +0.0 for rows past the node's end), the re-addition reads its 256 bytes from there; the two sources of addends
        // do NOT join before the additions (joined, the gathered path's wait for its load -- vmcnt(0): everything in
        // flight, the group just requested included, a full memory round trip at every eighth parked segment -- was paid
        // by the parked path too; picking one of eight registers by a run-time rank went through scratch memory).
        auto readd = [&](uint32_t seg, int32_t gef, int32_t rank) {
            ++fallbacks;
            const unsigned long long c_in = dbg ? clock64() : 0ull;
            const int slot = fs_ef_slot(gef);
            if (slot >= 0) {  // parked by k_fs_fold: contiguous, requested a group ahead
                while (rank >= pa_base + kFsAhead) {  // uniform: group B has arrived and becomes A, the next one is requested
                    fs_wave_lds_sync();  // the reads of the old group A are done
#pragma unroll
                    for (int k = 0; k < kFsAhead; ++k) lpark[k * 64 + (int)lane] = pb[k];
                    pa_base += kFsAhead;
                    request_b((uint32_t)(pa_base + kFsAhead));
                    fs_wave_lds_sync();
                }
                const unsigned long long q3 = dbg ? clock64() : 0ull;
                if (dbg) cy_pre += q3 - c_in;
                add64(lpark + (rank - pa_base) * 64);
                if (dbg) {
                    asm volatile("" ::"v"(s));
                    cy_add += clock64() - q3;
                }
            } else {
                const uint32_t r0 = seg * kFsSeg, rows_here = min(kFsSeg, len - r0);
                const float vv = (lane < rows_here) ? fs_value<MODE>(X[(size_t)perm[a + r0 + lane] * d + c], mu) : 0.0f;  // (+0.0 past the end)
                if (dbg && lane == 0) atomicAdd(dbg + 7, 1u);
                fs_wave_lds_sync();
                ladd[lane] = vv;
                fs_wave_lds_sync();
                add64(ladd);
            }
            if (dbg) {
                asm volatile("" ::"v"(s));
                cyc_readd += clock64() - c_in;
            }
        };
        // One stream: the failing lane's eight segments are SCANNED, not walked.  Walking them one by one (four
        // v_readlane and ~25 dependent scalar-ish instructions per segment, whether it holds or not) was ~1100 of the
        // ~2000 cycles a failing lane costs -- and on zero-mean columns 10-28 % of the segments fail, one wave per column
        // paying for each in series.  The batch's summaries go to LDS once (at its first failing lane), lanes 0..7 fetch
        // the failing lane's eight (one ds_read_b128), a three-step prefix gives every segment its incoming S, one ballot
        // finds the first that does not hold: the segments in front of it are applied in one step, it is re-added, the
        // ballot is repeated behind it.
        bool staged = false;
        auto walk_lane = [&](uint32_t good) {
            const unsigned long long w_in = dbg ? clock64() : 0ull;
            ++n_walks;
            if (!staged) {  // uniform
                fs_wave_lds_sync();  // the previous batch's reads of lsum are done
#pragma unroll
                for (int j = 0; j < kFsSpl; ++j) lsum[lane * kFsSpl + (uint32_t)j] = m[j];
                fs_wave_lds_sync();
                staged = true;
            }
            const uint32_t pb_g = (uint32_t)__builtin_amdgcn_readlane((int)pbefore, (int)good),
                           pk_g = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)good);
            const uint32_t jl = lane & 7u;
            FsS g = lsum[good * kFsSpl + jl];  // every group of eight lanes holds a copy; lanes 0..7 decide
            if (dbg) {
                asm volatile("" : "+v"(g.d), "+v"(g.lo), "+v"(g.hi), "+v"(g.ef));
                cyc_lds += clock64() - w_in;
            }
            const uint32_t seg_first = t0 + kFsSpl * good;
            const uint32_t nvalid = min((uint32_t)kFsSpl, nseg - seg_first);  // this lane's segments inside the node (>= 1)
            int32_t incl8 = g.d;
#define VQ_FS_ADD8(CTRL, K) { const int32_t t = __builtin_amdgcn_update_dpp(0, incl8, CTRL, 0xF, 0xF, true); if (jl >= K) incl8 += t; }
            VQ_FS_ADD8(0x111, 1u)
            VQ_FS_ADD8(0x112, 2u)
            VQ_FS_ADD8(0x114, 4u)
#undef VQ_FS_ADD8
            const int32_t before8 = incl8 - g.d;
            const bool usable = !(g.ef & 1);
            const int g_e = fs_ef_e(g.ef);
            const uint32_t beyond = ~0u << nvalid;  // (nvalid <= 8)
            uint32_t start2 = 0;
            int32_t base2 = 0;
            for (;;) {
                const uint32_t sb1 = __float_as_uint(s), se1 = (sb1 >> 23) & 0xFFu;
                const int32_t mag1 = (int32_t)((sb1 & 0x7FFFFFu) | 0x800000u);
                const int32_t S1 = (sb1 >> 31) ? -mag1 : mag1;
                const bool ok = (se1 != 0u) && (se1 != 255u) && usable && ((int)se1 - 127 == g_e) && fs_inside(S1 + before8 - base2, g.lo, g.hi);
                const uint32_t bad8 = (uint32_t)__ballot(!ok) & 0xFFu;
                const uint32_t js = (uint32_t)__builtin_ctz((bad8 | beyond) & (~0u << start2));  // first that does not hold, or nvalid
                if (js > start2) {
                    const int32_t S2 = S1 + __builtin_amdgcn_readlane(incl8, (int)js - 1) - base2;
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se1 << 23) | (m2a & 0x7FFFFFu));
                }
                if (js >= nvalid) break;
                const int32_t gef = __builtin_amdgcn_readlane(g.ef, (int)js);
                if (dbg && lane == 0) {
                    const uint32_t sb2 = __float_as_uint(s), se2 = (sb2 >> 23) & 0xFFu;  // the S the failing segment meets
                    const int why = (se2 == 0u || se2 == 255u) ? 6 : (gef & 1) ? 3 : ((int)se2 - 127 != fs_ef_e(gef)) ? 4 : 5;
                    atomicAdd(dbg + why, 1u);
                }
                readd(seg_first + js, gef, (int32_t)(pb_g + (uint32_t)__builtin_popcount(pk_g & ((1u << js) - 1u))));
                base2 = __builtin_amdgcn_readlane(incl8, (int)js);
                start2 = js + 1;
                if (start2 >= nvalid) break;
            }
            if (dbg) {
                asm volatile("" ::"v"(s));
                cyc_walk += clock64() - w_in;
            }
        };
        // Two streams (an exact tie somewhere in the batch -- every batch of zero-mean data, whose sums stay small against
        // their addends): the same, with the segments' parity transducers composed over the eight lanes; the scan is
        // repeated behind every segment that does not hold, the segments already consumed scanning as the identity --
        // the wave's own loop at the scale of one lane.  (Round 3 walked the eight segments one by one through
        // v_readlane and SGPRs: 580 cycles per segment visited, holding or not, 4600 of the 9500 a failing lane cost.)
        auto walk_lane2 = [&](uint32_t good) {
            if (!staged) {  // uniform
                fs_wave_lds_sync();
#pragma unroll
                for (int j = 0; j < kFsSpl; ++j) lsum[lane * kFsSpl + (uint32_t)j] = m[j], lsum2[lane * kFsSpl + (uint32_t)j] = m2[j];
                fs_wave_lds_sync();
                staged = true;
            }
            const uint32_t pb_g = (uint32_t)__builtin_amdgcn_readlane((int)pbefore, (int)good),
                           pk_g = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)good);
            const uint32_t jl = lane & 7u;
            const unsigned long long q0 = dbg ? clock64() : 0ull;
            FsS g = lsum[good * kFsSpl + jl], g2 = lsum2[good * kFsSpl + jl];
            if (dbg) {
                asm volatile("" : "+v"(g.d), "+v"(g2.d));
                cy_fetch += clock64() - q0;
            }
            const uint32_t seg_first = t0 + kFsSpl * good;
            const uint32_t nvalid = min((uint32_t)kFsSpl, nseg - seg_first);
            const bool usable = !(g.ef & 1);
            const int g_e = fs_ef_e(g.ef);
            const uint32_t beyond = ~0u << nvalid;
            uint32_t start2 = 0;
            for (;;) {
                const unsigned long long q1 = dbg ? clock64() : 0ull;
                ++n_iter;
                const uint32_t sb1 = __float_as_uint(s), se1 = (sb1 >> 23) & 0xFFu;
                const int32_t mag1 = (int32_t)((sb1 & 0x7FFFFFu) | 0x800000u);
                const int32_t S1 = (sb1 >> 31) ? -mag1 : mag1;
                const bool in = jl >= start2;
                FsD v;
                v.d0 = in ? g.d : 0, v.d1 = in ? g2.d : 0;
                fs_scan_incl8_d(v, jl);
                const int32_t incl_d = (S1 & 1) ? v.d1 : v.d0;  // delta from segment start2 through this one, for the actual parity of S
                int32_t before8 = __builtin_amdgcn_update_dpp(0, incl_d, 0x111, 0xF, 0xF, true);  // row_shr:1
                if (jl == 0u) before8 = 0;
                const int32_t Sin = S1 + before8;
                const bool podd = (Sin & 1) != 0;
                const bool ok = (se1 != 0u) && (se1 != 255u) && usable && ((int)se1 - 127 == g_e) &&
                                fs_inside(Sin, podd ? g2.lo : g.lo, podd ? g2.hi : g.hi);
                const uint32_t bad8 = (uint32_t)__ballot(!ok) & 0xFFu;
                uint32_t js = (uint32_t)__builtin_ctz((bad8 | beyond) & (~0u << start2));
                const unsigned long long q2 = dbg ? clock64() : 0ull;
                if (dbg) {
                    asm volatile("" : "+s"(js));
                    cy_scan += q2 - q1;
                }
                if (js > start2) {
                    const int32_t S2 = S1 + __builtin_amdgcn_readlane(incl_d, (int)js - 1);
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se1 << 23) | (m2a & 0x7FFFFFu));
                }
                if (dbg) {
                    asm volatile("" ::"v"(s));
                    cy_apply += clock64() - q2;
                }
                if (js >= nvalid) break;
                const int32_t gef = __builtin_amdgcn_readlane(g.ef, (int)js);
                if (dbg && lane == 0) {
                    const uint32_t sb2 = __float_as_uint(s), se2 = (sb2 >> 23) & 0xFFu;
                    const int why = (se2 == 0u || se2 == 255u) ? 6 : (gef & 1) ? 3 : ((int)se2 - 127 != fs_ef_e(gef)) ? 4 : 5;
                    atomicAdd(dbg + why, 1u);
                }
                readd(seg_first + js, gef, (int32_t)(pb_g + (uint32_t)__builtin_popcount(pk_g & ((1u << js) - 1u))));
                start2 = js + 1;
                if (start2 >= nvalid) break;
            }
        };
        const uint64_t past = nl < 64 ? (~0ull << nl) : 0ull;  // lanes behind the batch's last
        if (!two) {
            // One stream (no exact tie anywhere in the batch: every batch of continuous data): the deltas simply add, so ONE
            // scan serves the whole batch -- behind a lane that did not hold, the S entering lane l is the S the walk
            // arrived at plus the deltas of the lanes in between.
            int32_t incl = mine.d0;
#define VQ_FS_ADD(CTRL, COND) { const int32_t t = __builtin_amdgcn_update_dpp(0, incl, CTRL, 0xF, 0xF, true); if (COND) incl += t; }
            VQ_FS_ADD(0x111, (lane & 15u) >= 1u)
            VQ_FS_ADD(0x112, (lane & 15u) >= 2u)
            VQ_FS_ADD(0x114, (lane & 15u) >= 4u)
            VQ_FS_ADD(0x118, (lane & 15u) >= 8u)
            VQ_FS_ADD(0x142, (lane & 16u) != 0u)
            VQ_FS_ADD(0x143, lane >= 32u)
#undef VQ_FS_ADD
            int32_t before = __shfl_up(incl, 1);
            if (lane == 0) before = 0;
            const bool lane_ok = (lane < nl) && !lane_bad;
            uint32_t start = 0;   // first lane of the batch not yet applied
            int32_t base_d = 0;   // inclusive delta of lane start - 1
            for (;;) {
                const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
                const bool s_normal = (se != 0u) && (se != 255u);
                const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
                const int32_t S = (sb >> 31) ? -mag : mag;
                const bool ok = s_normal && lane_ok && ((int)se - 127 == lane_e) && fs_inside(S + before - base_d, mine.lo0, mine.hi0);
                const uint64_t below = start ? ((~0ull) >> (64 - start)) : 0ull;
                const uint64_t bad_mask = (__ballot(!ok) | past) & ~below;
                const uint32_t good = bad_mask ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;  // lanes start .. good-1 hold (uniform)
                if (good > start) {
                    const int32_t S2 = S + __builtin_amdgcn_readlane(incl, (int)good - 1) - base_d;
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2a & 0x7FFFFFu));
                }
                if (good >= nl) break;
                walk_lane(good);
                base_d = __builtin_amdgcn_readlane(incl, (int)good);
                start = good + 1;
                if (start >= nl) break;
            }
        } else {
            // exact ties in the batch: which stream a lane's run takes depends on the parity of the S entering it, so the
            // scan is repeated behind every lane that did not hold (lanes already consumed scan as the identity)
            uint32_t start = 0;
            for (;;) {
                const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
                const bool s_normal = (se != 0u) && (se != 255u);
                const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
                const int32_t S = (sb >> 31) ? -mag : mag;
                const bool in = lane >= start;
                const unsigned long long sc_in = dbg ? clock64() : 0ull;
                FsD v;
                v.d0 = in ? mine.d0 : 0, v.d1 = in ? mine.d1 : 0;
                fs_scan_incl_d(v, lane);
                const int32_t incl_d = (S & 1) ? v.d1 : v.d0;  // delta from position `start`, for the actual parity of S
                int32_t before = __shfl_up(incl_d, 1);
                if (lane == 0) before = 0;
                const int32_t Sin = S + before;
                const bool podd = (Sin & 1) != 0;
                bool ok = s_normal && (lane < nl) && !lane_bad && ((int)se - 127 == lane_e);
                ok = ok && fs_inside(Sin, podd ? mine.lo1 : mine.lo0, podd ? mine.hi1 : mine.hi0);
                const uint64_t below = start ? ((~0ull) >> (64 - start)) : 0ull;
                const uint64_t bad_mask = (__ballot(!ok) | past) & ~below;
                const uint32_t good = bad_mask ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;
                if (good > start) {
                    const int32_t S2 = S + __builtin_amdgcn_readlane(incl_d, (int)good - 1);
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2a & 0x7FFFFFu));
                }
                if (dbg) {
                    asm volatile("" ::"v"(s));
                    cyc_lds += clock64() - sc_in;  // (two streams: scan + test + apply)
                }
                if (good >= nl) break;
                const unsigned long long w_in = dbg ? clock64() : 0ull;
                ++n_walks;
                walk_lane2(good);
                if (dbg) {
                    asm volatile("" ::"v"(s));
                    cyc_walk += clock64() - w_in;
                }
                start = good + 1;
                if (start >= nl) break;
            }
        }
    }
    if (lane == 0) {
        if (MODE == 0) na.centroid[(size_t)node * d + c] = s / (float)len;  // T::from_usize(n)
        else na.var[(size_t)node * d + c] = s;
        if (n_fallback && fallbacks) atomicAdd(n_fallback, fallbacks);
        if (dbg) {
            atomicAdd(dbg + 0, 1u);
            atomicAdd(dbg + 1, fallbacks);
            atomicMax(dbg + 2, fallbacks);
            atomicAdd(dbg + 8, (uint32_t)((clock64() - cyc_all) >> 6));
            atomicAdd(dbg + 9, (uint32_t)(cyc_walk >> 6));
            atomicAdd(dbg + 10, (uint32_t)(cyc_readd >> 6));
            atomicAdd(dbg + 11, (uint32_t)(cyc_lds >> 6));
            atomicAdd(dbg + 12, n_walks);
            atomicMax(dbg + 13, (uint32_t)((clock64() - cyc_all) >> 6));
            if (blockIdx.x == 0 && blockIdx.y == 0 && n_walks)
                printf("[vqhip-dev] chain(0,0) mode %d: %u failing lanes, %u scan passes, %u re-added; cycles per failing lane: fetch %.0f; per scan pass: "
                       "scan+test %.0f, apply %.0f; per re-added segment: before the additions %.0f, additions %.0f; all %.0f k\n",
                       MODE, n_walks, n_iter, fallbacks, (double)cy_fetch / n_walks, (double)cy_scan / n_iter, (double)cy_apply / n_iter,
                       (double)cy_pre / (fallbacks ? fallbacks : 1), (double)cy_add / (fallbacks ? fallbacks : 1), (double)(clock64() - cyc_all) / 1e3);
        }
    }
}

// ---- the chain with its operands in LDS ahead of time (round 5) -------------------------------------------------------
// k_fs_chain above keeps the next batches of summaries in registers and fetches parked addends eight segments at a time
// on demand.  Measured (VQHIP_TSVQ_DEBUG, C4): a batch of 512 summaries that holds costs ~6000 cycles where its
// instructions are ~2000, and the first re-addition of a batch waits 600 .. 25000 cycles for its addends (the groups of
// eight are requested one after the other).  Here a workgroup is two waves.  The LOADER keeps a queue of LDS-DMA loads
// (global_load_lds_dwordx4: no destination registers) filled -- the summaries of the batch after next, the parked addends
// of the next batch's first PCAP parked segments, listed from the summaries that have just landed -- and hands batches to
// the WALKER through two LDS words; the walker reads nothing from memory on its way: summaries, parked addends and the
// per-lane parking counts are in LDS rings when a batch is handed over.  The loads are written in inline assembly: the
// compiler neither knows nor tracks them, the loader waits with s_waitcnt vmcnt(0) once per batch (the compiler's own
// waits for ordinary loads can only become more conservative, never wrong: loads return in order) --
// profiles/ubench/lds_dma.hip checks where the data lands and the ordering on the device.  A ring three batches deep
// with exact wait counts (every iteration padded to the same number of DMA instructions) was built too and measured no
// faster: one batch of look-ahead covers the round trip once the loader runs beside the walker.
// Everything else -- lane runs of eight segments, wave scan, the eight-lane scan of a failing lane, 64 dependent
// additions per re-added segment -- is k_fs_chain's; every sum is still the reference's bit pattern.
// (M0 = the LDS address; gfx9 and later need M0 for nothing else a compute kernel of this file does -- no LDS bounds, no
// v_movrel, no ds_gws; the compiler's only own use, checked in the ISA, is the s_sendmsg of a printf in the DEBUG
// instantiations, set right in front of it -- so no value lives in M0 across the statement)
__device__ __forceinline__ void fs_dma16(const void *gp, uint32_t lds_byte_addr) {  // lane l: 16 bytes from gp to lds_byte_addr + 16 l
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void fs_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int MODE, bool DBG>
__global__ __launch_bounds__(128) void k_fs_chain4(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                   const uint32_t *__restrict__ fast_nodes, const uint32_t *__restrict__ tile_base,
                                                   NodeArrays na, const FsS *__restrict__ summ, const FsS *__restrict__ summ_odd,
                                                   const float *__restrict__ side, uint32_t *__restrict__ n_fallback,
                                                   const LevelInfo *__restrict__ lv, uint32_t *__restrict__ dbg_arg,
                                                   const uint32_t *__restrict__ only_sampled) {
    uint32_t *const dbg = DBG ? dbg_arg : nullptr;
    if (only_sampled && only_sampled[blockIdx.y / kFsCols] == 0u) return;
    constexpr uint32_t kBatch = 64 * kFsSpl;
    constexpr int A = 1, B = 2;                        // batches ahead: parked addends, summaries
    constexpr int NS = B + 1, NP = A + 1;              // ring slots
    constexpr int P4 = 4;                              // parked DMA instructions per batch at most, four segments each
    constexpr uint32_t PCAP = 4 * P4;                  // parked segments of a batch fetched ahead (the rest: on demand)
    __shared__ __attribute__((aligned(16))) FsS sring[NS][kBatch];       // [slot][j * 64 + lane]: segment 8 lane + j of the batch
    __shared__ __attribute__((aligned(16))) float pring[NP][PCAP * 64];  // [slot][rank among the batch's parked][row]
    __shared__ int plist[NP][kBatch];                                   // side slots of the batch's parked segments, in order
    __shared__ uint2 pinfo[NP][64];                                     // per lane: parked segments in front of the lane's, its own (bit j)
    __shared__ __attribute__((aligned(16))) float ladd[64];             // addends of a segment fetched on demand
    __shared__ uint32_t sync_ready, sync_done;                          // batches the loader has made ready / the walker is done with
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t node = fast_nodes[blockIdx.x], c = blockIdx.y, lane = threadIdx.x & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0: the walker, 1: the loader
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nseg = (len + kFsSeg - 1) / kFsSeg, nb = (nseg + kBatch - 1) / kBatch;
    const size_t seg0 = (size_t)tile_base[blockIdx.x] * kFsSegsPerTile;
    const FsS *sp = summ + (size_t)c * na.fs_seg_stride + seg0, *sp2 = summ_odd + (size_t)c * na.fs_seg_stride + seg0;
    if (threadIdx.x == 0) sync_ready = 0u, sync_done = 0u;
    __syncthreads();
    // the flags: plain LDS words, ordered against the LDS data around them by LDS-only fences (a release / acquire over all
    // address spaces would also wait for the loader's DMA queue -- the very thing that must stay in flight)
    auto flag_wait = [&](uint32_t *flag, uint32_t want) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    auto flag_set = [&](uint32_t *flag, uint32_t v) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    const uint32_t pring_addr = (uint32_t)(uintptr_t)&pring[0][0];
    if (role == 1u) {
        // ================= the loader: DMA queue, parked lists ==========================================================
        const uint32_t sring_addr = (uint32_t)(uintptr_t)&sring[0][0];
        // eight DMA instructions: the summaries of batch b (clamped at the node's last segment: a batch past the end re-reads it)
        auto issue_summ = [&](uint32_t b) {
            const uint32_t base = sring_addr + (b % (uint32_t)NS) * (uint32_t)(kBatch * sizeof(FsS));
            if ((b + 1u) * kBatch <= nseg) {  // uniform: a whole batch, one address
                const FsS *p0 = sp + (b * kBatch + kFsSpl * lane);
#pragma unroll
                for (int j = 0; j < kFsSpl; ++j) fs_dma16(p0 + j, base + (uint32_t)j * 1024u);
            } else {
#pragma unroll
                for (int j = 0; j < kFsSpl; ++j)
                    fs_dma16(sp + min(b * kBatch + kFsSpl * lane + (uint32_t)j, nseg - 1u), base + (uint32_t)j * 1024u);
            }
        };
        // batch b's summaries are in their slot: its parked segments listed in order, the addends of the first PCAP requested
        // (sixteen lanes per segment, four segments per DMA instruction; ranks past the batch's count repeat its last)
        auto issue_parked = [&](uint32_t b) {
            const uint32_t ss = b % (uint32_t)NS, ps = b % (uint32_t)NP;
            uint32_t efv[kFsSpl];
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j) efv[j] = fs_vgpr((uint32_t)sring[ss][(uint32_t)j * 64u + lane].ef);  // (all eight reads in flight)
            const uint32_t left = nseg - min(nseg, b * kBatch + kFsSpl * lane);  // this lane's segments inside the node, if < 8
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j) pk |= ((efv[j] >> 10) & 0x1FFFFFu) != 0u && (uint32_t)j < left ? (1u << j) : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(pk);
            uint32_t pincl = pc;
#define VQ_FS_ADD(CTRL, COND) { const int32_t t = __builtin_amdgcn_update_dpp(0, (int32_t)pincl, CTRL, 0xF, 0xF, true); if (COND) pincl += (uint32_t)t; }
            VQ_FS_ADD(0x111, (lane & 15u) >= 1u)
            VQ_FS_ADD(0x112, (lane & 15u) >= 2u)
            VQ_FS_ADD(0x114, (lane & 15u) >= 4u)
            VQ_FS_ADD(0x118, (lane & 15u) >= 8u)
            VQ_FS_ADD(0x142, (lane & 16u) != 0u)
            VQ_FS_ADD(0x143, lane >= 32u)
#undef VQ_FS_ADD
            const uint32_t pbefore = pincl - pc, ptotal = (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);
            pinfo[ps][lane] = make_uint2(pbefore, pk);
            const uint32_t pdst = pring_addr + ps * (PCAP * 256u);
            if (ptotal == 0u) return;  // uniform: most batches
            uint32_t r = pbefore;
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j)
                if (pk & (1u << j)) plist[ps][r++] = (int)((efv[j] >> 10) & 0x1FFFFFu) - 1;
            fs_wave_lds_sync();
            const uint32_t have = min(ptotal, PCAP);
            uint32_t slot[P4];
#pragma unroll
            for (int i = 0; i < P4; ++i) slot[i] = fs_vgpr((uint32_t)plist[ps][min(4u * (uint32_t)i + (lane >> 4), have - 1u)]);
#pragma unroll
            for (int i = 0; i < P4; ++i)
                if (4u * (uint32_t)i < have)  // uniform
                    fs_dma16(reinterpret_cast<const char *>(side + (size_t)slot[i] * kFsSeg) + 16u * (lane & 15u), pdst + (uint32_t)i * 1024u);
        };
#pragma unroll
        for (int b = 0; b < B; ++b) issue_summ((uint32_t)b);
        fs_wait_vm0();
        fs_wave_lds_sync();
#pragma unroll
        for (int b = 0; b < A; ++b) issue_parked((uint32_t)b);
        unsigned long long l_all = dbg ? clock64() : 0ull, l_vm = 0ull, l_done = 0ull;  // VQHIP_TSVQ_DEBUG
        for (uint32_t t = 0; t < nb; ++t) {
            const unsigned long long q0 = dbg ? clock64() : 0ull;
            // batch t is complete once its parked addends and (for the list built next) the summaries of batch t + A have
            // landed: everything issued so far
            fs_wait_vm0();
            if (dbg) l_vm += clock64() - q0;
            flag_set(&sync_ready, t + 1u);
            if (t + (uint32_t)A >= nb) continue;  // nothing left to request
            const unsigned long long q1 = dbg ? clock64() : 0ull;
            if (t > 0u) flag_wait(&sync_done, t);  // the slots written next are batch t - 1's
            if (dbg) l_done += clock64() - q1;
            issue_parked(t + (uint32_t)A);
            issue_summ(t + (uint32_t)B);
        }
        fs_wait_vm0();  // (nothing may land in LDS after the wave has gone)
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
            printf("[vqhip-dev] chain4(0,0) mode %d loader: %u batches, %.0f k cycles: waiting for DMA %.0f k, for the walker %.0f k\n", MODE, nb,
                   (double)(clock64() - l_all) / 1e3, (double)l_vm / 1e3, (double)l_done / 1e3);
        return;
    }

    // ===================== the walker: k_fs_chain's batch loop over operands that are already in LDS ====================
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    uint32_t fallbacks = 0, n_walks = 0;
    const unsigned long long cyc_all = dbg ? clock64() : 0ull;
    unsigned long long w_ready = 0ull, w_walk = 0ull;  // VQHIP_TSVQ_DEBUG: waiting for the loader, inside failing lanes
    auto add64 = [&](const float *lds64) {  // 64 additions in row order, the addends by broadcast LDS reads (k_fs_chain)
        const f32x4_t *l4 = reinterpret_cast<const f32x4_t *>(lds64);
        f32x4_t rq[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) rq[g] = l4[g];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            s = s + rq[g][0];
            s = s + rq[g][1];
            s = s + rq[g][2];
            s = s + rq[g][3];
        }
    };
    for (uint32_t t = 0; t < nb; ++t) {
        const uint32_t t0 = t * kBatch, cnt = min(kBatch, nseg - t0), nl = (cnt + kFsSpl - 1) / kFsSpl;
        const unsigned long long w0 = dbg ? clock64() : 0ull;
        flag_wait(&sync_ready, t + 1u);
        if (dbg) w_ready += clock64() - w0;
        const uint32_t ss = t % (uint32_t)NS, ps = t % (uint32_t)NP;
        FsS m[kFsSpl];
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) m[j] = sring[ss][(uint32_t)j * 64u + lane];
        const uint2 pi = pinfo[ps][lane];
        if (cnt < kBatch) {  // uniform: the node's last batch -- what lies past its end is unusable
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j)
                if (kFsSpl * lane + (uint32_t)j >= cnt) m[j].d = m[j].lo = m[j].hi = 0, m[j].ef = 1;
        }
        const uint32_t pbefore = pi.x, pk = pi.y;
        if (dbg && lane == 63) atomicAdd(dbg + 14, pbefore + (uint32_t)__builtin_popcount(pk));
        // flags of the lane's run: any summary unusable, binades differ, any odd stream (bit 1; with bit 0 set the run is
        // unusable anyway and the two-stream path is merely taken for nothing)
        uint32_t ef_or = (uint32_t)m[0].ef, ef_x = 0u;
#pragma unroll
        for (int j = 1; j < kFsSpl; ++j) ef_or |= (uint32_t)m[j].ef, ef_x |= (uint32_t)(m[j].ef ^ m[0].ef);
        const int lane_bad = (int)((ef_or & 1u) | ((ef_x & 0x3FCu) ? 1u : 0u));
        const int lane_e = fs_ef_e(m[0].ef);
        const bool two = __ballot((ef_or & 2u) != 0u) != 0ull;
        // ---- this lane's run: its segments composed ----
        FsT mine;
        mine.d0 = m[0].d, mine.lo0 = m[0].lo, mine.hi0 = m[0].hi;
#pragma unroll
        for (int j = 1; j < kFsSpl; ++j) {
            mine.lo0 = min(mine.lo0, mine.d0 + m[j].lo);
            mine.hi0 = max(mine.hi0, mine.d0 + m[j].hi);
            mine.d0 = mine.d0 + m[j].d;
        }
        mine.d1 = mine.d0, mine.lo1 = mine.lo0, mine.hi1 = mine.hi0;
        FsS m2[kFsSpl];  // the odd streams (only set and read when `two`)
        if (two) {  // uniform (the first batch of most chains: small sums cut few bits of an addend): the odd streams by
                    // ordinary loads, the run composed again as a parity transducer
            // (all eight loads unconditional and in flight together: a load under `if` is waited for on the spot -- eight
            // memory round trips in a row; what a segment without a tie reads there is never used)
            FsS o[kFsSpl];
            const FsS *const p2 = sp2 + min(t0 + kFsSpl * lane, nseg - 1u);
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j) o[j] = p2[min((uint32_t)j, nseg - 1u - min(t0 + kFsSpl * lane, nseg - 1u))];
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j) {
                o[j].d = (int32_t)fs_vgpr((uint32_t)o[j].d);
                const bool tie = (m[j].ef & 3) == 2;
                m2[j] = m[j];
                m2[j].d = tie ? o[j].d : m[j].d, m2[j].lo = tie ? o[j].lo : m[j].lo, m2[j].hi = tie ? o[j].hi : m[j].hi;
            }
            mine.d0 = m[0].d, mine.lo0 = m[0].lo, mine.hi0 = m[0].hi, mine.d1 = m2[0].d, mine.lo1 = m2[0].lo, mine.hi1 = m2[0].hi;
#pragma unroll
            for (int j = 1; j < kFsSpl; ++j) {
                FsT gj;
                gj.d0 = m[j].d, gj.lo0 = m[j].lo, gj.hi0 = m[j].hi, gj.d1 = m2[j].d, gj.lo1 = m2[j].lo, gj.hi1 = m2[j].hi;
                mine = fs_compose(mine, gj);
            }
        }
        // the 64 rows of segment `seg` re-added in the reference's order (uniform: every lane carries the same s)
        uint32_t pr_base = 0;  // rank of the parked segment at the front of this batch's ring slot
        auto readd = [&](uint32_t seg, int32_t gef, uint32_t rank) {
            ++fallbacks;
            const int slot = fs_ef_slot(gef);
            if (slot >= 0) {  // parked: its addends are in the ring, or the ring is moved on to it
                if (rank >= pr_base + PCAP) {
                    // Behind what was fetched ahead: the walk only moves forward, so the slot is refilled with the PCAP parked
                    // segments from this one on (failures come in runs: under a sampled guess the segments between the
                    // guessed and the true crossing were all folded under the wrong binade) -- one round trip per run,
                    // not per segment.  The walker has nothing else in flight: vmcnt(0) is this request.
                    const uint32_t ptotal = (uint32_t)__builtin_amdgcn_readlane((int)(pbefore + (uint32_t)__builtin_popcount(pk)), 63);
                    fs_wave_lds_sync();
#pragma unroll
                    for (int i = 0; i < P4; ++i) {
                        const int sl = plist[ps][min(rank + 4u * (uint32_t)i + (lane >> 4), ptotal - 1u)];
                        fs_dma16(reinterpret_cast<const char *>(side + (size_t)sl * kFsSeg) + 16u * (lane & 15u),
                                 pring_addr + ps * (PCAP * 256u) + (uint32_t)i * 1024u);
                    }
                    fs_wait_vm0();
                    fs_wave_lds_sync();
                    pr_base = rank;
                    if (dbg && lane == 0) atomicAdd(dbg + 7, 1u);
                }
                add64(&pring[ps][(rank - pr_base) * 64u]);
            } else {
                float vv;
                {
                    const uint32_t r0 = seg * kFsSeg, rows_here = min(kFsSeg, len - r0);
                    vv = (lane < rows_here) ? fs_value<MODE>(X[(size_t)perm[a + r0 + lane] * d + c], mu) : 0.0f;  // (+0.0 past the end)
                    if (dbg && lane == 0) atomicAdd(dbg + 7, 1u);
                }
                fs_wave_lds_sync();
                ladd[lane] = vv;
                fs_wave_lds_sync();
                add64(ladd);
            }
        };
        // a failing lane's eight segments, scanned over eight lanes (k_fs_chain's walk_lane; the summaries come from the ring)
        auto walk_lane = [&](uint32_t good) {
            const uint32_t pb_g = (uint32_t)__builtin_amdgcn_readlane((int)pbefore, (int)good),
                           pk_g = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)good);
            const uint32_t jl = lane & 7u;
            FsS g = sring[ss][jl * 64u + good];  // every group of eight lanes holds a copy; lanes 0..7 decide
            const uint32_t seg_first = t0 + kFsSpl * good;
            const uint32_t nvalid = min((uint32_t)kFsSpl, nseg - seg_first);  // this lane's segments inside the node (>= 1)
            if (jl >= nvalid) g.d = g.lo = g.hi = 0, g.ef = 1;
            int32_t incl8 = g.d;
#define VQ_FS_ADD8(CTRL, K) { const int32_t t = __builtin_amdgcn_update_dpp(0, incl8, CTRL, 0xF, 0xF, true); if (jl >= K) incl8 += t; }
            VQ_FS_ADD8(0x111, 1u)
            VQ_FS_ADD8(0x112, 2u)
            VQ_FS_ADD8(0x114, 4u)
#undef VQ_FS_ADD8
            const int32_t before8 = incl8 - g.d;
            const bool usable = !(g.ef & 1);
            const int g_e = fs_ef_e(g.ef);
            const uint32_t beyond = ~0u << nvalid;  // (nvalid <= 8)
            uint32_t start2 = 0;
            int32_t base2 = 0;
            for (;;) {
                const uint32_t sb1 = __float_as_uint(s), se1 = (sb1 >> 23) & 0xFFu;
                const int32_t mag1 = (int32_t)((sb1 & 0x7FFFFFu) | 0x800000u);
                const int32_t S1 = (sb1 >> 31) ? -mag1 : mag1;
                const bool ok = (se1 != 0u) && (se1 != 255u) && usable && ((int)se1 - 127 == g_e) && fs_inside(S1 + before8 - base2, g.lo, g.hi);
                const uint32_t bad8 = (uint32_t)__ballot(!ok) & 0xFFu;
                const uint32_t js = (uint32_t)__builtin_ctz((bad8 | beyond) & (~0u << start2));  // first that does not hold, or nvalid
                if (js > start2) {
                    const int32_t S2 = S1 + __builtin_amdgcn_readlane(incl8, (int)js - 1) - base2;
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se1 << 23) | (m2a & 0x7FFFFFu));
                }
                if (js >= nvalid) break;
                const int32_t gef = __builtin_amdgcn_readlane(g.ef, (int)js);
                if (dbg && lane == 0) {
                    const uint32_t sb2 = __float_as_uint(s), se2 = (sb2 >> 23) & 0xFFu;  // the S the failing segment meets
                    const int why = (se2 == 0u || se2 == 255u) ? 6 : (gef & 1) ? 3 : ((int)se2 - 127 != fs_ef_e(gef)) ? 4 : 5;
                    atomicAdd(dbg + why, 1u);
                }
                readd(seg_first + js, gef, pb_g + (uint32_t)__builtin_popcount(pk_g & ((1u << js) - 1u)));
                base2 = __builtin_amdgcn_readlane(incl8, (int)js);
                start2 = js + 1;
                if (start2 >= nvalid) break;
            }
        };
        // two streams (an exact tie somewhere in the batch): the same with the segments' parity transducers composed over the
        // eight lanes, the scan repeated behind every segment that does not hold (k_fs_chain's walk_lane2)
        auto walk_lane2 = [&](uint32_t good) {
            const uint32_t pb_g = (uint32_t)__builtin_amdgcn_readlane((int)pbefore, (int)good),
                           pk_g = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)good);
            const uint32_t jl = lane & 7u;
            const uint32_t seg_first = t0 + kFsSpl * good;
            const uint32_t nvalid = min((uint32_t)kFsSpl, nseg - seg_first);
            FsS g = sring[ss][jl * 64u + good];
            // the failing lane's odd streams out of its registers (lane jl takes segment jl's): no memory round trip
            FsS g2 = g;
#pragma unroll
            for (int j = 0; j < kFsSpl; ++j) {
                const int32_t od = __builtin_amdgcn_readlane(m2[j].d, (int)good), ol = __builtin_amdgcn_readlane(m2[j].lo, (int)good),
                              oh = __builtin_amdgcn_readlane(m2[j].hi, (int)good);
                if (jl == (uint32_t)j) g2.d = od, g2.lo = ol, g2.hi = oh;
            }
            if (jl >= nvalid) g.d = g.lo = g.hi = 0, g.ef = 1, g2 = g;
            const bool usable = !(g.ef & 1);
            const int g_e = fs_ef_e(g.ef);
            const uint32_t beyond = ~0u << nvalid;
            uint32_t start2 = 0;
            for (;;) {
                const uint32_t sb1 = __float_as_uint(s), se1 = (sb1 >> 23) & 0xFFu;
                const int32_t mag1 = (int32_t)((sb1 & 0x7FFFFFu) | 0x800000u);
                const int32_t S1 = (sb1 >> 31) ? -mag1 : mag1;
                const bool in = jl >= start2;
                FsD v;
                v.d0 = in ? g.d : 0, v.d1 = in ? g2.d : 0;
                fs_scan_incl8_d(v, jl);
                const int32_t incl_d = (S1 & 1) ? v.d1 : v.d0;  // delta from segment start2 through this one, for the actual parity of S
                int32_t before8 = __builtin_amdgcn_update_dpp(0, incl_d, 0x111, 0xF, 0xF, true);  // row_shr:1
                if (jl == 0u) before8 = 0;
                const int32_t Sin = S1 + before8;
                const bool podd = (Sin & 1) != 0;
                const bool ok = (se1 != 0u) && (se1 != 255u) && usable && ((int)se1 - 127 == g_e) &&
                                fs_inside(Sin, podd ? g2.lo : g.lo, podd ? g2.hi : g.hi);
                const uint32_t bad8 = (uint32_t)__ballot(!ok) & 0xFFu;
                const uint32_t js = (uint32_t)__builtin_ctz((bad8 | beyond) & (~0u << start2));
                if (js > start2) {
                    const int32_t S2 = S1 + __builtin_amdgcn_readlane(incl_d, (int)js - 1);
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se1 << 23) | (m2a & 0x7FFFFFu));
                }
                if (js >= nvalid) break;
                const int32_t gef = __builtin_amdgcn_readlane(g.ef, (int)js);
                if (dbg && lane == 0) {
                    const uint32_t sb2 = __float_as_uint(s), se2 = (sb2 >> 23) & 0xFFu;
                    const int why = (se2 == 0u || se2 == 255u) ? 6 : (gef & 1) ? 3 : ((int)se2 - 127 != fs_ef_e(gef)) ? 4 : 5;
                    atomicAdd(dbg + why, 1u);
                }
                readd(seg_first + js, gef, pb_g + (uint32_t)__builtin_popcount(pk_g & ((1u << js) - 1u)));
                start2 = js + 1;
                if (start2 >= nvalid) break;
            }
        };
        const uint64_t past = nl < 64 ? (~0ull << nl) : 0ull;  // lanes behind the batch's last
        // Lanes whose run holds an exact tie (an odd stream that differs): at a chain's start -- small sums cut few bits of an
        // addend -- and hardly anywhere else.  While such a lane lies ahead, the parity of S decides which stream a run
        // takes and the wave-wide scan is repeated behind every lane that did not hold; behind the last of them the deltas
        // simply add and ONE scan serves the rest of the batch.
        const uint64_t two_mask = two ? __ballot((ef_or & 2u) != 0u) : 0ull;
        uint32_t start = 0;  // first lane of the batch not yet applied
        while (start < nl && (two_mask >> start) != 0ull) {
            const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
            const bool s_normal = (se != 0u) && (se != 255u);
            const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
            const int32_t S = (sb >> 31) ? -mag : mag;
            const bool in = lane >= start;
            FsD v;
            v.d0 = in ? mine.d0 : 0, v.d1 = in ? mine.d1 : 0;
            fs_scan_incl_d(v, lane);
            const int32_t incl_d = (S & 1) ? v.d1 : v.d0;  // delta from position `start`, for the actual parity of S
            int32_t before = __shfl_up(incl_d, 1);
            if (lane == 0) before = 0;
            const int32_t Sin = S + before;
            const bool podd = (Sin & 1) != 0;
            bool ok = s_normal && (lane < nl) && !lane_bad && ((int)se - 127 == lane_e);
            ok = ok && fs_inside(Sin, podd ? mine.lo1 : mine.lo0, podd ? mine.hi1 : mine.hi0);
            const uint64_t below = start ? ((~0ull) >> (64 - start)) : 0ull;
            const uint64_t bad_mask = (__ballot(!ok) | past) & ~below;
            const uint32_t good = bad_mask ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;
            if (good > start) {
                const int32_t S2 = S + __builtin_amdgcn_readlane(incl_d, (int)good - 1);
                const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2a & 0x7FFFFFu));
            }
            if (good >= nl) {
                start = nl;
                break;
            }
            if ((two_mask >> good) & 1ull) walk_lane2(good);
            else walk_lane(good);
            start = good + 1;
        }
        if (start < nl) {
            // one stream from `start` on
            int32_t incl = lane >= start ? mine.d0 : 0;
            const int32_t own = incl;
#define VQ_FS_ADD(CTRL, COND) { const int32_t t = __builtin_amdgcn_update_dpp(0, incl, CTRL, 0xF, 0xF, true); if (COND) incl += t; }
            VQ_FS_ADD(0x111, (lane & 15u) >= 1u)
            VQ_FS_ADD(0x112, (lane & 15u) >= 2u)
            VQ_FS_ADD(0x114, (lane & 15u) >= 4u)
            VQ_FS_ADD(0x118, (lane & 15u) >= 8u)
            VQ_FS_ADD(0x142, (lane & 16u) != 0u)
            VQ_FS_ADD(0x143, lane >= 32u)
#undef VQ_FS_ADD
            const int32_t before = incl - own;
            const bool lane_ok = (lane < nl) && !lane_bad;
            int32_t base_d = 0;  // inclusive delta of lane start - 1
            for (;;) {
                const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
                const bool s_normal = (se != 0u) && (se != 255u);
                const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
                const int32_t S = (sb >> 31) ? -mag : mag;
                const bool ok = s_normal && lane_ok && ((int)se - 127 == lane_e) && fs_inside(S + before - base_d, mine.lo0, mine.hi0);
                const uint64_t below = start ? ((~0ull) >> (64 - start)) : 0ull;
                const uint64_t bad_mask = (__ballot(!ok) | past) & ~below;
                const uint32_t good = bad_mask ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;  // lanes start .. good-1 hold (uniform)
                if (good > start) {
                    const int32_t S2 = S + __builtin_amdgcn_readlane(incl, (int)good - 1) - base_d;
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2a & 0x7FFFFFu));
                }
                if (good >= nl) break;
                const unsigned long long q2 = dbg ? clock64() : 0ull;
                ++n_walks;
                walk_lane(good);
                if (dbg) {
                    asm volatile("" ::"v"(s));
                    w_walk += clock64() - q2;
                }
                base_d = __builtin_amdgcn_readlane(incl, (int)good);
                start = good + 1;
                if (start >= nl) break;
            }
        }
        flag_set(&sync_done, t + 1u);
    }
    if (lane == 0) {
        if (MODE == 0) na.centroid[(size_t)node * d + c] = s / (float)len;  // T::from_usize(n)
        else na.var[(size_t)node * d + c] = s;
        if (n_fallback && fallbacks) atomicAdd(n_fallback, fallbacks);
        if (dbg) {
            atomicAdd(dbg + 0, 1u);
            atomicAdd(dbg + 1, fallbacks);
            atomicMax(dbg + 2, fallbacks);
            atomicAdd(dbg + 8, (uint32_t)((clock64() - cyc_all) >> 6));
            atomicMax(dbg + 13, (uint32_t)((clock64() - cyc_all) >> 6));
            if (blockIdx.x == 0 && blockIdx.y == 0)
                printf("[vqhip-dev] chain4(0,0) mode %d walker: %u batches, %.0f k cycles: waiting for the loader %.0f k, %u failing lanes %.0f k, %u re-added\n",
                       MODE, nb, (double)(clock64() - cyc_all) / 1e3, (double)w_ready / 1e3, n_walks, (double)w_walk / 1e3, fallbacks);
        }
    }
}

// ---- round 5: re-additions looked up, not executed --------------------------------------------------------------------
// On zero-mean columns the running sum is a random walk that keeps crossing binade edges near zero: 10-30 % of the
// 64-row segments are parked (their guess comes close to an edge), half of those really leave the binade their summary
// was folded under, and k_fs_chain above pays ~2000 cycles in series for each (64 dependent additions, an eight-lane scan
// pass, a wave-wide rescan) on the ONE wave a column has.  Two kernels take that work off the column's wave:
//
// k_fs_prep, TABLES.  For a parked segment let F be the map "incoming f32 sum -> sum after the segment's 64 additions".
// Write an incoming sum of the guessed binade by its raw bits r = c0 + 32 m + i (c0 = the guess's bits rounded down to a
// multiple of 32: same sign and exponent).  Half a wave per parked segment, chip-wide: lane i really adds the 64 addends
// to the input c0 + i (T[i]: exact by construction), and
//     F(c0 + 32 m + i) = T[i] +- m * 32 ulp_in        for every m in [mlo, mhi]      (-: negative sums)
// because fl(y + g) = fl(y) + g whenever g is an even multiple of the grid y is rounded on: true for g = 32 ulp_in at
// every step if (1) the partial sums of the shifted input have the same (sign, exponent) as those of c0 at every step
// -- fl(s + x) is monotone in s, so it is enough that the two END points c0 + 32 mlo and c0 + 32 mhi + 31 do: lanes
// 0..15 / 16..31 run those chains for 16 geometrically spaced mlo / mhi, next to a reference chain from c0 (for m < 0 the
// chains from c0 .. c0 + 31 themselves must be on the reference's itinerary too) -- and (2) no partial sum is more than
// four binades above the input (else m = 0 only).  tools/fs_table_sim.py checks the claim by brute force; the trees stay
// bit-identical to the oracle's (tests/test_gpu_tsvq.py, test_gpu_allrows.py).  No table for a segment whose summary
// holds for every sum within 2^11 ulps of the guess, nor under a sampled guess (off by far more than any window).
//
// k_fs_prep, ITEMS.  A batch of 512 segment summaries of one column becomes a list of items, one wave per (batch,
// column): the segments between two terminators composed into ONE run (a segmented scan over the lanes) and written as
// the raw-bit range of incoming sums it holds for plus the raw-bit delta it applies, per parity of the sum -- sign,
// exponent, "stays inside the binade" and the parity streams are all in those two compares.
//
// k_fs_chain3 is then one pass over a column's items: test + apply a run (ten instructions); a parked segment first
// tries its own summary, then its table (one dependent LDS read: T[r & 31] +- m 2^(e-18)), then -- a sum outside
// [mlo, mhi] or in another binade than guessed: a few per cent, the guess is an f64 prefix and the f32 chain has
// drifted from it by its own rounding errors -- the 64 additions from the parked addends, as before.
// end points of the validity test, in periods of 32 ulps, on either side of the table's 32 candidates
__constant__ int32_t kFsTabM[16] = {1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 512, 2048};
constexpr int32_t kFsTabSlack = 1 << 11;  // a summary that holds this far around the guess needs no table
constexpr uint32_t kFsTabN = 32;          // candidates per table: two tables per wave (half a wave each), period 32 ulps

__device__ __forceinline__ void fs_tables_wave(uint32_t lane, uint32_t wave, uint32_t n_waves, const float *__restrict__ side,
                                               const uint4 *__restrict__ side_meta, uint32_t pass_tag,
                                               const uint32_t *__restrict__ side_count, uint32_t side_cap,
                                               float *__restrict__ tab, int4 *__restrict__ tmeta, float *lad /* LDS: [2][128] of this wave */) {
    const uint32_t n_slots = min(*side_count, side_cap), half = lane >> 5, hl = lane & 31u;
    if (2u * wave >= n_slots) return;
    // A pair's operands a pair AHEAD of its additions, none of them in the way: the two segments' 64 addends each by ONE
    // LDS-DMA load into this wave's double buffer (lanes 0..15 / 16..31 fetch the 16-byte pieces of the first / second
    // segment; no destination registers, no copies), the two slot records by ordinary loads that are consumed a trip later.
    // Fetched when needed, a pair was three dependent round trips (slot record -> summary -> addends) in front of ~700
    // instructions, on four waves per SIMD (`SQ_WAIT_ANY` 67 % of the wave cycles).
    const uint32_t lad_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lad);  // (wave-uniform: M0 takes a scalar)
    auto slot_of = [&](uint32_t pair, uint32_t h) { return min(2u * pair + h, n_slots - 1u); };
    auto fetch = [&](uint32_t pair, uint32_t buf) {
        const uint32_t sl = slot_of(pair, (lane >> 4) & 1u);
        fs_dma16(reinterpret_cast<const char *>(side + (size_t)sl * kFsSeg) + 16u * (lane & 15u),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(lad_addr + buf * 1024u)));  // (lanes 32..63: the same again behind it)
    };
    uint32_t pair = wave, buf = 0;
    fetch(pair, 0);
    uint4 sm = side_meta[2 * slot_of(pair, half)], sm2 = side_meta[2 * slot_of(pair, half) + 1];
    for (; 2u * pair < n_slots; pair += n_waves, buf ^= 1u) {
        fs_wait_vm0();  // this pair's addends are in LDS (and its records in their registers)
        fs_wave_lds_sync();
        const uint32_t nxt = (2u * (pair + n_waves) < n_slots) ? pair + n_waves : pair;
        fetch(nxt, buf ^ 1u);
        const uint4 sm_n = side_meta[2 * slot_of(nxt, half)], sm2_n = side_meta[2 * slot_of(nxt, half) + 1];
        const uint32_t slot = slot_of(pair, half);  // (an odd count: the last pair's second half repeats the first's slot)
        const bool dup = 2u * pair + half >= n_slots;
        const uint32_t gb = sm.x, ex = (gb >> 23) & 0xFFu;
        // tmeta = {c0, mlo, mhi, bits of +-2^(e-18)}; mlo > mhi: no table.  None for: the unused rest of a wave's chunk of
        // slots (another pass's tag), a sampled guess, no normal guess (a node's first segment: the sum starts at 0) or
        // 2^(e-18) not normal, a summary that holds well around the guess
        bool none = sm.y != pass_tag || sm.w != 0u || ex < 24u || ex > 240u;
        if (!none && sm2.z == 0u) {
            const int32_t lo = (int32_t)sm2.x, hi = (int32_t)sm2.y, mo = (int32_t)(gb & 0x7FFFFFu);
            none = (gb >> 31) ? (mo - hi - kFsTabSlack > 0 && mo - lo + kFsTabSlack <= 0x7FFFFF)
                              : (mo + lo - kFsTabSlack > 0 && mo + hi + kFsTabSlack <= 0x7FFFFF);
        }
        const uint32_t tag_here = sm.y;
        sm = sm_n, sm2 = sm2_n;
        if (__ballot(!none) == 0ull) {  // uniform: neither half has a table to build
            if (hl == 0 && !dup && tag_here == pass_tag) tmeta[slot] = make_int4(0, 1, 0, 0);
            continue;
        }
        // Lane roles inside the half: both rows of sixteen lanes are complete tests of their own -- lanes 0..7 of a row run the
        // LOWER end points of eight window sizes, lanes 8..15 the UPPER end points of the same eight, and the reference
        // they are compared with is the row's first lane's own chain (c0 / c0 + 16: inside every window), handed along by DPP
        // row_newbcast: no third chain, five instructions per addend.
        const uint32_t c0 = gb & ~(kFsTabN - 1u);
        const uint32_t row = hl >> 4, pos = hl & 15u, kk = 8u * row + (pos & 7u);
        const int32_t M = kFsTabM[kk];
        const uint32_t rB = (pos < 8u) ? c0 - kFsTabN * (uint32_t)M : c0 + kFsTabN * (uint32_t)M + (kFsTabN - 1u);
        const bool inB = ((rB ^ c0) >> 23) == 0u;  // the end point has the guess's sign and exponent
        float sA = __uint_as_float(c0 + hl), sB = __uint_as_float(inB ? rB : c0);
        uint32_t mism = 0u;
        float amax = 0.0f;
        // (the two halves add different segments: sixteen broadcast reads of the half's 256 bytes)
        const f32x4_t *ad = reinterpret_cast<const f32x4_t *>(lad + buf * 256u + half * kFsSeg);
        f32x4_t av[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = ad[u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sA = sA + av[u][w];
                sB = sB + av[u][w];
                const uint32_t bR = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(sA), 0x150, 0xF, 0xF, true);  // row_newbcast:0
                mism |= __float_as_uint(sB) ^ bR;
                amax = fmaxf(amax, fabsf(__uint_as_float(bR)));  // (an inf stays an inf; a NaN can only follow one)
            }
        }
        const bool sameB = inB && (mism & 0xFF800000u) == 0u;
        const uint32_t bal = (uint32_t)(__ballot(sameB) >> (32u * half));
        const uint32_t m_lo = (bal & 0xFFu) | (((bal >> 16) & 0xFFu) << 8), m_hi = ((bal >> 8) & 0xFFu) | ((bal >> 24) << 8);
        int32_t mlo = m_lo ? -kFsTabM[31 - __builtin_clz(m_lo)] : 0, mhi = m_hi ? kFsTabM[31 - __builtin_clz(m_hi)] : 0;
        // The candidates c0 .. c0 + 31 themselves must share one itinerary -- T[i] stands in for inputs on either side, and
        // the second row's reference is c0 + 16, not c0 -- which the smallest upper window (c0 + 63 against c0) proves;
        // without it only m = 0 is left
        if (!(m_hi & 1u)) mlo = mhi = 0;
        // 32 ulp_in must be an even multiple of the coarsest grid met: no partial sum more than four binades above the input
        const bool high = (int)((__float_as_uint(amax) >> 23) & 0xFFu) - (int)ex > 4;
        if ((uint32_t)(__ballot(high) >> (32u * half)) != 0u) mlo = mhi = 0;
        if (none) mlo = 1, mhi = 0;
        if (!dup && tag_here == pass_tag) {
            if (!none) tab[(size_t)slot * kFsTabN + hl] = sA;
            if (hl == 0) tmeta[slot] = make_int4((int32_t)c0, mlo, mhi, (int32_t)((gb & 0x80000000u) | ((ex - 18u) << 23)));
        }
    }
    fs_wait_vm0();  // (the last trip's prefetch: nothing may land in LDS after the wave has gone)
}

// A run of consecutive segments as ONE parity transducer, with the (sign, binade) key its summaries were folded under
// (kFsEAny: no segment yet; kFsEMixed: summaries of different keys -- such a run never passes its test).
constexpr int32_t kFsEAny = -100000, kFsEMixed = -200000;
struct FsR {
    FsT t;
    int32_t e;
};
__device__ __forceinline__ FsR fs_run_none() {
    FsR r;
    r.t.d0 = r.t.d1 = r.t.lo0 = r.t.lo1 = r.t.hi0 = r.t.hi1 = 0;
    r.e = kFsEAny;
    return r;
}
__device__ __forceinline__ FsR fs_run_compose(const FsR &f, const FsR &g) {  // f first, then g
    FsR h;
    h.t = fs_compose(f.t, g.t);
    h.e = (f.e == kFsEAny) ? g.e : ((g.e == kFsEAny || g.e == f.e) ? f.e : kFsEMixed);
    return h;
}

constexpr uint32_t kFsBatch = 64 * kFsSpl;         // segments per batch: one lane's eight are one 512-row tile
constexpr uint32_t kFsItemCap = kFsBatch + 1;      // items of a batch: at most one per segment, plus the end (the two-item form of
                                                   // a parked segment -- types 3 and 4 -- is not emitted: its columns keep k_fs_chain)
// An item is a run and what ends it, 32 bytes: {rmin0, rmax0, dm0, rmin1}, {rmax1, dm1, info, 0}.  With r the raw bits
// of the running sum and p = r & 1 (the parity of S: the same as that of |S|), the run holds iff rmin_p <= r <= rmax_p
// (unsigned: one range per sign and binade) and leaves the sum at r + dm_p.  info = type | tseg << 3 | slot << 12:
//   0 end of the batch           3 parked segment whose own summary is the run: tried first, then its table
//   1 parked segment: its table  4 nothing (closes the run in front of a type-3 item)
//   2 segment without a usable summary: its rows are gathered and added
// tseg = the segment the terminator stands for, inside the batch (types 0: the batch's segment count).
enum : uint32_t { kFsItEnd = 0, kFsItTable = 1, kFsItGather = 2, kFsItSeg = 3, kFsItNop = 4 };
__device__ __forceinline__ int32_t fs_item_info(uint32_t type, uint32_t tseg, uint32_t slot) { return (int32_t)(type | (tseg << 3) | (slot << 12)); }
struct FsItem {
    int4 a, b;
};
__device__ __forceinline__ FsItem fs_item(const FsR &r, int32_t info) {
    FsItem it;
    uint32_t mn0 = 0u, mx0 = 0xFFFFFFFFu, mn1 = 0u, mx1 = 0xFFFFFFFFu;  // no segment: every sum passes, nothing moves
    int32_t dm0 = 0, dm1 = 0;
    if (r.e != kFsEAny) {
        const bool neg = (r.e >> 8) & 1;
        const uint32_t base = (neg ? 0x80000000u : 0u) | ((uint32_t)((r.e & 0xFF) - 1) << 23);
        const int32_t lim = 0x7FFFFF;
        const bool sane = r.e != kFsEMixed && r.t.lo0 >= -lim && r.t.lo1 >= -lim && r.t.hi0 <= lim && r.t.hi1 <= lim;
        // mantissa offsets o = |S| - 2^23 the run holds for (fs_inside): S + lo > 2^23, S + hi <= 2^24 - 1, mirrored for S < 0
        mn0 = base + (uint32_t)(neg ? r.t.hi0 + 1 : 1 - r.t.lo0), mx0 = base + (uint32_t)(neg ? lim + r.t.lo0 : lim - r.t.hi0);
        mn1 = base + (uint32_t)(neg ? r.t.hi1 + 1 : 1 - r.t.lo1), mx1 = base + (uint32_t)(neg ? lim + r.t.lo1 : lim - r.t.hi1);
        dm0 = neg ? -r.t.d0 : r.t.d0, dm1 = neg ? -r.t.d1 : r.t.d1;
        if (!sane) mn0 = mn1 = 1u, mx0 = mx1 = 0u;
    }
    it.a = make_int4((int32_t)mn0, (int32_t)mx0, dm0, (int32_t)mn1);
    it.b = make_int4((int32_t)mx1, dm1, info, 0);
    return it;
}

// m / m2: the lane's eight summaries (even / odd stream) of segments 8 lane .. 8 lane + 7 of the batch; cnt: the batch's segments
__device__ __forceinline__ void fs_items_wave(uint32_t lane, uint32_t c, uint32_t d, uint32_t G, uint32_t cnt, FsS (&m)[kFsSpl],
                                              FsS (&m2)[kFsSpl], int4 *__restrict__ items, uint32_t *__restrict__ ihdr, bool seg_first) {
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j)
        if (kFsSpl * lane + (uint32_t)j >= cnt) m[j].d = m[j].lo = m[j].hi = 0, m[j].ef = -1, m2[j] = m[j];  // past the node's end: ef = -1 marks it
    // terminators: parked segments (bit j of pk; of pu if their own summary is usable and worth trying first: under a
    // sampled guess and in the variance pass most parked segments do hold -- the guess is the rough part -- while the
    // mean pass of a zero-mean column, with its exact guess, parks what really leaves its binade: 96 % fail) and segments
    // without a usable summary (gathered)
    uint32_t pk = 0, pu = 0, tm = 0;
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j) {
        const bool live = m[j].ef != -1;
        const bool parked = live && fs_ef_slot(m[j].ef) >= 0, bad = live && (m[j].ef & 1);
        pk |= parked ? (1u << j) : 0u;
        pu |= (parked && !bad && seg_first && kFsItemCap > 2 * kFsBatch) ? (1u << j) : 0u;  // (two items per segment need the room)
        tm |= (parked || bad) ? (1u << j) : 0u;
    }
    auto seg_run = [&](int j) {
        FsR r;
        r.t.d0 = m[j].d, r.t.lo0 = m[j].lo, r.t.hi0 = m[j].hi, r.t.d1 = m2[j].d, r.t.lo1 = m2[j].lo, r.t.hi1 = m2[j].hi;
        r.e = fs_ef_key(m[j].ef);
        return r;
    };
    FsR agg = fs_run_none();  // the segments behind the lane's last terminator (all of them if it has none)
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j) {
        const FsR nx = fs_run_compose(agg, seg_run(j));
        const bool is_t = (tm >> j) & 1u, live = m[j].ef != -1;
        agg = is_t ? fs_run_none() : (live ? nx : agg);
    }
    FsR v = agg;
    int vf = tm != 0u ? 1 : 0;
    {
#define VQ_FS_DPP(XX, CTRL) __builtin_amdgcn_update_dpp(0, XX, CTRL, 0xF, 0xF, true)
#define VQ_FS_SEG(CTRL, COND)                                                                                            \
    {                                                                                                                    \
        FsR p;                                                                                                           \
        p.t.d0 = VQ_FS_DPP(v.t.d0, CTRL), p.t.d1 = VQ_FS_DPP(v.t.d1, CTRL), p.t.lo0 = VQ_FS_DPP(v.t.lo0, CTRL);          \
        p.t.lo1 = VQ_FS_DPP(v.t.lo1, CTRL), p.t.hi0 = VQ_FS_DPP(v.t.hi0, CTRL), p.t.hi1 = VQ_FS_DPP(v.t.hi1, CTRL);      \
        p.e = VQ_FS_DPP(v.e, CTRL);                                                                                      \
        const int pf = VQ_FS_DPP(vf, CTRL);                                                                              \
        const FsR cv = fs_run_compose(p, v);                                                                             \
        const bool take = (COND) && !vf;                                                                                 \
        v = take ? cv : v, vf = take ? pf : vf;                                                                          \
    }
        VQ_FS_SEG(0x111, (lane & 15u) >= 1u)
        VQ_FS_SEG(0x112, (lane & 15u) >= 2u)
        VQ_FS_SEG(0x114, (lane & 15u) >= 4u)
        VQ_FS_SEG(0x118, (lane & 15u) >= 8u)
        VQ_FS_SEG(0x142, (lane & 16u) != 0u)
        VQ_FS_SEG(0x143, lane >= 32u)
#undef VQ_FS_SEG
#undef VQ_FS_DPP
    }
    FsR open;  // the run open when this lane starts: the inclusive value of the lane in front
    open.t.d0 = __shfl_up(v.t.d0, 1), open.t.d1 = __shfl_up(v.t.d1, 1), open.t.lo0 = __shfl_up(v.t.lo0, 1);
    open.t.lo1 = __shfl_up(v.t.lo1, 1), open.t.hi0 = __shfl_up(v.t.hi0, 1), open.t.hi1 = __shfl_up(v.t.hi1, 1);
    open.e = __shfl_up(v.e, 1);
    if (lane == 0) open = fs_run_none();
    // items of this lane: one per terminator, two where a usable parked segment follows a run that is not empty
    uint32_t mine_cnt = 0;
    {
        bool empty = open.e == kFsEAny;
#pragma unroll
        for (int j = 0; j < kFsSpl; ++j) {
            const bool is_t = (tm >> j) & 1u, live = m[j].ef != -1;
            mine_cnt += is_t ? (((pu >> j) & 1u) && !empty ? 2u : 1u) : 0u;
            empty = is_t ? true : (live ? false : empty);
        }
    }
    uint32_t incl = mine_cnt;
    {
#define VQ_FS_ADD(CTRL, COND) { const int32_t t = __builtin_amdgcn_update_dpp(0, (int32_t)incl, CTRL, 0xF, 0xF, true); if (COND) incl += (uint32_t)t; }
        VQ_FS_ADD(0x111, (lane & 15u) >= 1u)
        VQ_FS_ADD(0x112, (lane & 15u) >= 2u)
        VQ_FS_ADD(0x114, (lane & 15u) >= 4u)
        VQ_FS_ADD(0x118, (lane & 15u) >= 8u)
        VQ_FS_ADD(0x142, (lane & 16u) != 0u)
        VQ_FS_ADD(0x143, lane >= 32u)
#undef VQ_FS_ADD
    }
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + 1u;
    int4 *const out = items + ((size_t)G * d + c) * kFsItemCap * 2;
    uint32_t at = incl - mine_cnt;
    auto put = [&](const FsR &r, int32_t info) {
        const FsItem it = fs_item(r, info);
        out[2 * at] = it.a, out[2 * at + 1] = it.b;
        ++at;
    };
    FsR r = open;
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j) {
        const bool is_t = (tm >> j) & 1u, live = m[j].ef != -1;
        const uint32_t tseg = kFsSpl * lane + (uint32_t)j;
        if (is_t) {
            const bool parked = (pk >> j) & 1u;
            const uint32_t slot = parked ? (uint32_t)fs_ef_slot(m[j].ef) : 0u;
            if ((pu >> j) & 1u) {
                if (r.e != kFsEAny) put(r, fs_item_info(kFsItNop, tseg, 0u));
                put(seg_run(j), fs_item_info(kFsItSeg, tseg, slot));
            } else {
                put(r, fs_item_info(parked ? kFsItTable : kFsItGather, tseg, slot));
            }
        }
        const FsR nx = fs_run_compose(r, seg_run(j));
        r = is_t ? fs_run_none() : (live ? nx : r);
    }
    if (lane == 63u) {
        put(r, fs_item_info(kFsItEnd, cnt, 0u));
        ihdr[(size_t)G * d + c] = n_items;
    }
}

// tables and items in ONE launch (both only need the fold's output): workgroups [0, n_tab_blocks) run the tables' body,
// the others take one batch x four adjacent columns each (their summaries share 64-byte sectors)
__global__ __launch_bounds__(256) void k_fs_prep(const float *__restrict__ side, const uint4 *__restrict__ side_meta, uint32_t pass_tag,
                                                 const uint32_t *__restrict__ side_count, uint32_t side_cap, float *__restrict__ tab,
                                                 int4 *__restrict__ tmeta, uint32_t n_tab_blocks, uint32_t d,
                                                 const uint32_t *__restrict__ fast_nodes, const uint32_t *__restrict__ tile_base,
                                                 const uint2 *__restrict__ batch_tab, NodeArrays na, const FsS *__restrict__ summ,
                                                 const FsS *__restrict__ summ_odd, int4 *__restrict__ items, uint32_t *__restrict__ ihdr,
                                                 const LevelInfo *__restrict__ lv, const uint32_t *__restrict__ policy, int seg_first_default) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (blockIdx.x < n_tab_blocks) {
        __shared__ __attribute__((aligned(16))) float lad[4][512];  // per wave: two 1 KB landing buffers of a pair's addends
        fs_tables_wave(lane, blockIdx.x * 4u + w, n_tab_blocks * 4u, side, side_meta, pass_tag, side_count, side_cap, tab, tmeta, &lad[w][0]);
        return;
    }
    // ---- items: one batch x four adjacent columns per workgroup, a wave per column.  The summaries are [column][segment]:
    // a lane's eight are one 128-byte line
    const uint32_t cq = (d + 3) / 4, q = blockIdx.x - n_tab_blocks, G = q / cq, c0 = (q - G * cq) * 4u, c = c0 + w;
    if (G >= lv->pad) return;  // (pad = the level's batches; launched over an upper bound)
    const bool seg_first = policy ? policy[c0 / kFsCols] != 0u : seg_first_default != 0;  // (policy: 1 = sampled guess allowed for the column block)
    if (seg_first) return;  // columns under a sampled guess keep k_fs_chain (uniform over the workgroup: four columns of one block of 32)
    if (c >= d) return;
    const uint2 bt = batch_tab[G];
    const uint32_t node = fast_nodes[bt.x], len = na.seg_len[node];
    const uint32_t nseg = (len + kFsSeg - 1) / kFsSeg, t0 = bt.y * kFsBatch, cnt = min(kFsBatch, nseg - t0);
    const size_t seg0 = (size_t)tile_base[bt.x] * kFsSegsPerTile;
    const FsS *gp = summ + (size_t)c * na.fs_seg_stride + seg0, *gp2 = summ_odd + (size_t)c * na.fs_seg_stride + seg0;
    FsS m[kFsSpl], m2[kFsSpl];
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j) m[j] = gp[min(t0 + kFsSpl * lane + (uint32_t)j, nseg - 1u)];  // unconditional, clamped
    // the odd streams of the runs with an exact tie, else a copy of the even ones
#pragma unroll
    for (int j = 0; j < kFsSpl; ++j) {
        m2[j] = m[j];
        if (kFsSpl * lane + (uint32_t)j < cnt && (m[j].ef & 3) == 2) {
            const FsS o = gp2[t0 + kFsSpl * lane + (uint32_t)j];
            m2[j].d = o.d, m2[j].lo = o.lo, m2[j].hi = o.hi;
        }
    }
    fs_items_wave(lane, c, d, G, cnt, m, m2, items, ihdr, seg_first);
}

// The exact chain, third form: one wave per (node, column), ONE pass over the column's items.  Items travel 64 at a time
// (a chunk): the chunk after the one being walked sits in LDS already, the one after that is in flight in registers;
// ALL tables of a chunk's parked segments (at most 64: 16 KB) are requested a chunk ahead and sit in LDS while the chunk
// is walked, their validity records (tmeta) next to the items.  The loop over a chunk's items has ONE branch per item:
// the common outcomes -- the run holds, and the item ends in nothing or in a table whose window holds the sum -- are
// computed without a test between them (the table's entry is read speculatively), everything else (a run that does not
// hold: walked segment by segment from the summaries in memory, or -- a parked segment's own summary -- followed by its
// table; a sum outside the table's window: the 64 additions from the parked addends; a segment without a summary: its
// rows gathered) leaves the loop for one item.  Every lane carries the same sum: all branches are uniform.
template <int MODE, bool DBG>
__global__ __launch_bounds__(64) void k_fs_chain3(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                  const uint32_t *__restrict__ fast_nodes, const uint32_t *__restrict__ tile_base,
                                                  const uint32_t *__restrict__ batch_base, NodeArrays na,
                                                  const FsS *__restrict__ summ, const FsS *__restrict__ summ_odd,
                                                  const float *__restrict__ side, const float *__restrict__ tab,
                                                  const int4 *__restrict__ tmeta, const int4 *__restrict__ items,
                                                  const uint32_t *__restrict__ ihdr, uint32_t *__restrict__ n_fallback,
                                                  const LevelInfo *__restrict__ lv, uint32_t *__restrict__ dbg_arg,
                                                  const uint32_t *__restrict__ skip_sampled) {
    uint32_t *const dbg = DBG ? dbg_arg : nullptr;
    // the chunk being walked / the next one: per item {run fields a, b; m = its table's validity record, or a record that
    // always / never holds for an item without a table: the loop tests nothing else}, one 48-byte record (one address)
    struct __attribute__((aligned(16))) LItem {
        int4 a, b, m;
    };
    __shared__ LItem lit[2][64];
    __shared__ int plist[64];                                               // side slots of the next chunk's parked items, in order
    __shared__ __attribute__((aligned(16))) float ltab[64 * kFsTabN];       // the tables of the chunk being walked
    __shared__ __attribute__((aligned(16))) float ladd[64];                 // addends of a segment being re-added
    __shared__ __attribute__((aligned(16))) FsS lwk[64], lwk2[64];          // summaries of a run being walked
    __shared__ uint32_t lhdr[1024];                                        // items per batch
    if (blockIdx.x >= lv->n_fast) return;
    if (skip_sampled && skip_sampled[blockIdx.y / kFsCols] != 0u) return;  // columns under a sampled guess keep k_fs_chain
    const uint32_t fidx = blockIdx.x, node = fast_nodes[fidx], c = blockIdx.y, lane = threadIdx.x;
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nseg = (len + kFsSeg - 1) / kFsSeg, nbat = (nseg + kFsBatch - 1) / kFsBatch, G0 = batch_base[fidx];
    const size_t seg0 = (size_t)tile_base[fidx] * kFsSegsPerTile;
    const FsS *sp = summ + (size_t)c * na.fs_seg_stride + seg0, *sp2 = summ_odd + (size_t)c * na.fs_seg_stride + seg0;
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    uint32_t n_readd = 0, n_hit = 0, n_miss = 0, n_slow = 0, n_parked = 0, n_own = 0, n_out = 0;
    unsigned long long cy_all = DBG ? clock64() : 0ull, cy_out = 0ull, cy_miss = 0ull, cy_stage = 0ull;  // VQHIP_TSVQ_DEBUG
    plist[lane] = 0;  // (clamped table requests read entries nobody wrote: slot 0 is a valid address)
    auto add64 = [&](const float *lds64) {  // 64 additions in row order, the addends by broadcast LDS reads
        const f32x4_t *l4 = reinterpret_cast<const f32x4_t *>(lds64);
        f32x4_t rq[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) rq[g] = l4[g];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            s = s + rq[g][0];
            s = s + rq[g][1];
            s = s + rq[g][2];
            s = s + rq[g][3];
        }
    };
    auto readd_from = [&](float vv) {
        fs_wave_lds_sync();
        ladd[lane] = vv;
        fs_wave_lds_sync();
        add64(ladd);
        ++n_readd;
    };
    auto gather_add = [&](uint32_t seg) {  // the 64 rows of a segment through `perm`, added in row order
        const uint32_t r0 = seg * kFsSeg, rows_here = min(kFsSeg, len - r0);
        readd_from((lane < rows_here) ? fs_value<MODE>(X[(size_t)perm[a + r0 + lane] * d + c], mu) : 0.0f);  // (+0.0 past the end)
    };
    // a run that did not hold: its segments [from, to) of the column one by one -- 64 summaries at a time through LDS
    auto walk = [&](uint32_t from, uint32_t to) {
        ++n_slow;
        for (uint32_t q0 = from; q0 < to; q0 += 64u) {
            const uint32_t q = min(q0 + lane, to - 1u);
            const FsS g = sp[q];
            FsS g2 = g;
            if ((g.ef & 3) == 2) g2 = sp2[q];
            fs_wave_lds_sync();
            lwk[lane] = g, lwk2[lane] = g2;
            fs_wave_lds_sync();
            const uint32_t nq = min(64u, to - q0);
            for (uint32_t k = 0; k < nq; ++k) {
                const FsS h = lwk[k], h2 = lwk2[k];
                const uint32_t sb2 = __float_as_uint(s), se2 = (sb2 >> 23) & 0xFFu;
                const int32_t mg2 = (int32_t)((sb2 & 0x7FFFFFu) | 0x800000u);
                const int32_t Sq = (sb2 >> 31) ? -mg2 : mg2;
                const bool oq = (Sq & 1) != 0;
                const bool okq = (se2 != 0u) && (se2 != 255u) && !(h.ef & 1) && ((int)se2 - 127 == fs_ef_e(h.ef)) &&
                                 fs_inside(Sq, oq ? h2.lo : h.lo, oq ? h2.hi : h.hi);
                if (__builtin_amdgcn_readfirstlane(okq ? 1 : 0)) {
                    const int32_t S2 = Sq + (oq ? h2.d : h.d);
                    const uint32_t m2a = (uint32_t)(S2 < 0 ? -S2 : S2);
                    s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se2 << 23) | (m2a & 0x7FFFFFu));
                } else {
                    gather_add(q0 + k);
                }
            }
        }
    };
    // items per batch: the first 1024 batches' counts sit in LDS (one pass at the start: a count read when the cursor gets
    // there would be a memory round trip per batch); longer columns read the rest from memory
    constexpr uint32_t kHdrLds = 1024;
    for (uint32_t b = lane; b < min(nbat, kHdrLds); b += 64) lhdr[b] = ihdr[(size_t)(G0 + b) * d + c];
    fs_wave_lds_sync();
    auto hdr = [&](uint32_t b) -> uint32_t {
        const uint32_t bb = min(b, nbat - 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(bb < kHdrLds ? lhdr[bb] : ihdr[(size_t)(G0 + bb) * d + c]));
    };
    // the chunks of the column in order: (batch, chunk inside the batch); a cursor names the next chunk to REQUEST
    uint32_t q_b = 0, q_k = 0, q_n = hdr(0);
    int4 rA, rB;              // the chunk in flight
    uint32_t rn = 0, rt = 0;  // its items (0: past the column's end), its batch's first segment
    auto request = [&]() {
        const bool live = q_b < nbat;
        const uint32_t bb = live ? q_b : nbat - 1u;
        rn = live ? min(64u, q_n - q_k * 64u) : 0u;
        rt = bb * kFsBatch;
        const uint32_t i = min(q_k * 64u + lane, kFsItemCap - 1u);
        const int4 *src = items + (((size_t)(G0 + bb) * d + c) * kFsItemCap + i) * 2;
        rA = src[0], rB = src[1];
        if (live) {
            if ((q_k + 1u) * 64u < q_n) ++q_k;
            else ++q_b, q_k = 0u, q_n = hdr(q_b);
        }
    };
    // stage the chunk in flight: registers -> LDS buffer bf (each parked item gets the byte offset of its table in `ltab`),
    // its parked items listed, ALL their tables (pf: a chunk ahead of their use) and their validity records requested
    float pf[32];
    int4 rM = make_int4(0, 1, 0, 0);
    uint32_t stype = 0, ctype = 0;  // the type of this lane's item in the chunk just staged / being walked
    auto stage = [&](int bf) -> uint32_t {
        const uint32_t type = (uint32_t)rB.z & 7u;
        const bool parked = lane < rn && (type == kFsItTable || type == kFsItSeg);
        const uint64_t pmask = __ballot(parked);
        const uint32_t prank = (uint32_t)__builtin_popcountll(pmask & ((1ull << lane) - 1ull));
        const uint32_t slot = parked ? ((uint32_t)rB.z >> 12) : 0u;
        // The record as the loop wants it: every per-parity field as "even value + (r & 1) * difference" (one v_mad_i32_i24
        // behind the parity bit instead of compare + select), ranges as (start, span) for ONE unsigned compare:
        //   a = {rmin0, span0, r + dm0 's dm0, rmin1 - rmin0},  b = {span1 - span0, dm1 - dm0, info, table offset | table flag}
        // A run whose differences do not fit 24 bits is made one that never holds (rmin0 = 1, span0 = 0 means r == 1 only: a
        // subnormal sum no run holds for anyway; then differences 0): it is walked.
        {
            uint32_t rmin0 = (uint32_t)rA.x, span0 = (uint32_t)rA.y - (uint32_t)rA.x, rmin1 = (uint32_t)rA.w, span1 = (uint32_t)rB.x - (uint32_t)rA.w;
            if ((uint32_t)rA.y < (uint32_t)rA.x) rmin0 = 1u, span0 = 0u;  // an empty range (first > last)
            if ((uint32_t)rB.x < (uint32_t)rA.w) rmin1 = 1u, span1 = 0u;
            int32_t dm0 = rA.z, dr = (int32_t)(rmin1 - rmin0), ds = (int32_t)(span1 - span0), dd = rB.y - rA.z;
            const int32_t lim24 = (1 << 23) - 1;
            if (dr < -lim24 || dr > lim24 || ds < -lim24 || ds > lim24 || dd < -lim24 || dd > lim24) rmin0 = 1u, span0 = 0u, dr = ds = dd = 0;
            lit[bf][lane].a = make_int4((int32_t)rmin0, (int32_t)span0, dm0, dr);
            // (b.w: the byte offset of the item's table in `ltab`, sign bit set for an item that ends in a table)
            lit[bf][lane].b = make_int4(ds, dd, rB.z, (int32_t)(parked ? (prank * kFsTabN * 4u) | (type == kFsItTable ? 0x80000000u : 0u) : 0u));
        }
        stype = type;
        fs_wave_lds_sync();  // (the previous chunk's table requests have read plist)
        if (parked) plist[prank] = (int)slot;
        rM = tmeta[slot];
        fs_wave_lds_sync();
        const uint32_t ptotal = (uint32_t)__builtin_popcountll(pmask);
        const uint32_t last = max(ptotal, 1u) - 1u;
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8)
            if ((uint32_t)(16 * g8) < ptotal) {  // uniform; a load fetches two tables (half a wave each)
#pragma unroll
                for (int k = 8 * g8; k < 8 * g8 + 8; ++k)
                    pf[k] = tab[(size_t)plist[min(2u * (uint32_t)k + (lane >> 5), last)] * kFsTabN + (lane & 31u)];
            }
        return ptotal;
    };
    request();
    uint32_t ptot_next = stage(0);      // chunk 0 -> buffer 0
    ctype = stype;
    uint32_t cn = rn, ct0 = rt;         // the chunk being walked: items, first segment of its batch
    request();                          // chunk 1 in flight
    int buf = 0;
    uint32_t pos = 0;                   // next segment of the batch not yet consumed when the chunk starts
    while (cn != 0u) {                  // uniform
        const unsigned long long q_st = DBG ? clock64() : 0ull;
        // this chunk: its validity records and tables (requested a chunk ago) into LDS; the NEXT chunk: out of its registers
        // into the other buffer, its tables requested; the chunk after it requested
        // (an item that ends in nothing passes the table test whatever the sum, a gathered one never does; a parked segment
        // whose own summary is tried first -- not on the columns this kernel serves -- takes the slow path for its table)
        // m = {c0 + 32 mlo, mlo, mhi - mlo, +-2^(e-18)}: with mml = (r' - m.x) >> 5 the window test is 0 <= mml <= m.z (another sign
        // or binade: mml far outside); m.z = -1 (no table, or not this kernel's to use) never passes.
        {
            const bool tb = ctype == kFsItTable, valid = tb && rM.y <= rM.z;
            lit[buf][lane].m = make_int4((int32_t)((uint32_t)rM.x + 32u * (uint32_t)rM.y), rM.y, valid ? rM.z - rM.y : -1, rM.w);
        }
        const uint32_t ptot_cur = ptot_next;
        fs_wave_lds_sync();  // the reads of ltab and of the other buffer (the chunk walked before this one) are done
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8)
            if ((uint32_t)(16 * g8) < ptot_cur) {  // uniform
#pragma unroll
                for (int k = 8 * g8; k < 8 * g8 + 8; ++k) ltab[k * 64 + (int)lane] = pf[k];  // tables 2k and 2k + 1
            }
        if (DBG) n_parked += ptot_cur;
        const uint32_t nn = rn, nt0 = rt;
        ptot_next = stage(buf ^ 1);
        const uint32_t ntype = stype;
        request();
        if (DBG) cy_stage += clock64() - q_st;
        // (the item index goes through fs_vgpr: an LDS read at a uniform address is moved to SGPRs -- eleven v_readfirstlane --
        // right behind the read, i.e. the wave waits for the item it has just asked for; as vector registers the next
        // item's fields are waited for when they are used, an item later)
        int4 A = lit[buf][fs_vgpr(0u)].a, B = lit[buf][fs_vgpr(0u)].b, M = lit[buf][fs_vgpr(0u)].m;
        for (uint32_t i = 0; i < cn; ++i) {
            const uint32_t inx = fs_vgpr(min(i + 1u, 63u));
            const int4 nA = lit[buf][inx].a, nB = lit[buf][inx].b, nM = lit[buf][inx].m;  // (do not depend on s: in flight behind this item)
            // The dependent chain from the sum to the next sum, kept short: parity bit -> one multiply-add per per-parity
            // field -> r' = r + dm -> table address -> LDS -> one addition.  The tests (run range: one unsigned compare;
            // table window: 0 <= mml <= span) hang off it and meet in ONE ballot: the only branch of an item.
            const uint32_t sb = __float_as_uint(s);
            const int32_t odd = (int32_t)(sb & 1u);
            const uint32_t rmin = (uint32_t)fs_mad24(odd, A.w, A.x), span = (uint32_t)fs_mad24(odd, B.x, A.y);
            const uint32_t s1 = (uint32_t)fs_mad24(odd, B.y, (int32_t)(sb + (uint32_t)A.z));
            const bool is_tab = B.w < 0;
            const int32_t mml = (int32_t)(s1 - (uint32_t)M.x) >> 5;
            const float tv = ltab[(((uint32_t)B.w & 0x7FFFFFFFu) >> 2) + (s1 & 31u)];
            // an item that ends in a table needs its window; one that ends in nothing (the batch's end) needs nothing more;
            // anything else (a gathered segment, forms this kernel does not serve) leaves the loop: its b.z type is not 0
            const bool tail_ok = is_tab ? (mml >= 0 && mml <= M.z) : (((uint32_t)B.z & 7u) == kFsItEnd);
            const bool fast = (sb - rmin) <= span && tail_ok;
            if (__ballot(fast) != 0ull) {  // (every lane holds the same sum: any == all)
                s = is_tab ? tv + (float)(mml + M.y) * __int_as_float(M.w) : __uint_as_float(s1);
                if (DBG) n_hit += is_tab ? 1u : 0u;
            } else {
                const unsigned long long q_o = DBG ? clock64() : 0ull;
                ++n_out;
                const uint32_t info = (uint32_t)__builtin_amdgcn_readfirstlane(B.z), ty = info & 7u, tseg = (info >> 3) & 511u;
                const bool ok = (sb - rmin) <= span;
                // where the run starts: behind what the item in front consumed
                uint32_t from = pos;
                if (i > 0u) {
                    const uint32_t pi = (uint32_t)__builtin_amdgcn_readfirstlane(lit[buf][i - 1u].b.z), pt = pi & 7u, ps = (pi >> 3) & 511u;
                    from = pt == kFsItEnd ? 0u : (pt == kFsItNop ? ps : ps + 1u);
                }
                // (a parked segment whose own summary comes first -- no such item on the columns this kernel serves -- is re-added
                // when that summary does not hold: m is its table record only for type 1)
                bool need_table = ty == kFsItTable, table_ok = ty == kFsItTable;
                if (__ballot(ok) != 0ull) s = __uint_as_float(s1);
                else if (ty == kFsItSeg) need_table = true;
                else walk(ct0 + from, ct0 + (ty == kFsItEnd ? info >> 3 : tseg));
                if (need_table) {
                    const uint32_t sb3 = __float_as_uint(s);
                    const int32_t m3 = (int32_t)(sb3 - (uint32_t)M.x) >> 5;
                    const bool hit3 = table_ok && m3 >= 0 && m3 <= M.z;
                    if (__ballot(hit3) != 0ull) {
                        const float tv3 = ltab[(((uint32_t)B.w & 0x7FFFFFFFu) >> 2) + (sb3 & 31u)];
                        s = tv3 + (float)(m3 + M.y) * __int_as_float(M.w);
                        if (DBG) ++n_hit;
                    } else {
                        const unsigned long long q_ms = DBG ? clock64() : 0ull;
                        readd_from(side[(size_t)(info >> 12) * kFsSeg + lane]);
                        if (DBG) {
                            asm volatile("" ::"v"(s));
                            cy_miss += clock64() - q_ms;
                        }
                        ++n_miss;
                    }
                } else if (ty == kFsItGather) {
                    gather_add(ct0 + tseg);
                }
                if (DBG) {
                    asm volatile("" ::"v"(s));
                    cy_out += clock64() - q_o;
                }
            }
            A = nA, B = nB, M = nM;
        }
        {  // what this chunk's last item consumed: where a run that continues in the next chunk starts
            const uint32_t li = (uint32_t)__builtin_amdgcn_readfirstlane(lit[buf][cn - 1u].b.z), lt = li & 7u, ls = (li >> 3) & 511u;
            pos = lt == kFsItEnd ? 0u : (lt == kFsItNop ? ls : ls + 1u);
        }
        buf ^= 1;
        cn = nn, ct0 = nt0, ctype = ntype;
    }
    if (lane == 0) {
        if (MODE == 0) na.centroid[(size_t)node * d + c] = s / (float)len;  // T::from_usize(n)
        else na.var[(size_t)node * d + c] = s;
        if (n_fallback && n_readd) atomicAdd(n_fallback, n_readd);
        if (dbg) {  // VQHIP_TSVQ_DEBUG: chains, re-added segments (most in one chain), table hits / misses, runs walked, core cycles
            atomicAdd(dbg + 0, 1u);
            atomicAdd(dbg + 1, n_readd);
            atomicMax(dbg + 2, n_readd);
            atomicAdd(dbg + 3, n_hit);
            atomicAdd(dbg + 4, n_miss);
            atomicAdd(dbg + 5, n_slow);
            atomicAdd(dbg + 6, n_own);
            atomicAdd(dbg + 7, n_out);
            atomicAdd(dbg + 14, n_parked);
            atomicAdd(dbg + 8, (uint32_t)((clock64() - cy_all) >> 6));
            atomicAdd(dbg + 9, (uint32_t)(cy_out >> 6));
            atomicAdd(dbg + 10, (uint32_t)(cy_miss >> 6));
            atomicAdd(dbg + 11, (uint32_t)(cy_stage >> 6));
            atomicMax(dbg + 13, (uint32_t)((clock64() - cy_all) >> 6));
        }
    }
}

// debug aid (VQHIP_TSVQ_CHECK=1): per (node, column) walk the segments one by one, compare the summary-applied
// sum with the row-by-row sum and report the first disagreement
template <int MODE>
__global__ __launch_bounds__(64) void k_fs_check(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ fast_nodes,
                                                 const uint32_t *__restrict__ tile_base, NodeArrays na,
                                                 const FsS *__restrict__ summ, const FsS *__restrict__ summ_odd,
                                                 const LevelInfo *__restrict__ lv) {
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t node = fast_nodes[blockIdx.x], c = blockIdx.y;
    if (threadIdx.x != 0) return;
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nseg = (len + kFsSeg - 1) / kFsSeg;
    const size_t seg0 = (size_t)tile_base[blockIdx.x] * kFsSegsPerTile;
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    for (uint32_t t = 0; t < nseg; ++t) {
        const FsS sm = summ[(size_t)c * na.fs_seg_stride + seg0 + t];
        FsS so = sm;
        if ((sm.ef & 3) == 2) so = summ_odd[(size_t)c * na.fs_seg_stride + seg0 + t];
        float seq = s;
        const uint32_t r0 = t * kFsSeg, r1 = min(len, r0 + kFsSeg);
        for (uint32_t r = r0; r < r1; ++r) seq = seq + fs_value<MODE>(X[(size_t)perm[a + r] * d + c], mu);
        const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
        if (((sm.ef & 1) == 0) && se != 0u && se != 255u && ((int)se - 127 == fs_ef_e(sm.ef))) {
            const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
            const int32_t S = (sb >> 31) ? -mag : mag;
            const bool odd = (S & 1) != 0;
            const int32_t D = odd ? so.d : sm.d, lo = odd ? so.lo : sm.lo, hi = odd ? so.hi : sm.hi;
            if (fs_inside(S, lo, hi)) {
                const int32_t S2 = S + D;
                const uint32_t m2 = (uint32_t)(S2 < 0 ? -S2 : S2);
                const float fast = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2 & 0x7FFFFFu));
                if (__float_as_uint(fast) != __float_as_uint(seq)) {
                    printf("[fs_check] mode %d node %u col %u segment %u: s=%.9g (S=%d odd=%d e=%d) summary D=%d lo=%d hi=%d -> %.9g, row by row %.9g\n",
                           MODE, node, c, t, s, S, (int)odd, fs_ef_e(sm.ef), D, lo, hi, fast, seq);
                    return;
                }
            }
        }
        s = seq;
    }
}

// f16 image of the node centroids (RNE, half::f16::from_f32 as src/tsvq.rs:249-253 applies per call)
__global__ __launch_bounds__(256) void k_tsvq_table_f16(const float *__restrict__ centroids, uint64_t total,
                                                        uint16_t *__restrict__ table) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < total) table[e] = __half_as_ushort(__float2half_rn(centroids[e]));
}

// out[row] = table[leaf[row]], 16 bytes per lane (d % 8 == 0)
__global__ __launch_bounds__(256) void k_tsvq_gather_f16v(const uint4 *__restrict__ table, uint32_t d8,
                                                          const int32_t *__restrict__ leaf, uint64_t n,
                                                          uint4 *__restrict__ out) {
    const uint64_t total = n * d8;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint64_t row = e / d8;
        const uint32_t t = (uint32_t)(e - row * d8);
        out[e] = table[(size_t)leaf[row] * d8 + t];
    }
}

}  // namespace

int launch_tsvq_table_f16(const float *centroids, uint32_t n_nodes, uint32_t d, uint16_t *table, hipStream_t stream) {
    const uint64_t total = (uint64_t)n_nodes * d;
    hipLaunchKernelGGL(k_tsvq_table_f16, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, centroids, total,
                       table);
    VQ_LAUNCH_CHECK("k_tsvq_table_f16");
    return VQHIP_OK;
}

int launch_tsvq_gather_table(const uint16_t *table, uint32_t d, const int32_t *leaf, uint64_t n, uint16_t *f16_out,
                             hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    const uint32_t d8 = d / 8;
    uint64_t blocks = (n * d8 + 255) / 256;
    if (blocks > (uint64_t)num_cus() * 16) blocks = (uint64_t)num_cus() * 16;
    hipLaunchKernelGGL(k_tsvq_gather_f16v, dim3((uint32_t)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(table), d8, leaf, n, reinterpret_cast<uint4 *>(f16_out));
    VQ_LAUNCH_CHECK("k_tsvq_gather_f16v");
    return VQHIP_OK;
}

// ---- device-side level planning -------------------------------------------------------------------
// The host used to decide, level by level, which nodes split, which take the tile-parallel sums and what their
// tile table looks like -- one stream synchronisation and ~15 small uploads per level (1.4 of 8.6 ms at 1M x 128,
// depth 8).  Two single-workgroup kernels do it on the device instead; the host only launches, over upper bounds.

// start of a level: the split list (+ remap level-local -> split-local), the fast / slow lists and the tile table
__device__ inline void plan_level_body(LevelInfo *__restrict__ lv, int can_split, int can_fast, uint32_t fs_min_rows, NodeArrays na,
                                       uint32_t *__restrict__ lvl_split, uint32_t *__restrict__ remap,
                                       uint32_t *__restrict__ fast_nodes, uint32_t *__restrict__ slow_nodes,
                                       uint32_t *__restrict__ tile_base, uint32_t *__restrict__ n_tiles_of,
                                       FsTile *__restrict__ tiles, uint32_t *__restrict__ batch_base, uint2 *__restrict__ batch_tab,
                                       uint32_t *sh, const PlanLds *pl = nullptr, uint32_t first_arg = 0, uint32_t count_arg = 0) {
    // (pl: the level's segments are in LDS -- plan_children_body has just made them; first / count come with them)
    const bool staged = pl != nullptr && count_arg <= kPlanNext;
    const uint32_t first = pl ? first_arg : lv->first, count = pl ? count_arg : lv->count;
    uint32_t n_split = 0, n_fast = 0, n_tiles = 0, n_bat = 0;
    for (uint32_t b = 0; b < count; b += 1024) {
        const uint32_t li = b + threadIdx.x;
        const bool in = li < count;
        const uint32_t node = first + li;
        const uint32_t len = in ? (staged ? pl->nlen[li] : na.seg_len[node]) : 0u;
        const uint32_t is_split = (in && can_split && len > 1) ? 1u : 0u;            // src/tsvq.rs:38-44
        const uint32_t is_fast = (in && can_fast && len >= fs_min_rows) ? 1u : 0u;
        const uint32_t nt = is_fast ? (len + kFsTile - 1) / kFsTile : 0u;
        uint32_t tot_s, tot_f, tot_t;
        const uint32_t ps = block_excl_scan(is_split, sh, &tot_s);
        const uint32_t pf = block_excl_scan(is_fast, sh, &tot_f);
        const uint32_t pt = block_excl_scan(nt, sh, &tot_t);
        // batches of 64 tiles (k_fs_prep / k_fs_chain3: the items of a column are kept per batch)
        uint32_t tot_b;
        const uint32_t pbt = block_excl_scan((nt + 63u) / 64u, sh, &tot_b);
        if (in && is_fast && batch_base) {
            batch_base[n_fast + pf] = n_bat + pbt;
            if (staged && n_fast + pf < kPlanFast) pl->fbat[n_fast + pf] = n_bat + pbt;
        }
        if (in) {
            remap[li] = is_split ? n_split + ps : kInactive;
            if (is_split) lvl_split[n_split + ps] = node;
            if (is_fast) {
                fast_nodes[n_fast + pf] = node;
                tile_base[n_fast + pf] = n_tiles + pt;
                n_tiles_of[n_fast + pf] = nt;
                if (staged && n_fast + pf < kPlanFast)
                    pl->fnode[n_fast + pf] = node, pl->flen[n_fast + pf] = len, pl->fstart[n_fast + pf] = pl->nstart[li], sh_tb(pl)[n_fast + pf] = n_tiles + pt;
            } else {
                slow_nodes[(li - (n_fast + pf))] = node;  // nodes before li that are not fast: li - (fast before li)
            }
        }
        n_split += tot_s;
        n_fast += tot_f;
        n_tiles += tot_t;
        n_bat += tot_b;
    }
    // tile table: tile T belongs to the fast node f with tile_base[f] <= T < tile_base[f] + nt[f].  The search runs on a
    // copy of tile_base in LDS when it fits (the usual case: a handful of long nodes): five dependent global loads per tile
    // were half of this kernel's 20 us; with the level's segments staged (k_plan_fused) nothing here is read from memory
    const bool all_lds = staged && n_fast <= kPlanFast;
    const bool in_lds = n_fast <= 1024u;
    if (all_lds) {
        __syncthreads();
    } else {
        __threadfence();
        __syncthreads();
        if (in_lds) {
            if (threadIdx.x < n_fast) sh[threadIdx.x] = tile_base[threadIdx.x];
            __syncthreads();
        }
    }
    const uint32_t *tb = all_lds ? sh_tb(pl) : sh;
    for (uint32_t T = threadIdx.x; T < n_tiles; T += 1024) {
        uint32_t lo = 0, hi = n_fast;  // last f with tile_base[f] <= T
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((in_lds ? tb[mid] : tile_base[mid]) <= T) lo = mid; else hi = mid;
        }
        const uint32_t node = all_lds ? pl->fnode[lo] : fast_nodes[lo], t = T - (in_lds ? tb[lo] : tile_base[lo]);
        const uint32_t len = all_lds ? pl->flen[lo] : na.seg_len[node];
        FsTile tl;
        tl.node = node;
        tl.t = t;
        tl.start = (all_lds ? pl->fstart[lo] : na.seg_start[node]) + t * kFsTile;
        tl.rows = min(kFsTile, len - t * kFsTile);
        tiles[T] = tl;
    }
    if (batch_tab) {  // batch B = batch B - batch_base[f] of the fast node f with batch_base[f] <= B (same search)
        const bool b_lds = n_fast <= 1024u;
        __syncthreads();
        if (b_lds && !all_lds) {
            if (threadIdx.x < n_fast) sh[threadIdx.x] = batch_base[threadIdx.x];
            __syncthreads();
        }
        const uint32_t *bb = all_lds ? pl->fbat : sh;
        for (uint32_t B = threadIdx.x; B < n_bat; B += 1024) {
            uint32_t lo = 0, hi = n_fast;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((b_lds ? bb[mid] : batch_base[mid]) <= B) lo = mid; else hi = mid;
            }
            batch_tab[B] = make_uint2(lo, B - (b_lds ? bb[lo] : batch_base[lo]));
        }
    }
    if (threadIdx.x == 0) {
        lv->pad = n_bat;
        lv->n_split = n_split;
        lv->n_fast = n_fast;
        lv->n_slow = count - n_fast;
        lv->n_tiles = n_tiles;
    }
}
__global__ __launch_bounds__(1024) void k_plan_level(LevelInfo *__restrict__ lv, int can_split, int can_fast, uint32_t fs_min_rows, NodeArrays na,
                                                     uint32_t *__restrict__ lvl_split, uint32_t *__restrict__ remap,
                                                     uint32_t *__restrict__ fast_nodes, uint32_t *__restrict__ slow_nodes,
                                                     uint32_t *__restrict__ tile_base, uint32_t *__restrict__ n_tiles_of,
                                                     FsTile *__restrict__ tiles, uint32_t *__restrict__ batch_base, uint2 *__restrict__ batch_tab) {
    __shared__ uint32_t sh[1024];
    plan_level_body(lv, can_split, can_fast, fs_min_rows, na, lvl_split, remap, fast_nodes, slow_nodes, tile_base, n_tiles_of, tiles, batch_base, batch_tab, sh);
}

// end of a level: children of the split nodes from nleft / nv (src/tsvq.rs:88-108), numbered in order
__device__ inline void plan_children_body(LevelInfo *__restrict__ lv, LevelInfo *__restrict__ lv_next,
                                          const uint32_t *__restrict__ lvl_split, NodeArrays na,
                                          int32_t *__restrict__ node_left, int32_t *__restrict__ node_right,
                                          uint32_t dcap, const uint32_t *__restrict__ Pb,
                                          const uint32_t *__restrict__ bsums, const uint32_t *__restrict__ flags, uint32_t *sh,
                                          const PlanLds *pl = nullptr, uint32_t nb = 0, uint32_t *next_first_out = nullptr,
                                          uint32_t *made_out = nullptr) {
    const bool sb_lds = pl != nullptr && nb <= kPlanSb;  // the scanned block totals are in LDS (scan_sums_body)
    const uint32_t n_split = lv->n_split, next_first = lv->first + lv->count;
    uint32_t made = 0, err = 0;
    for (uint32_t b = 0; b < n_split; b += 1024) {
        const uint32_t j = b + threadIdx.x;
        const bool in = j < n_split;
        uint32_t node = 0, start = 0, len = 0, nl = 0;
        if (in) {
            node = lvl_split[j];
            start = na.seg_start[node];
            len = na.seg_len[node];
            // lefts of the node = scan(end) - scan(start), the exclusive scan completed with the block totals on the fly
            const uint32_t e = start + len - 1;
            nl = (Pb[e] + (sb_lds ? pl->sb[e >> 10] : bsums[e >> 10]) + flags[e]) - (Pb[start] + (sb_lds ? pl->sb[start >> 10] : bsums[start >> 10]));
            na.nleft[node] = nl;
            if (na.nv[node] == 0) err = 1;
        }
        const uint32_t nr = len - nl;
        const uint32_t has_l = (in && nl != 0 && nl < len) ? 1u : 0u, has_r = (in && nr != 0 && nr < len) ? 1u : 0u;
        uint32_t tot;
        const uint32_t pos = block_excl_scan(has_l + has_r, sh, &tot);
        if (in) {
            const uint32_t il = made + pos, ir = il + has_l;
            na.child_local[2 * node] = has_l ? il : kInactive;
            na.child_local[2 * node + 1] = has_r ? ir : kInactive;
            if (has_l && next_first + il < dcap) {
                node_left[node] = (int32_t)(next_first + il);
                na.seg_start[next_first + il] = start;
                na.seg_len[next_first + il] = nl;
                if (pl && il < kPlanNext) pl->nstart[il] = start, pl->nlen[il] = nl;
            }
            if (has_r && next_first + ir < dcap) {
                node_right[node] = (int32_t)(next_first + ir);
                na.seg_start[next_first + ir] = start + nl;
                na.seg_len[next_first + ir] = nr;
                if (pl && ir < kPlanNext) pl->nstart[ir] = start + nl, pl->nlen[ir] = nr;
            }
        }
        made += tot;
    }
    const int any_err = __syncthreads_or((int)err);
    if (threadIdx.x == 0) {
        if (any_err) lv->error = 1;
        lv_next->first = next_first;
        lv_next->count = (next_first + made <= dcap) ? made : 0u;  // cannot happen (dcap bounds the tree); no overrun if it did
        if (next_first + made > dcap) lv->error = 2;
    }
    if (next_first_out) *next_first_out = next_first;
    if (made_out) *made_out = (next_first + made <= dcap) ? made : 0u;
}
// The three single-workgroup steps between two levels in ONE launch (each kernel boundary of the build costs ~5 us, and a
// level had 24 of them): the scan of the partition's block totals and the children of level L's split nodes
// (src/tsvq.rs:88-108), then -- the children's segment lengths now known -- the plan of level L + 1.  k_scatter runs
// behind it (it needs nleft / child_local / the scanned totals, not the plan).
__global__ __launch_bounds__(1024) void k_plan_fused(LevelInfo *__restrict__ lv, LevelInfo *__restrict__ lv_next,
                                                     uint32_t *__restrict__ lvl_split, NodeArrays na,
                                                     int32_t *__restrict__ node_left, int32_t *__restrict__ node_right,
                                                     uint32_t dcap, const uint32_t *__restrict__ Pb, uint32_t *__restrict__ bsums,
                                                     uint32_t nb, const uint32_t *__restrict__ flags, int next_can_split,
                                                     int can_fast, uint32_t fs_min_rows, uint32_t *__restrict__ lvl_split_next,
                                                     uint32_t *__restrict__ remap_next, uint32_t *__restrict__ fast_nodes,
                                                     uint32_t *__restrict__ slow_nodes, uint32_t *__restrict__ tile_base,
                                                     uint32_t *__restrict__ n_tiles_of, FsTile *__restrict__ tiles,
                                                     uint32_t *__restrict__ batch_base, uint2 *__restrict__ batch_tab) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t l_sb[kPlanSb], l_nstart[kPlanNext], l_nlen[kPlanNext], l_fnode[kPlanFast], l_fstart[kPlanFast], l_flen[kPlanFast],
        l_ftile[kPlanFast], l_fbat[kPlanFast];
    PlanLds pl;
    pl.sb = l_sb, pl.nstart = l_nstart, pl.nlen = l_nlen, pl.fnode = l_fnode, pl.fstart = l_fstart, pl.flen = l_flen, pl.ftile = l_ftile, pl.fbat = l_fbat;
    scan_sums_body(bsums, nb, sh, l_sb);
    if (nb > kPlanSb) __threadfence();  // (else the next phase reads the totals from LDS; k_scatter reads them behind this kernel)
    __syncthreads();
    uint32_t next_first = 0, made = 0;
    plan_children_body(lv, lv_next, lvl_split, na, node_left, node_right, dcap, Pb, bsums, flags, sh, &pl, nb, &next_first, &made);
    if (made > kPlanNext) __threadfence();  // (uniform; else the next level's segments are in LDS)
    __syncthreads();
    plan_level_body(lv_next, next_can_split, can_fast, fs_min_rows, na, lvl_split_next, remap_next, fast_nodes, slow_nodes, tile_base,
                    n_tiles_of, tiles, batch_base, batch_tab, sh, &pl, next_first, made);
}

// ---- host driver of the build ------------------------------------------------------------
// Device scratch of a build, kept per host thread between builds (their hipMalloc calls were 2-3 ms of a
// 13 ms build); dropped when it exceeds 6 GiB (of 288: a 1M x 384 build holds 1.3 GiB, and re-allocating it was 3.4 of
// that build's 10 ms) or the device changes.
struct TsvqBuildWs {
    int device = -1;
    DevBuf b_perm[2], b_nodeof[2], b_vals, b_flags, b_scan, b_bsums, b_hist, b_lvl, b_remap, b_lvl2, b_remap2;
    DevBuf b_seg_start, b_seg_len, b_split, b_nv, b_nleft, b_median, b_selp, b_selr, b_child, b_cent, b_var, b_left, b_right;
    DevBuf b_fs_tiles, b_fs_nodes, b_fs_base, b_fs_nt, b_fs_sum, b_fs_pref, b_fs_summ2, b_fs_summ, b_lvl_slow, b_fs_fb, b_fs_side, b_fs_mom, b_lv;
    DevBuf b_fs_smeta, b_fs_tab, b_fs_tmeta;  // k_fs_prep: guess + pass tag per parked slot, the tables, their validity
    DevBuf b_fs_bbase, b_fs_btab, b_fs_items, b_fs_ihdr;  // batches per fast node, batch table, the items per (batch, column)
    DevBuf b_selbin;                                      // the median selection's linear bins per node
    DevBuf *all[47] = {&b_selbin, &b_fs_bbase, &b_fs_btab, &b_fs_items, &b_fs_ihdr, &b_fs_smeta, &b_fs_tab, &b_fs_tmeta, &b_lvl2, &b_remap2, &b_perm[0], &b_perm[1], &b_nodeof[0], &b_nodeof[1], &b_vals, &b_flags, &b_scan, &b_bsums, &b_hist,
                       &b_lvl, &b_remap, &b_seg_start, &b_seg_len, &b_split, &b_nv, &b_nleft, &b_median, &b_selp, &b_selr,
                       &b_child, &b_cent, &b_var, &b_left, &b_right, &b_fs_tiles, &b_fs_nodes, &b_fs_base, &b_fs_nt, &b_fs_sum, &b_fs_pref, &b_fs_summ2,
                       &b_fs_summ, &b_lvl_slow, &b_fs_fb, &b_fs_side, &b_fs_mom, &b_lv};
    // pinned host staging of the node download.  A pageable destination of a few MB is pinned by the runtime for the
    // copy; when that memory is later unmapped (a std::vector or numpy array of 4 MB goes back to the OS) the driver
    // invalidates the mapping by evicting and restoring the process' queues, and the next kernel launched -- the next
    // build's first -- starts 10-30 ms late (the round-1 "depth-12 anomaly", profiles/r2/tsvq_anomaly.txt)
    void *h_stage = nullptr;
    size_t h_bytes = 0;
    int ensure_host(size_t need) {
        if (need <= h_bytes) return VQHIP_OK;
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr;
        h_bytes = 0;
        VQ_HIP(hipHostMalloc(&h_stage, need, hipHostMallocDefault));
        h_bytes = need;
        return VQHIP_OK;
    }
    size_t total() const {
        size_t t = 0;
        for (const DevBuf *b : all) t += b->bytes;
        return t;
    }
    void release() {
        for (DevBuf *b : all) b->release();
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr;
        h_bytes = 0;
    }
    ~TsvqBuildWs() { release(); }
};

static thread_local bool tl_tsvq_conservative = false;  // the repeat of a build whose speculation failed
int tsvq_build_device(const float *X, uint64_t n64, uint32_t d, uint32_t max_depth, uint32_t cap,
                      float *centroids_out, int32_t *left_out, int32_t *right_out, int32_t *n_nodes_out,
                      hipStream_t stream, TsvqPolicyCache *policy_cache) {
    if (n64 >= (1ull << 31)) return fail(VQHIP_ERR_UNSUPPORTED, "TSVQ build supports < 2^31 rows per device");
    const uint32_t n = (uint32_t)n64;
    const uint64_t need_cap = (max_depth < 31) ? std::min<uint64_t>((1ull << (max_depth + 1)) - 1, 2ull * n - 1) : 2ull * n - 1;
    if (cap < need_cap)
        return fail(VQHIP_ERR_INVALID_INPUT, "node capacity %u < required %llu", cap, (unsigned long long)need_cap);
    const uint32_t dcap = (uint32_t)need_cap;
    // levels: the root is level 0; a tree over n rows is at most n - 1 levels deep whatever max_depth says
    const uint32_t n_levels = (uint32_t)std::min<uint64_t>((uint64_t)max_depth, (uint64_t)n - 1) + 1;

    // VQHIP_TSVQ_TIMING=<ms>: host-side timeline of every build slower than <ms> (where did an outlier spend its time?)
    static const char *timing_env = getenv("VQHIP_TSVQ_TIMING");
    struct Marks {
        double limit_ms = -1.0;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        std::vector<std::pair<const char *, double>> v;
        void mark(const char *what) {
            if (limit_ms >= 0.0) v.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        ~Marks() {
            if (limit_ms < 0.0 || v.empty() || v.back().second < limit_ms) return;
            fprintf(stderr, "[vqhip] tsvq build timeline (ms):");
            for (auto &m : v) fprintf(stderr, " %s %.2f", m.first, m.second);
            fprintf(stderr, "\n");
        }
    } marks;
    marks.limit_ms = timing_env ? atof(timing_env) : -1.0;
    static thread_local TsvqBuildWs ws;
    {
        int dev = 0;
        VQ_HIP(hipGetDevice(&dev));
        if (ws.device != dev) ws.release();
        ws.device = dev;
    }
    struct WsTrim {  // keep the scratch for the next build unless it is large
        TsvqBuildWs &w;
        ~WsTrim() {
            if (w.total() > (6ull << 30)) w.release();
        }
    } ws_trim{ws};
    // widest level: min(2^level, n) nodes
    auto level_width = [&](uint32_t L) -> uint32_t { return (L >= 31) ? n : (uint32_t)std::min<uint64_t>(1ull << L, n); };
    const uint32_t wmax = level_width(n_levels - 1);
    static const char *minrows_env = getenv("VQHIP_TSVQ_FAST_MIN_ROWS");  // nodes at least this long take the tile-parallel emulation
    // The plain chain costs one dependent v_add_f32 per row and column chain (~4.9 ns per row, measured, whatever d is
    // while its workgroups fit the chip); the emulation costs about what streaming the level's rows at ~1.7 TB/s does.
    // A node of `len` rows is cheaper through the emulation when len * 4.9 ns > n * d * 4 B / 1.7 TB/s.
    const uint32_t fs_model_rows = (uint32_t)std::min<double>(4.0e9, 4.8e-4 * (double)d * (double)n);
    const uint32_t fs_min_rows = minrows_env ? (uint32_t)std::max(4096, atoi(minrows_env)) : std::max(kFsMinRows, fs_model_rows);
    const uint32_t fast_max = n / fs_min_rows + 1, tiles_max = n / kFsTile + fast_max + 1;
    for (int q = 0; q < 2; ++q) {
        VQ_TRY(ws.b_perm[q].ensure((size_t)n * 4));
        VQ_TRY(ws.b_nodeof[q].ensure((size_t)n * 4));
    }
    VQ_TRY(ws.b_vals.ensure((size_t)n * 4));
    VQ_TRY(ws.b_flags.ensure((size_t)n * 4));
    VQ_TRY(ws.b_scan.ensure((size_t)n * 4));
    const uint32_t nblk = (n + 1023) / 1024;
    VQ_TRY(ws.b_bsums.ensure((size_t)nblk * 4));
    VQ_TRY(ws.b_lvl.ensure((size_t)wmax * 4));
    VQ_TRY(ws.b_remap.ensure((size_t)wmax * 4));
    VQ_TRY(ws.b_lvl2.ensure((size_t)wmax * 4));
    VQ_TRY(ws.b_remap2.ensure((size_t)wmax * 4));
    VQ_TRY(ws.b_lvl_slow.ensure((size_t)wmax * 4));
    // median: three radix rounds of 11 / 11 / 10 bits (2048 bins per node and rank) while the widest level's histograms
    // stay small, else four rounds of 8 bits
    static const char *radix_env = getenv("VQHIP_TSVQ_RADIX8");  // =1: four 8-bit rounds (A/B)
    const bool radix11 = wmax <= 4096 && !(radix_env && radix_env[0] == '1');
    const uint32_t hist_bins = radix11 ? 2048u : 256u;
    VQ_TRY(ws.b_hist.ensure(((size_t)wmax * 2 * hist_bins + 2 * (size_t)wmax) * 4));  // + the candidate counters (k_select_collect)
    VQ_TRY(ws.b_seg_start.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_seg_len.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_split.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_nv.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_nleft.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_median.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_selbin.ensure((size_t)dcap * sizeof(float2)));
    VQ_TRY(ws.b_selp.ensure((size_t)dcap * 8));
    VQ_TRY(ws.b_selr.ensure((size_t)dcap * 8));
    VQ_TRY(ws.b_child.ensure((size_t)dcap * 8));
    VQ_TRY(ws.b_left.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_right.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_cent.ensure((size_t)dcap * d * 4));
    VQ_TRY(ws.b_var.ensure((size_t)dcap * d * 4));
    VQ_TRY(ws.b_lv.ensure((size_t)(n_levels + 1) * sizeof(LevelInfo)));
    VQ_TRY(ws.b_fs_fb.ensure(8 + 64 * 2 * 64 + 4096));  // [0] tile re-additions (diagnostic), [1] side-buffer slots handed out, [2..] VQHIP_TSVQ_DEBUG counters: 8 per (level < 64, pass)
    NodeArrays na;
    na.seg_start = ws.b_seg_start.as<uint32_t>();
    na.seg_len = ws.b_seg_len.as<uint32_t>();
    na.split_dim = ws.b_split.as<uint32_t>();
    na.nv = ws.b_nv.as<uint32_t>();
    na.nleft = ws.b_nleft.as<uint32_t>();
    na.median = ws.b_median.as<float>();
    na.sel_bin = ws.b_selbin.as<float2>();
    na.sel_prefix = ws.b_selp.as<uint32_t>();
    na.sel_rank = ws.b_selr.as<uint32_t>();
    na.child_local = ws.b_child.as<uint32_t>();
    na.centroid = ws.b_cent.as<float>();
    na.var = ws.b_var.as<float>();
    na.fs_seg_stride = 0;
    int32_t *node_left = ws.b_left.as<int32_t>(), *node_right = ws.b_right.as<int32_t>();
    LevelInfo *lv = ws.b_lv.as<LevelInfo>();

    static const char *nopark = getenv("VQHIP_TSVQ_NOPARK");
    static const char *seq_env = getenv("VQHIP_TSVQ_SEQSUM");  // =1: plain chain everywhere (A/B)
    const bool can_fast = (d % 4 == 0) && !(seq_env && seq_env[0] == '1');  // 16-byte row parts
    static const char *samp_env = getenv("VQHIP_TSVQ_SAMPLE");  // rows read for the mean pass's binade guess: 1/N (default 1/4)
    const uint32_t fs_sample = samp_env ? (uint32_t)std::max(1, std::min(16, atoi(samp_env))) : 4u;  // (1/8 until k_fs_chain4: 17 us a level against 30, but twice the segments folded under the wrong binade -- C4 3.90 against 3.84 ms)
    // sampling policy per block of 32 columns (k_fs_policy), behind the diagnostics in b_fs_fb; a forced VQHIP_TSVQ_SAMPLE
    // applies to every column
    const uint32_t n_cblk = (d + kFsCols - 1) / kFsCols;
    const bool adaptive_sampling = !samp_env && can_fast && n_cblk <= 1024;
    const float park_rel = fs_sample > 1 ? 3.0f * 0.6f * sqrtf((float)fs_sample / (float)kFsTile) : 0.0f;  // k_fs_fold: parking margin
    // segment summaries [segment slot][column]: eight slots per tile, a node's segments contiguous
    const size_t seg_slots = (size_t)tiles_max * kFsSegsPerTile;
    na.fs_seg_stride = (uint32_t)seg_slots;  // (a multiple of 8: a tile's eight summaries of a column are one aligned 128-byte line)
    // parked segments per pass (256 bytes each; beyond the cap the re-addition gathers its rows, two dependent loads): half
    // of all segments (zero-mean columns park up to a third at the deeper levels), 256 MB at most
    const uint32_t side_cap = (nopark && nopark[0] == '1') ? 0u : (uint32_t)std::min<uint64_t>(seg_slots * d / 2 + 1, 1ull << 20);
    // VQHIP_TSVQ_CHAIN=1: round 4's chain (re-additions executed on the column's wave) instead of the table form (A/B)
    static const char *chain_env = getenv("VQHIP_TSVQ_CHAIN");
    const bool use_tables = side_cap > 0 && !(chain_env && chain_env[0] == '1');
    const uint32_t batches_max = tiles_max / 64 + fast_max + 1;
    if (can_fast) {
        VQ_TRY(ws.b_fs_tiles.ensure((size_t)tiles_max * sizeof(FsTile)));
        VQ_TRY(ws.b_fs_nodes.ensure((size_t)fast_max * 4));
        VQ_TRY(ws.b_fs_base.ensure((size_t)fast_max * 4));
        VQ_TRY(ws.b_fs_nt.ensure((size_t)fast_max * 4));
        if (n >= fs_min_rows) {
            VQ_TRY(ws.b_fs_sum.ensure((size_t)tiles_max * d * 8));
            VQ_TRY(ws.b_fs_pref.ensure((size_t)tiles_max * d * 4));
            VQ_TRY(ws.b_fs_mom.ensure((size_t)tiles_max * d * sizeof(double2)));
            VQ_TRY(ws.b_fs_summ.ensure(seg_slots * d * sizeof(FsS)));
            VQ_TRY(ws.b_fs_summ2.ensure(seg_slots * d * sizeof(FsS)));
            VQ_TRY(ws.b_fs_side.ensure(std::max<size_t>((size_t)side_cap * kFsSeg * 4, 16)));
            if (use_tables) {
                VQ_TRY(ws.b_fs_tab.ensure(std::max<size_t>((size_t)side_cap * kFsTabN * 4, 16)));
                VQ_TRY(ws.b_fs_smeta.ensure(std::max<size_t>((size_t)side_cap * 2 * sizeof(uint4), 16)));  // two records per slot
                VQ_TRY(ws.b_fs_tmeta.ensure(std::max<size_t>((size_t)side_cap * sizeof(int4), 16)));
                VQ_TRY(ws.b_fs_bbase.ensure((size_t)fast_max * 4));
                VQ_TRY(ws.b_fs_btab.ensure((size_t)batches_max * sizeof(uint2)));
                VQ_TRY(ws.b_fs_items.ensure((size_t)batches_max * d * kFsItemCap * 2 * sizeof(int4)));
                VQ_TRY(ws.b_fs_ihdr.ensure((size_t)batches_max * d * 4));
            }
        }
    } else {
        VQ_TRY(ws.b_fs_tiles.ensure(16));
        VQ_TRY(ws.b_fs_nodes.ensure(16));
        VQ_TRY(ws.b_fs_base.ensure(16));
        VQ_TRY(ws.b_fs_nt.ensure(16));
    }
    const bool have_fast = can_fast && n >= fs_min_rows;

    marks.mark("allocated");
    // initial state (k_build_init), then the sampling policy; nothing comes from the host, so nothing waits for it here
    static_assert(sizeof(LevelInfo) == 32 && offsetof(LevelInfo, count) == 4, "k_build_init writes LevelInfo[0].count as word 1");
    hipLaunchKernelGGL(k_build_init, dim3(std::min<uint32_t>((n + 255) / 256, (uint32_t)num_cus() * 16)), dim3(256), 0, stream,
                       ws.b_perm[0].as<uint32_t>(), ws.b_nodeof[0].as<uint32_t>(), n, ws.b_left.as<int32_t>(), ws.b_right.as<int32_t>(), dcap,
                       ws.b_lv.as<uint32_t>(), (uint32_t)((n_levels + 1) * sizeof(LevelInfo) / 4), na.seg_start, na.seg_len,
                       ws.b_fs_fb.as<uint32_t>(), (uint32_t)((8 + 64 * 2 * 64) / 4), ws.b_hist.as<uint32_t>(), wmax * 2 * hist_bins + 2 * wmax);
    VQ_LAUNCH_CHECK("k_build_init");
    // the sampling policy: from the data set's cache if an earlier build left it there, else the kernel (into the cache's
    // buffer when there is one)
    uint32_t *policy_buf = ws.b_fs_fb.as<uint32_t>() + 2 + 64 * 2 * 16;
    bool policy_cached = false, any_sampled = true, any_exact = true;
    if (adaptive_sampling && n >= fs_min_rows) {
        if (policy_cache) {
            std::lock_guard<std::mutex> lk(policy_cache->mu);
            if (policy_cache->valid && policy_cache->n_cblk == n_cblk) {
                policy_cached = true;
                any_sampled = any_exact = false;
                for (uint32_t v : policy_cache->host) (v ? any_sampled : any_exact) = true;
            } else {
                VQ_TRY(policy_cache->dev.ensure((size_t)n_cblk * 4));
            }
            policy_buf = policy_cache->dev.as<uint32_t>();
        }
        if (!policy_cached) {
            hipLaunchKernelGGL(k_fs_policy, dim3(n_cblk), dim3(1024), 0, stream, X, n, d, policy_buf);
            VQ_LAUNCH_CHECK("k_fs_policy");
        }
    }
    marks.mark("queued");
    marks.mark("setup");
    int cur = 0;
    const uint32_t ncb = (d + kFsCols - 1) / kFsCols;
    uint32_t *const lvl_split_buf[2] = {ws.b_lvl.as<uint32_t>(), ws.b_lvl2.as<uint32_t>()};  // by level parity: k_plan_fused writes
    uint32_t *const remap_buf[2] = {ws.b_remap.as<uint32_t>(), ws.b_remap2.as<uint32_t>()};   // level L + 1's while k_scatter still reads level L's
    uint32_t *slow_nodes = ws.b_lvl_slow.as<uint32_t>();
    const FsTile *tl = ws.b_fs_tiles.as<FsTile>();
    double *ts = ws.b_fs_sum.as<double>();
    float *tp = ws.b_fs_pref.as<float>();
    double2 *mom = ws.b_fs_mom.as<double2>();
    FsS *sm = ws.b_fs_summ.as<FsS>(), *sm2 = ws.b_fs_summ2.as<FsS>();
    const uint32_t *fn = ws.b_fs_nodes.as<uint32_t>(), *fb = ws.b_fs_base.as<uint32_t>(), *fc = ws.b_fs_nt.as<uint32_t>();
    float *side = ws.b_fs_side.as<float>();
    float *tabs = ws.b_fs_tab.as<float>();
    uint4 *smeta = use_tables ? ws.b_fs_smeta.as<uint4>() : nullptr;
    int4 *tmeta = ws.b_fs_tmeta.as<int4>();
    const bool have_prep = use_tables && can_fast && n >= fs_min_rows;
    uint32_t *bbase = have_prep ? ws.b_fs_bbase.as<uint32_t>() : nullptr;
    uint2 *btab = have_prep ? ws.b_fs_btab.as<uint2>() : nullptr;
    int4 *items = ws.b_fs_items.as<int4>();
    uint32_t *ihdr = ws.b_fs_ihdr.as<uint32_t>();
    // tags of this build's passes: never a value an earlier pass (of this or an earlier build over the same scratch) used
    static std::atomic<uint32_t> g_pass_tag{1};
    uint32_t *fbk = ws.b_fs_fb.as<uint32_t>();
    const uint32_t *policy = adaptive_sampling ? policy_buf : nullptr;

    // sequential-order column sums of the level's nodes: long nodes through the tile-parallel exact emulation (k_fs_*),
    // the rest through the plain chain kernel.  Grids are upper bounds; the kernels read the level's counts.
    static const bool fs_debug = getenv("VQHIP_TSVQ_DEBUG") != nullptr;
    auto colsum = [&](int mode, const LevelInfo *lvp, uint32_t ub_nodes, const uint32_t *perm, bool with_fast, bool with_slow) -> int {
        const uint32_t lvl_idx = (uint32_t)(lvp - lv);
        uint32_t *dbg = (fs_debug && lvl_idx < 64) ? fbk + 2 + (lvl_idx * 2 + (uint32_t)mode) * 16 : nullptr;
        const uint32_t ub_fast = (have_fast && with_fast) ? std::min(ub_nodes, fast_max) : 0u;
        // few nodes: 16 columns per workgroup (more chains in flight); many: 32 (fewer, fuller workgroups)
        const uint32_t g16 = (d + 15) / 16, g32 = (d + 31) / 32;
        static const char *narrow_env = getenv("VQHIP_TSVQ_NARROW_WGS");
        const uint64_t narrow_max = narrow_env ? (uint64_t)atoi(narrow_env) : (uint64_t)num_cus();  // measured: 16-column workgroups pay only while they leave CUs idle otherwise
        const bool narrow = (uint64_t)ub_nodes * g16 <= narrow_max;
        if (!with_slow) {
            // (speculation, below: no node of this level is expected on the plain chain)
        } else if (narrow) {
            if (mode == 0) hipLaunchKernelGGL((k_seg_colsum<0, 16>), dim3(ub_nodes, g16), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
            else hipLaunchKernelGGL((k_seg_colsum<1, 16>), dim3(ub_nodes, g16), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
        } else {
            if (mode == 0) hipLaunchKernelGGL((k_seg_colsum<0, 32>), dim3(ub_nodes, g32), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
            else hipLaunchKernelGGL((k_seg_colsum<1, 32>), dim3(ub_nodes, g32), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
        }
        VQ_LAUNCH_CHECK("k_seg_colsum");
        if (ub_fast == 0) return VQHIP_OK;
        const uint32_t ub_tiles = std::min(tiles_max, n / kFsTile + ub_fast);
        // k_fs_fold: persistent single-wave workgroups, two per SIMD (the kernel's register budget)
        const dim3 tgrid(ub_tiles * ncb), xgrid(std::min<uint32_t>(ub_tiles * ncb, (uint32_t)num_cus() * 8));
        const dim3 pgrid(ub_fast, (d + kFsPrefCols - 1) / kFsPrefCols), cgrid(ub_fast, d);
        uint32_t tag = g_pass_tag.fetch_add(1);
        if (tag == 0) tag = g_pass_tag.fetch_add(1);
        // k_fs_prep: the tables (four waves per workgroup striding over the parked slots: the count is the device's) and
        // the items (one workgroup per batch x four columns, over an upper bound of the level's batches)
        const uint32_t n_tab_blocks = std::min<uint32_t>((side_cap + 3) / 4, (uint32_t)num_cus() * 4);
        const uint32_t ub_batches = std::min(batches_max, ub_tiles / 64 + ub_fast);
        const dim3 bgrid(n_tab_blocks + ub_batches * ((d + 3) / 4));
        // Mean pass: columns with an exact guess (k_fs_policy: zero-mean columns, where 10-30 % of the segments leave their
        // binade) through the tables + items + one-pass chain of round 5; columns under a sampled guess (most parked segments
        // hold there, and the guess is off by far more than a table's window) and the variance pass (monotone sums: a
        // handful of crossings per column) keep round 4's chain.
        // (a cached policy tells the host which of the two has columns at all: no empty launches)
        const bool new_any = use_tables && mode == 0 && (policy != nullptr ? any_exact : fs_sample == 1);
        const bool old_any = !(use_tables && mode == 0) || (policy != nullptr ? any_sampled : fs_sample > 1);
        const uint32_t *only_sampled = (new_any && old_any) ? policy : nullptr;
        // k_fs_chain4 (loader + walker, operands in LDS ahead of time) unless VQHIP_TSVQ_CHAIN4=0 (A/B: k_fs_chain)
        static const char *c4_env = getenv("VQHIP_TSVQ_CHAIN4");
        const bool use_chain4 = !(c4_env && c4_env[0] == '0');
        if (mode == 0) {
            // the binade guesses: f64 sums of every 4th group of rows of each tile where k_fs_policy allows (|mean| >= sigma),
            // of every row elsewhere; prefix over the node's tiles
            hipLaunchKernelGGL(k_fs_tile_sums<0>, tgrid, dim3(256), 0, stream, X, d, perm, tl, na, ts, fs_sample, policy, lvp);
            hipLaunchKernelGGL(k_fs_prefix<false>, pgrid, dim3(1024), 0, stream, d, fn, fb, fc, na, ts, mom, tp, lvp, fbk + 1);
            hipLaunchKernelGGL(k_fs_fold<0>, xgrid, dim3(64), 0, stream, X, d, perm, tl, lvp, na, tp, sm, sm2, side, side_cap, fbk + 1, mom, park_rel, policy, new_any ? smeta : (uint4 *)nullptr, tag);
            if (new_any) {
                hipLaunchKernelGGL(k_fs_prep, bgrid, dim3(256), 0, stream, side, smeta, tag, fbk + 1, side_cap, tabs, tmeta, n_tab_blocks, d, fn, fb, btab, na, sm, sm2, items, ihdr, lvp, policy, 0);
                if (dbg) hipLaunchKernelGGL((k_fs_chain3<0, true>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, bbase, na, sm, sm2, side, tabs, tmeta, items, ihdr, fbk, lvp, dbg, policy);
                else hipLaunchKernelGGL((k_fs_chain3<0, false>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, bbase, na, sm, sm2, side, tabs, tmeta, items, ihdr, fbk, lvp, dbg, policy);
            }
            if (old_any) {
                // (VQHIP_TSVQ_DEBUG counts one form per pass: the new one where both run)
                uint32_t *const dbg0 = (dbg && !new_any) ? dbg : nullptr;
                if (!use_chain4) {
                    if (dbg0) hipLaunchKernelGGL((k_fs_chain<0, true>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg0, only_sampled);
                    else hipLaunchKernelGGL((k_fs_chain<0, false>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, (uint32_t *)nullptr, only_sampled);
                } else {
                    if (dbg0) hipLaunchKernelGGL((k_fs_chain4<0, true>), cgrid, dim3(128), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg0, only_sampled);
                    else hipLaunchKernelGGL((k_fs_chain4<0, false>), cgrid, dim3(128), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, (uint32_t *)nullptr, only_sampled);
                }
            }
        } else {
            // the guess comes from the sums the mean pass of the same level left behind (same tile table: every node
            // long enough for the emulation has more than one row, so it is a split node whenever the level splits)
            hipLaunchKernelGGL(k_fs_prefix<true>, pgrid, dim3(1024), 0, stream, d, fn, fb, fc, na, ts, mom, tp, lvp, fbk + 1);
            hipLaunchKernelGGL(k_fs_fold<1>, xgrid, dim3(64), 0, stream, X, d, perm, tl, lvp, na, tp, sm, sm2, side, side_cap, fbk + 1, (double2 *)nullptr, 0.0f, (const uint32_t *)nullptr, (uint4 *)nullptr, 0u);
            if (!use_chain4) {
                if (dbg) hipLaunchKernelGGL((k_fs_chain<1, true>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg, (const uint32_t *)nullptr);
                else hipLaunchKernelGGL((k_fs_chain<1, false>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg, (const uint32_t *)nullptr);
            } else {
                if (dbg) hipLaunchKernelGGL((k_fs_chain4<1, true>), cgrid, dim3(128), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg, (const uint32_t *)nullptr);
                else hipLaunchKernelGGL((k_fs_chain4<1, false>), cgrid, dim3(128), 0, stream, X, d, perm, fn, fb, na, sm, sm2, side, fbk, lvp, dbg, (const uint32_t *)nullptr);
            }
        }
        VQ_LAUNCH_CHECK("k_fs_*");
        if (getenv("VQHIP_TSVQ_CHECK")) {
            if (mode == 0) hipLaunchKernelGGL(k_fs_check<0>, cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, lvp);
            else hipLaunchKernelGGL(k_fs_check<1>, cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, sm2, lvp);
            VQ_HIP(hipStreamSynchronize(stream));
        }
        return VQHIP_OK;
    };

    uint32_t levels_run = 0;
    // Long nodes only get shorter down the tree (a child has fewer rows than its parent), so once a level has none the
    // seven k_fs_* launches per level can stop -- each empty launch is ~5 us, 130 us of a depth-8 build of 1M rows.  Which
    // level that is the host learns from ONE small read-back (a synchronisation: ~20 us of idle device) at the first
    // level where evenly split nodes would be too short, repeated only while long nodes are still found.
    bool fast_possible = have_fast;
    uint32_t ask_from = 0;
    while (ask_from < 31 && (uint64_t)(n >> ask_from) >= fs_min_rows) ++ask_from;
    static const char *noask_env = getenv("VQHIP_TSVQ_NO_LEVEL_ASK");  // =1: always launch both halves (A/B)
    // Speculation: a median split halves a node (up to ties on the split dimension), so above level `ask_from` every node is
    // long and from it on none is -- the build launches ONLY the kernels of the path each level is expected to take (an empty
    // launch costs ~5 us: ten k_seg_colsum launches over no nodes at C4) and does not ask.  The level table read back at the
    // end says whether a level held a node of the other kind (heavy ties: uneven children); then the build is repeated
    // with both paths launched everywhere, and a data set remembers that (policy cache).
    static const char *spec_env = getenv("VQHIP_TSVQ_SPECULATE");  // =0: never
    bool cache_says_no = false;
    if (policy_cache) {
        std::lock_guard<std::mutex> lk(policy_cache->mu);
        cache_says_no = policy_cache->no_speculation;
    }
    const bool speculate = have_fast && !tl_tsvq_conservative && !cache_says_no && !(spec_env && spec_env[0] == '0') &&
                           !(noask_env && noask_env[0] == '1') && !getenv("VQHIP_TSVQ_DEBUG");
    hipLaunchKernelGGL(k_plan_level, dim3(1), dim3(1024), 0, stream, &lv[0], max_depth > 0 ? 1 : 0, can_fast ? 1 : 0, fs_min_rows, na, lvl_split_buf[0],
                       remap_buf[0], ws.b_fs_nodes.as<uint32_t>(), slow_nodes, ws.b_fs_base.as<uint32_t>(), ws.b_fs_nt.as<uint32_t>(),
                       ws.b_fs_tiles.as<FsTile>(), bbase, btab);
    VQ_LAUNCH_CHECK("k_plan_level");
    for (uint32_t L = 0; L < n_levels; ++L) {
        uint32_t ub_nodes = level_width(L);
        const bool ask = !speculate && fast_possible && L >= ask_from && !(noask_env && noask_env[0] == '1');
        const bool lvl_fast = speculate ? L < ask_from : fast_possible, lvl_slow = speculate ? L >= ask_from : true;
        if (ub_nodes > 1024 || ask) {
            // wide levels: read the level's node count (one small copy + synchronisation) instead of launching the
            // per-node grids over 2^L mostly absent nodes; also ends the loop when the tree has stopped growing
            LevelInfo h;
            VQ_HIP(hipMemcpyAsync(&h, &lv[L], sizeof(LevelInfo), hipMemcpyDeviceToHost, stream));
            VQ_HIP(hipStreamSynchronize(stream));
            marks.mark("level-sync");
            if (h.count == 0) break;
            ub_nodes = std::min(ub_nodes, h.count);
            if (h.n_fast == 0) fast_possible = false;
        }
        ++levels_run;
        const LevelInfo *lvp = &lv[L];
        const bool splits = L < max_depth;  // depth left at this level (src/tsvq.rs:38)
        uint32_t *lvl_split = lvl_split_buf[L & 1], *remap = remap_buf[L & 1];
        uint32_t *perm = ws.b_perm[cur].as<uint32_t>(), *node_of = ws.b_nodeof[cur].as<uint32_t>();
        // means of every node of the level (tsvq.rs:36)
        VQ_TRY(colsum(0, lvp, ub_nodes, perm, lvl_fast, lvl_slow));
        if (!splits) break;
        // variances + split dimension (tsvq.rs:46-66)
        VQ_TRY(colsum(1, lvp, ub_nodes, perm, lvl_fast, lvl_slow));
        hipLaunchKernelGGL(k_pick_split, dim3((ub_nodes + 3) / 4), dim3(256), 0, stream, lvl_split, lvp, d, na);
        VQ_LAUNCH_CHECK("k_pick_split");
        // median (tsvq.rs:68-81): radix select (three rounds of 11 / 11 / 10 bits) of the two middle order statistics; the first round gathers the
        // split dimension's values
        {
            const uint32_t hblocks = std::min<uint32_t>((n + 2047) / 2048, (uint32_t)num_cus() * 4);
            const uint32_t hchunk = (n + hblocks - 1) / hblocks;
            auto round = [&](auto first_c, auto bits_c, uint32_t shift, uint32_t width) {
                constexpr bool kFirst = decltype(first_c)::value;
                constexpr int kBits = decltype(bits_c)::value;
                hipLaunchKernelGGL((k_select_hist<kFirst, kBits>), dim3(hblocks), dim3(256), 0, stream, X, d, perm, ws.b_vals.as<float>(), n, hchunk,
                                   node_of, remap, lvl_split, na, shift, ws.b_hist.as<uint32_t>(), width);
                hipLaunchKernelGGL((k_select_pick<kFirst, kBits>), dim3(ub_nodes * 2), dim3(64), 0, stream, lvl_split, lvp, na, shift,
                                   ws.b_hist.as<uint32_t>());
            };
            using T = std::true_type;
            using F = std::false_type;
            if (radix11) {
                static const char *rounds_env = getenv("VQHIP_TSVQ_RADIX_ROUNDS");  // =1: three histogram rounds on the keys' bits instead of linear bins + candidate lists (A/B)
                if (rounds_env && rounds_env[0] == '1') {
                    round(T{}, std::integral_constant<int, 11>{}, 21u, 11u);
                    round(F{}, std::integral_constant<int, 11>{}, 10u, 11u);
                    round(F{}, std::integral_constant<int, 11>{}, 0u, 10u);  // (the low 10 bits: bins 1024.. stay empty)
                } else {
                    // one round over k_pick_split's linear bins, then the chosen bin's keys collected per (node, rank) -- b_scan /
                    // b_flags are free until k_flags_scan -- and the rank picked among them
                    hipLaunchKernelGGL((k_select_hist<true, 11, true>), dim3(hblocks), dim3(256), 0, stream, X, d, perm, ws.b_vals.as<float>(), n, hchunk,
                                       node_of, remap, lvl_split, na, 21u, ws.b_hist.as<uint32_t>(), 11u);
                    hipLaunchKernelGGL((k_select_pick<true, 11>), dim3(ub_nodes * 2), dim3(64), 0, stream, lvl_split, lvp, na, 21u, ws.b_hist.as<uint32_t>());
                    uint32_t *cnt = ws.b_hist.as<uint32_t>() + (size_t)wmax * 2 * hist_bins;
                    hipLaunchKernelGGL(k_select_collect, dim3(hblocks), dim3(256), 0, stream, ws.b_vals.as<float>(), n, hchunk, node_of, remap, lvl_split, na,
                                       ws.b_scan.as<uint32_t>(), ws.b_flags.as<uint32_t>(), cnt);
                    hipLaunchKernelGGL(k_select_final, dim3(ub_nodes * 2), dim3(1024), 0, stream, lvl_split, lvp, na, ws.b_scan.as<uint32_t>(),
                                       ws.b_flags.as<uint32_t>(), cnt);
                }
            } else {
                round(T{}, std::integral_constant<int, 8>{}, 24u, 8u);
                round(F{}, std::integral_constant<int, 8>{}, 16u, 8u);
                round(F{}, std::integral_constant<int, 8>{}, 8u, 8u);
                round(F{}, std::integral_constant<int, 8>{}, 0u, 8u);
            }
            VQ_LAUNCH_CHECK("k_select_*");
        }
        // partition (tsvq.rs:84-85): flags + in-block scan; block totals, children (tsvq.rs:88-108) and the NEXT level's
        // plan in one launch; stable scatter
        hipLaunchKernelGGL(k_flags_scan, dim3(nblk), dim3(256), 0, stream, ws.b_vals.as<float>(), n, node_of, remap, lvl_split, na,
                           ws.b_flags.as<uint32_t>(), ws.b_scan.as<uint32_t>(), ws.b_bsums.as<uint32_t>());
        VQ_LAUNCH_CHECK("k_flags_scan");
        hipLaunchKernelGGL(k_plan_fused, dim3(1), dim3(1024), 0, stream, &lv[L], &lv[L + 1], lvl_split, na, node_left, node_right, dcap,
                           ws.b_scan.as<uint32_t>(), ws.b_bsums.as<uint32_t>(), nblk, ws.b_flags.as<uint32_t>(), (L + 1 < max_depth) ? 1 : 0,
                           can_fast ? 1 : 0, fs_min_rows, lvl_split_buf[(L + 1) & 1], remap_buf[(L + 1) & 1], ws.b_fs_nodes.as<uint32_t>(),
                           slow_nodes, ws.b_fs_base.as<uint32_t>(), ws.b_fs_nt.as<uint32_t>(), ws.b_fs_tiles.as<FsTile>(), bbase, btab);
        VQ_LAUNCH_CHECK("k_plan_fused");
        hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, n, perm, node_of, remap, lvl_split, na,
                           ws.b_scan.as<uint32_t>(), ws.b_bsums.as<uint32_t>(), ws.b_flags.as<uint32_t>(), ws.b_perm[cur ^ 1].as<uint32_t>(),
                           ws.b_nodeof[cur ^ 1].as<uint32_t>());
        VQ_LAUNCH_CHECK("k_scatter");
        cur ^= 1;
    }

    // level table -> host: node count, error flags; then the nodes
    std::vector<LevelInfo> hlv(n_levels + 1);
    marks.mark("launched");
    VQ_HIP(hipMemcpyAsync(hlv.data(), lv, (size_t)(n_levels + 1) * sizeof(LevelInfo), hipMemcpyDeviceToHost, stream));
    VQ_HIP(hipStreamSynchronize(stream));
    marks.mark("kernels-done");
    if (speculate) {
        bool wrong = false;
        for (uint32_t L = 0; L < n_levels && hlv[L].count; ++L) wrong = wrong || (L < ask_from ? hlv[L].n_slow != 0 : hlv[L].n_fast != 0);
        if (wrong) {  // a level held nodes of the kind that was not launched: their sums are missing -- once more, both paths
            if (policy_cache) {
                std::lock_guard<std::mutex> lk(policy_cache->mu);
                policy_cache->no_speculation = true;
            }
            if (getenv("VQHIP_TSVQ_VERBOSE")) fprintf(stderr, "[vqhip] tsvq build: a level mixed long and short nodes; repeated with both column-sum paths\n");
            tl_tsvq_conservative = true;
            const int rc = tsvq_build_device(X, n64, d, max_depth, cap, centroids_out, left_out, right_out, n_nodes_out, stream, policy_cache);
            tl_tsvq_conservative = false;
            return rc;
        }
    }
    uint32_t total = 0;
    for (uint32_t L = 0; L < n_levels; ++L) {
        if (hlv[L].error == 1)
            return fail(VQHIP_ERR_INVALID_INPUT,
                        "TSVQ: every value on a split dimension is NaN (the reference panics here, src/tsvq.rs:77-78)");
        if (hlv[L].error) return fail(VQHIP_ERR_FAILURE, "TSVQ node count exceeded its bound");
        if (hlv[L].count == 0) break;
        total = hlv[L].first + hlv[L].count;
    }
    (void)levels_run;
    if (total == 0 || total > dcap) return fail(VQHIP_ERR_FAILURE, "TSVQ build produced %u nodes (bound %u)", total, dcap);
    if (getenv("VQHIP_TSVQ_DEBUG")) {
        uint32_t fbn[2 + 64 * 2 * 16];
        VQ_HIP(hipMemcpyAsync(fbn, fbk, sizeof(fbn), hipMemcpyDeviceToHost, stream));
        VQ_HIP(hipStreamSynchronize(stream));
        fprintf(stderr, "[vqhip] tsvq build: %u re-added 64-row segments in the exact column sums\n", fbn[0]);
        if (policy) {
            uint32_t pol[1024];
            VQ_HIP(hipMemcpy(pol, policy, (size_t)n_cblk * 4, hipMemcpyDeviceToHost));
            uint32_t on = 0;
            for (uint32_t q = 0; q < n_cblk; ++q) on += pol[q] ? 1u : 0u;
            fprintf(stderr, "[vqhip]   sampled binade guess (1/%u rows) allowed for %u of %u column blocks\n", fs_sample, on, n_cblk);
        }
        for (uint32_t q = 0; q < 128; ++q) {
            const uint32_t *c8 = fbn + 2 + q * 16;
            if (c8[0] && use_tables && !(q & 1) && c8[7] + c8[3] + c8[14] != 0 && c8[12] == 0) {
                fprintf(stderr, "[vqhip]   level %u %s: %u chains, %u segments parked: %u held by their own summary, %u looked up in their tables, %u outside the "
                                "table's window or without one (%.2f %% of the parked: re-added from the parked addends); %u segments re-added in all (most in one chain %u), "
                                "%u runs walked segment by segment\n",
                        q / 2, (q & 1) ? "variance" : "mean", c8[0], c8[14], c8[6], c8[3], c8[4], 100.0 * c8[4] / (c8[14] ? c8[14] : 1), c8[1], c8[2], c8[5]);
                fprintf(stderr, "[vqhip]     core cycles per chain (average; longest %.0f k): %.0f k, of which %u items off the fast path %.0f k (table misses "
                                "%.0f k, %.0f each), staging chunks %.0f k\n",
                        c8[13] * 64.0 / 1e3, c8[8] * 64.0 / c8[0] / 1e3, c8[7] / c8[0], c8[9] * 64.0 / c8[0] / 1e3, c8[10] * 64.0 / c8[0] / 1e3,
                        c8[10] * 64.0 / (c8[4] ? c8[4] : 1), c8[11] * 64.0 / c8[0] / 1e3);
                continue;
            }
            if (c8[0])
                fprintf(stderr, "[vqhip]   level %u %s: %u chains, %u segments re-added (most in one chain %u): unusable summary %u, "
                                "other binade than guessed %u, prefix leaves the binade %u, sum not normal %u; %u gathered (not parked)\n",
                        q / 2, (q & 1) ? "variance" : "mean", c8[0], c8[1], c8[2], c8[3], c8[4], c8[5], c8[6], c8[7]);
            if (c8[0] && c8[12])
                fprintf(stderr, "[vqhip]     core cycles per chain (average; longest %.0f k): %.0f k, of which in %u failing lanes %.0f k "
                                "(%.0f per lane; wave-wide scan pass / summary fetch %.0f; re-additions %.0f per segment); %u segments parked\n",
                        c8[13] * 64.0 / 1e3, c8[8] * 64.0 / c8[0] / 1e3, c8[12] / c8[0], c8[9] * 64.0 / c8[0] / 1e3, c8[9] * 64.0 / c8[12],
                        c8[11] * 64.0 / c8[12], c8[10] * 64.0 / (c8[1] ? c8[1] : 1), c8[14]);
        }
    }
    // nodes -> host, then BFS -> pre-order (the oracle's numbering)
    const size_t cent_bytes = (size_t)total * d * 4, idx_bytes = (((size_t)total * 4) + 15) & ~(size_t)15;
    VQ_TRY(ws.ensure_host(cent_bytes + 2 * idx_bytes));
    const float *cent = static_cast<const float *>(ws.h_stage);
    const int32_t *hl = reinterpret_cast<const int32_t *>(static_cast<const char *>(ws.h_stage) + cent_bytes);
    const int32_t *hr = reinterpret_cast<const int32_t *>(static_cast<const char *>(ws.h_stage) + cent_bytes + idx_bytes);
    VQ_HIP(hipMemcpyAsync(ws.h_stage, na.centroid, cent_bytes, hipMemcpyDeviceToHost, stream));
    VQ_HIP(hipMemcpyAsync(const_cast<int32_t *>(hl), node_left, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
    VQ_HIP(hipMemcpyAsync(const_cast<int32_t *>(hr), node_right, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
    VQ_HIP(hipStreamSynchronize(stream));
    marks.mark("downloaded");
    std::vector<int32_t> order;  // pre-order list of BFS ids
    order.reserve(total);
    std::vector<int32_t> stack = {0};
    while (!stack.empty()) {
        const int32_t id = stack.back();
        stack.pop_back();
        order.push_back(id);
        if (hr[id] >= 0) stack.push_back(hr[id]);
        if (hl[id] >= 0) stack.push_back(hl[id]);
    }
    std::vector<int32_t> newid(total, -1);
    for (uint32_t q = 0; q < order.size(); ++q) newid[order[q]] = (int32_t)q;
    for (uint32_t q = 0; q < order.size(); ++q) {
        const int32_t id = order[q];
        memcpy(centroids_out + (size_t)q * d, cent + (size_t)id * d, (size_t)d * 4);
        left_out[q] = hl[id] >= 0 ? newid[hl[id]] : -1;
        right_out[q] = hr[id] >= 0 ? newid[hr[id]] : -1;
    }
    *n_nodes_out = (int32_t)order.size();
    if (policy_cache && !policy_cached && adaptive_sampling && n >= fs_min_rows) {  // (the stream is idle: the nodes have been downloaded)
        std::vector<uint32_t> h(n_cblk);
        VQ_HIP(hipMemcpy(h.data(), policy_buf, (size_t)n_cblk * 4, hipMemcpyDeviceToHost));
        std::lock_guard<std::mutex> lk(policy_cache->mu);
        policy_cache->host = h;
        policy_cache->n_cblk = n_cblk;
        policy_cache->valid = true;
    }
    return VQHIP_OK;
}

int launch_tsvq_node_norms(const float *centroids, uint32_t n_nodes, uint32_t d, float *cnorm, hipStream_t stream) {
    hipLaunchKernelGGL(k_tsvq_node_norms, dim3((n_nodes + 63) / 64), dim3(64), 0, stream, centroids, n_nodes, d, cnorm);
    VQ_LAUNCH_CHECK("k_tsvq_node_norms");
    return VQHIP_OK;
}

template <int METRIC, int RB>
static int launch_descend_lds(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                              const int32_t *left, const int32_t *right, int32_t *leaf, hipStream_t stream) {
    static PerDeviceOnce attr_set;
    if (attr_set.needed()) {
        VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tsvq_descend_lds<METRIC, RB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    hipLaunchKernelGGL((k_tsvq_descend_lds<METRIC, RB>), dim3((uint32_t)((n + RB - 1) / RB)), dim3(RB),
                       (size_t)d * RB * 4, stream, X, n, d, centroids, cnorm, left, right, leaf);
    VQ_LAUNCH_CHECK("k_tsvq_descend_lds");
    return VQHIP_OK;
}

template <int METRIC>
static int dispatch_descend(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                            const int32_t *left, const int32_t *right, int32_t *leaf, hipStream_t stream, bool *done) {
    const size_t budget = 150 * 1024;
    *done = true;
    const bool x_aligned = (reinterpret_cast<uintptr_t>(X) & 15) == 0;
#define VQ_DESCEND_REG(DV)                                                                                          \
    if (d == DV && x_aligned) {                                                                                     \
        hipLaunchKernelGGL((k_tsvq_descend_reg<METRIC, DV>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, \
                           X, n, centroids, cnorm, left, right, leaf);                                              \
        VQ_LAUNCH_CHECK("k_tsvq_descend_reg");                                                                      \
        return VQHIP_OK;                                                                                            \
    }
    VQ_DESCEND_REG(32) VQ_DESCEND_REG(64) VQ_DESCEND_REG(96) VQ_DESCEND_REG(128)
#undef VQ_DESCEND_REG
    if ((size_t)d * 256 * 4 <= budget) return launch_descend_lds<METRIC, 256>(X, n, d, centroids, cnorm, left, right, leaf, stream);
    if ((size_t)d * 128 * 4 <= budget) return launch_descend_lds<METRIC, 128>(X, n, d, centroids, cnorm, left, right, leaf, stream);
    if ((size_t)d * 64 * 4 <= budget) return launch_descend_lds<METRIC, 64>(X, n, d, centroids, cnorm, left, right, leaf, stream);
    // long vectors (d = 768, 1536, ...): fewer rows per workgroup rather than the per-row global-memory walk, which
    // re-reads its row uncoalesced at every level (measured 50 GB/s at d = 768)
    if ((size_t)d * 32 * 4 <= budget) return launch_descend_lds<METRIC, 32>(X, n, d, centroids, cnorm, left, right, leaf, stream);
    if ((size_t)d * 16 * 4 <= budget) return launch_descend_lds<METRIC, 16>(X, n, d, centroids, cnorm, left, right, leaf, stream);
    *done = false;
    return VQHIP_OK;
}

int launch_tsvq_encode(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                       const int32_t *left, const int32_t *right, int metric, int32_t *leaf, hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    bool done = false;
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(centroids) & 15) == 0);
    if (vec) {
        switch (metric) {
        case VQHIP_SQUARED_EUCLIDEAN: VQ_TRY(dispatch_descend<VQHIP_SQUARED_EUCLIDEAN>(X, n, d, centroids, cnorm, left, right, leaf, stream, &done)); break;
        case VQHIP_EUCLIDEAN: VQ_TRY(dispatch_descend<VQHIP_EUCLIDEAN>(X, n, d, centroids, cnorm, left, right, leaf, stream, &done)); break;
        case VQHIP_MANHATTAN: VQ_TRY(dispatch_descend<VQHIP_MANHATTAN>(X, n, d, centroids, cnorm, left, right, leaf, stream, &done)); break;
        case VQHIP_COSINE: VQ_TRY(dispatch_descend<VQHIP_COSINE>(X, n, d, centroids, cnorm, left, right, leaf, stream, &done)); break;
        case VQHIP_COSINE_UNCLAMPED: VQ_TRY(dispatch_descend<VQHIP_COSINE_UNCLAMPED>(X, n, d, centroids, cnorm, left, right, leaf, stream, &done)); break;
        default: return fail(VQHIP_ERR_INVALID_INPUT, "unknown metric %d", metric);
        }
    }
    if (!done) {
        hipLaunchKernelGGL(k_tsvq_descend, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, X, n, d, centroids,
                           left, right, metric, leaf);
        VQ_LAUNCH_CHECK("k_tsvq_descend");
    }
    return VQHIP_OK;
}

int launch_tsvq_gather_f16(const float *centroids, uint32_t d, const int32_t *leaf, uint64_t n, uint16_t *f16_out,
                           hipStream_t stream) {
    if (n == 0) return VQHIP_OK;
    uint64_t blocks = (n * d + 255) / 256;
    if (blocks > (uint64_t)num_cus() * 8) blocks = (uint64_t)num_cus() * 8;
    hipLaunchKernelGGL(k_tsvq_gather_f16, dim3((uint32_t)blocks), dim3(256), 0, stream, centroids, d, leaf, n, f16_out);
    VQ_LAUNCH_CHECK("k_tsvq_gather_f16");
    return VQHIP_OK;
}

}  // namespace vqhip
