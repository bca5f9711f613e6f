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
#include <chrono>
#include <utility>
#include <vector>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vqhip {
namespace {

constexpr uint32_t kInactive = 0xFFFFFFFFu;
constexpr uint32_t kCsBlock = 20;  // rows the consumer of the column-sum kernel stages per register block (5 x ds_read_b128)
constexpr uint32_t kCsAhead = 6;   // tiles a loader wave keeps in flight in registers

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
    uint32_t pad;
};

struct NodeArrays {
    uint32_t *seg_start, *seg_len, *split_dim, *nv, *nleft;
    float *median;
    uint32_t *sel_prefix, *sel_rank;  // [cap][2]
    uint32_t *child_local;            // [cap][2] level-local index of children in the NEXT level
    float *centroid;                  // [cap][d]
    float *var;                       // [cap][d]
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
__global__ __launch_bounds__(1024) void k_seg_colsum(const float *__restrict__ X, uint32_t d,
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
    }
}

// vals[i] = X[perm[i]][split_dim(node of i)]; nv[node] ends as the non-NaN count
__global__ __launch_bounds__(256) void k_gather_vals(const float *__restrict__ X, uint32_t d, uint32_t n,
                                                     const uint32_t *__restrict__ perm,
                                                     const uint32_t *__restrict__ node_of,
                                                     const uint32_t *__restrict__ remap,
                                                     const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                     float *__restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t li = node_of[i];
    if (li != kInactive) li = remap[li];  // level-local node index -> index among the level's split nodes
    if (li == kInactive) return;
    const uint32_t node = lvl_node[li];
    const float x = X[(size_t)perm[i] * d + na.split_dim[node]];
    vals[i] = x;
    if (x != x) atomicSub(&na.nv[node], 1u);  // nv starts at seg_len (k_pick_split); NaNs are rare
}

// ranks of the two order statistics the median needs (tsvq.rs:77-81)
__global__ void k_select_init(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv, NodeArrays na) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= lv->n_split) return;
    const uint32_t node = lvl_node[li];
    const uint32_t nv = na.nv[node];
    if (nv == 0) {
        na.sel_rank[2 * node] = na.sel_rank[2 * node + 1] = 0;
        return;
    }
    const uint32_t h = nv / 2;
    na.sel_rank[2 * node + 0] = (nv % 2 == 0) ? h - 1 : h;
    na.sel_rank[2 * node + 1] = h;
}

// one radix-select round: histogram of the byte at `shift` among keys matching the prefix.
// Positions are grouped by node, so a workgroup's contiguous chunk touches very few nodes:
// histograms are privatised in LDS (8 slots keyed by the level-local node index, claimed with
// a CAS) and flushed once; the rare slot collision falls back to a global atomic.  Unprivatised,
// the root level is 1M atomics on 512 words (0.64 ms per round).
constexpr uint32_t kHistSlots = 8;
__global__ __launch_bounds__(256) void k_select_hist(const float *__restrict__ vals, uint32_t n,
                                                     uint32_t chunk,
                                                     const uint32_t *__restrict__ node_of,
                                                     const uint32_t *__restrict__ remap,
                                                     const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                     uint32_t shift, uint32_t *__restrict__ hist) {
    __shared__ uint32_t tags[kHistSlots];
    __shared__ uint32_t lh[kHistSlots][2][256];
    for (uint32_t e = threadIdx.x; e < kHistSlots * 512; e += 256) (&lh[0][0][0])[e] = 0u;
    if (threadIdx.x < kHistSlots) tags[threadIdx.x] = kInactive;
    __syncthreads();
    const uint32_t hi_mask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
    const uint32_t i0 = blockIdx.x * chunk;
    const uint32_t i1 = min(n, i0 + chunk);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256) {
        uint32_t li = node_of[i];
        if (li != kInactive) li = remap[li];
        if (li == kInactive) continue;
        const float x = vals[i];
        if (x != x) continue;
        const uint32_t node = lvl_node[li];
        const uint32_t key = order_key(x);
        const uint32_t slot = li & (kHistSlots - 1);
        uint32_t owner = tags[slot];
        if (owner == kInactive) {
            const uint32_t old = atomicCAS(&tags[slot], kInactive, li);
            owner = (old == kInactive) ? li : old;
        }
#pragma unroll
        for (uint32_t sel = 0; sel < 2; ++sel)
            if ((key & hi_mask) == na.sel_prefix[2 * node + sel]) {
                const uint32_t bin = (key >> shift) & 255u;
                if (owner == li) atomicAdd(&lh[slot][sel][bin], 1u);
                else atomicAdd(&hist[((size_t)li * 2 + sel) * 256 + bin], 1u);
            }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < kHistSlots * 512; e += 256) {
        const uint32_t slot = e >> 9, rest = e & 511;
        const uint32_t c = (&lh[0][0][0])[e];
        const uint32_t li = tags[slot];
        if (c != 0 && li != kInactive) atomicAdd(&hist[(size_t)li * 512 + rest], c);
    }
}

// one wave per (node, which of the two ranks): lane l owns bins 4l..4l+3, a wave prefix scan finds
// the bin holding the rank (the serial walk over 256 dependent loads cost 25 us per launch)
__global__ __launch_bounds__(64) void k_select_pick(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv,
                                                    NodeArrays na, uint32_t shift, uint32_t *__restrict__ hist) {
    const uint32_t idx = blockIdx.x, lane = threadIdx.x;
    if (idx >= lv->n_split * 2) return;
    const uint32_t li = idx >> 1, sel = idx & 1;
    const uint32_t node = lvl_node[li];
    uint4 *h4 = reinterpret_cast<uint4 *>(hist + ((size_t)li * 2 + sel) * 256);
    const uint4 c = h4[lane];
    h4[lane] = make_uint4(0u, 0u, 0u, 0u);  // ready for the next round
    const uint32_t rank = na.sel_rank[2 * node + sel];
    const uint32_t mine = c.x + c.y + c.z + c.w;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += up;
    }
    const uint32_t excl = incl - mine;
    // first bin b with rank < cumulative(b); the serial rule stops at bin 255 if none does
    const bool here = (rank >= excl) && (rank < incl);
    const uint64_t m = __ballot(here);
    uint32_t b = 255, new_rank;
    if (na.nv[node] == 0) {
        b = 0;
        new_rank = rank;
    } else if (m) {
        const int src = __builtin_ctzll(m);
        uint32_t r = rank - excl, bb = 4 * lane;
        if (r >= c.x) {
            r -= c.x;
            ++bb;
            if (r >= c.y) {
                r -= c.y;
                ++bb;
                if (r >= c.z) {
                    r -= c.z;
                    ++bb;
                }
            }
        }
        b = (uint32_t)__shfl((int)bb, src);
        new_rank = (uint32_t)__shfl((int)r, src);
    } else {  // rank beyond every bin: walk ends at bin 255 with the counts of bins 0..254 removed
        const uint32_t total = (uint32_t)__shfl((int)incl, 63), last = (uint32_t)__shfl((int)c.w, 63);
        new_rank = rank - (total - last);
    }
    if (lane == 0) {
        na.sel_rank[2 * node + sel] = new_rank;
        na.sel_prefix[2 * node + sel] |= b << shift;
    }
}

__global__ void k_median(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv, NodeArrays na) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= lv->n_split) return;
    const uint32_t node = lvl_node[li];
    const uint32_t nv = na.nv[node];
    if (nv == 0) {
        na.median[node] = 0.0f;
        return;
    }
    const float lo = key_to_float(na.sel_prefix[2 * node + 0]);
    const float hi = key_to_float(na.sel_prefix[2 * node + 1]);
    if (nv % 2 == 0) {
        const float s2 = lo + hi;
        na.median[node] = s2 / 2.0f;
    } else {
        na.median[node] = hi;
    }
}

__global__ __launch_bounds__(256) void k_flags(const float *__restrict__ vals, uint32_t n,
                                               const uint32_t *__restrict__ node_of,
                                               const uint32_t *__restrict__ remap,
                                               const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                               uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t li = node_of[i];
    if (li != kInactive) li = remap[li];
    uint32_t f = 0;
    if (li != kInactive) f = (vals[i] <= na.median[lvl_node[li]]) ? 1u : 0u;  // NaN -> right
    flags[i] = f;
}

// exclusive scan of u32 flags: 1024 elements per block
__global__ __launch_bounds__(256) void k_scan_blocks(const uint32_t *__restrict__ in, uint32_t n,
                                                     uint32_t *__restrict__ out, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t sh[256];
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = (base + q < n) ? in[base + q] : 0u;
        s += v[q];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (base + q < n) out[base + q] = excl;
        excl += v[q];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ block_sums, uint32_t nb) {
    // single workgroup; nb <= a few thousand: serial chunks of 1024
    __shared__ uint32_t sh[1024];
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint32_t t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + sh[threadIdx.x] - v;
        const uint32_t tot = sh[1023];
        __syncthreads();
        carry += tot;
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(uint32_t *__restrict__ out, uint32_t n,
                                                    const uint32_t *__restrict__ block_sums) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += block_sums[i >> 10];
}

// lefts per node = P[a+len] - P[a] (P exclusive; the last element adds its own flag)
__global__ void k_nleft(const uint32_t *__restrict__ lvl_node, const LevelInfo *__restrict__ lv, NodeArrays na,
                        const uint32_t *__restrict__ P, const uint32_t *__restrict__ flags) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= lv->n_split) return;
    const uint32_t node = lvl_node[li];
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    na.nleft[node] = (P[a + len - 1] + flags[a + len - 1]) - P[a];
}

// stable partition of every split node's segment (tsvq.rs:84-85)
__global__ __launch_bounds__(256) void k_scatter(uint32_t n, const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ node_of,
                                                 const uint32_t *__restrict__ remap,
                                                 const uint32_t *__restrict__ lvl_node, NodeArrays na,
                                                 const uint32_t *__restrict__ P,
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
    const uint32_t lr = P[i] - P[a];
    const uint32_t f = flags[i];
    const uint32_t pos = f ? (a + lr) : (a + na.nleft[node] + (i - a - lr));
    perm2[pos] = perm[i];
    node_of2[pos] = na.child_local[2 * node + (f ? 0 : 1)];
}

__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ perm, uint32_t *__restrict__ node_of, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        perm[i] = i;
        node_of[i] = 0;
    }
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
constexpr uint32_t kFsTile = 512;        // rows per tile
constexpr uint32_t kFsCols = 32;         // columns per workgroup (one 128-byte line per row)
constexpr uint32_t kFsMinRows = 16384;   // shorter nodes always keep the plain sequential kernel (most tiles of a short node sit on a binade crossing and are re-added)

struct FsTile {
    uint32_t node, t;      // node id, tile index inside the node
    uint32_t start, rows;  // position of the tile's first row in perm, rows in the tile (<= kFsTile)
};
struct FsSumm {
    int32_t d0, d1, lo0, hi0, lo1, hi1, e, flag;  // flag != 0: no usable summary
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
#pragma unroll 4
    for (uint32_t i = 0; i < kFsTile / 32; i += samp) {
        const uint32_t r = rr + 32 * i;
        if (r < tl.rows && col_ok) {
            const float4 v = *reinterpret_cast<const float4 *>(X + (size_t)perm[tl.start + r] * d + c0 + 4 * q);
            acc[0] += (double)fs_value<MODE>(v.x, mu[0]);
            acc[1] += (double)fs_value<MODE>(v.y, mu[1]);
            acc[2] += (double)fs_value<MODE>(v.z, mu[2]);
            acc[3] += (double)fs_value<MODE>(v.w, mu[3]);
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

// exclusive prefix over the tiles of a node, per column, in place: 32 chunk lanes x 32 columns per
// workgroup (chunk sums -> LDS -> offsets -> rewrite); only a guess is needed, so f64 order is free
__global__ __launch_bounds__(1024) void k_fs_prefix(uint32_t d, const uint32_t *__restrict__ tile_base,
                                                    const uint32_t *__restrict__ n_tiles_of, double *__restrict__ tile_sum,
                                                    const LevelInfo *__restrict__ lv, uint32_t *__restrict__ side_count) {
    __shared__ double part[32][kFsCols + 1];
    // the side buffer's slot counter of this pass (k_fs_transduce hands slots out; the previous pass's chain is done)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *side_count = 0u;
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t c = blockIdx.y * kFsCols + (threadIdx.x & 31), lt = threadIdx.x >> 5;
    const uint32_t base = tile_base[blockIdx.x], nt = n_tiles_of[blockIdx.x];
    const uint32_t chunk = (nt + 31) / 32, t0 = lt * chunk, t1 = min(nt, t0 + chunk);
    const bool col_ok = c < d;
    double local = 0.0;
    if (col_ok)
        for (uint32_t t = t0; t < t1; ++t) local += tile_sum[(size_t)(base + t) * d + c];
    part[lt][threadIdx.x & 31] = local;
    __syncthreads();
    if (!col_ok) return;
    double run = 0.0;
    for (uint32_t q = 0; q < lt; ++q) run += part[q][threadIdx.x & 31];
    for (uint32_t t = t0; t < t1; ++t) {
        double *p = tile_sum + (size_t)(base + t) * d + c;
        const double v = *p;
        *p = run;
        run += v;
    }
}

// Variance pass: its guess needs no pass over the rows.  The mean pass left S1 = sum (x - m0) and S2 = sum (x - m0)^2
// per (tile, column), m0 = the node's first row (so that the moments are O(sigma) whatever the data's offset), and
//     sum over the tile of (x - mu)^2  =  S2 - 2 delta S1 + rows delta^2,      delta = mu - m0,
// evaluated in f64 -- good to ~1e-6 relative against the f32 (x - mu)^2 terms the chain will add, ample for a
// binade guess.  Exclusive prefix over the node's tiles as in k_fs_prefix.
__global__ __launch_bounds__(1024) void k_fs_prefix_var(const float *__restrict__ X, uint32_t d,
                                                        const uint32_t *__restrict__ perm,
                                                        const uint32_t *__restrict__ fast_nodes,
                                                        const uint32_t *__restrict__ tile_base,
                                                        const uint32_t *__restrict__ n_tiles_of, NodeArrays na,
                                                        const double2 *__restrict__ tile_mom,
                                                        double *__restrict__ tile_sum,
                                                        const LevelInfo *__restrict__ lv, uint32_t *__restrict__ side_count) {
    __shared__ double part[32][kFsCols + 1];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *side_count = 0u;  // as in k_fs_prefix
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t c = blockIdx.y * kFsCols + (threadIdx.x & 31), lt = threadIdx.x >> 5;
    const uint32_t node = fast_nodes[blockIdx.x];
    const uint32_t base = tile_base[blockIdx.x], nt = n_tiles_of[blockIdx.x], len = na.seg_len[node];
    const uint32_t chunk = (nt + 31) / 32, t0 = lt * chunk, t1 = min(nt, t0 + chunk);
    const bool col_ok = c < d;
    double delta = 0.0;
    if (col_ok) delta = (double)na.centroid[(size_t)node * d + c] - (double)X[(size_t)perm[na.seg_start[node]] * d + c];
    auto tile_q = [&](uint32_t t) {
        const double2 m = tile_mom[(size_t)(base + t) * d + c];
        const double rows = (double)min(kFsTile, len - t * kFsTile);
        return m.y - 2.0 * delta * m.x + rows * delta * delta;
    };
    double local = 0.0;
    if (col_ok)
        for (uint32_t t = t0; t < t1; ++t) local += tile_q(t);
    part[lt][threadIdx.x & 31] = local;
    __syncthreads();
    if (!col_ok) return;
    double run = 0.0;
    for (uint32_t q = 0; q < lt; ++q) run += part[q][threadIdx.x & 31];
    for (uint32_t t = t0; t < t1; ++t) {
        tile_sum[(size_t)(base + t) * d + c] = run;
        run += tile_q(t);
    }
}

struct FsAcc {  // transducer summary of a run of rows: for even / odd incoming S
    long long d[2], lo[2], hi[2];
};
struct FsSeg {  // the same for one 64-row segment: everything fits 32 bits (|q| < 2^24 is enforced)
    int32_t d[2], lo[2], hi[2];
};

// summaries of all tiles in parallel, under the binade guessed from the f64 prefix.  Persistent workgroups
// (two per CU: the 512 x 32 tile takes 66 KB of LDS) walk the (tile, column block) items.  Per item: the
// 16 independent 16-byte loads per thread (whole 128-byte lines) fetched during the PREVIOUS item are
// parked column-major in LDS, then thread (column, segment) folds its 64 addends from LDS into the parity
// transducer.  The next item's loads are issued inside that fold -- perm indices first, the rows they
// name half-way through -- so the two dependent HBM round trips hide behind the arithmetic (a workgroup
// that did load, park, fold in sequence spent 2/3 of its time waiting: 265 -> 1xx us per pass at 1M x 128).
template <int MODE>
__global__ __launch_bounds__(256) void k_fs_transduce(const float *__restrict__ X, uint32_t d,
                                                      const uint32_t *__restrict__ perm,
                                                      const FsTile *__restrict__ tiles, const LevelInfo *__restrict__ lv, NodeArrays na,
                                                      const double *__restrict__ tile_pref,
                                                      FsSumm *__restrict__ summ, float *__restrict__ side,
                                                      uint32_t side_cap, uint32_t *__restrict__ side_count,
                                                      double2 *__restrict__ tile_mom, float park_rel_arg,
                                                      const uint32_t *__restrict__ policy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_lds[];
    float(*lds_v)[kFsTile + 1] = reinterpret_cast<float(*)[kFsTile + 1]>(fs_lds);             // [32][513]
    FsSeg(*seg_acc)[kFsCols] = reinterpret_cast<FsSeg(*)[kFsCols]>(fs_lds + kFsCols * (kFsTile + 1) * 4);  // [8][32]
    __shared__ int seg_bad[8][kFsCols];
    __shared__ int col_slot[kFsCols];
    __shared__ float2 seg_mom[8][kFsCols];  // MODE 0: sum (x - m0), sum (x - m0)^2 of a segment (the variance pass's guess)
    const uint32_t ncb = (d + kFsCols - 1) / kFsCols;  // the last column block may be short (d % 4 == 0)
    const uint32_t n_items = lv->n_tiles * ncb;
    const uint32_t q = threadIdx.x & 7, rr = threadIdx.x >> 3;     // load role: 16-byte part q of rows rr + 32 i
    const uint32_t cl = threadIdx.x & 31, seg = threadIdx.x >> 5;  // fold role: column cl, rows 64 seg ..
    uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    uint32_t prow[16];
    float4 v[16];
    // per-item metadata in three generations: _s = just requested (item after next), _n = next item, _c = current
    float mu4_s[4] = {0.f, 0.f, 0.f, 0.f}, mu4[4] = {0.f, 0.f, 0.f, 0.f};
    double pref_s = 0.0, pref = 0.0;
    uint32_t row0_s = 0;  // MODE 0: the node's first row; its values m0 centre the moments
    float m0 = 0.0f, m0_next = 0.0f;
    auto issue_perm = [&](const FsTile &t, uint32_t it) {  // indices, means and the f64 guess of item `it` -> prow, *_s
        const uint32_t tid = it / ncb, c0n = (it - tid * ncb) * kFsCols;
        if (MODE == 0) row0_s = perm[na.seg_start[t.node]];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t r = rr + 32 * i;
            prow[i] = perm[t.start + min(r, t.rows - 1u)];  // clamped, not predicated: rows past the end park as zeros
        }
        // a short last column block: the missing parts / columns read the block's first ones instead (valid
        // addresses, values never used: their summaries are not written)
        const uint32_t cq = (c0n + 4 * q < d) ? c0n + 4 * q : c0n;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mu4_s[i] = na.centroid[(size_t)t.node * d + cq + i];
        }
        pref_s = tile_pref[(size_t)tid * d + ((c0n + cl < d) ? c0n + cl : c0n)];
    };
    auto issue_rows = [&](uint32_t it, uint32_t row0) {  // the rows prow names -> v; the node's first row -> m0_next
        const uint32_t tid = it / ncb, c0n = (it - tid * ncb) * kFsCols;
        const uint32_t cq = (c0n + 4 * q < d) ? c0n + 4 * q : c0n;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4 *>(X + (size_t)prow[i] * d + cq);
        if (MODE == 0) m0_next = X[(size_t)row0 * d + ((c0n + cl < d) ? c0n + cl : c0n)];
    };
    // Pipeline (round 2): the rows of item i+1 are requested as soon as item i has been parked in LDS -- they travel
    // during the whole fold of item i -- and the row indices of item i+2 right behind them.  (Round 1 requested the
    // rows half-way through the fold: each workgroup had loads in flight for under half of its time, 2.7 TB/s.)
    FsTile tl = tiles[item / ncb];
    issue_perm(tl, item);
#pragma unroll
    for (int i = 0; i < 4; ++i) mu4[i] = mu4_s[i];
    pref = pref_s;
    issue_rows(item, row0_s);
    m0 = m0_next;
    uint32_t next = item + gridDim.x;
    FsTile tl_next = tiles[(next < n_items ? next : item) / ncb];
    if (next < n_items) issue_perm(tl_next, next);
    uint32_t next2 = next + gridDim.x;
    FsTile tl_next2 = tiles[(next2 < n_items ? next2 : item) / ncb];
  for (;;) {
    const uint32_t tile_id = item / ncb, c0 = (item - tile_id * ncb) * kFsCols;
    const uint32_t rows = tl.rows;
    {   // park the fetched rows (waits for the loads issued during the previous item)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t r = rr + 32 * i;
            const bool live = r < rows;  // rows past the node's end park as +0: a zero addend leaves the transducer as it is
            lds_v[4 * q + 0][r] = live ? fs_value<MODE>(v[i].x, mu4[0]) : 0.0f;
            lds_v[4 * q + 1][r] = live ? fs_value<MODE>(v[i].y, mu4[1]) : 0.0f;
            lds_v[4 * q + 2][r] = live ? fs_value<MODE>(v[i].z, mu4[2]) : 0.0f;
            lds_v[4 * q + 3][r] = live ? fs_value<MODE>(v[i].w, mu4[3]) : 0.0f;
        }
    }
    const uint32_t c = c0 + cl;
    const float s_guess = (float)pref;
    __syncthreads();
    const bool has_next = next < n_items;  // uniform
    float mu4_n[4] = {0.f, 0.f, 0.f, 0.f};
    double pref_n = 0.0;
    if (has_next) {
        // metadata of the next item (requested an item ago), then its rows, then the indices of the one after
#pragma unroll
        for (int i = 0; i < 4; ++i) mu4_n[i] = mu4_s[i];
        pref_n = pref_s;
        issue_rows(next, row0_s);
        if (next2 < n_items) issue_perm(tl_next2, next2);
    }
    const uint32_t gb = __float_as_uint(s_guess), ge = (gb >> 23) & 0xFFu;
    const int e = (int)ge - 127;
    // scale = 2^(23-e): needs a normal guess and a representable power of two
    bool bad = (ge == 0u) || (ge == 255u) || (23 - e > 126) || (23 - e < -126);
    const float scale = bad ? 1.0f : __uint_as_float((uint32_t)(23 - e + 127) << 23);
    // per 64-row segment everything fits 32 bits: |q| < 2^24 is enforced (an addend of 2 s or more
    // cannot leave s in its binade), so |prefix| < 2^30
    int32_t dd[2] = {0, 0}, lo[2] = {0, 0}, hi[2] = {0, 0};
    const uint32_t i0 = seg * 64;
    int badi = bad ? 1 : 0;
    // Fast fold, ONE stream: with q = x / ulp(s) and S = s / ulp(s) an integer, fl(s + x) / ulp = S + rne(q)
    // whenever q is not exactly half-way between two integers -- whatever the parity of S.  So the two parity
    // streams of the transducer coincide until a tie shows up, and a tie-free segment (all of them on continuous
    // data; the test is exact: q - rne(q) = +-1/2) costs 10 VALU operations per addend instead of 27.  A segment
    // that does hold a tie is folded again by the two-stream code below.
    int32_t d1 = 0, lo1 = 0, hi1 = 0;
    float tmax = 0.0f, sy = 0.0f, sy2 = 0.0f;
    uint32_t imax = 0u;
    auto fold1 = [&](uint32_t ib0, uint32_t ib1) {
#pragma unroll 1
    for (uint32_t ib = ib0; ib < ib1; ib += 8) {
      float qv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) qv[u] = lds_v[cl][i0 + ib + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float qq = qv[u] * scale;                 // exact (power of two) unless it overflows: caught by imax
        const float r = __builtin_rintf(qq);            // v_rndne_f32
        tmax = fmaxf(tmax, fabsf(qq - r));              // qq - r is exact; a NaN is caught by imax
        imax = max(imax, __float_as_uint(qq) & 0x7FFFFFFFu);
        d1 += (int32_t)r;
        lo1 = min(lo1, d1);
        hi1 = max(hi1, d1);
        if (MODE == 0) {                                // f32 moments of the segment around the node's first row
            const float y = qv[u] - m0;
            sy = sy + y;
            sy2 = __builtin_fmaf(y, y, sy2);
        }
      }
    }
    };
    // two streams (even / odd incoming S), exact tie handling; 64 addends per thread, 8 LDS reads in flight,
    // branch-free
    const float scale2 = scale + scale;  // 2^(24-e): exact (23 - e <= 126 was checked)
    auto fold2 = [&](uint32_t ib0, uint32_t ib1) {
#pragma unroll 1
    for (uint32_t ib = ib0; ib < ib1; ib += 8) {
      float qv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) qv[u] = lds_v[cl][i0 + ib + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // q = a + f, a = floor(q), 0 <= f < 1, classified EXACTLY from 2q (exact: |q| < 2^24): with i2 = floor(2q)
        // a = i2 >> 1, and f is above / at / below one half as (i2 odd, 2q not an integer) / (i2 odd, 2q an
        // integer) / (i2 even).  (q - floor(q) itself is inexact for -1 < q < 0: -0.49999997 would read as a tie.)
        const float q2 = qv[u] * scale2;
        const bool in_range = fabsf(q2) < 33554432.0f;  // else |q| >= 2^24, inf or NaN: cannot stay in the binade
        badi |= in_range ? 0 : 1;
        const float qq = in_range ? q2 : 0.0f;
        const float fl = floorf(qq);
        const int32_t i2 = (int32_t)fl, sticky = qq != fl ? 1 : 0;
        const int32_t ai = i2 >> 1, half = i2 & 1;
        const int32_t up = half & sticky, tie = half & (sticky ^ 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int32_t base = dd[p] + ai;
            const int32_t inc = base + (up | (tie & (p + base)));  // tie: round to the even S
            dd[p] = inc;
            lo[p] = min(lo[p], inc);
            hi[p] = max(hi[p], inc);
        }
      }
    }
    };
    fold1(0, 64);
    if (imax >= 0x4B800000u) badi = 1;  // |q| >= 2^24, inf or NaN: cannot stay in the binade
    if (!badi && tmax == 0.5f) {         // a tie in this segment: the exact two-stream fold
        fold2(0, 64);
    } else {
        dd[0] = dd[1] = d1;
        lo[0] = lo[1] = lo1;
        hi[0] = hi[1] = hi1;
    }
    bad = badi != 0;
    FsSeg acc;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        acc.d[p] = dd[p];
        acc.lo[p] = lo[p];
        acc.hi[p] = hi[p];
    }
    if (MODE == 0) seg_mom[seg][cl] = make_float2(sy, sy2);
    seg_acc[seg][cl] = acc;
    seg_bad[seg][cl] = bad ? 1 : 0;
    __syncthreads();
    if (seg == 0) {
        auto widen = [](const FsSeg &x) {
            FsAcc w;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                w.d[p] = x.d[p];
                w.lo[p] = x.lo[p];
                w.hi[p] = x.hi[p];
            }
            return w;
        };
        if (MODE == 0 && tile_mom && c < d) {
            double a1 = 0.0, a2 = 0.0;
            for (int g = 0; g < 8; ++g) {
                a1 += (double)seg_mom[g][threadIdx.x].x;
                a2 += (double)seg_mom[g][threadIdx.x].y;
            }
            // rows past the node's end were parked as +0 and entered the moments as (0 - m0): take them out
            const double pad = (double)(kFsTile - rows), m0d = (double)m0;
            tile_mom[(size_t)tile_id * d + c] = make_double2(a1 + pad * m0d, a2 - pad * m0d * m0d);
        }
        FsAcc f = widen(seg_acc[0][threadIdx.x]);
        int anybad = seg_bad[0][threadIdx.x];
        for (int g = 1; g < 8; ++g) {  // f then g, in row order
            const FsAcc gg = widen(seg_acc[g][threadIdx.x]);
            anybad |= seg_bad[g][threadIdx.x];
            FsAcc h;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bool odd = ((p + f.d[p]) & 1) != 0;  // selects, not gg.d[p2]: a run-time index sends the struct to scratch
                const long long gd = odd ? gg.d[1] : gg.d[0], gl = odd ? gg.lo[1] : gg.lo[0], gh = odd ? gg.hi[1] : gg.hi[0];
                h.d[p] = f.d[p] + gd;
                const long long l2 = f.d[p] + gl, h2 = f.d[p] + gh;
                h.lo[p] = f.lo[p] < l2 ? f.lo[p] : l2;
                h.hi[p] = f.hi[p] > h2 ? f.hi[p] : h2;
            }
            f = h;
        }
        const long long lim = 1ll << 28;
        if (f.hi[0] > lim || f.hi[1] > lim || f.lo[0] < -lim || f.lo[1] < -lim) anybad = 1;
        // Will the chain have to re-add this tile?  Predict it from the guessed S (margin 2^14 of the
        // 2^23-wide binade) and, if so, park the tile's addends column-contiguous in the side buffer:
        // the re-addition then reads 2 KB instead of gathering 4 bytes from each of 512 rows.
        int slot = -1;
        {
            const int32_t mag = (int32_t)((gb & 0x7FFFFFu) | 0x800000u);
            const long long Sg = (gb >> 31) ? -(long long)mag : (long long)mag;
            const long long lo2 = f.lo[0] < f.lo[1] ? f.lo[0] : f.lo[1], hi2 = f.hi[0] > f.hi[1] ? f.hi[0] : f.hi[1];
            // 0.2 % of the binade where the f64 guess is good to ~1e-5 (exact tile sums, the variance pass's moments).
            // A guess from every r-th row is off by ~sigma sqrt(r N): three sigmas of that (park_rel = 3 sqrt(r / 512)
            // sigma / mean for sigma / mean ~ 0.6, N = 512 (t + 1) rows so far) -- a tile whose guess is that close to
            // a binade edge is likely to be re-added, and a parked tile is read back in one go instead of 512 gathers.
            long long margin = 1ll << 14;
            const float park_rel = (policy && !policy[c0 / kFsCols]) ? 0.0f : park_rel_arg;  // exact sums for this column block
            if (park_rel > 0.0f) {
                const long long m2 = (long long)(park_rel * __builtin_amdgcn_rsqf((float)(tl.t + 1u)) * (float)(Sg < 0 ? -Sg : Sg));
                margin = m2 > margin ? (m2 < (1ll << 22) ? m2 : (1ll << 22)) : margin;
            }
            const bool leaves = (Sg > 0) ? (Sg + lo2 < (1ll << 23) + margin || Sg + hi2 > (1ll << 24) - margin)
                                         : (Sg + hi2 > -(1ll << 23) - margin || Sg + lo2 < -(1ll << 24) + margin);
            if ((anybad || leaves) && side_cap && c < d) {
                const uint32_t got = atomicAdd(side_count, 1u);
                if (got < side_cap) slot = (int)got;
            }
        }
        col_slot[threadIdx.x] = slot;
        FsSumm o;
        o.d0 = (int32_t)f.d[0];
        o.d1 = (int32_t)f.d[1];
        o.lo0 = (int32_t)f.lo[0];
        o.hi0 = (int32_t)f.hi[0];
        o.lo1 = (int32_t)f.lo[1];
        o.hi1 = (int32_t)f.hi[1];
        o.e = e;
        o.flag = (anybad ? 1 : 0) | ((slot + 1) << 1);  // bit 0: unusable; bits 1..: side slot + 1
        if (c < d) summ[(size_t)tile_id * d + c] = o;
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t cc = 0; cc < kFsCols; ++cc) {  // rare: park the flagged columns (already in LDS)
        const int slot = col_slot[cc];
        if (slot < 0) continue;
        float *dst = side + (size_t)slot * kFsTile;
        for (uint32_t i = threadIdx.x; i < rows; i += 256) dst[i] = lds_v[cc][i];
    }
    if (!has_next) break;
    __syncthreads();  // lds_v, seg_acc and col_slot are rewritten by the next item
    item = next;
    next = next2;
    next2 += gridDim.x;
    tl = tl_next;
    tl_next = tl_next2;
    tl_next2 = tiles[(next2 < n_items ? next2 : item) / ncb];
    m0 = m0_next;
    pref = pref_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) mu4[i] = mu4_n[i];
  }
}

// the exact chain: one wave per (node, column).  The tile summaries are themselves parity
// transducers, so 64 tiles at a time are composed with a wave scan: lane l learns the exact S
// entering tile t0+l (as if every earlier tile of the batch held), checks its own tile (same
// binade as the guess, prefixes inside it), and the first lane that fails marks where the batch
// stops: the tiles before it are applied in one step, that tile is re-added row by row in the
// reference's order (the additions run through v_readlane), and the scan resumes behind it.
struct FsPair {
    int32_t d[2], lo[2], hi[2];
};
__device__ __forceinline__ FsPair fs_compose(const FsPair &f, const FsPair &g) {  // f first, then g
    FsPair h;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool odd = ((p + f.d[p]) & 1) != 0;  // selects: a run-time array index would send g to scratch
        const int32_t gd = odd ? g.d[1] : g.d[0], gl = odd ? g.lo[1] : g.lo[0], gh = odd ? g.hi[1] : g.hi[0];
        // saturating enough: summaries are clamped to |.| <= 2^28 and a batch is stopped at the first
        // tile that leaves the binade, so valid prefixes stay below 2^25
        h.d[p] = f.d[p] + gd;
        h.lo[p] = min(f.lo[p], f.d[p] + gl);
        h.hi[p] = max(f.hi[p], f.d[p] + gh);
    }
    return h;
}

// DBG (VQHIP_TSVQ_DEBUG): the instantiation with the counters and timers; the production one carries none of it
template <int MODE, bool DBG>
__global__ __launch_bounds__(64) void k_fs_chain(const float *__restrict__ X, uint32_t d,
                                                 const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ fast_nodes,
                                                 const uint32_t *__restrict__ tile_base, NodeArrays na,
                                                 const FsSumm *__restrict__ summ, const float *__restrict__ side,
                                                 uint32_t *__restrict__ n_fallback, const LevelInfo *__restrict__ lv,
                                                 uint32_t *__restrict__ dbg_arg) {
    uint32_t *const dbg = DBG ? dbg_arg : nullptr;  // folds every `if (dbg)` below away when !DBG
    // dbg (VQHIP_TSVQ_DEBUG): 8 counters of this (level, pass): chains, re-added tiles, most in one chain, and the
    // first reason the re-added tile failed: unusable summary / other binade than guessed / prefix leaves the binade /
    // running sum not a normal number
    __shared__ __attribute__((aligned(16))) float stage[kFsTile];  // addends of the tile being re-added
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t node = fast_nodes[blockIdx.x], c = blockIdx.y, lane = threadIdx.x;
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nt = (len + kFsTile - 1) / kFsTile, base = tile_base[blockIdx.x];
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    uint32_t fallbacks = 0;
    uint32_t t0 = 0;  // first tile not yet applied
    // Summaries of a batch sit 32 B x d apart (one cache line each): a restart behind a re-added tile would wait a
    // full memory latency for them, and so would the re-addition for its 512 addends.  So the next batch (from the
    // tile behind the failing one, or the following 64) is requested before the re-addition starts, a second batch is
    // always in flight behind it, and the addends of the first tile the transducer PREDICTED to fail (it parked them,
    // flag bits 1..) are requested at the start of the batch, next to the scan.
    auto load_summ = [&](uint32_t tstart) {
        FsSumm m;
        m.flag = 1;
        m.e = 0;
        m.d0 = m.d1 = m.lo0 = m.lo1 = m.hi0 = m.hi1 = 0;
        if (tstart + lane < nt) m = summ[(size_t)(base + tstart + lane) * d + c];
        return m;
    };
    FsSumm cur = load_summ(0), spec = load_summ(64);
    uint32_t spec_t = 64;
    // dbg only: where the time of this chain goes (100 MHz ticks)
    const uint64_t tk0 = dbg ? wall_clock64() : 0;
    uint32_t tk_batch = 0, tk_wait_s = 0, tk_redo = 0, tk_wait_v = 0, n_batch = 0;
    while (t0 < nt) {
        const uint32_t cnt = min(64u, nt - t0);
        const uint64_t tka = dbg ? wall_clock64() : 0;
        const FsSumm mine = cur;
        if (dbg) {
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the summaries are here
            tk_wait_s += (uint32_t)(wall_clock64() - tka);
            ++n_batch;
        }
        const int myslot = (mine.flag >> 1) - 1;  // lanes past the node's tiles carry flag 1: no slot
        const uint64_t pmask = __ballot(myslot >= 0);
        const int pf = pmask ? (int)__builtin_ctzll(pmask) : -1;
        float vp[8];
        if (pf >= 0) {
            const float *src = side + (size_t)__shfl(myslot, pf) * kFsTile;
#pragma unroll
            for (int i = 0; i < 8; ++i) vp[i] = src[i * 64 + lane];
        }
        const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
        const bool s_normal = (se != 0u) && (se != 255u);
        const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
        const int32_t S = (sb >> 31) ? -mag : mag;
        // exclusive scan of the tile transducers over the lanes (Hillis-Steele on inclusive, then shift)
        FsPair incl;
        incl.d[0] = mine.d0, incl.d[1] = mine.d1;
        incl.lo[0] = mine.lo0, incl.lo[1] = mine.lo1;
        incl.hi[0] = mine.hi0, incl.hi[1] = mine.hi1;
        // in-row steps by DPP row_shr (lane i <- lane i - off of its 16-lane row), then the totals of rows 0 / 2 into
        // rows 1 / 3 (row_bcast:15) and of lane 31 into rows 2 and 3 (row_bcast:31): six steps, no LDS round trips
        // (ds_bpermute shuffles made a batch cost 1.15 us)
#define VQ_FS_STEP(CTRL, COND)                                                                   \
        {                                                                                        \
            FsPair prev;                                                                         \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                      \
                prev.d[p] = __builtin_amdgcn_update_dpp(0, incl.d[p], CTRL, 0xF, 0xF, true);     \
                prev.lo[p] = __builtin_amdgcn_update_dpp(0, incl.lo[p], CTRL, 0xF, 0xF, true);   \
                prev.hi[p] = __builtin_amdgcn_update_dpp(0, incl.hi[p], CTRL, 0xF, 0xF, true);   \
            }                                                                                    \
            if (COND) incl = fs_compose(prev, incl);                                             \
        }
        VQ_FS_STEP(0x111, (lane & 15u) >= 1u)   // row_shr:1
        VQ_FS_STEP(0x112, (lane & 15u) >= 2u)   // row_shr:2
        VQ_FS_STEP(0x114, (lane & 15u) >= 4u)   // row_shr:4
        VQ_FS_STEP(0x118, (lane & 15u) >= 8u)   // row_shr:8
        VQ_FS_STEP(0x142, (lane & 16u) != 0u)   // row_bcast:15 -> rows 1 and 3
        VQ_FS_STEP(0x143, lane >= 32u)          // row_bcast:31 -> rows 2 and 3
#undef VQ_FS_STEP
        // delta from the batch start to the start of my tile, for the actual parity of S
        const int odd = S & 1;
        int32_t before = __shfl_up(incl.d[odd], 1);
        if (lane == 0) before = 0;
        const int32_t Sin = S + before;
        const int podd = Sin & 1;
        const int32_t lo = podd ? mine.lo1 : mine.lo0, hi = podd ? mine.hi1 : mine.hi0;
        // every prefix must stay STRICTLY inside the binade on the zero side: a sum that rounds to exactly
        // +-2^23 on this grid may have had a smaller magnitude, which the finer grid below represents
        // differently (it may also be exact -- then the tile is merely re-added)
        bool ok = s_normal && (lane < cnt) && ((mine.flag & 1) == 0) && ((int)se - 127 == mine.e);
        ok = ok && ((S > 0) ? (Sin + lo > (1 << 23) && Sin + hi <= (1 << 24) - 1)
                            : (Sin + hi < -(1 << 23) && Sin + lo >= -((1 << 24) - 1)));
        const uint64_t bad_mask = __ballot(!ok) | (cnt < 64 ? (~0ull << cnt) : 0ull);
        const uint32_t good = bad_mask ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;  // tiles t0 .. t0+good-1 hold
        if (good > 0) {
            const int32_t total = __shfl(incl.d[odd], (int)good - 1);
            const int32_t S2 = S + total;
            const uint32_t m2 = (uint32_t)(S2 < 0 ? -S2 : S2);
            s = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2 & 0x7FFFFFu));
            t0 += good;
        }
        const bool redo = good < cnt;  // tile t0 (the first that failed) is re-added in the reference's order
        const int32_t flag = __shfl(mine.flag, (int)(redo ? good : 0u));
        const int32_t fe = __shfl(mine.e, (int)(redo ? good : 0u));
        {   // the batches behind this one
            const uint32_t nt0 = t0 + (redo ? 1u : 0u);
            if (nt0 == spec_t) cur = spec;
            else cur = load_summ(nt0);
            spec_t = nt0 + 64;
            spec = load_summ(spec_t);
        }
        if (dbg) tk_batch += (uint32_t)(wall_clock64() - tka);
        const uint64_t tkr = dbg ? wall_clock64() : 0;
        if (redo) {
            ++fallbacks;
            if (dbg && lane == 0) {
                const int why = !s_normal ? 6 : (flag & 1) ? 3 : ((int)se - 127 != fe) ? 4 : 5;
                atomicAdd(dbg + why, 1u);
            }
            const uint32_t r0 = t0 * kFsTile;
            float v[8];
            const int slot = (flag >> 1) - 1;
            if ((int)good == pf) {  // the predicted tile: already here
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = vp[i];
            } else if (slot >= 0) {  // parked by k_fs_transduce: contiguous
                const float *src = side + (size_t)slot * kFsTile;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = src[i * 64 + lane];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t r = r0 + (uint32_t)i * 64 + lane;
                    v[i] = (r < len) ? fs_value<MODE>(X[(size_t)perm[a + r] * d + c], mu) : 0.0f;
                }
            }
            // The 512 additions in row order.  Through v_readlane each addend went VGPR -> SGPR -> v_add with the
            // hazard wait in between: 29 cycles per addition, 7 us per tile (measured).  Staged in LDS instead, every
            // lane reads the same four addends per ds_read_b128 (a broadcast) and carries the same running sum: the
            // chain runs at the add's own latency (~10 cycles), the next 16 addends are read while these are added.
            // Rows past the node's end are +0.0 (the sum is never -0.0 once a real row is in: no bit changes).
            __syncthreads();  // one wave per block: orders the previous tile's reads before these writes
            const uint32_t rows_here = min((uint32_t)kFsTile, len - r0);
            if (dbg) {
                __builtin_amdgcn_s_waitcnt(0);
                tk_wait_v += (uint32_t)(wall_clock64() - tkr);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) stage[i * 64 + lane] = ((uint32_t)i * 64 + lane < rows_here) ? v[i] : 0.0f;  // a parked tile holds its real rows only
            __syncthreads();
            float4 qa[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const float4 *>(stage + 4 * u);
            for (uint32_t r = 0; r < rows_here; r += 32) {  // whole groups of 16: the tail adds zeros
#pragma unroll
                for (int u = 0; u < 4; ++u) qb[u] = *reinterpret_cast<const float4 *>(stage + ((r + 16 + 4 * u) & (kFsTile - 1)));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s = s + qa[u].x;
                    s = s + qa[u].y;
                    s = s + qa[u].z;
                    s = s + qa[u].w;
                }
                if (r + 16 >= rows_here) break;
#pragma unroll
                for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const float4 *>(stage + ((r + 32 + 4 * u) & (kFsTile - 1)));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s = s + qb[u].x;
                    s = s + qb[u].y;
                    s = s + qb[u].z;
                    s = s + qb[u].w;
                }
            }
            t0 += 1;
            if (dbg) tk_redo += (uint32_t)(wall_clock64() - tkr);
        }
    }
    if (dbg && lane == 0) {
        const uint32_t tot = (uint32_t)(wall_clock64() - tk0);
        if (atomicMax(dbg + 8, tot) < tot) {  // (racy between chains of similar length: diagnostic only)
            dbg[9] = n_batch, dbg[10] = tk_batch, dbg[11] = tk_wait_s, dbg[12] = fallbacks, dbg[13] = tk_redo, dbg[14] = tk_wait_v;
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
        }
    }
}

// debug aid (VQHIP_TSVQ_CHECK=1): per (node, column) walk the tiles one by one, compare the summary-applied
// sum with the row-by-row sum and report the first disagreement
template <int MODE>
__global__ __launch_bounds__(64) void k_fs_check(const float *__restrict__ X, uint32_t d, const uint32_t *__restrict__ perm,
                                                 const uint32_t *__restrict__ fast_nodes,
                                                 const uint32_t *__restrict__ tile_base, NodeArrays na,
                                                 const FsSumm *__restrict__ summ, const LevelInfo *__restrict__ lv) {
    if (blockIdx.x >= lv->n_fast) return;
    const uint32_t node = fast_nodes[blockIdx.x], c = blockIdx.y;
    if (threadIdx.x != 0) return;
    const uint32_t a = na.seg_start[node], len = na.seg_len[node];
    const uint32_t nt = (len + kFsTile - 1) / kFsTile, base = tile_base[blockIdx.x];
    const float mu = (MODE == 1) ? na.centroid[(size_t)node * d + c] : 0.0f;
    float s = (MODE == 0) ? 0.0f : -0.0f;
    for (uint32_t t = 0; t < nt; ++t) {
        const FsSumm sm = summ[(size_t)(base + t) * d + c];
        float seq = s;
        const uint32_t r0 = t * kFsTile, r1 = min(len, r0 + kFsTile);
        for (uint32_t r = r0; r < r1; ++r) seq = seq + fs_value<MODE>(X[(size_t)perm[a + r] * d + c], mu);
        const uint32_t sb = __float_as_uint(s), se = (sb >> 23) & 0xFFu;
        bool ok = ((sm.flag & 1) == 0) && se != 0u && se != 255u && ((int)se - 127 == sm.e);
        float fast = seq;
        if (ok) {
            const int32_t mag = (int32_t)((sb & 0x7FFFFFu) | 0x800000u);
            const int32_t S = (sb >> 31) ? -mag : mag;
            const int odd = S & 1;
            const int32_t D = odd ? sm.d1 : sm.d0, lo = odd ? sm.lo1 : sm.lo0, hi = odd ? sm.hi1 : sm.hi0;
            ok = (S > 0) ? (S + lo > (1 << 23) && S + hi <= (1 << 24) - 1) : (S + hi < -(1 << 23) && S + lo >= -((1 << 24) - 1));
            if (ok) {
                const int32_t S2 = S + D;
                const uint32_t m2 = (uint32_t)(S2 < 0 ? -S2 : S2);
                fast = __uint_as_float((S2 < 0 ? 0x80000000u : 0u) | (se << 23) | (m2 & 0x7FFFFFu));
                if (__float_as_uint(fast) != __float_as_uint(seq)) {
                    printf("[fs_check] mode %d node %u col %u tile %u: s=%.9g (S=%d odd=%d e=%d) summary D=%d lo=%d hi=%d -> %.9g, row by row %.9g\n",
                           MODE, node, c, t, s, S, odd, sm.e, D, lo, hi, fast, seq);
                    // find the first element where the transducer deviates
                    float run = s;
                    int32_t Sr = S;
                    const float scale = __uint_as_float((uint32_t)(23 - sm.e + 127) << 23);
                    for (uint32_t r = r0; r < r1; ++r) {
                        const float v = fs_value<MODE>(X[(size_t)perm[a + r] * d + c], mu);
                        run = run + v;
                        const float q = v * scale;
                        const float tt = fabsf(q), at = floorf(tt), ft = tt - at;
                        const bool neg = q < 0.0f, frac = ft != 0.0f;
                        const int32_t ai = neg ? -(int32_t)at - (frac ? 1 : 0) : (int32_t)at;
                        const int32_t tie = ft == 0.5f, up = (neg ? (frac && ft < 0.5f) : (ft > 0.5f));
                        const int32_t bse = Sr + ai;
                        Sr = bse + (up | (tie & bse));
                        const uint32_t rb = __float_as_uint(run);
                        const int32_t rm = (int32_t)((rb & 0x7FFFFFu) | 0x800000u);
                        const int32_t Rr = (rb >> 31) ? -rm : rm;
                        if (Rr != Sr || ((rb >> 23) & 0xFFu) != se) {
                            printf("[fs_check]   first deviation at row %u: v=%.9g q=%.9g ai=%d up=%d tie=%d  transducer S=%d, float S=%d (exp %u vs %u)\n",
                                   r - r0, v, q, ai, up, tie, Sr, Rr, (rb >> 23) & 0xFFu, se);
                            break;
                        }
                    }
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

// exclusive scan of one value per thread over the workgroup (1024 threads); returns the total
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *sh, uint32_t *total) {
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t add = (t >= off) ? sh[t - off] : 0u;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const uint32_t incl = sh[t];
    *total = sh[1023];
    __syncthreads();
    return incl - v;
}

// start of a level: the split list (+ remap level-local -> split-local), the fast / slow lists and the tile table
__global__ __launch_bounds__(1024) void k_plan_level(LevelInfo *__restrict__ lv, int can_split, int can_fast, uint32_t fs_min_rows, NodeArrays na,
                                                     uint32_t *__restrict__ lvl_split, uint32_t *__restrict__ remap,
                                                     uint32_t *__restrict__ fast_nodes, uint32_t *__restrict__ slow_nodes,
                                                     uint32_t *__restrict__ tile_base, uint32_t *__restrict__ n_tiles_of,
                                                     FsTile *__restrict__ tiles) {
    __shared__ uint32_t sh[1024];
    const uint32_t first = lv->first, count = lv->count;
    uint32_t n_split = 0, n_fast = 0, n_tiles = 0;
    for (uint32_t b = 0; b < count; b += 1024) {
        const uint32_t li = b + threadIdx.x;
        const bool in = li < count;
        const uint32_t node = first + li;
        const uint32_t len = in ? na.seg_len[node] : 0u;
        const uint32_t is_split = (in && can_split && len > 1) ? 1u : 0u;            // src/tsvq.rs:38-44
        const uint32_t is_fast = (in && can_fast && len >= fs_min_rows) ? 1u : 0u;
        const uint32_t nt = is_fast ? (len + kFsTile - 1) / kFsTile : 0u;
        uint32_t tot_s, tot_f, tot_t;
        const uint32_t ps = block_excl_scan(is_split, sh, &tot_s);
        const uint32_t pf = block_excl_scan(is_fast, sh, &tot_f);
        const uint32_t pt = block_excl_scan(nt, sh, &tot_t);
        if (in) {
            remap[li] = is_split ? n_split + ps : kInactive;
            if (is_split) lvl_split[n_split + ps] = node;
            if (is_fast) {
                fast_nodes[n_fast + pf] = node;
                tile_base[n_fast + pf] = n_tiles + pt;
                n_tiles_of[n_fast + pf] = nt;
            } else {
                slow_nodes[(li - (n_fast + pf))] = node;  // nodes before li that are not fast: li - (fast before li)
            }
        }
        n_split += tot_s;
        n_fast += tot_f;
        n_tiles += tot_t;
    }
    __threadfence();
    __syncthreads();
    // tile table: tile T belongs to the fast node f with tile_base[f] <= T < tile_base[f] + nt[f]
    for (uint32_t T = threadIdx.x; T < n_tiles; T += 1024) {
        uint32_t lo = 0, hi = n_fast;  // last f with tile_base[f] <= T
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile_base[mid] <= T) lo = mid; else hi = mid;
        }
        const uint32_t node = fast_nodes[lo], t = T - tile_base[lo];
        const uint32_t len = na.seg_len[node];
        FsTile tl;
        tl.node = node;
        tl.t = t;
        tl.start = na.seg_start[node] + t * kFsTile;
        tl.rows = min(kFsTile, len - t * kFsTile);
        tiles[T] = tl;
    }
    if (threadIdx.x == 0) {
        lv->n_split = n_split;
        lv->n_fast = n_fast;
        lv->n_slow = count - n_fast;
        lv->n_tiles = n_tiles;
    }
}

// end of a level: children of the split nodes from nleft / nv (src/tsvq.rs:88-108), numbered in order
__global__ __launch_bounds__(1024) void k_plan_children(LevelInfo *__restrict__ lv, LevelInfo *__restrict__ lv_next,
                                                        const uint32_t *__restrict__ lvl_split, NodeArrays na,
                                                        int32_t *__restrict__ node_left, int32_t *__restrict__ node_right,
                                                        uint32_t dcap) {
    __shared__ uint32_t sh[1024];
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
            nl = na.nleft[node];
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
            }
            if (has_r && next_first + ir < dcap) {
                node_right[node] = (int32_t)(next_first + ir);
                na.seg_start[next_first + ir] = start + nl;
                na.seg_len[next_first + ir] = nr;
            }
        }
        made += tot;
    }
    sh[threadIdx.x] = err;
    __syncthreads();
    for (uint32_t off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] |= sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (sh[0]) lv->error = 1;
        lv_next->first = next_first;
        lv_next->count = (next_first + made <= dcap) ? made : 0u;  // cannot happen (dcap bounds the tree); no overrun if it did
        if (next_first + made > dcap) lv->error = 2;
    }
}

// ---- host driver of the build ------------------------------------------------------------
// Device scratch of a build, kept per host thread between builds (their hipMalloc calls were 2-3 ms of a
// 13 ms build); dropped when it exceeds 1 GiB or the device changes.
struct TsvqBuildWs {
    int device = -1;
    DevBuf b_perm[2], b_nodeof[2], b_vals, b_flags, b_scan, b_bsums, b_hist, b_lvl, b_remap;
    DevBuf b_seg_start, b_seg_len, b_split, b_nv, b_nleft, b_median, b_selp, b_selr, b_child, b_cent, b_var, b_left, b_right;
    DevBuf b_fs_tiles, b_fs_nodes, b_fs_base, b_fs_nt, b_fs_sum, b_fs_summ, b_lvl_slow, b_fs_fb, b_fs_side, b_fs_mom, b_lv;
    DevBuf *all[35] = {&b_perm[0], &b_perm[1], &b_nodeof[0], &b_nodeof[1], &b_vals, &b_flags, &b_scan, &b_bsums, &b_hist,
                       &b_lvl, &b_remap, &b_seg_start, &b_seg_len, &b_split, &b_nv, &b_nleft, &b_median, &b_selp, &b_selr,
                       &b_child, &b_cent, &b_var, &b_left, &b_right, &b_fs_tiles, &b_fs_nodes, &b_fs_base, &b_fs_nt, &b_fs_sum,
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

int tsvq_build_device(const float *X, uint64_t n64, uint32_t d, uint32_t max_depth, uint32_t cap,
                      float *centroids_out, int32_t *left_out, int32_t *right_out, int32_t *n_nodes_out,
                      hipStream_t stream) {
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
            if (w.total() > (1ull << 30)) w.release();
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
    VQ_TRY(ws.b_lvl_slow.ensure((size_t)wmax * 4));
    VQ_TRY(ws.b_hist.ensure((size_t)wmax * 2 * 256 * 4));
    VQ_TRY(ws.b_seg_start.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_seg_len.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_split.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_nv.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_nleft.ensure((size_t)dcap * 4));
    VQ_TRY(ws.b_median.ensure((size_t)dcap * 4));
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
    na.sel_prefix = ws.b_selp.as<uint32_t>();
    na.sel_rank = ws.b_selr.as<uint32_t>();
    na.child_local = ws.b_child.as<uint32_t>();
    na.centroid = ws.b_cent.as<float>();
    na.var = ws.b_var.as<float>();
    int32_t *node_left = ws.b_left.as<int32_t>(), *node_right = ws.b_right.as<int32_t>();
    LevelInfo *lv = ws.b_lv.as<LevelInfo>();

    static const char *nopark = getenv("VQHIP_TSVQ_NOPARK");
    const uint32_t side_cap = (nopark && nopark[0] == '1') ? 0u : 32768u;  // parked tiles per pass (64 MB); beyond it the re-addition gathers
    static const char *seq_env = getenv("VQHIP_TSVQ_SEQSUM");  // =1: plain chain everywhere (A/B)
    const bool can_fast = (d % 4 == 0) && !(seq_env && seq_env[0] == '1');  // 16-byte row parts
    static const char *samp_env = getenv("VQHIP_TSVQ_SAMPLE");  // rows read for the mean pass's binade guess: 1/N (default 1/8)
    const uint32_t fs_sample = samp_env ? (uint32_t)std::max(1, std::min(16, atoi(samp_env))) : 8u;
    // sampling policy per block of 32 columns (k_fs_policy), behind the diagnostics in b_fs_fb; a forced VQHIP_TSVQ_SAMPLE
    // applies to every column
    const uint32_t n_cblk = (d + kFsCols - 1) / kFsCols;
    const bool adaptive_sampling = !samp_env && can_fast && n_cblk <= 1024;
    const float park_rel = fs_sample > 1 ? 3.0f * 0.6f * sqrtf((float)fs_sample / (float)kFsTile) : 0.0f;  // k_fs_transduce: parking margin
    const size_t fs_lds_bytes = (size_t)kFsCols * (kFsTile + 1) * 4 + 8 * kFsCols * sizeof(FsSeg);
    if (can_fast) {
        VQ_TRY(ws.b_fs_tiles.ensure((size_t)tiles_max * sizeof(FsTile)));
        VQ_TRY(ws.b_fs_nodes.ensure((size_t)fast_max * 4));
        VQ_TRY(ws.b_fs_base.ensure((size_t)fast_max * 4));
        VQ_TRY(ws.b_fs_nt.ensure((size_t)fast_max * 4));
        if (n >= fs_min_rows) {
            VQ_TRY(ws.b_fs_sum.ensure((size_t)tiles_max * d * 8));
            VQ_TRY(ws.b_fs_summ.ensure((size_t)tiles_max * d * sizeof(FsSumm)));
            VQ_TRY(ws.b_fs_mom.ensure((size_t)tiles_max * d * sizeof(double2)));
            VQ_TRY(ws.b_fs_side.ensure((size_t)side_cap * kFsTile * 4));
        }
        static PerDeviceOnce fs_attr;
        if (fs_attr.needed()) {
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fs_transduce<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fs_lds_bytes));
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fs_transduce<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fs_lds_bytes));
            fs_attr.done();
        }
    } else {
        VQ_TRY(ws.b_fs_tiles.ensure(16));
        VQ_TRY(ws.b_fs_nodes.ensure(16));
        VQ_TRY(ws.b_fs_base.ensure(16));
        VQ_TRY(ws.b_fs_nt.ensure(16));
    }
    const bool have_fast = can_fast && n >= fs_min_rows;

    marks.mark("allocated");
    // initial state: identity permutation, every row in node 0 (the root), no children anywhere
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, stream, ws.b_perm[0].as<uint32_t>(),
                       ws.b_nodeof[0].as<uint32_t>(), n);
    VQ_LAUNCH_CHECK("k_iota");
    VQ_HIP(hipMemsetAsync(ws.b_lv.p, 0, (size_t)(n_levels + 1) * sizeof(LevelInfo), stream));
    VQ_HIP(hipMemsetAsync(ws.b_left.p, 0xFF, (size_t)dcap * 4, stream));
    VQ_HIP(hipMemsetAsync(ws.b_right.p, 0xFF, (size_t)dcap * 4, stream));
    VQ_HIP(hipMemsetAsync(ws.b_fs_fb.p, 0, 8 + 64 * 2 * 64, stream));
    if (adaptive_sampling && n >= fs_min_rows) {
        hipLaunchKernelGGL(k_fs_policy, dim3(n_cblk), dim3(1024), 0, stream, X, n, d, ws.b_fs_fb.as<uint32_t>() + 2 + 64 * 2 * 16);
        VQ_LAUNCH_CHECK("k_fs_policy");
    }
    marks.mark("queued");
    {
        const uint32_t root[2] = {0u, n};  // seg_start[0], seg_len[0]; lv[0] = {first 0, count 1}
        const uint32_t one = 1u;
        VQ_HIP(hipMemcpyAsync(na.seg_start, &root[0], 4, hipMemcpyHostToDevice, stream));
        VQ_HIP(hipMemcpyAsync(na.seg_len, &root[1], 4, hipMemcpyHostToDevice, stream));
        VQ_HIP(hipMemcpyAsync(&lv[0].count, &one, 4, hipMemcpyHostToDevice, stream));
        marks.mark("uploaded");
        VQ_HIP(hipStreamSynchronize(stream));  // stack sources; also the only synchronisation before the final download
    }
    marks.mark("setup");
    int cur = 0;
    const uint32_t ncb = (d + kFsCols - 1) / kFsCols;
    uint32_t *lvl_split = ws.b_lvl.as<uint32_t>(), *remap = ws.b_remap.as<uint32_t>(), *slow_nodes = ws.b_lvl_slow.as<uint32_t>();
    const FsTile *tl = ws.b_fs_tiles.as<FsTile>();
    double *ts = ws.b_fs_sum.as<double>();
    FsSumm *sm = ws.b_fs_summ.as<FsSumm>();
    double2 *mom = ws.b_fs_mom.as<double2>();
    const uint32_t *fn = ws.b_fs_nodes.as<uint32_t>(), *fb = ws.b_fs_base.as<uint32_t>(), *fc = ws.b_fs_nt.as<uint32_t>();
    float *side = ws.b_fs_side.as<float>();
    uint32_t *fbk = ws.b_fs_fb.as<uint32_t>();
    const uint32_t *policy = adaptive_sampling ? fbk + 2 + 64 * 2 * 16 : nullptr;

    // sequential-order column sums of the level's nodes: long nodes through the tile-parallel exact emulation (k_fs_*),
    // the rest through the plain chain kernel.  Grids are upper bounds; the kernels read the level's counts.
    static const bool fs_debug = getenv("VQHIP_TSVQ_DEBUG") != nullptr;
    auto colsum = [&](int mode, const LevelInfo *lvp, uint32_t ub_nodes, const uint32_t *perm) -> int {
        const uint32_t lvl_idx = (uint32_t)(lvp - lv);
        uint32_t *dbg = (fs_debug && lvl_idx < 64) ? fbk + 2 + (lvl_idx * 2 + (uint32_t)mode) * 16 : nullptr;
        const uint32_t ub_fast = have_fast ? std::min(ub_nodes, fast_max) : 0u;
        // few nodes: 16 columns per workgroup (more chains in flight); many: 32 (fewer, fuller workgroups)
        const uint32_t g16 = (d + 15) / 16, g32 = (d + 31) / 32;
        static const char *narrow_env = getenv("VQHIP_TSVQ_NARROW_WGS");
        const uint64_t narrow_max = narrow_env ? (uint64_t)atoi(narrow_env) : (uint64_t)num_cus();  // measured: 16-column workgroups pay only while they leave CUs idle otherwise
        const bool narrow = (uint64_t)ub_nodes * g16 <= narrow_max;
        if (narrow) {
            if (mode == 0) hipLaunchKernelGGL((k_seg_colsum<0, 16>), dim3(ub_nodes, g16), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
            else hipLaunchKernelGGL((k_seg_colsum<1, 16>), dim3(ub_nodes, g16), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
        } else {
            if (mode == 0) hipLaunchKernelGGL((k_seg_colsum<0, 32>), dim3(ub_nodes, g32), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
            else hipLaunchKernelGGL((k_seg_colsum<1, 32>), dim3(ub_nodes, g32), dim3(1024), 0, stream, X, d, perm, slow_nodes, lvp, na);
        }
        VQ_LAUNCH_CHECK("k_seg_colsum");
        if (ub_fast == 0) return VQHIP_OK;
        const uint32_t ub_tiles = std::min(tiles_max, n / kFsTile + ub_fast);
        const dim3 tgrid(ub_tiles * ncb), xgrid(std::min<uint32_t>(ub_tiles * ncb, (uint32_t)num_cus() * 2));  // persistent: two workgroups fit a CU's LDS
        const dim3 pgrid(ub_fast, ncb), cgrid(ub_fast, d);
        if (mode == 0) {
            hipLaunchKernelGGL(k_fs_tile_sums<0>, tgrid, dim3(256), 0, stream, X, d, perm, tl, na, ts, fs_sample, policy, lvp);
            hipLaunchKernelGGL(k_fs_prefix, pgrid, dim3(1024), 0, stream, d, fb, fc, ts, lvp, fbk + 1);
            hipLaunchKernelGGL(k_fs_transduce<0>, xgrid, dim3(256), fs_lds_bytes, stream, X, d, perm, tl, lvp, na, ts, sm, side, side_cap, fbk + 1, mom, park_rel, policy);
            if (dbg) hipLaunchKernelGGL((k_fs_chain<0, true>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, side, fbk, lvp, dbg);
            else hipLaunchKernelGGL((k_fs_chain<0, false>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, side, fbk, lvp, dbg);
        } else {
            // the guess comes from the moments the mean pass of the same level left behind (same tile table: every node
            // long enough for the emulation has more than one row, so it is a split node whenever the level splits)
            hipLaunchKernelGGL(k_fs_prefix_var, pgrid, dim3(1024), 0, stream, X, d, perm, fn, fb, fc, na, mom, ts, lvp, fbk + 1);
            hipLaunchKernelGGL(k_fs_transduce<1>, xgrid, dim3(256), fs_lds_bytes, stream, X, d, perm, tl, lvp, na, ts, sm, side, side_cap, fbk + 1, (double2 *)nullptr, 0.0f, (const uint32_t *)nullptr);
            if (dbg) hipLaunchKernelGGL((k_fs_chain<1, true>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, side, fbk, lvp, dbg);
            else hipLaunchKernelGGL((k_fs_chain<1, false>), cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, side, fbk, lvp, dbg);
        }
        VQ_LAUNCH_CHECK("k_fs_*");
        if (getenv("VQHIP_TSVQ_CHECK")) {
            if (mode == 0) hipLaunchKernelGGL(k_fs_check<0>, cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, lvp);
            else hipLaunchKernelGGL(k_fs_check<1>, cgrid, dim3(64), 0, stream, X, d, perm, fn, fb, na, sm, lvp);
            VQ_HIP(hipStreamSynchronize(stream));
        }
        return VQHIP_OK;
    };

    uint32_t levels_run = 0;
    for (uint32_t L = 0; L < n_levels; ++L) {
        uint32_t ub_nodes = level_width(L);
        if (ub_nodes > 1024) {
            // wide levels: read the level's node count (one small copy + synchronisation) instead of launching the
            // per-node grids over 2^L mostly absent nodes; also ends the loop when the tree has stopped growing
            LevelInfo h;
            VQ_HIP(hipMemcpyAsync(&h, &lv[L], sizeof(LevelInfo), hipMemcpyDeviceToHost, stream));
            VQ_HIP(hipStreamSynchronize(stream));
            marks.mark("level-sync");
            if (h.count == 0) break;
            ub_nodes = h.count;
        }
        ++levels_run;
        const LevelInfo *lvp = &lv[L];
        const bool can_split = L + 1 < n_levels || L < max_depth;  // depth left at this level (src/tsvq.rs:38)
        const bool splits = L < max_depth;
        (void)can_split;
        uint32_t *perm = ws.b_perm[cur].as<uint32_t>(), *node_of = ws.b_nodeof[cur].as<uint32_t>();
        hipLaunchKernelGGL(k_plan_level, dim3(1), dim3(1024), 0, stream, &lv[L], splits ? 1 : 0, can_fast ? 1 : 0, fs_min_rows, na, lvl_split, remap,
                           ws.b_fs_nodes.as<uint32_t>(), slow_nodes, ws.b_fs_base.as<uint32_t>(), ws.b_fs_nt.as<uint32_t>(),
                           ws.b_fs_tiles.as<FsTile>());
        VQ_LAUNCH_CHECK("k_plan_level");
        // means of every node of the level (tsvq.rs:36)
        VQ_TRY(colsum(0, lvp, ub_nodes, perm));
        if (!splits) break;
        // variances + split dimension (tsvq.rs:46-66)
        VQ_TRY(colsum(1, lvp, ub_nodes, perm));
        const uint32_t nb64 = (ub_nodes + 63) / 64;
        hipLaunchKernelGGL(k_pick_split, dim3((ub_nodes + 3) / 4), dim3(256), 0, stream, lvl_split, lvp, d, na);
        VQ_LAUNCH_CHECK("k_pick_split");
        // median (tsvq.rs:68-81)
        hipLaunchKernelGGL(k_gather_vals, dim3((n + 255) / 256), dim3(256), 0, stream, X, d, n, perm, node_of, remap, lvl_split, na,
                           ws.b_vals.as<float>());
        VQ_LAUNCH_CHECK("k_gather_vals");
        hipLaunchKernelGGL(k_select_init, dim3(nb64), dim3(64), 0, stream, lvl_split, lvp, na);
        VQ_LAUNCH_CHECK("k_select_init");
        VQ_HIP(hipMemsetAsync(ws.b_hist.p, 0, (size_t)ub_nodes * 2 * 256 * 4, stream));
        for (int shift = 24; shift >= 0; shift -= 8) {
            const uint32_t hblocks = std::min<uint32_t>((n + 2047) / 2048, (uint32_t)num_cus() * 4);
            const uint32_t hchunk = (n + hblocks - 1) / hblocks;
            hipLaunchKernelGGL(k_select_hist, dim3(hblocks), dim3(256), 0, stream, ws.b_vals.as<float>(), n, hchunk, node_of, remap,
                               lvl_split, na, (uint32_t)shift, ws.b_hist.as<uint32_t>());
            VQ_LAUNCH_CHECK("k_select_hist");
            hipLaunchKernelGGL(k_select_pick, dim3(ub_nodes * 2), dim3(64), 0, stream, lvl_split, lvp, na, (uint32_t)shift,
                               ws.b_hist.as<uint32_t>());
            VQ_LAUNCH_CHECK("k_select_pick");
        }
        hipLaunchKernelGGL(k_median, dim3(nb64), dim3(64), 0, stream, lvl_split, lvp, na);
        VQ_LAUNCH_CHECK("k_median");
        // partition (tsvq.rs:84-85)
        hipLaunchKernelGGL(k_flags, dim3((n + 255) / 256), dim3(256), 0, stream, ws.b_vals.as<float>(), n, node_of, remap, lvl_split,
                           na, ws.b_flags.as<uint32_t>());
        VQ_LAUNCH_CHECK("k_flags");
        hipLaunchKernelGGL(k_scan_blocks, dim3(nblk), dim3(256), 0, stream, ws.b_flags.as<uint32_t>(), n,
                           ws.b_scan.as<uint32_t>(), ws.b_bsums.as<uint32_t>());
        VQ_LAUNCH_CHECK("k_scan_blocks");
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, ws.b_bsums.as<uint32_t>(), nblk);
        VQ_LAUNCH_CHECK("k_scan_sums");
        hipLaunchKernelGGL(k_scan_apply, dim3((n + 255) / 256), dim3(256), 0, stream, ws.b_scan.as<uint32_t>(), n,
                           ws.b_bsums.as<uint32_t>());
        VQ_LAUNCH_CHECK("k_scan_apply");
        hipLaunchKernelGGL(k_nleft, dim3(nb64), dim3(64), 0, stream, lvl_split, lvp, na, ws.b_scan.as<uint32_t>(),
                           ws.b_flags.as<uint32_t>());
        VQ_LAUNCH_CHECK("k_nleft");
        // children (tsvq.rs:88-108)
        hipLaunchKernelGGL(k_plan_children, dim3(1), dim3(1024), 0, stream, &lv[L], &lv[L + 1], lvl_split, na, node_left, node_right, dcap);
        VQ_LAUNCH_CHECK("k_plan_children");
        hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, n, perm, node_of, remap, lvl_split, na,
                           ws.b_scan.as<uint32_t>(), ws.b_flags.as<uint32_t>(), ws.b_perm[cur ^ 1].as<uint32_t>(),
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
        fprintf(stderr, "[vqhip] tsvq build: %u tile re-additions in the exact column sums\n", fbn[0]);
        if (policy) {
            uint32_t pol[1024];
            VQ_HIP(hipMemcpy(pol, policy, (size_t)n_cblk * 4, hipMemcpyDeviceToHost));
            uint32_t on = 0;
            for (uint32_t q = 0; q < n_cblk; ++q) on += pol[q] ? 1u : 0u;
            fprintf(stderr, "[vqhip]   sampled binade guess (1/%u rows) allowed for %u of %u column blocks\n", fs_sample, on, n_cblk);
        }
        for (uint32_t q = 0; q < 128; ++q) {
            const uint32_t *c8 = fbn + 2 + q * 16;
            if (c8[0])
                fprintf(stderr, "[vqhip]   level %u %s: %u chains, %u tiles re-added (most in one chain %u): unusable summary %u, "
                                "other binade than guessed %u, prefix leaves the binade %u, sum not normal %u\n",
                        q / 2, (q & 1) ? "variance" : "mean", c8[0], c8[1], c8[2], c8[3], c8[4], c8[5], c8[6]);
            if (c8[0])  // the slowest chain of the pass, 10 ns ticks of the 100 MHz clock
                fprintf(stderr, "[vqhip]     slowest chain: %.1f us = %u batches %.1f us (waiting for summaries %.1f) + %u re-additions %.1f us (waiting for addends %.1f)\n",
                        c8[8] * 0.01, c8[9], c8[10] * 0.01, c8[11] * 0.01, c8[12], c8[13] * 0.01, c8[14] * 0.01);
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
