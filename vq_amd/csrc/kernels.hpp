// kernels.hpp -- launch wrappers of the gfx950 kernels (implemented in k_*.hip).
// Every wrapper enqueues on `stream` and returns a vqhip status.
#pragma once

#include <mutex>
#include <vector>

#include "common.hpp"

namespace vqhip {

// ---- code width --------------------------------------------------------------------
// A code is one byte per (row, subspace) while k <= 256 and a little-endian u16 above; every buffer the
// ABI types `uint8_t *codes` holds n*m*code_bytes(k) bytes. The width is a function of k alone so that no
// kernel needs an extra argument for it.
__host__ __device__ __forceinline__ uint32_t code_bytes(uint32_t k) { return k <= 256u ? 1u : 2u; }
__device__ __forceinline__ uint32_t load_code(const uint8_t *codes, size_t i, uint32_t k) {
    return k <= 256u ? (uint32_t)codes[i] : (uint32_t)reinterpret_cast<const uint16_t *>(codes)[i];
}
__device__ __forceinline__ void store_code(uint8_t *codes, size_t i, uint32_t j, uint32_t k) {
    if (k <= 256u) codes[i] = (uint8_t)j;
    else reinterpret_cast<uint16_t *>(codes)[i] = (uint16_t)j;
}
constexpr uint32_t kMaxCentroids = 65536;
constexpr uint32_t kX32MaxGroups = 16;  // centroid groups of the bf16 X32 screen at k > 256

// ---- prepared codebook ------------------------------------------------------------
// Device-side view of m codebooks of k centroids of sub_dim floats, plus what the
// assignment kernels derive from it once per codebook change.
struct CodebookView {
    uint32_t m = 0, k = 0, sd = 0;
    const float *cb = nullptr;       // [m][k][sd]
    // MFMA screen operands (null when the shape has no MFMA instantiation)
    uint32_t nt = 0, ks = 0;         // 16-centroid tiles (padded, power of two), k-steps = sd/4
    const float *prepA = nullptr;    // [m][nt][ks][64]   -2*c in MFMA A-operand lane order
    const float *prepCn = nullptr;   // [m][nt*16]        |c|^2 (padding = +inf)
    const float *meta = nullptr;     // [m][4]            {max|c|, margin coefficient, -, -}
    const float *cnsqrt = nullptr;   // [m][k]            sqrt(sum c^2) (cosine's norm_b)
    const uint32_t *prepA32 = nullptr;  // [m][ceil(k/32)][NMF][4][64] same, 32x32x16 MFMA lane order
    const float *cen = nullptr;      // [m][sd+4]  X32, L2: {mu[sd], max|c-mu|, margin coefficient, -, -}
    const float *cn32 = nullptr;     // [m][ceil(k/32)*32]  X32, L2: |c-mu|^2, finite padding
};

// MFMA screen availability for a shape
bool screen_supported(uint32_t sd, uint32_t k);
void screen_tiling(uint32_t sd, uint32_t k, uint32_t *nt, uint32_t *ks);

// per-MFMA accumulation error (in units of 2^-24 (|C| + sum|ab|)) that the bf16 margin coefficients budget:
// >= kBf16ModelUlps = 18.1, the bound that follows from the bit-exact adder model (mfma_model.hpp)
constexpr float kBf16AssumedUlps = 20.0f;
// one-time device check that the hardware equals that model (k_selftest.hip); trusted = zero mismatches
int bf16_mfma_selftest(float *ratio32, float *ratio16, int *trusted);
// d[t] = one v_mfma_f32_32x32x16_bf16 of (a[t][0..16), b[t][0..16), c[t]); host buffers (diagnostics)
int mfma_bf16_probe(const uint16_t *a, const uint16_t *b, const float *c, uint64_t trials, float *d);
// software model of that instruction (mfma_model.hpp): on the host, and compared with the hardware on the device
void mfma_bf16_model_host(const uint16_t *a, const uint16_t *b, const float *c, uint64_t trials, float *d);
int mfma_bf16_model_check(uint64_t trials, uint64_t seed, uint64_t *mismatches, uint64_t *first_bad, uint64_t *fail_ids,
                          uint32_t fail_cap);
void mfma_bf16_model_case(uint64_t seed, uint64_t trial, uint16_t *a, uint16_t *b, float *c);

// bf16-split screen (k_screen_bf16.hip)
uint32_t x32_padded_sd(uint32_t sd);  // sub_dim of the X32 kernel serving `sd` (zero padding for 5..63), 0 = none
bool screen_bf16_x32_supported(uint32_t sd, uint32_t k);
bool screen_bf16_fused_update_supported(uint32_t sd, uint32_t k);
void screen_bf16_x32_tiling(uint32_t sd, uint32_t k, uint32_t *nt32_per_group, uint32_t *groups);
uint32_t screen_bf16_x32_mfmas(uint32_t sd);
int launch_prepare_bf16_x32(const CodebookView &v, uint32_t *prepA32, int cosine, float *cbc, float *cen,
                            float *cn32, hipStream_t stream);

int launch_prepare_codebook(const CodebookView &v, float *prepA, float *prepCn, float *meta,
                            float *cnsqrt, hipStream_t stream);

// ---- assignment -------------------------------------------------------------------
// LDS row pitch of k_codes_transpose's [256][pitch] tile: a multiple of 4 above m that is never a multiple of 128
// bytes (m = 124: m + 4 would put every row's byte s into one bank, a 32-way conflict on the scatter)
__host__ __device__ constexpr uint32_t codes_transpose_pitch(uint32_t m) { return m + 4 + (((m + 4) % 128 == 0) ? 4u : 0u); }

struct AssignArgs {
    const float *X = nullptr;  // [n][d]
    uint64_t n = 0;
    uint32_t d = 0;
    int metric = VQHIP_SQUARED_EUCLIDEAN;
    const uint32_t *sub_list = nullptr;  // device [n_sub] subspace ids to process
    uint32_t n_sub = 0;
    uint8_t *codes = nullptr;  // [n][m]
    // optional scratch [m][codes_t_pitch] (pitch = n rounded up to 256): the single-pass bf16 screen writes a subspace's
    // codes contiguously there and a transposition forms [n][m] (byte stores m apart cost 15 x the code bytes in
    // write traffic at m = 96); nullptr = the screen writes [n][m] itself
    uint8_t *codes_t = nullptr;
    uint64_t codes_t_pitch = 0;
    // per-subspace work lists of rows needing the exact re-check (screen -> exact)
    uint32_t *wl_rows = nullptr;   // [m][wl_stride]
    uint32_t *wl_count = nullptr;  // [m]
    uint64_t wl_stride = 0;
    // segmented form (wave-private segments, no atomics): wl_seg [m][n_seg][2] = {first slot, count};
    // a screen launch that uses it sets n_seg (> 0) for the re-check launch that follows
    uint32_t *wl_seg = nullptr;
    uint32_t wl_seg_cap = 0;
    void *part = nullptr;            // grouped X32 screen: [m][G][n] uint4 partial verdicts
    mutable uint32_t n_seg = 0;
    // fused update (training, screen_bf16_fused_update_supported shapes): partial slabs [chunk][n_sub][k][sd] /
    // [chunk][n_sub][k] that the screen fills with the sums / counts of the rows it proves; it reports the chunks it
    // used (acc_chunk_cap = slabs available / n_sub)
    float *acc_sums = nullptr;
    uint32_t *acc_counts = nullptr;
    uint32_t acc_chunk_cap = 0;
    mutable uint32_t acc_chunks = 0;
    // vqhip_kmeans_run: device flags checked by the fused screen (active [m] bytes, *halt != 0 = run paused)
    const uint8_t *gate_active = nullptr;
    const uint32_t *gate_halt = nullptr;
};

// exact VALU scan of every centroid (reference op order); if use_worklist, only the rows
// listed per subspace in wl_rows/wl_count are processed
int launch_assign_exact(const CodebookView &cb, const AssignArgs &a, bool use_worklist,
                        hipStream_t stream);
// MFMA screen: writes a provisional code for every row and appends the rows whose winner is
// not provably the reference's to the work lists (wl_count must be zeroed before)
int launch_assign_screen(const CodebookView &cb, const AssignArgs &a, hipStream_t stream);
int launch_assign_screen_bf16(const CodebookView &cb, const AssignArgs &a, hipStream_t stream);

// out[i] = metric(a[i], b[i]) in the reference's arithmetic (Distance::compute)
int launch_distance_batch(int metric, const float *a, const float *b, uint64_t n, uint32_t d,
                          float *out, hipStream_t stream);

// ---- centroid update --------------------------------------------------------------
struct UpdatePlan {
    uint32_t m = 0, k = 0, sd = 0;
    uint32_t subs_per_chunk = 0;  // subspaces whose accumulators share one workgroup's LDS
    uint32_t k_range = 0;         // clusters per workgroup (= k unless k*(sd+2) words exceed the LDS)
    uint32_t n_k_ranges = 1;
    uint32_t n_sub_chunks = 0;
    uint32_t n_row_chunks = 0;
    uint32_t owned_waves = 0;     // > 0: wave-owned (atomic-free) accumulate, waves per workgroup = subspaces x row splits
    uint32_t owned_row_split = 1; // waves of a workgroup that share a subspace, each with its own part of the chunk's rows and its own slab
    size_t partial_floats = 0;    // per row chunk: m*k*sd sums
    size_t partial_counts = 0;    // per row chunk: m*k counts
};
int plan_update(uint32_t m, uint32_t k, uint32_t sd, uint64_t n, UpdatePlan *plan);

// per-workgroup LDS accumulation of sums/counts by code -> partial slabs
int launch_accumulate(const UpdatePlan &p, const float *X, uint64_t n, uint32_t d,
                      const uint8_t *codes, const uint8_t *active, float *partial_sums,
                      uint32_t *partial_counts, hipStream_t stream);
// the rows the screen handed to the exact re-check (segmented work lists), added into `n_patch` more partial slabs
// behind the `first_chunk` the fused screen filled; codes must already hold the re-check's answers
int launch_accumulate_listed(uint32_t m, uint32_t k, uint32_t sd, const float *X, uint32_t d, const uint8_t *codes,
                             const uint32_t *sub_list, uint32_t n_sub, const uint32_t *wl_rows, uint64_t wl_stride,
                             const uint32_t *wl_seg, uint32_t n_seg, uint32_t first_chunk, uint32_t n_patch,
                             float *partial_sums, uint32_t *partial_counts, hipStream_t stream);
// fixed-order f64 combination of the fused path's slabs [chunk][position in the active list][k][sd]
int launch_reduce_partials_pos(uint32_t m, uint32_t k, uint32_t sd, const float *partial_sums, const uint32_t *partial_counts,
                               uint32_t n_chunks, uint32_t n_sub, const int32_t *sub_pos, double *slab, hipStream_t stream,
                               const uint8_t *gate_active = nullptr, const uint32_t *gate_halt = nullptr,
                               uint32_t *clear_changed = nullptr);  // device-driven run: also clears the iteration's `changed` flags [m]
// fixed-order f64 combination of the partial slabs -> slab [m][k][sd+1] (last = count)
int launch_reduce_partials(const UpdatePlan &p, const float *partial_sums,
                           const uint32_t *partial_counts, const uint8_t *active, double *slab,
                           hipStream_t stream);
// exact (reference-order) cluster sums: stable bucket by code + sequential f32 chains
int launch_exact_sums(uint32_t m, uint32_t k, uint32_t sd, const float *X, uint64_t n, uint32_t d,
                      const uint8_t *codes, const uint8_t *active, void *workspace,
                      size_t workspace_bytes, double *slab, hipStream_t stream);
size_t exact_sums_workspace_bytes(uint32_t m, uint32_t k, uint64_t n);
// means + 1e-6 convergence test; exact_div: slab sums are f32-exact values -> f32 divide
int launch_finalize(uint32_t m, uint32_t k, uint32_t sd, const double *slab, const uint8_t *active,
                    float *centroids, uint32_t *counts, uint32_t *changed, int exact_div,
                    hipStream_t stream);
// the same + the end of one iteration of a device-driven run (vqhip_kmeans_run): retire converged subspaces, count the
// iteration, raise *halt on an empty cluster; `changed` must have been cleared by the gated launch_reduce_partials_pos
int launch_finalize_run(uint32_t m, uint32_t k, uint32_t sd, const double *slab, uint8_t *active, float *centroids,
                        uint32_t *counts, uint32_t *changed, uint32_t *halt, uint32_t *iters, uint32_t *done_blocks,
                        hipStream_t stream);
int launch_reduce_finalize_run(uint32_t m, uint32_t k, uint32_t sd, const float *partial_sums, const uint32_t *partial_counts,
                               uint32_t n_chunks, uint32_t n_sub, const int32_t *sub_pos, double *slab, uint8_t *active, float *centroids,
                               uint32_t *counts, uint32_t *changed, uint32_t *halt, uint32_t *iters, uint32_t *done_blocks,
                               uint32_t *chg_scratch, hipStream_t stream);
// centroids[s][j] = X[rows[s*k+j]][s*sd ..]
int launch_gather_rows(const float *X, uint32_t d, uint32_t m, uint32_t k, uint32_t sd,
                       const uint64_t *rows, float *centroids, hipStream_t stream);

// sharded init: bits of the owned rows among global ids `rows`, zero words elsewhere
int launch_gather_rows_owned(const float *X, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const uint64_t *rows,
                             uint64_t row_offset, uint64_t n_local, uint32_t *out_bits, hipStream_t stream);

// ---- RCCL (comm.hip): the exchange step of row-sharded training --------------------------
struct Comm;
int comm_unique_id(uint8_t *id128);
int comm_create(const uint8_t *id128, int world, int rank, Comm **out);  // id128 == NULL: one rank, no RCCL
int comm_adopt(void *nccl_comm, Comm **out);
void comm_info(const Comm *c, int *world, int *rank);
int comm_destroy(Comm *c);
int comm_allreduce_f64(Comm *c, double *buf, size_t count, hipStream_t stream);
int comm_allreduce_u32(Comm *c, uint32_t *buf, size_t count, hipStream_t stream);
// the ranks of one process (a host thread per GPU): a direct fixed-order exchange through peer access (comm.hip)
struct LocalGroup;
int local_group_create(int world, LocalGroup **out);
void local_group_destroy(LocalGroup *g);
int comm_create_local(LocalGroup *g, int rank, Comm **out);  // collective over the group's ranks, each on its own thread
void comm_abort(Comm *c, const char *text);  // poison an in-process group: blocked peers return at once (no-op otherwise)
void comm_abort_rccl(Comm *c);                 // ncclCommAbort on an owned communicator (one-process RCCL team's failure path)
int comm_kind(const Comm *c);                                // 0 identity, 1 RCCL, 2 in-process exchange

// ---- TSVQ -----------------------------------------------------------------------------
// build on a device-resident matrix; outputs are HOST arrays in pre-order (see vqhip.h)
// what k_fs_policy found out about a data set's columns (which blocks of 32 may take a sampled binade guess): a property
// of the rows, kept with a library-owned (immutable) data set from its first build on -- later builds skip the kernel
// and the host knows which of the two mean-pass chains have columns to work on
struct TsvqPolicyCache {
    std::mutex mu;
    bool valid = false;
    bool no_speculation = false;  // a build found levels mixing long and short nodes: launch both column-sum paths at once
    uint32_t n_cblk = 0;
    DevBuf dev;
    std::vector<uint32_t> host;
};
int tsvq_build_device(const float *X, uint64_t n, uint32_t d, uint32_t max_depth, uint32_t cap,
                      float *centroids_out, int32_t *left_out, int32_t *right_out, int32_t *n_nodes_out,
                      hipStream_t stream, TsvqPolicyCache *policy_cache = nullptr);
// latency path for a handful of rows (k_small.hip): rows / outputs are device-visible pinned host pointers
int launch_pq_encode_small(const float *rows_dev, uint32_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, int metric,
                           const float *cb, const float *cnsqrt, uint8_t *codes_dev, uint16_t *f16_dev,
                           hipStream_t stream);
// one Lloyd iteration of a small problem in two launches (k_lloyd_small.hip)
bool lloyd_small_supported(uint64_t n, uint32_t m, uint32_t k, uint32_t sd);
size_t lloyd_small_workspace(uint64_t n, uint32_t m, uint32_t k, uint32_t sd, size_t *cnt_bytes, size_t *flag_bytes);
int launch_lloyd_small(const float *X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, float *cb, uint8_t *codes,
                       float *psum, uint32_t *pcnt, uint32_t *counts, uint32_t *changed, const uint8_t *active, uint32_t *run_flags,
                       uint32_t *run_iters, uint32_t it, hipStream_t stream);
void lloyd_small_run_result(uint32_t m, const uint8_t *start_active, const uint32_t *flags, const uint32_t *iters, bool *paused,
                            uint8_t *active_out, uint32_t *changed_out);
int launch_lloyd_small_slab(const float *X, uint64_t n, uint32_t d, uint32_t m, uint32_t k, uint32_t sd, const float *cb, uint8_t *codes,
                            float *psum, uint32_t *pcnt, const uint8_t *active, const uint32_t *gate_halt, uint32_t *changed, double *slab,
                            hipStream_t stream);
bool tsvq_small_supported(uint32_t d);
int launch_tsvq_encode_small(const float *rows_dev, uint32_t n, uint32_t d, int metric, const float *centroids,
                             const float *cnorm, const int32_t *left, const int32_t *right, int32_t *leaf_dev,
                             uint16_t *f16_dev, hipStream_t stream);

// asymmetric-distance search over stored codes (k_adc.hip)
int launch_adc_search(const float *cb, uint32_t m, uint32_t k, uint32_t sd, int metric, const uint8_t *codes, uint64_t n,
                      const float *queries_dev, uint32_t nq, uint32_t topk, float *lut_ws, float *dist_ws,
                      void *state_ws, unsigned long long *cand_ws, uint32_t *idx_out_dev, float *dist_out_dev,
                      hipStream_t stream, uint32_t qgroup);
uint32_t adc_query_batch();
uint32_t adc_query_group(uint64_t n, uint32_t nq);  // queries that go through one set of launches (a multiple of the batch)
size_t adc_state_bytes(uint32_t qgroup);
size_t adc_cand_bytes(uint32_t qgroup);
// the threshold pass (one scan of the codes, candidates only): all queries in one set of launches; redo_dev[q] = 1 where
// query q has to be repeated through launch_adc_search
bool adc_fast_eligible(uint32_t m, uint32_t k, uint64_t n, uint32_t topk);
size_t adc_fast_lut_bytes(uint32_t m, uint32_t k, uint32_t nq);
size_t adc_fast_state_bytes(uint32_t m, uint32_t k, uint32_t nq);
size_t adc_fast_cand_bytes(uint32_t m, uint32_t k, uint32_t nq);
int launch_adc_search_fast(const float *cb, uint32_t m, uint32_t k, uint32_t sd, int metric, const uint8_t *codes, uint64_t n,
                           const float *queries_dev, uint32_t nq, uint32_t topk, float *lut_ws, void *state_ws,
                           unsigned long long *cand_ws, uint32_t *idx_out_dev, float *dist_out_dev, uint32_t *redo_dev,
                           hipStream_t stream);

// prepared per-node data of the screened squared-L2 / Euclidean descent (k_tsvq_screen.hip)
struct TsvqScreen {
    const float *w = nullptr;     // [n_int][d]  c_left - c_right of every two-child node; cosine: [n_int][2][d] unit vectors of the children
    const int4 *info = nullptr;   // [n_int]     {code_l, code_r, bits(b), bits(|w|)}; code >= 0 slot, < 0 leaf -1-code; cosine: {.., .., bits(margin M, NaN = exact-only), 0}
    const int32_t *slot_node = nullptr;  // [n_int] node index of a slot
    const int32_t *node_slot = nullptr;  // [n_nodes] slot of a two-child node, -1 otherwise (the continuation's re-screen)
    int32_t start_slot = 0;       // slot the root resolves to
    const float *mu = nullptr;    // [d]         root centroid
    uint32_t n_int = 0, n_nodes = 0;  // n_int: slots resident in LDS (the levels nearest the root)
    uint32_t n_slots = 0;         // all two-child nodes of the tree (>= n_int; more: the deeper ones are read from L2)
    float R = 0.0f;               // >= max_node |c - mu|
    float coef_a = 0.0f, coef_b = 0.0f;
    uint2 *wl = nullptr;          // [n] undecided (row, node)
    // two counters used in turn: a call appends under wl_count[turn] and its continuation kernel, which is the last to
    // read it, clears the OTHER one for the next call -- the memset in front of every call was one launch in four
    uint32_t *wl_count = nullptr;
    mutable uint32_t turn = 0;    // counter of the next call (flipped by launch_tsvq_screen_encode, under the handle's lock)
};
size_t tsvq_screen_lds_bytes(uint32_t n_int, uint32_t nv, uint32_t d);
uint32_t tsvq_screen_width(uint32_t d);  // instantiated width serving d (zero padding for other multiples of 4), 0 = none
bool tsvq_screen_supported(uint32_t n_int, uint32_t n_nodes, uint32_t d, int metric);
int launch_tsvq_screen_encode(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                              const int32_t *left, const int32_t *right, int metric, const TsvqScreen &s, int32_t *leaf,
                              hipStream_t stream, const uint16_t *table16 = nullptr, uint16_t *f16_out = nullptr);
int launch_tsvq_gather_f16(const float *centroids, uint32_t d, const int32_t *leaf, uint64_t n, uint16_t *f16_out,
                           hipStream_t stream);
int launch_tsvq_node_norms(const float *centroids, uint32_t n_nodes, uint32_t d, float *cnorm, hipStream_t stream);
int launch_tsvq_encode(const float *X, uint64_t n, uint32_t d, const float *centroids, const float *cnorm,
                       const int32_t *left, const int32_t *right, int metric, int32_t *leaf, hipStream_t stream);
int launch_tsvq_table_f16(const float *centroids, uint32_t n_nodes, uint32_t d, uint16_t *table, hipStream_t stream);
int launch_tsvq_gather_table(const uint16_t *table, uint32_t d, const int32_t *leaf, uint64_t n, uint16_t *f16_out,
                             hipStream_t stream);

// ---- outputs / misc ---------------------------------------------------------------
int launch_gather_f16(const CodebookView &cb, const uint8_t *codes, uint64_t n, uint16_t *f16_out,
                      hipStream_t stream);
int launch_decode_f32(const CodebookView &cb, const uint8_t *codes, uint64_t n, float *out,
                      hipStream_t stream);
int launch_dequant_f16(const uint16_t *in, uint64_t count, float *out, hipStream_t stream);
int launch_synth_uniform(float *X, uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset,
                         hipStream_t stream);
void synth_uniform_host(float *out, uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset);

}  // namespace vqhip
