/*
 * vq_oracle.h -- CPU restatement of the reference's k-means / nearest-centroid path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or load this library, and only as the checker
 * (or as the labelled CPU baseline).  Nothing under vq_amd/ may call it.
 *
 * Every function restates one piece of CogitatorTech/vq (crate `vq` 0.2.1) in plain C
 * and cites the reference file:line it follows.  Arithmetic is IEEE binary32, one
 * rounding per operation, no FMA contraction, sums strictly in index order -- the
 * library MUST be compiled with  -ffp-contract=off  and without -ffast-math
 * (oracle/Makefile does).  The reference's `simd` (hsdlib) branches are NOT restated:
 * hsdlib's source is absent from the reference mount (empty submodule
 * habedi/hsdlib@main, no pinned commit), so the oracle follows the scalar fallbacks.
 *
 * Parity pin: the reference ships no golden vectors for this path and cannot be
 * built in this environment (Rust; no cargo/rustc).  The oracle is pinned against
 * every RNG-independent known-answer case the reference's own tests hold
 * (tests/test_oracle_kat.py lists them with file:line) and cross-checked against an
 * independent numpy restatement (tests/test_oracle_numpy.py).  Beyond those cases --
 * i.e. for non-trivial centroids/codes -- PARITY IS UNPINNED against a running
 * reference; RNG-dependent behaviour (rand 0.9 StdRng draws) is injected by the
 * caller (init_rows / reseed_rows) instead of being reproduced.
 */
#ifndef VQ_ORACLE_H
#define VQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Distance metric ids -- same order as the reference enum, src/core/distance.rs:8-17. */
enum {
    VQO_SQUARED_EUCLIDEAN = 0,
    VQO_EUCLIDEAN = 1,
    VQO_MANHATTAN = 2,
    VQO_COSINE = 3,
    VQO_COSINE_UNCLAMPED = 4 /* opt-in, unpinned: the `simd` build's 1 - similarity (no EPSILON rule, no clamp) */
};

/* Status codes.  0 ok; the others mirror VqError variants (src/core/error.rs:4-28). */
enum {
    VQO_OK = 0,
    VQO_ERR_EMPTY_INPUT = 1,        /* VqError::EmptyInput */
    VQO_ERR_DIMENSION_MISMATCH = 2, /* VqError::DimensionMismatch */
    VQO_ERR_INVALID_PARAMETER = 3,  /* VqError::InvalidParameter */
    VQO_ERR_RESEED_EXHAUSTED = 4,   /* caller supplied too few reseed rows (oracle-only) */
    VQO_ERR_REFERENCE_PANICS = 5,   /* input on which the reference panics (tsvq.rs:77-78) */
    VQO_ERR_ALLOC = 6
};

/* ---- primitives ------------------------------------------------------------------ */

/* Vector::dot, src/core/vector.rs:110-122 : fold(0, acc + a*b). */
float vqo_dot(const float *a, const float *b, size_t n);
/* Vector::norm, src/core/vector.rs:126-128 : sqrt(dot(self,self)). */
float vqo_norm(const float *a, size_t n);
/* Vector::distance2, src/core/vector.rs:135-143 : fold(0, acc + (a-b)*(a-b)). */
float vqo_distance2(const float *a, const float *b, size_t n);
/* Distance::compute scalar paths, src/core/distance.rs:48-64, 76-82, 94, 107-119. */
float vqo_distance(int metric, const float *a, const float *b, size_t n);
/* mean_vector, src/core/vector.rs:332-348 : column sums in row order, then / (float)n.
 * rows are `stride` floats apart. */
int vqo_mean_vector(const float *rows, size_t n, size_t d, size_t stride, float *out);

/* half::f16::from_f32 / to_f32 (IEEE round-to-nearest-even), src/pq.rs:194,208. */
uint16_t vqo_f32_to_f16(float x);
float vqo_f16_to_f32(uint16_t h);

/* find_nearest_centroid, src/core/vector.rs:352-363 : squared L2, strict '<', first min
 * wins.  centroids is [k][sd] contiguous. */
size_t vqo_find_nearest(const float *x, const float *centroids, size_t k, size_t sd);
/* the argmin loop of ProductQuantizer::quantize, src/pq.rs:183-191, any metric. */
size_t vqo_find_nearest_metric(int metric, const float *x, const float *centroids,
                               size_t k, size_t sd);

/* ---- Lloyd / LBG ------------------------------------------------------------------ */

/*
 * One iteration of the loop body of lbg_quantize, src/core/vector.rs:415-453, on the
 * sub-vectors data[i*stride .. i*stride+sd).
 *   centroids   [k][sd]  in: current, out: updated (empty clusters keep their old
 *                        value -- the caller applies reseeds, vector.rs:448-452)
 *   assign_out  [n]      optional: cluster index per row (vector.rs:417-429)
 *   counts_out  [k]      optional: members per cluster (vector.rs:432-435)
 *   changed_out          1 iff some non-empty cluster moved by >= 1e-6 in a component
 *                        (vector.rs:438-447, 232-240)
 * threads > 1 parallelises the assignment over rows (rayon par_iter, vector.rs:417-423);
 * the update stays serial like the reference.
 */
int vqo_lloyd_step(const float *data, size_t n, size_t stride, size_t sd, size_t k,
                   float *centroids, uint32_t *assign_out, uint32_t *counts_out,
                   int *changed_out, int threads);

/*
 * lbg_quantize, src/core/vector.rs:390-461, with the two RNG draws injected:
 *   init_rows   [k]      row ids that `choose_multiple` would return (vector.rs:412-413)
 *   reseed_rows [n_reseed] row ids that successive `choose` calls would return, consumed
 *                        in order (cluster index ascending within an iteration,
 *                        vector.rs:448-452)
 *   centroids_out [k][sd]
 *   iters_out   optional: loop iterations executed
 *   reseeds_used_out optional
 */
int vqo_lloyd(const float *data, size_t n, size_t stride, size_t sd, size_t k,
              size_t max_iters, const uint64_t *init_rows, const uint64_t *reseed_rows,
              size_t n_reseed, float *centroids_out, size_t *iters_out,
              size_t *reseeds_used_out, int threads);

/*
 * ProductQuantizer::new, src/pq.rs:83-141 (validation 91-117, per-subspace LBG 120-132).
 *   rows [n][d] contiguous;  init_rows [m][k];  reseed_rows [m][n_reseed_per_sub]
 *   codebooks_out [m][k][d/m];  iters_out optional [m]
 */
int vqo_pq_fit(const float *rows, size_t n, size_t d, size_t m, size_t k, size_t max_iters,
               const uint64_t *init_rows, const uint64_t *reseed_rows,
               size_t n_reseed_per_sub, float *codebooks_out, size_t *iters_out,
               int threads);

/*
 * ProductQuantizer::quantize over a batch, src/pq.rs:167-199.
 *   codes_out [n][m] optional: the internal best_idx per subspace (pq.rs:183-191)
 *   f16_out   [n][d] optional: selected centroid values as f16 bits (pq.rs:193-195)
 * threads > 1 parallelises over rows (the reference itself encodes on one thread).
 */
int vqo_pq_encode(int metric, const float *rows, size_t n, size_t d, size_t m, size_t k,
                  const float *codebooks, uint32_t *codes_out, uint16_t *f16_out,
                  int threads);

/* ---- TSVQ ------------------------------------------------------------------------- */

/*
 * TSVQNode::build, src/tsvq.rs:31-115, flattened to arrays in pre-order (node 0 = root,
 * then the whole left subtree, then the right one).  Capacity needed: 2^(max_depth+1)-1.
 *   centroids [cap][d];  left/right [cap] child index or -1;  n_nodes_out
 *   node_rows_out optional [cap]: training rows that reached each node
 */
/*
 * Asymmetric distance search over stored codes.  NOT a restatement: the reference has no such
 * function (it keeps f16 reconstructions, src/pq.rs:165-199; SURVEY.md 8(f) N3 lists it as the
 * next tier).  This fixes the semantics the GPU path is tested against, built from restated
 * pieces: t_s(q, j) = vqo_distance2 (src/core/vector.rs:135-143) or the L1 kernel
 * (src/core/distance.rs:85-95) between the query's sub-vector s and centroid j;
 * D(q, i) = t_0 + t_1 + ... + t_{m-1} over codes[i][s], in subspace order, f32;
 * result = the topk rows by (D, row) ascending, NaN last; Euclidean reports sqrt(D).
 * Returns VQO_ERR_INVALID_PARAMETER for cosine (not a sum over subspaces).
 */
int vqo_adc_search(int metric, const float *codebooks, size_t m, size_t k, size_t sd,
                   const uint8_t *codes, size_t n, const float *queries, size_t nq, size_t topk,
                   uint32_t *idx_out, float *dist_out);
int vqo_adc_search16(int metric, const float *codebooks, size_t m, size_t k, size_t sd,
                     const uint16_t *codes, size_t n, const float *queries, size_t nq, size_t topk,
                     uint32_t *idx_out, float *dist_out);

int vqo_tsvq_build(const float *rows, size_t n, size_t d, size_t max_depth, size_t cap,
                   float *centroids, int32_t *left, int32_t *right, int32_t *n_nodes_out,
                   uint64_t *node_rows_out);

/* TSVQNode::find_leaf + TSVQ::quantize, src/tsvq.rs:117-132, 239-255, over a batch.
 *   leaf_out [n] optional; f16_out [n][d] optional */
int vqo_tsvq_encode(int metric, const float *rows, size_t n, size_t d, const float *centroids,
                    const int32_t *left, const int32_t *right, int32_t *leaf_out,
                    uint16_t *f16_out, int threads);

/* number of OpenMP threads the library would use for threads<=0 (host core count) */
int vqo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* VQ_ORACLE_H */
