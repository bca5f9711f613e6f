/*
 * vq_oracle.c -- CPU restatement of the reference's k-means / nearest-centroid path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE -- see vq_oracle.h for the rules and the
 * parity-pin statement.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 *
 * All `file:line` citations are into /root/reference (CogitatorTech/vq 0.2.1).
 */
#include "vq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

int vqo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int resolve_threads(int threads) {
    if (threads <= 0) return vqo_max_threads();
    return threads;
}

/* ------------------------------------------------------------------ primitives ---- */

/* src/core/vector.rs:110-122 */
float vqo_dot(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        acc = acc + p;
    }
    return acc;
}

/* src/core/vector.rs:126-128 */
float vqo_norm(const float *a, size_t n) { return sqrtf(vqo_dot(a, a, n)); }

/* src/core/vector.rs:135-143 */
float vqo_distance2(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float diff = a[i] - b[i];
        float sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}

/* src/core/distance.rs:76-82.  `.sum()` folds from -0.0 on the pinned toolchain (1.85);
 * every term is >= +0 so the result equals the fold from +0.0 except for n == 0, where it
 * is -0.0 -- which compares equal to +0.0 everywhere it is used. */
static float squared_euclidean(const float *a, const float *b, size_t n) {
    float acc = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        float diff = a[i] - b[i];
        float sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}

/* src/core/distance.rs:94 */
static float manhattan(const float *a, const float *b, size_t n) {
    float acc = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        float diff = a[i] - b[i];
        acc = acc + fabsf(diff);
    }
    return acc;
}

/* src/core/distance.rs:107-119 */
static float cosine_distance(const float *a, const float *b, size_t n) {
    float dot = -0.0f, sa = -0.0f, sb = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        dot = dot + p;
    }
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * a[i];
        sa = sa + p;
    }
    for (size_t i = 0; i < n; ++i) {
        float p = b[i] * b[i];
        sb = sb + p;
    }
    float norm_a = sqrtf(sa);
    float norm_b = sqrtf(sb);
    const float EPSILON = 1e-10f;
    if (norm_a < EPSILON || norm_b < EPSILON) return 1.0f;
    float denom = norm_a * norm_b;
    float q = dot / denom;
    float v = 1.0f - q;
    /* f32::clamp(0.0, 1.0): NaN stays NaN */
    if (v < 0.0f) return 0.0f;
    if (v > 1.0f) return 1.0f;
    return v;
}

/* the reference's `simd` build: `1.0 - similarity` with no EPSILON rule and no clamp (src/core/distance.rs:97-105).
 * hsdlib's summation order is not in the reference tree: the three sums stay the scalar path's.  UNPINNED. */
static float cosine_distance_unclamped(const float *a, const float *b, size_t n) {
    float dot = -0.0f, sa = -0.0f, sb = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        dot = dot + p;
    }
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * a[i];
        sa = sa + p;
    }
    for (size_t i = 0; i < n; ++i) {
        float p = b[i] * b[i];
        sb = sb + p;
    }
    float denom = sqrtf(sa) * sqrtf(sb);
    float q = dot / denom;
    return 1.0f - q;
}

/* src/core/distance.rs:48-64 (length check is the caller's job here) */
float vqo_distance(int metric, const float *a, const float *b, size_t n) {
    switch (metric) {
    case VQO_SQUARED_EUCLIDEAN: return squared_euclidean(a, b, n);
    case VQO_EUCLIDEAN: return sqrtf(squared_euclidean(a, b, n)); /* distance.rs:58 */
    case VQO_MANHATTAN: return manhattan(a, b, n);
    case VQO_COSINE: return cosine_distance(a, b, n);
    case VQO_COSINE_UNCLAMPED: return cosine_distance_unclamped(a, b, n);
    default: return NAN;
    }
}

/* src/core/vector.rs:332-348 */
int vqo_mean_vector(const float *rows, size_t n, size_t d, size_t stride, float *out) {
    if (n == 0) return VQO_ERR_EMPTY_INPUT;
    float nf = (float)n; /* T::from_usize */
    for (size_t t = 0; t < d; ++t) out[t] = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float *v = rows + i * stride;
        for (size_t t = 0; t < d; ++t) out[t] = out[t] + v[t];
    }
    for (size_t t = 0; t < d; ++t) out[t] = out[t] / nf;
    return VQO_OK;
}

/* half::f16::from_f32 -- IEEE binary32 -> binary16, round to nearest even; NaN keeps its
 * sign, gets the quiet bit and the top 10 payload bits (src/pq.rs:194, src/tsvq.rs:252). */
uint16_t vqo_f32_to_f16(float value) {
    uint32_t x;
    memcpy(&x, &value, 4);
    uint32_t sign = x & 0x80000000u;
    uint32_t exp = x & 0x7F800000u;
    uint32_t man = x & 0x007FFFFFu;
    if (exp == 0x7F800000u) { /* inf / nan */
        uint32_t nan_bit = man == 0 ? 0 : 0x0200u;
        return (uint16_t)((sign >> 16) | 0x7C00u | nan_bit | (man >> 13));
    }
    uint32_t half_sign = sign >> 16;
    int32_t unbiased_exp = (int32_t)(exp >> 23) - 127;
    int32_t half_exp = unbiased_exp + 15;
    if (half_exp >= 0x1F) return (uint16_t)(half_sign | 0x7C00u); /* overflow -> inf */
    if (half_exp <= 0) { /* subnormal or zero */
        if (14 - half_exp > 24) return (uint16_t)half_sign; /* underflow to signed zero */
        man = man | 0x00800000u;
        uint32_t half_man = man >> (14 - half_exp);
        uint32_t round_bit = 1u << (13 - half_exp);
        if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) half_man += 1;
        return (uint16_t)(half_sign | half_man);
    }
    uint32_t he = (uint32_t)half_exp << 10;
    uint32_t hm = man >> 13;
    uint32_t round_bit = 0x00001000u;
    if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0)
        return (uint16_t)((half_sign | he | hm) + 1); /* may carry into the exponent */
    return (uint16_t)(half_sign | he | hm);
}

/* half::f16::to_f32 (exact), src/pq.rs:208 */
float vqo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x03FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while ((man & 0x0400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x03FFu) << 13);
        }
    } else if (exp == 0x1F) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + (127 - 15)) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* src/core/vector.rs:352-363 */
size_t vqo_find_nearest(const float *x, const float *centroids, size_t k, size_t sd) {
    size_t best_idx = 0;
    float best_dist = vqo_distance2(x, centroids, sd);
    for (size_t j = 1; j < k; ++j) {
        float dist = vqo_distance2(x, centroids + j * sd, sd);
        if (dist < best_dist) {
            best_dist = dist;
            best_idx = j;
        }
    }
    return best_idx;
}

/* src/pq.rs:183-191 */
size_t vqo_find_nearest_metric(int metric, const float *x, const float *centroids, size_t k,
                               size_t sd) {
    size_t best_idx = 0;
    float best_dist = vqo_distance(metric, x, centroids, sd);
    for (size_t j = 1; j < k; ++j) {
        float dist = vqo_distance(metric, x, centroids + j * sd, sd);
        if (dist < best_dist) {
            best_dist = dist;
            best_idx = j;
        }
    }
    return best_idx;
}

/* ----------------------------------------------------------------- Lloyd / LBG ---- */

/* src/core/vector.rs:232-240 */
static int approx_eq(const float *a, const float *b, size_t n, float epsilon) {
    for (size_t i = 0; i < n; ++i) {
        float diff = a[i] - b[i];
        if (!(fabsf(diff) < epsilon)) return 0;
    }
    return 1;
}

int vqo_lloyd_step(const float *data, size_t n, size_t stride, size_t sd, size_t k,
                   float *centroids, uint32_t *assign_out, uint32_t *counts_out,
                   int *changed_out, int threads) {
    if (n == 0) return VQO_ERR_EMPTY_INPUT;
    if (k == 0 || n < k) return VQO_ERR_INVALID_PARAMETER;
    uint32_t *assign = assign_out ? assign_out : (uint32_t *)malloc(n * sizeof(uint32_t));
    float *sums = (float *)calloc(k * sd, sizeof(float));
    uint32_t *counts = (uint32_t *)calloc(k, sizeof(uint32_t));
    if (!assign || !sums || !counts) {
        if (!assign_out) free(assign);
        free(sums);
        free(counts);
        return VQO_ERR_ALLOC;
    }
    int nt = resolve_threads(threads);
    (void)nt;

    /* assignment, src/core/vector.rs:417-429 (par_iter over rows when `parallel`) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (long long i = 0; i < (long long)n; ++i)
        assign[i] = (uint32_t)vqo_find_nearest(data + (size_t)i * stride, centroids, k, sd);

    /* bucket + mean, src/core/vector.rs:432-447 with 368-384.  Walking the rows once in
     * ascending order and adding each into its cluster's running sum performs, for every
     * cluster, exactly the additions of mean_vector_by_indices in the same order. */
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = assign[i];
        const float *v = data + i * stride;
        float *s = sums + (size_t)c * sd;
        for (size_t t = 0; t < sd; ++t) s[t] = s[t] + v[t];
        counts[c] += 1;
    }
    int changed = 0;
    const float EPSILON = 1e-6f; /* vector.rs:439 */
    for (size_t j = 0; j < k; ++j) {
        if (counts[j] != 0) {
            float nf = (float)counts[j]; /* indices.len() as f32, vector.rs:373 */
            float *s = sums + j * sd;
            for (size_t t = 0; t < sd; ++t) s[t] = s[t] / nf;
            if (!approx_eq(s, centroids + j * sd, sd, EPSILON)) changed = 1;
            memcpy(centroids + j * sd, s, sd * sizeof(float));
        }
        /* empty cluster: the caller reseeds (vector.rs:448-452); `changed` untouched */
    }
    if (counts_out) memcpy(counts_out, counts, k * sizeof(uint32_t));
    if (changed_out) *changed_out = changed;
    if (!assign_out) free(assign);
    free(sums);
    free(counts);
    return VQO_OK;
}

int vqo_lloyd(const float *data, size_t n, size_t stride, size_t sd, size_t k,
              size_t max_iters, const uint64_t *init_rows, const uint64_t *reseed_rows,
              size_t n_reseed, float *centroids_out, size_t *iters_out,
              size_t *reseeds_used_out, int threads) {
    /* validation, src/core/vector.rs:396-410 */
    if (n == 0) return VQO_ERR_EMPTY_INPUT;
    if (k == 0) return VQO_ERR_INVALID_PARAMETER;
    if (n < k) return VQO_ERR_INVALID_PARAMETER;
    /* init, vector.rs:412-413 with the draw injected */
    for (size_t j = 0; j < k; ++j) {
        if (init_rows[j] >= n) return VQO_ERR_INVALID_PARAMETER;
        memcpy(centroids_out + j * sd, data + (size_t)init_rows[j] * stride, sd * sizeof(float));
    }
    uint32_t *counts = (uint32_t *)malloc(k * sizeof(uint32_t));
    if (!counts) return VQO_ERR_ALLOC;
    size_t used = 0, iters = 0;
    int rc = VQO_OK;
    for (size_t it = 0; it < max_iters; ++it) { /* vector.rs:415 */
        int changed = 0;
        rc = vqo_lloyd_step(data, n, stride, sd, k, centroids_out, NULL, counts, &changed, threads);
        if (rc != VQO_OK) break;
        iters++;
        for (size_t j = 0; j < k; ++j) { /* vector.rs:448-452, ascending j */
            if (counts[j] == 0) {
                if (used >= n_reseed || reseed_rows == NULL) {
                    rc = VQO_ERR_RESEED_EXHAUSTED;
                    break;
                }
                uint64_t r = reseed_rows[used++];
                if (r >= n) {
                    rc = VQO_ERR_INVALID_PARAMETER;
                    break;
                }
                memcpy(centroids_out + j * sd, data + (size_t)r * stride, sd * sizeof(float));
            }
        }
        if (rc != VQO_OK) break;
        if (!changed) break; /* vector.rs:455-457 */
    }
    free(counts);
    if (iters_out) *iters_out = iters;
    if (reseeds_used_out) *reseeds_used_out = used;
    return rc;
}

int vqo_pq_fit(const float *rows, size_t n, size_t d, size_t m, size_t k, size_t max_iters,
               const uint64_t *init_rows, const uint64_t *reseed_rows,
               size_t n_reseed_per_sub, float *codebooks_out, size_t *iters_out,
               int threads) {
    /* src/pq.rs:91-117 */
    if (n == 0) return VQO_ERR_EMPTY_INPUT;
    if (m == 0) return VQO_ERR_INVALID_PARAMETER; /* the reference divides by zero here */
    if (d < m) return VQO_ERR_INVALID_PARAMETER;
    if (d % m != 0) return VQO_ERR_INVALID_PARAMETER;
    size_t sd = d / m;
    for (size_t s = 0; s < m; ++s) { /* src/pq.rs:121-132, sequential over subspaces */
        size_t it = 0;
        int rc = vqo_lloyd(rows + s * sd, n, d, sd, k, max_iters, init_rows + s * k,
                           reseed_rows ? reseed_rows + s * n_reseed_per_sub : NULL,
                           n_reseed_per_sub, codebooks_out + s * k * sd, &it, NULL, threads);
        if (iters_out) iters_out[s] = it;
        if (rc != VQO_OK) return rc;
    }
    return VQO_OK;
}

int vqo_pq_encode(int metric, const float *rows, size_t n, size_t d, size_t m, size_t k,
                  const float *codebooks, uint32_t *codes_out, uint16_t *f16_out,
                  int threads) {
    if (m == 0 || d % m != 0 || k == 0) return VQO_ERR_INVALID_PARAMETER;
    size_t sd = d / m;
    int nt = resolve_threads(threads);
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (long long ii = 0; ii < (long long)n; ++ii) {
        size_t i = (size_t)ii;
        const float *v = rows + i * d;
        for (size_t s = 0; s < m; ++s) { /* src/pq.rs:177-196 */
            const float *cb = codebooks + s * k * sd;
            size_t best = vqo_find_nearest_metric(metric, v + s * sd, cb, k, sd);
            if (codes_out) codes_out[i * m + s] = (uint32_t)best;
            if (f16_out)
                for (size_t t = 0; t < sd; ++t)
                    f16_out[i * d + s * sd + t] = vqo_f32_to_f16(cb[best * sd + t]);
        }
    }
    return VQO_OK;
}

/* ------------------------------------------------------------------------ TSVQ ---- */

/* f32::total_cmp key: maps the bit pattern to a signed integer with the same order */
static int32_t total_order_key(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

static int cmp_total(const void *pa, const void *pb) {
    int32_t a = total_order_key(*(const float *)pa), b = total_order_key(*(const float *)pb);
    return (a > b) - (a < b);
}

typedef struct {
    const float *rows;
    size_t d;
    size_t cap;
    float *centroids;
    int32_t *left, *right;
    uint64_t *node_rows;
    int32_t n_nodes;
    int rc;
} tsvq_ctx;

/* src/tsvq.rs:31-115.  idx = the node's training rows in their original relative order
 * (the partition at 84-85 is stable and build_from_refs clones in that order). */
static int32_t tsvq_build_node(tsvq_ctx *c, const uint64_t *idx, size_t n, size_t depth) {
    if (c->rc != VQO_OK) return -1;
    if (n == 0) {
        c->rc = VQO_ERR_EMPTY_INPUT;
        return -1;
    }
    if ((size_t)c->n_nodes >= c->cap) {
        c->rc = VQO_ERR_INVALID_PARAMETER;
        return -1;
    }
    size_t d = c->d;
    int32_t id = c->n_nodes++;
    float *mu = c->centroids + (size_t)id * d;
    c->left[id] = -1;
    c->right[id] = -1;
    if (c->node_rows) c->node_rows[id] = n;

    /* mean_vector, src/tsvq.rs:36 -> src/core/vector.rs:332-348 */
    float nf = (float)n;
    for (size_t t = 0; t < d; ++t) mu[t] = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float *v = c->rows + (size_t)idx[i] * d;
        for (size_t t = 0; t < d; ++t) mu[t] = mu[t] + v[t];
    }
    for (size_t t = 0; t < d; ++t) mu[t] = mu[t] / nf;

    if (depth == 0 || n <= 1) return id; /* src/tsvq.rs:38-44 */

    /* per-dimension un-normalised variance, src/tsvq.rs:46-57 (dim outer, rows inner) and
     * split dim, 59-66: NaN filtered, max_by returns the LAST maximum, none -> 0 */
    size_t split_dim = 0;
    int have = 0;
    float best = 0.0f;
    for (size_t t = 0; t < d; ++t) {
        float var = -0.0f;
        for (size_t i = 0; i < n; ++i) {
            float diff = c->rows[(size_t)idx[i] * d + t] - mu[t];
            float sq = diff * diff;
            var = var + sq;
        }
        if (var != var) continue; /* is_nan */
        if (!have || !(var < best)) { /* later element wins unless strictly smaller */
            best = var;
            split_dim = t;
            have = 1;
        }
    }

    /* median of the non-NaN values on split_dim, src/tsvq.rs:68-81 */
    float *values = (float *)malloc(n * sizeof(float));
    uint64_t *lidx = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *ridx = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!values || !lidx || !ridx) {
        free(values);
        free(lidx);
        free(ridx);
        c->rc = VQO_ERR_ALLOC;
        return -1;
    }
    size_t nv = 0;
    for (size_t i = 0; i < n; ++i) {
        float x = c->rows[(size_t)idx[i] * d + split_dim];
        if (x == x) values[nv++] = x;
    }
    if (nv == 0) { /* values[len/2 - 1] with len == 0: the reference panics */
        free(values);
        free(lidx);
        free(ridx);
        c->rc = VQO_ERR_REFERENCE_PANICS;
        return -1;
    }
    qsort(values, nv, sizeof(float), cmp_total);
    float median;
    if (nv % 2 == 0) {
        float s2 = values[nv / 2 - 1] + values[nv / 2];
        median = s2 / 2.0f;
    } else {
        median = values[nv / 2];
    }
    free(values);

    /* stable partition, src/tsvq.rs:84-85 (NaN <= median is false -> right) */
    size_t nl = 0, nr = 0;
    for (size_t i = 0; i < n; ++i) {
        float x = c->rows[(size_t)idx[i] * d + split_dim];
        if (x <= median) lidx[nl++] = idx[i];
        else ridx[nr++] = idx[i];
    }
    /* children, src/tsvq.rs:88-108 */
    if (nl != 0 && nl < n) {
        int32_t l = tsvq_build_node(c, lidx, nl, depth - 1);
        c->left[id] = l;
    }
    if (c->rc == VQO_OK && nr != 0 && nr < n) {
        int32_t r = tsvq_build_node(c, ridx, nr, depth - 1);
        c->right[id] = r;
    }
    free(lidx);
    free(ridx);
    return id;
}

int vqo_tsvq_build(const float *rows, size_t n, size_t d, size_t max_depth, size_t cap,
                   float *centroids, int32_t *left, int32_t *right, int32_t *n_nodes_out,
                   uint64_t *node_rows_out) {
    if (n == 0) return VQO_ERR_EMPTY_INPUT; /* src/tsvq.rs:196-198 */
    uint64_t *idx = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!idx) return VQO_ERR_ALLOC;
    for (size_t i = 0; i < n; ++i) idx[i] = i;
    tsvq_ctx c = {rows, d, cap, centroids, left, right, node_rows_out, 0, VQO_OK};
    tsvq_build_node(&c, idx, n, max_depth);
    free(idx);
    if (n_nodes_out) *n_nodes_out = c.n_nodes;
    return c.rc;
}

int vqo_tsvq_encode(int metric, const float *rows, size_t n, size_t d, const float *centroids,
                    const int32_t *left, const int32_t *right, int32_t *leaf_out,
                    uint16_t *f16_out, int threads) {
    int nt = resolve_threads(threads);
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (long long ii = 0; ii < (long long)n; ++ii) {
        size_t i = (size_t)ii;
        const float *v = rows + i * d;
        int32_t node = 0;
        for (;;) { /* src/tsvq.rs:117-132 */
            int32_t l = left[node], r = right[node];
            if (l >= 0 && r >= 0) {
                float dl = vqo_distance(metric, v, centroids + (size_t)l * d, d);
                float dr = vqo_distance(metric, v, centroids + (size_t)r * d, d);
                node = (dl <= dr) ? l : r;
            } else if (l >= 0) {
                node = l;
            } else if (r >= 0) {
                node = r;
            } else {
                break;
            }
        }
        if (leaf_out) leaf_out[i] = node;
        if (f16_out) /* src/tsvq.rs:248-253 */
            for (size_t t = 0; t < d; ++t)
                f16_out[i * d + t] = vqo_f32_to_f16(centroids[(size_t)node * d + t]);
    }
    return VQO_OK;
}

/* ---- ADC search (defines the semantics; no reference counterpart, see vq_oracle.h) -------- */
typedef struct {
    float d;
    uint32_t i;
} adc_pair;

static int adc_cmp(const void *pa, const void *pb) {
    const adc_pair *a = (const adc_pair *)pa, *b = (const adc_pair *)pb;
    const int an = a->d != a->d, bn = b->d != b->d; /* NaN last */
    if (an != bn) return an - bn;
    if (!an) {
        if (a->d < b->d) return -1;
        if (a->d > b->d) return 1;
    }
    return (a->i > b->i) - (a->i < b->i);
}

static int adc_search_impl(int metric, const float *codebooks, size_t m, size_t k, size_t sd,
                           const uint8_t *codes8, const uint16_t *codes16, size_t n, const float *queries, size_t nq, size_t topk,
                   uint32_t *idx_out, float *dist_out) {
    if (metric == VQO_COSINE || metric == VQO_COSINE_UNCLAMPED || topk == 0 || topk > n || m == 0 || k == 0 || (codes8 && k > 256) || k > 65536)
        return VQO_ERR_INVALID_PARAMETER;
    float *lut = (float *)malloc(m * k * sizeof(float));
    adc_pair *all = (adc_pair *)malloc(n * sizeof(adc_pair));
    if (!lut || !all) {
        free(lut);
        free(all);
        return VQO_ERR_ALLOC;
    }
    for (size_t q = 0; q < nq; ++q) {
        const float *x = queries + q * m * sd;
        for (size_t s = 0; s < m; ++s)
            for (size_t j = 0; j < k; ++j) {
                const float *c = codebooks + (s * k + j) * sd;
                lut[s * k + j] = (metric == VQO_MANHATTAN) ? manhattan(x + s * sd, c, sd)
                                                           : vqo_distance2(x + s * sd, c, sd);
            }
        for (size_t i = 0; i < n; ++i) {
#define VQO_CODE(ii) (codes8 ? (size_t)codes8[ii] : (size_t)codes16[ii])
            float acc = lut[VQO_CODE(i * m)];
            for (size_t s = 1; s < m; ++s) {
                const float t = lut[s * k + VQO_CODE(i * m + s)];
                acc = acc + t;
            }
#undef VQO_CODE
            all[i].d = acc;
            all[i].i = (uint32_t)i;
        }
        qsort(all, n, sizeof(adc_pair), adc_cmp);
        for (size_t r = 0; r < topk; ++r) {
            idx_out[q * topk + r] = all[r].i;
            dist_out[q * topk + r] = (metric == VQO_EUCLIDEAN) ? sqrtf(all[r].d) : all[r].d;
        }
    }
    free(lut);
    free(all);
    return VQO_OK;
}

int vqo_adc_search(int metric, const float *codebooks, size_t m, size_t k, size_t sd,
                   const uint8_t *codes, size_t n, const float *queries, size_t nq, size_t topk,
                   uint32_t *idx_out, float *dist_out) {
    return adc_search_impl(metric, codebooks, m, k, sd, codes, NULL, n, queries, nq, topk, idx_out, dist_out);
}

/* the same over two-byte codes (k up to 65536; the library's code width above 256 centroids) */
int vqo_adc_search16(int metric, const float *codebooks, size_t m, size_t k, size_t sd,
                     const uint16_t *codes, size_t n, const float *queries, size_t nq, size_t topk,
                     uint32_t *idx_out, float *dist_out) {
    return adc_search_impl(metric, codebooks, m, k, sd, NULL, codes, n, queries, nq, topk, idx_out, dist_out);
}
