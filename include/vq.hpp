// vq.hpp -- C++ host mirror of the reference's public interface for the hot path, over libvqhip.
//
// The reference is a Rust crate (no Rust toolchain in this image), so the compiled-language host
// side above the C ABI (include/vqhip.h) is this header: same type and method names, argument
// meaning, defaults and error behaviour as
//   vq::Distance           src/core/distance.rs:8-64
//   vq::VqError            src/core/error.rs:4-28   (what() == the Rust Display text)
//   vq::ProductQuantizer   src/pq.rs:83-210         (new / getters / Quantizer::quantize / dequantize)
//   vq::TSVQ               src/tsvq.rs:195-266
//   vq::lbg_quantize       src/core/vector.rs:390-461
// Validation happens before anything touches the device (same order and messages as the
// reference); the bodies run on the MI355X.  Random draws (src/core/vector.rs:412-413, 448-452) come
// from the documented SplitMix64 sampler also used by the Python mirror (vq_amd/rng.py) -- rand's
// StdRng stream is not reproducible outside Rust -- or from the caller (`init_rows`).
//
// Header-only, C++17; link with -lvqhip.  Thread-safe per object like the reference's plain-data types (`Send + Sync`,
// src/pq.rs:39-45): the `const` methods (`quantize`, `dequantize`, the getters) may be called on one object from any
// number of threads at once -- every libvqhip handle carries its own lock and hands its stream's tail over between
// threads (include/vqhip.h "Threads"; held by tests/cpp/test_vq_hpp.cpp's std::thread case).
#ifndef VQ_HPP
#define VQ_HPP

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "vqhip.h"

namespace vq {

// ---------------------------------------------------------------------------- errors ----
class VqError : public std::runtime_error {
   public:
    enum class Kind { DimensionMismatch, EmptyInput, InvalidParameter, InvalidData, FfiError };
    Kind kind;
    std::size_t expected = 0, found = 0;  // DimensionMismatch
    std::string parameter, reason;        // InvalidParameter

    static VqError DimensionMismatch(std::size_t expected, std::size_t found) {
        VqError e(Kind::DimensionMismatch, "Dimension mismatch: expected " + std::to_string(expected) + ", found " +
                                               std::to_string(found));
        e.expected = expected;
        e.found = found;
        return e;
    }
    static VqError EmptyInput() { return VqError(Kind::EmptyInput, "Empty input: at least one vector is required"); }
    static VqError InvalidParameter(const std::string &parameter, const std::string &reason) {
        VqError e(Kind::InvalidParameter, "Invalid parameter '" + parameter + "': " + reason);
        e.parameter = parameter;
        e.reason = reason;
        return e;
    }
    static VqError InvalidData(const std::string &what) { return VqError(Kind::InvalidData, "Invalid data: " + what); }
    static VqError FfiError(const std::string &what) { return VqError(Kind::FfiError, "FFI error: " + what); }

   private:
    VqError(Kind k, const std::string &msg) : std::runtime_error(msg), kind(k) {}
};

namespace detail {
inline void check(int status) {
    if (status != VQHIP_OK) throw VqError::FfiError(vqhip_last_error());
}
}  // namespace detail

// ------------------------------------------------------------------------------- f16 ----
// `half::f16` stand-in: the bits the device wrote (IEEE binary16, round-to-nearest-even)
struct f16 {
    std::uint16_t bits = 0;
    float to_f32() const {
        const std::uint32_t s = (std::uint32_t)(bits & 0x8000u) << 16, e = (bits >> 10) & 0x1Fu, m = bits & 0x3FFu;
        std::uint32_t u;
        if (e == 0) {
            if (m == 0) {
                u = s;
            } else {  // subnormal: normalise
                int sh = 0;
                std::uint32_t mm = m;
                while (!(mm & 0x400u)) {
                    mm <<= 1;
                    ++sh;
                }
                u = s | ((std::uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FFu) << 13);
            }
        } else if (e == 31) {
            u = s | 0x7F800000u | (m << 13);
        } else {
            u = s | ((e + 112u) << 23) | (m << 13);
        }
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    bool operator==(const f16 &o) const { return bits == o.bits; }
};

// -------------------------------------------------------------------------- Distance ----
class Distance {
   public:
    enum Kind : int {
        SquaredEuclidean = VQHIP_SQUARED_EUCLIDEAN,
        Euclidean = VQHIP_EUCLIDEAN,
        Manhattan = VQHIP_MANHATTAN,
        CosineDistance = VQHIP_COSINE,
    };
    constexpr Distance(Kind k = Euclidean) : kind_(k) {}
    constexpr Kind kind() const { return kind_; }
    constexpr bool operator==(const Distance &o) const { return kind_ == o.kind_; }
    // src/core/distance.rs:22-29
    const char *name() const {
        switch (kind_) {
            case SquaredEuclidean: return "squared_euclidean";
            case Euclidean: return "euclidean";
            case Manhattan: return "manhattan";
            default: return "cosine";
        }
    }
    // src/core/distance.rs:48-64 (scalar kernels; one pair per call, evaluated on the device)
    float compute(const float *a, std::size_t a_len, const float *b, std::size_t b_len) const {
        if (a_len != b_len) throw VqError::DimensionMismatch(a_len, b_len);
        float out = 0.0f;
        if (a_len == 0) return kind_ == CosineDistance ? 1.0f : 0.0f;
        detail::check(vqhip_distance_batch((int)kind_, a, b, 1, (std::uint32_t)a_len, &out));
        return out;
    }
    float compute(const std::vector<float> &a, const std::vector<float> &b) const {
        return compute(a.data(), a.size(), b.data(), b.size());
    }

   private:
    Kind kind_;
};

// ------------------------------------------------------------------------------- rng ----
// vq_amd/rng.py in C++: SplitMix64, Lemire's bounded integers, Floyd's sampling
class HostRng {
   public:
    explicit HostRng(std::uint64_t seed) : state_(seed) {}
    std::uint64_t next_u64() {
        state_ += 0x9E3779B97F4A7C15ull;
        std::uint64_t z = state_;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    std::uint64_t below(std::uint64_t n) {
        const std::uint64_t threshold = (0 - n) % n;
        for (;;) {
            const unsigned __int128 m = (unsigned __int128)next_u64() * n;
            if ((std::uint64_t)m >= threshold) return (std::uint64_t)(m >> 64);
        }
    }
    std::uint64_t choose(std::uint64_t n) { return below(n); }
    std::vector<std::uint64_t> choose_multiple(std::uint64_t n, std::uint64_t k) {
        std::unordered_set<std::uint64_t> chosen;
        std::vector<std::uint64_t> out;
        out.reserve(k);
        for (std::uint64_t j = n - k; j < n; ++j) {
            const std::uint64_t t = below(j + 1), pick = chosen.count(t) ? j : t;
            chosen.insert(pick);
            out.push_back(pick);
        }
        return out;
    }

   private:
    std::uint64_t state_;
};

namespace detail {

struct DatasetDel {
    void operator()(vqhip_dataset *p) const { vqhip_dataset_destroy(p); }
};
struct KMeansDel {
    void operator()(vqhip_kmeans *p) const { vqhip_kmeans_destroy(p); }
};
struct EncoderDel {
    void operator()(vqhip_pq_encoder *p) const { vqhip_pq_encoder_destroy(p); }
};
struct TsvqDel {
    void operator()(vqhip_tsvq *p) const { vqhip_tsvq_destroy(p); }
};
struct MTsvqDel {
    void operator()(vqhip_mtsvq *p) const { (void)vqhip_mtsvq_destroy(p); }
};

// `&[&[f32]]` -> contiguous [n][dim] with the reference's checks (src/pq.rs:91-104, src/tsvq.rs:196-210)
inline std::vector<float> flatten(const std::vector<std::vector<float>> &rows, std::size_t *dim) {
    if (rows.empty()) throw VqError::EmptyInput();
    *dim = rows[0].size();
    for (const auto &r : rows)
        if (r.size() != *dim) throw VqError::DimensionMismatch(*dim, r.size());
    std::vector<float> flat(rows.size() * *dim);
    for (std::size_t i = 0; i < rows.size(); ++i)
        if (*dim) std::memcpy(flat.data() + i * *dim, rows[i].data(), *dim * sizeof(float));
    return flat;
}

// src/core/vector.rs:396-410
inline void check_lbg_params(std::size_t n, std::size_t k) {
    if (n == 0) throw VqError::EmptyInput();
    if (k == 0) throw VqError::InvalidParameter("k", "must be greater than 0");
    if (n < k)
        throw VqError::InvalidParameter(
            "k", "not enough data points (" + std::to_string(n) + ") for " + std::to_string(k) + " clusters");
}

// control flow of lbg_quantize (src/core/vector.rs:412-460) for all m subspaces of a resident
// data set; the device does each Lloyd iteration, the host keeps the RNG (seed + s per subspace,
// src/pq.rs:130) and the convergence / empty-cluster logic
inline std::vector<float> fit_codebooks(vqhip_dataset *ds, std::uint64_t n, std::uint32_t dim, std::uint32_t m,
                                        std::uint32_t k, std::size_t max_iters, std::uint64_t seed,
                                        const std::uint64_t *init_rows = nullptr) {
    std::vector<HostRng> rngs;
    for (std::uint32_t s = 0; s < m; ++s) rngs.emplace_back(seed + s);
    std::vector<std::uint64_t> init((std::size_t)m * k);
    if (init_rows) {
        std::memcpy(init.data(), init_rows, init.size() * 8);
    } else {
        for (std::uint32_t s = 0; s < m; ++s) {
            const auto rows = rngs[s].choose_multiple(n, k);
            std::memcpy(init.data() + (std::size_t)s * k, rows.data(), (std::size_t)k * 8);
        }
    }
    vqhip_kmeans *raw = nullptr;
    check(vqhip_kmeans_create(ds, m, k, &raw));
    std::unique_ptr<vqhip_kmeans, KMeansDel> km(raw);
    check(vqhip_kmeans_init_from_rows(km.get(), init.data()));
    std::vector<std::uint8_t> active(m, 1), changed(m, 0);
    std::vector<std::uint32_t> counts((std::size_t)m * k);
    for (std::size_t it = 0; it < max_iters; ++it) {
        bool any = false;
        for (auto a : active) any = any || a;
        if (!any) break;
        check(vqhip_kmeans_step(km.get(), counts.data(), changed.data()));
        for (std::uint32_t s = 0; s < m; ++s) {
            if (!active[s]) continue;
            for (std::uint32_t j = 0; j < k; ++j)  // empty clusters in ascending j, vector.rs:448-452
                if (counts[(std::size_t)s * k + j] == 0)
                    check(vqhip_kmeans_patch_from_row(km.get(), s, j, rngs[s].choose(n)));
            if (!changed[s]) active[s] = 0;  // vector.rs:455-457
        }
        check(vqhip_kmeans_set_active(km.get(), active.data()));
    }
    std::vector<float> cb((std::size_t)m * k * (dim / m));
    check(vqhip_kmeans_get_centroids(km.get(), cb.data()));
    return cb;
}

struct MDatasetDel {
    void operator()(vqhip_mdataset *p) const { (void)vqhip_mdataset_destroy(p); }
};
struct MKMeansDel {
    void operator()(vqhip_mkmeans *p) const { (void)vqhip_mkmeans_destroy(p); }
};
struct MEncoderDel {
    void operator()(vqhip_mpq_encoder *p) const { (void)vqhip_mpq_encoder_destroy(p); }
};

// the same control flow with the training batch partitioned over several GPUs of THIS process (include/vqhip.h "one
// call, one process, several GPUs": the ranks of the row-sharded fit are worker threads inside the library; each Lloyd
// iteration all-reduces the per-cluster sums).  The loop's decisions are taken on the device (vqhip_mkmeans_run): it
// comes back when the iterations are used up, when every subspace has converged, or paused behind an iteration that
// left a cluster empty -- the reseed row is this side's draw (vector.rs:448-452), named by its global row id.
inline std::vector<float> fit_codebooks_multi(vqhip_mdataset *ds, std::uint64_t n, std::uint32_t dim, std::uint32_t m,
                                              std::uint32_t k, std::size_t max_iters, std::uint64_t seed,
                                              const std::uint64_t *init_rows = nullptr) {
    std::vector<HostRng> rngs;
    for (std::uint32_t s = 0; s < m; ++s) rngs.emplace_back(seed + s);
    std::vector<std::uint64_t> init((std::size_t)m * k);
    if (init_rows) {
        std::memcpy(init.data(), init_rows, init.size() * 8);
    } else {
        for (std::uint32_t s = 0; s < m; ++s) {
            const auto rows = rngs[s].choose_multiple(n, k);
            std::memcpy(init.data() + (std::size_t)s * k, rows.data(), (std::size_t)k * 8);
        }
    }
    vqhip_mkmeans *raw = nullptr;
    check(vqhip_mkmeans_create(ds, m, k, &raw));
    std::unique_ptr<vqhip_mkmeans, MKMeansDel> km(raw);
    check(vqhip_mkmeans_init_from_rows(km.get(), init.data()));
    std::vector<std::uint8_t> active(m, 1), changed(m, 0);
    std::vector<std::uint32_t> counts((std::size_t)m * k), iters(m);
    std::size_t done = 0;
    for (;;) {
        bool any = false;
        for (auto a : active) any = any || a;
        if (!any || done >= max_iters) break;
        int paused = 0;
        check(vqhip_mkmeans_run(km.get(), (std::uint32_t)(max_iters - done), iters.data(), counts.data(), changed.data(), &paused));
        std::uint32_t most = 1;
        for (auto x : iters) most = x > most ? x : most;
        done += most;
        check(vqhip_mkmeans_get_active(km.get(), active.data()));  // subspaces that converged inside the call are retired
        if (!paused) continue;
        for (std::uint32_t s = 0; s < m; ++s) {
            if (!active[s]) continue;
            for (std::uint32_t j = 0; j < k; ++j)  // empty clusters in ascending j, vector.rs:448-452
                if (counts[(std::size_t)s * k + j] == 0)
                    check(vqhip_mkmeans_patch_from_row(km.get(), s, j, rngs[s].choose(n)));
            if (!changed[s]) active[s] = 0;  // vector.rs:455-457
        }
        check(vqhip_mkmeans_set_active(km.get(), active.data()));
    }
    std::vector<float> cb((std::size_t)m * k * (dim / m));
    check(vqhip_mkmeans_get_centroids(km.get(), cb.data()));
    return cb;
}

}  // namespace detail

// ------------------------------------------------------------------ ProductQuantizer ----
class ProductQuantizer {
   public:
    // ProductQuantizer::new, src/pq.rs:83-141
    ProductQuantizer(const std::vector<std::vector<float>> &training_data, std::size_t m, std::size_t k,
                     std::size_t max_iters, Distance distance, std::uint64_t seed) {
        std::size_t dim = 0;
        const std::vector<float> flat = detail::flatten(training_data, &dim);
        init(flat.data(), training_data.size(), dim, m, k, max_iters, distance, seed);
    }
    // same, training rows already contiguous [n][dim]
    ProductQuantizer(const float *rows, std::size_t n, std::size_t dim, std::size_t m, std::size_t k,
                     std::size_t max_iters, Distance distance, std::uint64_t seed) {
        if (n == 0) throw VqError::EmptyInput();
        init(rows, n, dim, m, k, max_iters, distance, seed);
    }
    // same call, the training batch partitioned over `devices` (ids of visible GPUs; all_devices() names them all):
    // still ONE call in ONE process -- the ranks are worker threads inside libvqhip (not in the reference: src/pq.rs:83-141
    // has no device argument; the Rust shim passes the visible devices itself)
    ProductQuantizer(const float *rows, std::size_t n, std::size_t dim, std::size_t m, std::size_t k,
                     std::size_t max_iters, Distance distance, std::uint64_t seed, const std::vector<int> &devices) {
        if (n == 0) throw VqError::EmptyInput();
        init(rows, n, dim, m, k, max_iters, distance, seed, devices);
    }
    static std::vector<int> all_devices() {
        std::vector<int> v;
        for (int i = 0; i < vqhip_device_count(); ++i) v.push_back(i);
        return v;
    }

    std::size_t num_subspaces() const { return m_; }
    std::size_t sub_dim() const { return sub_dim_; }
    std::size_t dim() const { return dim_; }
    const char *distance_metric() const { return distance_.name(); }
    const std::vector<float> &codebooks() const { return codebooks_; }  // [m][k][sub_dim]
    std::size_t num_centroids() const { return k_; }

    // Quantizer::quantize, src/pq.rs:167-199
    std::vector<f16> quantize(const float *vector, std::size_t len) const {
        if (len != dim_) throw VqError::DimensionMismatch(dim_, len);
        std::vector<f16> out(dim_);
        detail::check(vqhip_pq_encode(enc_.get(), vector, 1, nullptr, reinterpret_cast<std::uint16_t *>(out.data())));
        return out;
    }
    std::vector<f16> quantize(const std::vector<float> &vector) const { return quantize(vector.data(), vector.size()); }
    // Quantizer::dequantize, src/pq.rs:201-209
    std::vector<float> dequantize(const std::vector<f16> &quantized) const {
        if (quantized.size() != dim_) throw VqError::DimensionMismatch(dim_, quantized.size());
        std::vector<float> out(dim_);
        for (std::size_t i = 0; i < dim_; ++i) out[i] = quantized[i].to_f32();
        return out;
    }

    // batch forms (ROADMAP.md:30 "batch quantization" is open upstream): rows [n][dim]
    std::vector<f16> quantize_batch(const float *rows, std::size_t n) const {
        std::vector<f16> out(n * dim_);
        if (n) detail::check(encode_raw(rows, n, nullptr, reinterpret_cast<std::uint16_t *>(out.data())));
        return out;
    }
    std::vector<std::uint8_t> encode(const float *rows, std::size_t n) const {  // best_idx per subspace, k <= 256
        if (k_ > 256) throw VqError::InvalidParameter("k", "one-byte codes need k <= 256: use encode_wide");
        std::vector<std::uint8_t> codes(n * m_);
        if (n) detail::check(encode_raw(rows, n, codes.data(), nullptr));
        return codes;
    }
    // any k: the library's one- or two-byte codes (vqhip.h "code width") widened to 32 bits
    std::vector<std::uint32_t> encode_wide(const float *rows, std::size_t n) const {
        std::vector<std::uint32_t> out(n * m_);
        if (!n) return out;
        if (vqhip_code_bytes((std::uint32_t)k_) == 1) {
            std::vector<std::uint8_t> c(n * m_);
            detail::check(vqhip_pq_encode(enc_.get(), rows, n, c.data(), nullptr));
            for (std::size_t i = 0; i < c.size(); ++i) out[i] = c[i];
        } else {
            std::vector<std::uint16_t> c(n * m_);
            detail::check(vqhip_pq_encode(enc_.get(), rows, n, reinterpret_cast<std::uint8_t *>(c.data()), nullptr));
            for (std::size_t i = 0; i < c.size(); ++i) out[i] = c[i];
        }
        return out;
    }

    // reconstruction from one-byte codes [n][m] -> [n][dim] f32 (un-rounded centroids), and the batch form of dequantize
    // (src/pq.rs:201-209): row blocks over the quantizer's devices when it was trained over several
    std::vector<float> decode(const std::uint8_t *codes, std::size_t n) const {
        if (k_ > 256) throw VqError::InvalidParameter("k", "one-byte codes need k <= 256");
        std::vector<float> out(n * dim_);
        if (!n) return out;
        detail::check(menc_ && n >= 65536 ? vqhip_mpq_decode(menc_.get(), codes, n, out.data())
                                          : vqhip_pq_decode(enc_.get(), codes, n, out.data()));
        return out;
    }
    std::vector<float> dequantize_batch(const f16 *quantized, std::size_t n) const {
        std::vector<float> out(n * dim_);
        if (!n) return out;
        const std::uint16_t *bits = reinterpret_cast<const std::uint16_t *>(quantized);
        detail::check(menc_ && n >= 65536 ? vqhip_mpq_dequantize_f16(menc_.get(), bits, n * dim_, out.data())
                                          : vqhip_dequantize_f16(bits, n * dim_, out.data()));
        return out;
    }

   private:
    // large batches of a quantizer trained over several devices: row blocks over the same devices
    int encode_raw(const float *rows, std::size_t n, std::uint8_t *codes, std::uint16_t *f16_out) const {
        if (menc_ && n >= 65536) return vqhip_mpq_encode(menc_.get(), rows, n, codes, f16_out);
        return vqhip_pq_encode(enc_.get(), rows, n, codes, f16_out);
    }
    void init(const float *rows, std::size_t n, std::size_t dim, std::size_t m, std::size_t k, std::size_t max_iters,
              Distance distance, std::uint64_t seed, const std::vector<int> &devices = {}) {
        if (m == 0) throw VqError::InvalidParameter("m", "must be greater than 0");
        if (dim < m) throw VqError::InvalidParameter("m", "must be at most the data dimension (" + std::to_string(dim) + ")");
        if (dim % m != 0)
            throw VqError::InvalidParameter("m", "dimension (" + std::to_string(dim) + ") must be divisible by m");
        detail::check_lbg_params(n, k);
        if (k > 65536) throw VqError::InvalidParameter("k", "codes are at most two bytes (k <= 65536)");
        m_ = m;
        k_ = k;
        dim_ = dim;
        sub_dim_ = dim / m;
        distance_ = distance;
        if (devices.size() > 1) {
            vqhip_mdataset *raw = nullptr;
            detail::check(vqhip_mdataset_from_host(rows, n, (std::uint32_t)dim, devices.data(), (int)devices.size(), &raw));
            std::unique_ptr<vqhip_mdataset, detail::MDatasetDel> ds(raw);
            codebooks_ = detail::fit_codebooks_multi(ds.get(), n, (std::uint32_t)dim, (std::uint32_t)m, (std::uint32_t)k, max_iters, seed);
            vqhip_mpq_encoder *me = nullptr;
            detail::check(vqhip_mpq_encoder_create(codebooks_.data(), (std::uint32_t)m, (std::uint32_t)k, (std::uint32_t)sub_dim_,
                                                   (int)distance.kind(), devices.data(), (int)devices.size(), &me));
            menc_.reset(me);
        } else {
            if (devices.size() == 1) detail::check(vqhip_set_device(devices[0]));
            vqhip_dataset *raw = nullptr;
            detail::check(vqhip_dataset_from_host(rows, n, (std::uint32_t)dim, &raw));
            std::unique_ptr<vqhip_dataset, detail::DatasetDel> ds(raw);
            codebooks_ = detail::fit_codebooks(ds.get(), n, (std::uint32_t)dim, (std::uint32_t)m, (std::uint32_t)k,
                                               max_iters, seed);
        }
        vqhip_pq_encoder *e = nullptr;
        detail::check(vqhip_pq_encoder_create(codebooks_.data(), (std::uint32_t)m, (std::uint32_t)k,
                                              (std::uint32_t)sub_dim_, (int)distance.kind(), &e));
        enc_.reset(e);
    }
    std::size_t m_ = 0, k_ = 0, dim_ = 0, sub_dim_ = 0;
    Distance distance_;
    std::vector<float> codebooks_;
    std::unique_ptr<vqhip_pq_encoder, detail::EncoderDel> enc_;
    std::unique_ptr<vqhip_mpq_encoder, detail::MEncoderDel> menc_;
};

// ------------------------------------------------------------------------------ TSVQ ----
class TSVQ {
   public:
    // TSVQ::new, src/tsvq.rs:195-223
    TSVQ(const std::vector<std::vector<float>> &training_data, std::size_t max_depth, Distance distance) {
        std::size_t dim = 0;
        const std::vector<float> flat = detail::flatten(training_data, &dim);
        init(flat.data(), training_data.size(), dim, max_depth, distance);
    }
    TSVQ(const float *rows, std::size_t n, std::size_t dim, std::size_t max_depth, Distance distance) {
        if (n == 0) throw VqError::EmptyInput();
        init(rows, n, dim, max_depth, distance);
    }
    // same, batch encodes in row blocks over `devices` (the build runs on devices[0]: SURVEY.md 8(e), "replicas only";
    // the tree is replicated, each device descends its own rows)
    TSVQ(const float *rows, std::size_t n, std::size_t dim, std::size_t max_depth, Distance distance, const std::vector<int> &devices) {
        if (n == 0) throw VqError::EmptyInput();
        if (!devices.empty()) detail::check(vqhip_set_device(devices[0]));
        init(rows, n, dim, max_depth, distance);
        if (devices.size() > 1) {
            vqhip_mtsvq *mt = nullptr;
            detail::check(vqhip_mtsvq_create(centroids_.data(), left_.data(), right_.data(), (std::uint32_t)left_.size(), (std::uint32_t)dim,
                                             (int)distance.kind(), devices.data(), (int)devices.size(), &mt));
            menc_.reset(mt);
        }
    }
    std::size_t dim() const { return dim_; }
    const char *distance_metric() const { return distance_.name(); }
    std::size_t num_nodes() const { return left_.size(); }
    const std::vector<float> &centroids() const { return centroids_; }  // [nodes][dim], pre-order
    const std::vector<std::int32_t> &left() const { return left_; }
    const std::vector<std::int32_t> &right() const { return right_; }

    // Quantizer::quantize, src/tsvq.rs:239-255
    std::vector<f16> quantize(const float *vector, std::size_t len) const {
        if (len != dim_) throw VqError::DimensionMismatch(dim_, len);
        std::vector<f16> out(dim_);
        detail::check(vqhip_tsvq_encode(enc_.get(), vector, 1, nullptr, reinterpret_cast<std::uint16_t *>(out.data())));
        return out;
    }
    std::vector<f16> quantize(const std::vector<float> &vector) const { return quantize(vector.data(), vector.size()); }
    // src/tsvq.rs:257-265
    std::vector<float> dequantize(const std::vector<f16> &quantized) const {
        if (quantized.size() != dim_) throw VqError::DimensionMismatch(dim_, quantized.size());
        std::vector<float> out(dim_);
        for (std::size_t i = 0; i < dim_; ++i) out[i] = quantized[i].to_f32();
        return out;
    }
    std::vector<std::int32_t> leaf_ids(const float *rows, std::size_t n) const {
        std::vector<std::int32_t> leaf(n);
        if (n) detail::check(encode_raw(rows, n, leaf.data(), nullptr));
        return leaf;
    }
    // batch forms: rows [n][dim] -> the leaf centroids as f16 (row i == quantize(rows[i])), and f16 -> f32
    std::vector<f16> quantize_batch(const float *rows, std::size_t n) const {
        std::vector<f16> out(n * dim_);
        if (n) detail::check(encode_raw(rows, n, nullptr, reinterpret_cast<std::uint16_t *>(out.data())));
        return out;
    }
    std::vector<float> dequantize_batch(const f16 *quantized, std::size_t n) const {
        std::vector<float> out(n * dim_);
        if (!n) return out;
        const std::uint16_t *bits = reinterpret_cast<const std::uint16_t *>(quantized);
        detail::check(menc_ && n >= 65536 ? vqhip_mtsvq_dequantize_f16(menc_.get(), bits, n * dim_, out.data())
                                          : vqhip_dequantize_f16(bits, n * dim_, out.data()));
        return out;
    }

   private:
    int encode_raw(const float *rows, std::size_t n, std::int32_t *leaf, std::uint16_t *f16_out) const {
        if (menc_ && n >= 65536) return vqhip_mtsvq_encode(menc_.get(), rows, n, leaf, f16_out);
        return vqhip_tsvq_encode(enc_.get(), rows, n, leaf, f16_out);
    }
    void init(const float *rows, std::size_t n, std::size_t dim, std::size_t max_depth, Distance distance) {
        dim_ = dim;
        distance_ = distance;
        vqhip_dataset *raw = nullptr;
        detail::check(vqhip_dataset_from_host(rows, n, (std::uint32_t)dim, &raw));
        std::unique_ptr<vqhip_dataset, detail::DatasetDel> ds(raw);
        const std::uint64_t by_rows = 2 * (std::uint64_t)n - 1;
        std::uint64_t cap = by_rows;
        if (max_depth < 40 && ((1ull << (max_depth + 1)) - 1) < cap) cap = (1ull << (max_depth + 1)) - 1;
        centroids_.assign((std::size_t)cap * dim, 0.0f);
        left_.assign(cap, -1);
        right_.assign(cap, -1);
        std::int32_t nodes = 0;
        detail::check(vqhip_tsvq_build(ds.get(), (std::uint32_t)max_depth, (std::uint32_t)cap, centroids_.data(),
                                       left_.data(), right_.data(), &nodes));
        centroids_.resize((std::size_t)nodes * dim);
        left_.resize(nodes);
        right_.resize(nodes);
        vqhip_tsvq *t = nullptr;
        detail::check(vqhip_tsvq_create(centroids_.data(), left_.data(), right_.data(), (std::uint32_t)nodes,
                                        (std::uint32_t)dim, (int)distance.kind(), &t));
        enc_.reset(t);
    }
    std::size_t dim_ = 0;
    Distance distance_;
    std::vector<float> centroids_;
    std::vector<std::int32_t> left_, right_;
    std::unique_ptr<vqhip_tsvq, detail::TsvqDel> enc_;
    std::unique_ptr<vqhip_mtsvq, detail::MTsvqDel> menc_;
};

// ---------------------------------------------------------------------- lbg_quantize ----
// src/core/vector.rs:390-461: k centroids of `data` (n vectors of equal length)
inline std::vector<std::vector<float>> lbg_quantize(const std::vector<std::vector<float>> &data, std::size_t k,
                                                    std::size_t max_iters, std::uint64_t seed) {
    if (data.empty()) throw VqError::EmptyInput();  // vector.rs:396-398
    if (k == 0) throw VqError::InvalidParameter("k", "must be greater than 0");
    if (data.size() < k)
        throw VqError::InvalidParameter("k", "not enough data points (" + std::to_string(data.size()) + ") for " +
                                                 std::to_string(k) + " clusters");
    if (k > 65536) throw VqError::InvalidParameter("k", "codes are at most two bytes (k <= 65536)");
    std::size_t dim = 0;
    const std::vector<float> flat = detail::flatten(data, &dim);
    vqhip_dataset *raw = nullptr;
    detail::check(vqhip_dataset_from_host(flat.data(), data.size(), (std::uint32_t)dim, &raw));
    std::unique_ptr<vqhip_dataset, detail::DatasetDel> ds(raw);
    const std::vector<float> cb =
        detail::fit_codebooks(ds.get(), data.size(), (std::uint32_t)dim, 1, (std::uint32_t)k, max_iters, seed);
    std::vector<std::vector<float>> out(k, std::vector<float>(dim));
    for (std::size_t j = 0; j < k; ++j) std::memcpy(out[j].data(), cb.data() + j * dim, dim * sizeof(float));
    return out;
}

// analogue of vq::get_simd_backend (src/lib.rs): names the device backend
inline std::string get_simd_backend() { return vqhip_backend(); }

}  // namespace vq
#endif  // VQ_HPP
