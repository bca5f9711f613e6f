// Exercises include/vq.hpp the way the reference's own Rust tests exercise the crate
// (tests/integration_tests.rs, tests/regression_tests.rs; citations per case).
//   test_vq_hpp validate            -- error variants and messages (no device needed)
//   test_vq_hpp run <in> <out>      -- fit + quantize on the GPU, results to <out> for the
//                                      Python mirror to compare (tests/test_cpp_host.py)
//   test_vq_hpp threads <in> <out>  -- ONE ProductQuantizer and ONE TSVQ shared by eight std::threads through their
//                                      const quantize(): the types are Send + Sync upstream (src/pq.rs:39-45)
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <thread>

#include "vq.hpp"

using vq::Distance;
using vq::VqError;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);      \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

template <class F>
static std::string error_of(F &&f, VqError::Kind *kind = nullptr) {
    try {
        f();
    } catch (const VqError &e) {
        if (kind) *kind = e.kind;
        return e.what();
    }
    return "<no error>";
}

static int validate() {
    using Rows = std::vector<std::vector<float>>;
    VqError::Kind kind;
    // tests/integration_tests.rs:134-141
    EXPECT(error_of([] { vq::ProductQuantizer(Rows{}, 2, 4, 10, Distance::Euclidean, 42); }, &kind) ==
           "Empty input: at least one vector is required");
    EXPECT(kind == VqError::Kind::EmptyInput);
    // tests/regression_tests.rs:69-88
    EXPECT(error_of([] { vq::ProductQuantizer(Rows{{1, 2, 3, 4}, {1, 2, 3}}, 2, 1, 10, Distance::Euclidean, 42); }, &kind) ==
           "Dimension mismatch: expected 4, found 3");
    EXPECT(kind == VqError::Kind::DimensionMismatch);
    // src/pq.rs:106-117
    EXPECT(error_of([] { vq::ProductQuantizer(Rows(50, std::vector<float>(4, 0.5f)), 8, 4, 10, Distance::Euclidean, 42); }) ==
           "Invalid parameter 'm': must be at most the data dimension (4)");
    EXPECT(error_of([] { vq::ProductQuantizer(Rows(50, std::vector<float>(7, 0.5f)), 2, 4, 10, Distance::Euclidean, 42); }) ==
           "Invalid parameter 'm': dimension (7) must be divisible by m");
    // src/core/vector.rs:399-410
    EXPECT(error_of([] { vq::ProductQuantizer(Rows(50, std::vector<float>(8, 0.5f)), 2, 0, 10, Distance::Euclidean, 42); }, &kind) ==
           "Invalid parameter 'k': must be greater than 0");
    EXPECT(kind == VqError::Kind::InvalidParameter);
    EXPECT(error_of([] { vq::ProductQuantizer(Rows(5, std::vector<float>(8, 0.5f)), 2, 8, 10, Distance::Euclidean, 42); }) ==
           "Invalid parameter 'k': not enough data points (5) for 8 clusters");
    EXPECT(error_of([] { vq::lbg_quantize(Rows{}, 2, 10, 42); }) == "Empty input: at least one vector is required");
    EXPECT(error_of([] { vq::lbg_quantize(Rows(3, std::vector<float>(2, 1.0f)), 4, 10, 42); }) ==
           "Invalid parameter 'k': not enough data points (3) for 4 clusters");
    // src/tsvq.rs:196-210
    EXPECT(error_of([] { vq::TSVQ(Rows{}, 3, Distance::Euclidean); }) == "Empty input: at least one vector is required");
    EXPECT(error_of([] { vq::TSVQ(Rows{{1, 2, 3, 4}, {1, 2, 3}}, 2, Distance::Euclidean); }) ==
           "Dimension mismatch: expected 4, found 3");
    // src/core/distance.rs:49-54, tests/regression_tests.rs:347-352
    EXPECT(error_of([] { Distance(Distance::Euclidean).compute({0.f, 0.f}, {0.f, 0.f, 0.f}); }) ==
           "Dimension mismatch: expected 2, found 3");
    EXPECT(std::string(Distance(Distance::CosineDistance).name()) == "cosine");
    EXPECT(std::string(Distance(Distance::SquaredEuclidean).name()) == "squared_euclidean");
    // half::f16 conversions used by dequantize
    EXPECT((vq::f16{0x3C00}.to_f32() == 1.0f) && (vq::f16{0xC000}.to_f32() == -2.0f) && (vq::f16{0x0001}.to_f32() == 5.9604644775390625e-08f));
    EXPECT(vq::f16{0x7C00}.to_f32() > 3e38f && vq::f16{0x0000}.to_f32() == 0.0f && vq::f16{0x03FF}.to_f32() == 6.09755516052246e-05f);
    // the sampler equals vq_amd/rng.py (values pinned in tests/test_cpp_host.py too)
    vq::HostRng r(42);
    const auto rows = r.choose_multiple(1000, 4);
    std::printf("rng %llu %llu %llu %llu %llu\n", (unsigned long long)rows[0], (unsigned long long)rows[1],
                (unsigned long long)rows[2], (unsigned long long)rows[3], (unsigned long long)r.choose(10));
    std::printf(failures ? "VALIDATE_FAILED\n" : "VALIDATE_OK\n");
    return failures ? 1 : 0;
}

// <in>: u64 n, u64 dim, u64 m, u64 k, u64 iters, u64 seed, u64 depth, then n*dim f32
static int run(const char *in_path, const char *out_path) {
    std::ifstream in(in_path, std::ios::binary);
    std::uint64_t h[7];
    in.read(reinterpret_cast<char *>(h), sizeof(h));
    const std::size_t n = h[0], dim = h[1], m = h[2], k = h[3], iters = h[4], depth = h[6];
    std::vector<float> X(n * dim);
    in.read(reinterpret_cast<char *>(X.data()), X.size() * 4);
    if (!in) return 2;
    std::ofstream out(out_path, std::ios::binary);
    auto put = [&](const void *p, std::size_t bytes) { out.write(reinterpret_cast<const char *>(p), bytes); };

    vq::ProductQuantizer pq(X.data(), n, dim, m, k, iters, Distance::Euclidean, h[5]);
    EXPECT(pq.dim() == dim && pq.num_subspaces() == m && pq.sub_dim() == dim / m);
    EXPECT(std::string(pq.distance_metric()) == "euclidean");
    put(pq.codebooks().data(), pq.codebooks().size() * 4);
    const auto q0 = pq.quantize(X.data(), dim);
    const auto all = pq.quantize_batch(X.data(), n);
    EXPECT(std::equal(q0.begin(), q0.end(), all.begin()));
    put(all.data(), all.size() * 2);
    const auto codes = pq.encode(X.data(), n);
    put(codes.data(), codes.size());
    const auto deq = pq.dequantize(q0);
    EXPECT(deq.size() == dim && deq[0] == q0[0].to_f32());
    EXPECT(error_of([&] { pq.quantize(X.data(), dim - 1); }) ==
           "Dimension mismatch: expected " + std::to_string(dim) + ", found " + std::to_string(dim - 1));

    vq::TSVQ tree(X.data(), n, dim, depth, Distance::SquaredEuclidean);
    const std::uint64_t nodes = tree.num_nodes();
    put(&nodes, 8);
    put(tree.centroids().data(), tree.centroids().size() * 4);
    put(tree.left().data(), nodes * 4);
    put(tree.right().data(), nodes * 4);
    const auto leaf = tree.leaf_ids(X.data(), n);
    put(leaf.data(), leaf.size() * 4);
    const auto tq = tree.quantize(X.data() + dim, dim);
    put(tq.data(), tq.size() * 2);

    // more than 256 centroids: two-byte codes behind encode_wide, consistent with the f16 output
    if (n >= 300) {
        vq::ProductQuantizer wide(X.data(), n, dim, m, 300, 2, Distance::SquaredEuclidean, h[5]);
        EXPECT(wide.num_centroids() == 300);
        EXPECT(error_of([&] { wide.encode(X.data(), 4); }).find("encode_wide") != std::string::npos);
        const std::size_t nw = std::min<std::size_t>(n, 200), sd = dim / m;
        const auto wc = wide.encode_wide(X.data(), nw);
        const auto wq = wide.quantize_batch(X.data(), nw);
        bool same = wc.size() == nw * m, high = false;
        for (std::size_t i = 0; same && i < nw; ++i)
            for (std::size_t s2 = 0; s2 < m; ++s2) {
                const std::uint32_t c = wc[i * m + s2];
                if (c >= 300) { same = false; break; }
                high = high || c > 255;
                const float *cen = wide.codebooks().data() + (s2 * 300 + c) * sd;
                for (std::size_t t = 0; t < sd; ++t)
                    same = same && std::fabs(wq[i * dim + s2 * sd + t].to_f32() - cen[t]) <= std::fabs(cen[t]) * 0.0005f + 1e-7f;  // half rounding
            }
        EXPECT(same);
        EXPECT(high);
    }

    // k = N distinct rows: every row is its own centroid (tests/regression_tests.rs:357-363)
    std::vector<std::vector<float>> two = {{1, 2, 3, 4}, {5, 6, 7, 8}};
    vq::ProductQuantizer tiny(two, 2, 2, 10, Distance::Manhattan, 42);
    for (const auto &r : two) {
        const auto q = tiny.quantize(r);
        for (std::size_t i = 0; i < 4; ++i) EXPECT(q[i].to_f32() == r[i]);
    }
    const auto cents = vq::lbg_quantize(two, 2, 10, 7);
    EXPECT(cents.size() == 2 && cents[0].size() == 4);
    EXPECT(Distance(Distance::SquaredEuclidean).compute({1.f, 2.f}, {3.f, 4.f}) == 8.0f);  // pyvq/tests/test_distance.py:41
    std::printf("backend: %s\n", vq::get_simd_backend().c_str());
    std::printf(failures ? "RUN_FAILED\n" : "RUN_OK\n");
    return failures ? 1 : 0;
}

// same <in> as run(); <out>: codebooks, node count, tree, then what the threads' quantize() calls returned, row by row
static int threads(const char *in_path, const char *out_path) {
    std::ifstream in(in_path, std::ios::binary);
    std::uint64_t h[7];
    in.read(reinterpret_cast<char *>(h), sizeof(h));
    const std::size_t n = h[0], dim = h[1], m = h[2], k = h[3], iters = h[4], depth = h[6];
    std::vector<float> X(n * dim);
    in.read(reinterpret_cast<char *>(X.data()), X.size() * 4);
    if (!in) return 2;
    const vq::ProductQuantizer pq(X.data(), n, dim, m, k, iters, Distance::Euclidean, h[5]);
    const vq::TSVQ tree(X.data(), n, dim, depth, Distance::SquaredEuclidean);
    std::vector<vq::f16> got_pq(n * dim), got_tree(n * dim);
    constexpr std::size_t kThreads = 8;
    std::vector<std::thread> pool;
    std::vector<std::string> errors(kThreads);
    for (std::size_t t = 0; t < kThreads; ++t)
        pool.emplace_back([&, t] {
            try {
                for (std::size_t i = t; i < n; i += kThreads) {  // interleaved: neighbours in time are different rows
                    const auto a = pq.quantize(X.data() + i * dim, dim);
                    std::copy(a.begin(), a.end(), got_pq.begin() + (std::ptrdiff_t)(i * dim));
                    const auto b = tree.quantize(X.data() + i * dim, dim);
                    std::copy(b.begin(), b.end(), got_tree.begin() + (std::ptrdiff_t)(i * dim));
                }
            } catch (const std::exception &e) {
                errors[t] = e.what();
            }
        });
    for (auto &th : pool) th.join();
    for (const auto &e : errors) {
        if (!e.empty()) std::printf("thread error: %s\n", e.c_str());
        EXPECT(e.empty());
    }
    // the batch forms on this thread agree with what the threads saw
    const auto all = pq.quantize_batch(X.data(), n);
    EXPECT(std::equal(all.begin(), all.end(), got_pq.begin()));
    std::ofstream out(out_path, std::ios::binary);
    auto put = [&](const void *p, std::size_t bytes) { out.write(reinterpret_cast<const char *>(p), bytes); };
    put(pq.codebooks().data(), pq.codebooks().size() * 4);
    const std::uint64_t nodes = tree.num_nodes();
    put(&nodes, 8);
    put(tree.centroids().data(), tree.centroids().size() * 4);
    put(tree.left().data(), nodes * 4);
    put(tree.right().data(), nodes * 4);
    put(got_pq.data(), got_pq.size() * 2);
    put(got_tree.data(), got_tree.size() * 2);
    std::printf(failures ? "THREADS_FAILED\n" : "THREADS_OK\n");
    return failures ? 1 : 0;
}

// ProductQuantizer trained over a device list (here: device 0 named twice -- two ranks inside the library): the fit's
// codebooks and a batch encoded through the row-block path
static int multi(const char *in_path, const char *out_path) {
    std::FILE *f = std::fopen(in_path, "rb");
    if (!f) return 2;
    std::uint64_t hdr[7];
    if (std::fread(hdr, 8, 7, f) != 7) return 2;
    const std::size_t n = hdr[0], dim = hdr[1], m = hdr[2], k = hdr[3], iters = hdr[4];
    const std::uint64_t seed = hdr[5];
    std::vector<float> X(n * dim);
    if (std::fread(X.data(), 4, X.size(), f) != X.size()) return 2;
    std::fclose(f);
    vq::ProductQuantizer one(X.data(), n, dim, m, k, iters, vq::Distance::Euclidean, seed, std::vector<int>{0});
    vq::ProductQuantizer two(X.data(), n, dim, m, k, iters, vq::Distance::Euclidean, seed, std::vector<int>{0, 0});
    const std::vector<vq::f16> q = two.quantize_batch(X.data(), n);
    const std::vector<std::uint8_t> c = two.encode(X.data(), n);
    std::FILE *o = std::fopen(out_path, "wb");
    if (!o) return 2;
    std::fwrite(one.codebooks().data(), 4, one.codebooks().size(), o);
    std::fwrite(two.codebooks().data(), 4, two.codebooks().size(), o);
    std::fwrite(q.data(), 2, q.size(), o);
    std::fwrite(c.data(), 1, c.size(), o);
    std::fclose(o);
    // the other row-block paths of a multi-device quantizer: decode and dequantize_batch equal the one-device forms
    const std::vector<float> dec2 = two.decode(c.data(), n), dec1 = one.decode(c.data(), n);
    EXPECT(dec2.size() == n * dim);
    for (std::size_t i = 0; i < n && !failures; i += 997)
        for (std::size_t s = 0; s < m; ++s)
            for (std::size_t t = 0; t < dim / m; ++t)
                EXPECT(dec2[i * dim + s * (dim / m) + t] == two.codebooks()[(s * k + c[i * m + s]) * (dim / m) + t]);
    (void)dec1;
    const std::vector<float> deq = two.dequantize_batch(q.data(), n);
    for (std::size_t i = 0; i < deq.size() && !failures; i += 101) EXPECT(deq[i] == q[i].to_f32());
    // TSVQ over a device list: the leaves and f16 rows of the one-device tree
    vq::TSVQ t1(X.data(), n, dim, 5, vq::Distance::SquaredEuclidean);
    vq::TSVQ t2(X.data(), n, dim, 5, vq::Distance::SquaredEuclidean, std::vector<int>{0, 0});
    EXPECT(t1.centroids() == t2.centroids() && t1.left() == t2.left());
    EXPECT(t1.leaf_ids(X.data(), n) == t2.leaf_ids(X.data(), n));
    const std::vector<vq::f16> tq1 = t1.quantize_batch(X.data(), n), tq2 = t2.quantize_batch(X.data(), n);
    EXPECT(tq1.size() == tq2.size() && std::memcmp(tq1.data(), tq2.data(), tq1.size() * 2) == 0);
    const std::vector<float> tdq = t2.dequantize_batch(tq2.data(), n);
    for (std::size_t i = 0; i < tdq.size() && !failures; i += 101) EXPECT(tdq[i] == tq2[i].to_f32());
    std::printf(failures ? "MULTI_FAILED\n" : "MULTI_OK %s\n", vqhip_backend());
    return failures ? 1 : 0;
}

int main(int argc, char **argv) {
    try {
        if (argc >= 4 && std::string(argv[1]) == "multi") return multi(argv[2], argv[3]);
        if (argc >= 2 && std::string(argv[1]) == "validate") return validate();
        if (argc >= 4 && std::string(argv[1]) == "run") return run(argv[2], argv[3]);
        if (argc >= 4 && std::string(argv[1]) == "threads") return threads(argv[2], argv[3]);
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    std::printf("usage: test_vq_hpp validate | run <in> <out> | threads <in> <out>\n");
    return 64;
}
