// multi.hip -- ONE call, ONE process, several GPUs: the drop-in constructor's multi-device path.
//
// `ProductQuantizer::new(training_data, m, k, max_iters, distance, seed)` (src/pq.rs:83-141) is one call in one process;
// the row-sharded fit of include/vqhip.h ("row-sharded training") wants one rank per GPU.  This layer supplies the ranks
// itself: a worker thread per device slot inside the library, each with its device current, its own stream (the
// library's per-thread stream), its block of the rows as a vqhip_dataset, its vqhip_kmeans and its communicator -- and
// runs the SAME per-rank entry points on all of them at once (vqhip_kmeans_init_from_global_rows, _run_sharded,
// _patch_from_global_row): the same f64-slab all-reduce per Lloyd iteration, the same rank agreement, the same pause /
// retire decisions as a multi-process job.  Encode shards host rows over the devices with no collective.
//   communicator: the in-process fixed-order exchange of comm.hip by default (deterministic; two slots may name the same
//   device, which is how the one-GPU boxes of this pool run the path); VQHIP_MULTI_COMM=rccl: ncclCommInitRank from the
//   worker threads (distinct devices only).
// Never a second process, never an exec: threads only.
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"

using namespace vqhip;

namespace {

// fn(rank) on every worker thread at once.  The first failure's status comes back on the calling thread, its text
// prefixed with the rank and device it happened on.  `on_fail` (optional) runs on the CALLING thread as soon as the first
// worker reports a failure, while the others may still be inside the call: it aborts their communicators, so that a rank
// blocked in a collective with the failed one returns instead of waiting.
class WorkerTeam {
   public:
    explicit WorkerTeam(const std::vector<int> &devices) : w_(devices.size()) {
        for (size_t r = 0; r < w_.size(); ++r) {
            w_[r].reset(new W());
            w_[r]->device = devices[r];
            w_[r]->th = std::thread([this, r] { loop((int)r); });
        }
    }
    ~WorkerTeam() {
        for (auto &w : w_) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                w->quit = true;
            }
            cv_.notify_all();
            w->th.join();
        }
    }
    int world() const { return (int)w_.size(); }
    int device(int r) const { return w_[(size_t)r]->device; }
    int run(const std::function<int(int)> &fn, const std::function<void()> &on_fail = nullptr) {
        std::lock_guard<std::mutex> one(call_mu_);  // one collective call at a time over a team
        std::unique_lock<std::mutex> lk(mu_);
        for (auto &w : w_) w->job = &fn, w->done = false, w->rc = VQHIP_OK, w->err.clear();
        n_done_ = 0, first_fail_ = -1;
        cv_.notify_all();
        bool told = false;
        for (;;) {
            cv_.wait(lk, [&] { return n_done_ == (int)w_.size() || (first_fail_ >= 0 && !told); });
            if (first_fail_ >= 0 && !told) {
                told = true;
                if (on_fail) {
                    lk.unlock();
                    on_fail();
                    lk.lock();
                }
            }
            if (n_done_ == (int)w_.size()) break;
        }
        if (first_fail_ < 0) return VQHIP_OK;
        const W &f = *w_[(size_t)first_fail_];
        int others = 0;
        for (auto &w : w_) others += (w->rc != VQHIP_OK) ? 1 : 0;
        if (w_.size() == 1) return fail(f.rc, "%s", f.err.c_str());
        return fail(f.rc, "rank %d (device %d): %s%s", first_fail_, f.device, f.err.c_str(),
                    others > 1 ? " [more than one rank failed]" : "");
    }

   private:
    struct W {
        int device = 0;
        std::thread th;
        const std::function<int(int)> *job = nullptr;
        bool done = true, quit = false;
        int rc = VQHIP_OK;
        std::string err;
    };
    void loop(int r) {
        W &w = *w_[(size_t)r];
        const int dev_rc = vqhip_set_device(w.device);
        const std::string dev_err = dev_rc == VQHIP_OK ? "" : vqhip_last_error();
        for (;;) {
            const std::function<int(int)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return w.quit || w.job; });
                if (!w.job) return;  // quit
                job = w.job;
            }
            int rc = dev_rc;
            std::string err = dev_err;
            if (rc == VQHIP_OK) {
                rc = (*job)(r);
                if (rc != VQHIP_OK) err = vqhip_last_error();
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                w.rc = rc, w.err = err, w.job = nullptr, w.done = true;
                ++n_done_;
                if (rc != VQHIP_OK && first_fail_ < 0) first_fail_ = r;
            }
            cv_.notify_all();
        }
    }
    std::vector<std::unique_ptr<W>> w_;
    std::mutex call_mu_, mu_;
    std::condition_variable cv_;
    int n_done_ = 0, first_fail_ = -1;
};

int check_devices(const int *devices, int n_devices, std::vector<int> *out) {
    if (n_devices < 1 || n_devices > 16) return fail(VQHIP_ERR_INVALID_INPUT, "n_devices = %d (1..16)", n_devices);
    if (!devices) return fail(VQHIP_ERR_NULL_PTR, "devices is NULL");
    const int have = vqhip_device_count();
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= have) return fail(VQHIP_ERR_INVALID_INPUT, "device %d out of range [0,%d)", devices[i], have);
        out->push_back(devices[i]);
    }
    return VQHIP_OK;
}

// contiguous row block of `rank`: the first n % world ranks get one more (vq_amd/sharded.py: shard_rows)
void shard_rows(uint64_t n, int world, int rank, uint64_t *off, uint64_t *cnt) {
    const uint64_t base = n / (uint64_t)world, rem = n % (uint64_t)world;
    *cnt = base + ((uint64_t)rank < rem ? 1 : 0);
    *off = (uint64_t)rank * base + std::min<uint64_t>((uint64_t)rank, rem);
}

}  // namespace

struct vqhip_mdataset {
    std::unique_ptr<WorkerTeam> team;
    std::vector<vqhip_dataset *> ds;
    std::vector<uint64_t> off, cnt;
    uint64_t n = 0;
    uint32_t d = 0;
};

struct vqhip_mkmeans {
    vqhip_mdataset *mds = nullptr;
    uint32_t m = 0, k = 0;
    std::vector<vqhip_kmeans *> km;
    std::vector<vqhip_comm *> comm;
    vqhip_comm_group *group = nullptr;
    int kind = 0;
};

struct vqhip_mpq_encoder {
    std::unique_ptr<WorkerTeam> team;
    std::vector<vqhip_pq_encoder *> enc;
    uint32_t m = 0, k = 0, d = 0;
};

struct vqhip_mtsvq {
    std::unique_ptr<WorkerTeam> team;
    std::vector<vqhip_tsvq *> t;
    uint32_t d = 0;
};

extern "C" {

static int mdataset_make(uint64_t n, uint32_t d, const int *devices, int n_devices, vqhip_mdataset **out,
                         const std::function<int(int, uint64_t, uint64_t, vqhip_dataset **)> &make) {
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    std::vector<int> devs;
    VQ_TRY(check_devices(devices, n_devices, &devs));
    if (n < (uint64_t)n_devices) return fail(VQHIP_ERR_INVALID_INPUT, "%llu rows over %d devices: every device needs a row", (unsigned long long)n, n_devices);
    std::unique_ptr<vqhip_mdataset> h(new vqhip_mdataset());
    h->team.reset(new WorkerTeam(devs));
    h->n = n, h->d = d;
    h->ds.assign(devs.size(), nullptr);
    h->off.resize(devs.size()), h->cnt.resize(devs.size());
    for (int r = 0; r < n_devices; ++r) shard_rows(n, n_devices, r, &h->off[(size_t)r], &h->cnt[(size_t)r]);
    vqhip_mdataset *p = h.get();
    const int rc = h->team->run([&](int r) { return make(r, p->off[(size_t)r], p->cnt[(size_t)r], &p->ds[(size_t)r]); });
    if (rc != VQHIP_OK) {
        const std::string keep = vqhip_last_error();
        (void)vqhip_mdataset_destroy(h.release());
        return fail(rc, "%s", keep.c_str());
    }
    *out = h.release();
    return VQHIP_OK;
}

int vqhip_mdataset_from_host(const float *rows, uint64_t n, uint32_t d, const int *devices, int n_devices, vqhip_mdataset **out) {
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    return mdataset_make(n, d, devices, n_devices, out, [&](int, uint64_t off, uint64_t cnt, vqhip_dataset **ds) {
        return vqhip_dataset_from_host(rows + off * d, cnt, d, ds);  // the devices' uploads run side by side
    });
}

int vqhip_mdataset_synthetic(uint64_t n, uint32_t d, uint64_t seed, const int *devices, int n_devices, vqhip_mdataset **out) {
    return mdataset_make(n, d, devices, n_devices, out, [&](int, uint64_t off, uint64_t cnt, vqhip_dataset **ds) {
        return vqhip_dataset_synthetic(cnt, d, seed, off, ds);
    });
}

int vqhip_mdataset_info(const vqhip_mdataset *ds, uint64_t *n, uint32_t *d, int *n_devices, uint64_t *rows_per_device) {
    if (!ds) return fail(VQHIP_ERR_NULL_PTR, "dataset is NULL");
    if (n) *n = ds->n;
    if (d) *d = ds->d;
    if (n_devices) *n_devices = ds->team->world();
    if (rows_per_device)
        for (size_t r = 0; r < ds->cnt.size(); ++r) rows_per_device[r] = ds->cnt[r];
    return VQHIP_OK;
}

int vqhip_mdataset_destroy(vqhip_mdataset *ds) {
    if (!ds) return VQHIP_OK;
    if (ds->team)
        (void)ds->team->run([&](int r) {
            const int rc = ds->ds[(size_t)r] ? vqhip_dataset_destroy(ds->ds[(size_t)r]) : VQHIP_OK;
            ds->ds[(size_t)r] = nullptr;
            return rc;
        });
    delete ds;
    return VQHIP_OK;
}

int vqhip_mkmeans_create(vqhip_mdataset *ds, uint32_t m, uint32_t k, vqhip_mkmeans **out) {
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (!ds) return fail(VQHIP_ERR_NULL_PTR, "dataset is NULL");
    const int world = ds->team->world();
    std::unique_ptr<vqhip_mkmeans> h(new vqhip_mkmeans());
    h->mds = ds, h->m = m, h->k = k;
    h->km.assign((size_t)world, nullptr);
    h->comm.assign((size_t)world, nullptr);
    const char *env = getenv("VQHIP_MULTI_COMM");
    const bool want_rccl = env && std::string(env) == "rccl";
    uint8_t uid[VQHIP_COMM_ID_BYTES];
    if (world > 1 && want_rccl) {
        VQ_TRY(vqhip_comm_unique_id(uid));
        h->kind = 1;
    } else if (world > 1) {
        VQ_TRY(vqhip_comm_group_create(world, &h->group));
        h->kind = 2;
    }
    vqhip_mkmeans *p = h.get();
    // (the communicator first: it is a collective over the workers, and a worker that failed before it would leave the
    // others waiting)
    int rc = ds->team->run([&](int r) {
        if (p->kind == 1) return vqhip_comm_create(uid, world, r, &p->comm[(size_t)r]);
        if (p->kind == 2) return vqhip_comm_create_local(p->group, r, &p->comm[(size_t)r]);
        return vqhip_comm_create(nullptr, 1, 0, &p->comm[(size_t)r]);
    });
    if (rc == VQHIP_OK) rc = ds->team->run([&](int r) { return vqhip_kmeans_create(ds->ds[(size_t)r], m, k, &p->km[(size_t)r]); });
    if (rc != VQHIP_OK) {
        const std::string keep = vqhip_last_error();
        (void)vqhip_mkmeans_destroy(h.release());
        return fail(rc, "%s", keep.c_str());
    }
    *out = h.release();
    return VQHIP_OK;
}

int vqhip_mkmeans_destroy(vqhip_mkmeans *km) {
    if (!km) return VQHIP_OK;
    (void)km->mds->team->run([&](int r) {
        (void)vqhip_synchronize();  // this rank's stream: nothing of the exchange is in flight when its buffers go
        if (km->km[(size_t)r]) (void)vqhip_kmeans_destroy(km->km[(size_t)r]);
        km->km[(size_t)r] = nullptr;
        return VQHIP_OK;
    });
    (void)km->mds->team->run([&](int r) {  // (behind every rank's synchronisation: peers read each other's published slabs)
        if (km->comm[(size_t)r]) (void)vqhip_comm_destroy(km->comm[(size_t)r]);
        km->comm[(size_t)r] = nullptr;
        return VQHIP_OK;
    });
    if (km->group) (void)vqhip_comm_group_destroy(km->group);
    delete km;
    return VQHIP_OK;
}

int vqhip_mkmeans_info(vqhip_mkmeans *km, int *world, int *comm_kind) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (world) *world = km->mds->team->world();
    if (comm_kind) *comm_kind = km->kind;
    return VQHIP_OK;
}

// a rank failed while the others may sit in a collective with it: abort every rank's communicator (an in-process
// group is poisoned -- the failing rank has done that itself already --, an RCCL communicator gets ncclCommAbort)
static void mkmeans_abort(vqhip_mkmeans *km) {
    if (const char *silent = getenv("VQHIP_TEST_FAIL_SILENT"); silent && silent[0] == '1') return;  // (tests: the peers must time out)
    for (vqhip_comm *c : km->comm)
        if (c) (void)vqhip_comm_abort(c);
}

#define VQ_M_ALL(km, expr)                                             \
    if (!(km)) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");       \
    return (km)->mds->team->run(                                       \
        [&](int r) {                                                   \
            vqhip_kmeans *h = (km)->km[(size_t)r];                     \
            vqhip_comm *c = (km)->comm[(size_t)r];                     \
            const uint64_t off = (km)->mds->off[(size_t)r];            \
            (void)h, (void)c, (void)off;                               \
            return (expr);                                             \
        },                                                             \
        [&] { if ((km)->mds->team->world() > 1) mkmeans_abort(km); })

int vqhip_mkmeans_set_engine(vqhip_mkmeans *km, int engine) { VQ_M_ALL(km, vqhip_kmeans_set_engine(h, engine)); }
// (the reference's summation order is one sequential chain over all rows: one device slot only)
int vqhip_mkmeans_set_exact_update(vqhip_mkmeans *km, int exact_update) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (exact_update && km->mds->team->world() > 1)
        return fail(VQHIP_ERR_UNSUPPORTED, "exact_update sums rows in one sequential chain: single GPU only");
    VQ_M_ALL(km, vqhip_kmeans_set_exact_update(h, exact_update));
}
int vqhip_mkmeans_set_centroids(vqhip_mkmeans *km, const float *centroids) { VQ_M_ALL(km, vqhip_kmeans_set_centroids(h, centroids)); }
int vqhip_mkmeans_set_active(vqhip_mkmeans *km, const uint8_t *active) { VQ_M_ALL(km, vqhip_kmeans_set_active(h, active)); }
int vqhip_mkmeans_init_from_rows(vqhip_mkmeans *km, const uint64_t *init_rows) {
    VQ_M_ALL(km, vqhip_kmeans_init_from_global_rows(h, c, init_rows, off));
}
int vqhip_mkmeans_patch_from_row(vqhip_mkmeans *km, uint32_t s, uint32_t j, uint64_t row) {
    VQ_M_ALL(km, vqhip_kmeans_patch_from_global_row(h, c, s, j, row, off));
}
// (every rank holds the same centroids and the same active set: rank 0's are handed out)
int vqhip_mkmeans_get_centroids(vqhip_mkmeans *km, float *centroids) { VQ_M_ALL(km, r == 0 ? vqhip_kmeans_get_centroids(h, centroids) : VQHIP_OK); }
int vqhip_mkmeans_get_active(vqhip_mkmeans *km, uint8_t *active) { VQ_M_ALL(km, r == 0 ? vqhip_kmeans_get_active(h, active) : VQHIP_OK); }

int vqhip_mkmeans_run(vqhip_mkmeans *km, uint32_t max_iters, uint32_t *iters_done, uint32_t *counts, uint8_t *changed, int *paused) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    const int world = km->mds->team->world();
    std::vector<int> pz((size_t)world, 0);
    const int rc = km->mds->team->run(
        [&](int r) {
            // counts / changed / iterations are global after the all-reduce: the same on every rank, rank 0 writes the caller's
            return vqhip_kmeans_run_sharded(km->km[(size_t)r], km->comm[(size_t)r], max_iters, r == 0 ? iters_done : nullptr,
                                            r == 0 ? counts : nullptr, r == 0 ? changed : nullptr, &pz[(size_t)r]);
        },
        [&] { if (world > 1) mkmeans_abort(km); });
    if (rc != VQHIP_OK) return rc;
    for (int r = 1; r < world; ++r)
        if (pz[(size_t)r] != pz[0]) return fail(VQHIP_ERR_FAILURE, "ranks disagree on the pause of a run (rank %d: %d, rank 0: %d)", r, pz[(size_t)r], pz[0]);
    if (paused) *paused = pz[0];
    return VQHIP_OK;
}

int vqhip_mpq_encoder_create(const float *codebooks, uint32_t m, uint32_t k, uint32_t sub_dim, int metric, const int *devices,
                             int n_devices, vqhip_mpq_encoder **out) {
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    std::vector<int> devs;
    VQ_TRY(check_devices(devices, n_devices, &devs));
    std::unique_ptr<vqhip_mpq_encoder> h(new vqhip_mpq_encoder());
    h->team.reset(new WorkerTeam(devs));
    h->enc.assign(devs.size(), nullptr);
    h->m = m, h->k = k, h->d = m * sub_dim;
    vqhip_mpq_encoder *p = h.get();
    const int rc = h->team->run([&](int r) { return vqhip_pq_encoder_create(codebooks, m, k, sub_dim, metric, &p->enc[(size_t)r]); });
    if (rc != VQHIP_OK) {
        const std::string keep = vqhip_last_error();
        (void)vqhip_mpq_encoder_destroy(h.release());
        return fail(rc, "%s", keep.c_str());
    }
    *out = h.release();
    return VQHIP_OK;
}

int vqhip_mpq_encoder_set_engine(vqhip_mpq_encoder *enc, int engine) {
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    return enc->team->run([&](int r) { return vqhip_pq_encoder_set_engine(enc->enc[(size_t)r], engine); });
}

// host rows in row blocks over the devices, no collective: each worker runs vqhip_pq_encode on its block (with its own
// transfer lanes for large blocks)
int vqhip_mpq_encode(vqhip_mpq_encoder *enc, const float *rows, uint64_t n, uint8_t *codes, uint16_t *f16_out) {
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    const int world = enc->team->world();
    const size_t cw = vqhip_code_bytes(enc->k);
    return enc->team->run([&](int r) {
        uint64_t off, cnt;
        shard_rows(n, world, r, &off, &cnt);
        if (cnt == 0) return (int)VQHIP_OK;
        return vqhip_pq_encode(enc->enc[(size_t)r], rows + off * enc->d, cnt, codes ? codes + off * enc->m * cw : nullptr,
                               f16_out ? f16_out + off * enc->d : nullptr);
    });
}

// the resident rows of a sharded data set through the encoder, `repeat` passes per device (codes in a scratch buffer of
// the device); codes_host [n][m] (optional): the last pass's codes, each device's block in place
int vqhip_mpq_encode_dataset(vqhip_mpq_encoder *enc, vqhip_mdataset *ds, uint32_t repeat, uint8_t *codes_host) {
    if (!enc || !ds) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (ds->team->world() != enc->team->world()) return fail(VQHIP_ERR_INVALID_INPUT, "encoder and data set name different device lists");
    for (int r = 0; r < ds->team->world(); ++r)
        if (ds->team->device(r) != enc->team->device(r)) return fail(VQHIP_ERR_INVALID_INPUT, "encoder and data set name different device lists");
    if (ds->d != enc->d) return fail(VQHIP_ERR_INVALID_INPUT, "data set dimension %u, encoder %u", ds->d, enc->d);
    const size_t cw = vqhip_code_bytes(enc->k);
    // (the encoder's workers run the passes: the handles are usable from any thread, the device is what must match)
    return enc->team->run([&](int r) {
        const void *rows = nullptr;
        uint64_t cnt = 0;
        VQ_TRY(vqhip_dataset_info(ds->ds[(size_t)r], &cnt, nullptr, &rows));
        DevBuf codes;
        VQ_TRY(codes.alloc((size_t)cnt * enc->m * cw));
        for (uint32_t it = 0; it < std::max(1u, repeat); ++it) VQ_TRY(vqhip_pq_encode_device(enc->enc[(size_t)r], rows, cnt, codes.p, nullptr));
        VQ_TRY(vqhip_synchronize());
        if (codes_host) VQ_HIP(hipMemcpy(codes_host + ds->off[(size_t)r] * enc->m * cw, codes.p, (size_t)cnt * enc->m * cw, hipMemcpyDeviceToHost));
        return (int)VQHIP_OK;
    });
}

// the row block of `rank` among `world` (what the handles above use; vq_amd/sharded.py: shard_rows)
int vqhip_shard_rows(uint64_t n, int world, int rank, uint64_t *offset, uint64_t *count) {
    if (world < 1 || rank < 0 || rank >= world || !offset || !count) return fail(VQHIP_ERR_INVALID_INPUT, "rank %d of %d", rank, world);
    shard_rows(n, world, rank, offset, count);
    return VQHIP_OK;
}

int vqhip_mpq_encoder_destroy(vqhip_mpq_encoder *enc) {
    if (!enc) return VQHIP_OK;
    if (enc->team)
        (void)enc->team->run([&](int r) {
            if (enc->enc[(size_t)r]) (void)vqhip_pq_encoder_destroy(enc->enc[(size_t)r]);
            enc->enc[(size_t)r] = nullptr;
            return VQHIP_OK;
        });
    delete enc;
    return VQHIP_OK;
}

// ---- the other two paths that shard by rows with no collective: TSVQ encode and dequantize / decode ----

// reconstruction from codes (the batch form of Quantizer::dequantize over stored codes, src/pq.rs:201-209): row blocks
int vqhip_mpq_decode(vqhip_mpq_encoder *enc, const uint8_t *codes, uint64_t n, float *out) {
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!codes || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    const int world = enc->team->world();
    const size_t cw = vqhip_code_bytes(enc->k);
    return enc->team->run([&](int r) {
        uint64_t off, cnt;
        shard_rows(n, world, r, &off, &cnt);
        if (cnt == 0) return (int)VQHIP_OK;
        return vqhip_pq_decode(enc->enc[(size_t)r], codes + off * enc->m * cw, cnt, out + off * enc->d);
    });
}

// f16 bits -> f32 over the encoder's devices (element blocks: the conversion is element-wise)
int vqhip_mpq_dequantize_f16(vqhip_mpq_encoder *enc, const uint16_t *f16_in, uint64_t count, float *out) {
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (count == 0) return VQHIP_OK;
    if (!f16_in || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    const int world = enc->team->world();
    return enc->team->run([&](int r) {
        uint64_t off, cnt;
        shard_rows(count, world, r, &off, &cnt);
        return cnt ? vqhip_dequantize_f16(f16_in + off, cnt, out + off) : (int)VQHIP_OK;
    });
}

// TSVQ::quantize for a batch (src/tsvq.rs:239-255) with the host rows in row blocks over the devices: the tree is
// replicated (a depth-8 tree at d = 128 is 260 KB), every device descends its own rows -- no collective
int vqhip_mtsvq_create(const float *centroids, const int32_t *left, const int32_t *right, uint32_t n_nodes, uint32_t d, int metric,
                       const int *devices, int n_devices, vqhip_mtsvq **out) {
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    std::vector<int> devs;
    VQ_TRY(check_devices(devices, n_devices, &devs));
    std::unique_ptr<vqhip_mtsvq> h(new vqhip_mtsvq());
    h->team.reset(new WorkerTeam(devs));
    h->t.assign(devs.size(), nullptr);
    h->d = d;
    vqhip_mtsvq *p = h.get();
    const int rc = h->team->run([&](int r) { return vqhip_tsvq_create(centroids, left, right, n_nodes, d, metric, &p->t[(size_t)r]); });
    if (rc != VQHIP_OK) {
        const std::string keep = vqhip_last_error();
        (void)vqhip_mtsvq_destroy(h.release());
        return fail(rc, "%s", keep.c_str());
    }
    *out = h.release();
    return VQHIP_OK;
}

int vqhip_mtsvq_encode(vqhip_mtsvq *t, const float *rows, uint64_t n, int32_t *leaf, uint16_t *f16_out) {
    if (!t) return fail(VQHIP_ERR_NULL_PTR, "tree is NULL");
    if (n == 0) return VQHIP_OK;
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    const int world = t->team->world();
    return t->team->run([&](int r) {
        uint64_t off, cnt;
        shard_rows(n, world, r, &off, &cnt);
        if (cnt == 0) return (int)VQHIP_OK;
        return vqhip_tsvq_encode(t->t[(size_t)r], rows + off * t->d, cnt, leaf ? leaf + off : nullptr, f16_out ? f16_out + off * t->d : nullptr);
    });
}

// f16 bits -> f32 (TSVQ::dequantize, src/tsvq.rs:257-265, for a batch) over the tree's devices
int vqhip_mtsvq_dequantize_f16(vqhip_mtsvq *t, const uint16_t *f16_in, uint64_t count, float *out) {
    if (!t) return fail(VQHIP_ERR_NULL_PTR, "tree is NULL");
    if (count == 0) return VQHIP_OK;
    if (!f16_in || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    const int world = t->team->world();
    return t->team->run([&](int r) {
        uint64_t off, cnt;
        shard_rows(count, world, r, &off, &cnt);
        return cnt ? vqhip_dequantize_f16(f16_in + off, cnt, out + off) : (int)VQHIP_OK;
    });
}

// (screened descent used on every device?, undecided rows summed over the devices) of the last batch
int vqhip_mtsvq_last_stats(vqhip_mtsvq *t, int *screened, uint64_t *undecided) {
    if (!t) return fail(VQHIP_ERR_NULL_PTR, "tree is NULL");
    const int world = t->team->world();
    std::vector<int> sc((size_t)world, 0);
    std::vector<uint64_t> un((size_t)world, 0);
    VQ_TRY(t->team->run([&](int r) { return vqhip_tsvq_last_stats(t->t[(size_t)r], &sc[(size_t)r], &un[(size_t)r]); }));
    int all = 1;
    uint64_t sum = 0;
    for (int r = 0; r < world; ++r) all = all && sc[(size_t)r], sum += un[(size_t)r];
    if (screened) *screened = all;
    if (undecided) *undecided = sum;
    return VQHIP_OK;
}

int vqhip_mtsvq_destroy(vqhip_mtsvq *t) {
    if (!t) return VQHIP_OK;
    if (t->team)
        (void)t->team->run([&](int r) {
            if (t->t[(size_t)r]) (void)vqhip_tsvq_destroy(t->t[(size_t)r]);
            t->t[(size_t)r] = nullptr;
            return VQHIP_OK;
        });
    delete t;
    return VQHIP_OK;
}

}  // extern "C"
