// comm.hip -- RCCL below the C ABI: the one exchange step of row-sharded k-means.
//
// The reference has no multi-device path; its only parallel loop is the rayon assignment over
// rows (src/core/vector.rs:417-423).  Sharding rows over the GPUs of a node generalises it: every
// rank assigns and sums its own rows, the per-cluster sums/counts (ONE f64 slab [m][k][sd+1],
// 278 KB at C2) are all-reduced over xGMI, and every rank derives the same means and the same
// convergence flags.  This file owns the communicator handle (vqhip_comm) and the three
// collectives the path needs, enqueued on the calling thread's stream -- a Rust (or C) host gets
// sharded training without a Python runtime.
//
// librccl is opened at first use (dlopen, by SONAME so that a process that already carries an
// RCCL -- PyTorch-ROCm bundles one -- shares it); libvqhip itself has no link-time dependency
// on it and single-GPU users never load it.  VQHIP_RCCL_LIB overrides the path.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "kernels.hpp"

namespace vqhip {
namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: only the failure path of a one-process team uses it
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

std::once_flag g_rccl_once;
RcclApi g_rccl;

void load_rccl() {
    const char *env = getenv("VQHIP_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !nm[0]) continue;
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
        g_rccl.error = dlerror();
    }
    if (!g_rccl.handle) return;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(g_rccl.handle, name);
        if (!p) g_rccl.error = std::string("librccl lacks ") + name;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(sym("ncclCommUserRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.CommCount ||
        !g_rccl.CommUserRank || !g_rccl.AllReduce || !g_rccl.GetErrorString) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
}

int rccl(const RcclApi **out) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle)
        return fail(VQHIP_ERR_RUNTIME, "RCCL is not available: %s", g_rccl.error.empty() ? "librccl.so.1 not found" : g_rccl.error.c_str());
    *out = &g_rccl;
    return VQHIP_OK;
}

#define VQ_NCCL(api, expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess)                                                                               \
            return ::vqhip::fail(VQHIP_ERR_RUNTIME, "%s failed: %s", #expr, (api)->GetErrorString(_r));       \
    } while (0)

}  // namespace

// ---- the ranks of ONE process (one host thread per GPU): a direct exchange instead of RCCL ----------------------------
// The drop-in constructor (`ProductQuantizer::new`: one call, one process, src/pq.rs:83-141) shards its rows over the
// node's GPUs with a worker thread per device (multi.hip).  Those ranks share an address space, so the one exchange
// step of a Lloyd iteration needs no communicator library: every rank publishes its slab in a buffer of its own
// device, an event says when it is complete, and every rank's stream -- behind the peers' events -- runs ONE kernel
// that reads all published slabs through peer access (xGMI) and adds them IN RANK ORDER: the same bits on every rank,
// the same bits run to run ("a fixed-rank-order direct exchange makes it reproducible").  Stream-
// ordered like ncclAllReduce: vqhip_kmeans_run_sharded queues ten iterations' exchanges back to back.  What the host
// threads must agree on is only that an event has been RECORDED before a peer waits for it (waiting for an event
// nobody has recorded yet is a no-op): one host barrier per call.  Published buffers alternate by call parity; before
// a rank overwrites the one it used two calls ago it waits for every peer's "read it" event of that call.
// Two ranks may name the same device (how the one-GPU boxes of this pool test the path); RCCL refuses that.
constexpr int kLocalMaxWorld = 16;
// Every host-side wait of the group is BOUNDED (VQHIP_COMM_TIMEOUT_S, default 60 s): a rank that left a collective call
// early -- an allocation or launch failure on its device only -- must not leave its peers asleep for ever.  The rank
// that fails poisons the group (comm_abort, called by the ABI layer on every error exit of a sharded entry point), which
// wakes the peers at once; the timeout is the net under that (a rank that died without saying so).  A poisoned group
// stays poisoned: every later collective on it returns VQHIP_ERR_RUNTIME with the first failure's text; the handles
// still destroy normally.
static double comm_timeout_s() {
    const char *e = getenv("VQHIP_COMM_TIMEOUT_S");
    if (e && e[0]) {
        char *end = nullptr;
        const double v = strtod(e, &end);
        if (end != e && v > 0) return v;
    }
    return 60.0;
}
struct LocalGroup {
    int world = 0;
    double timeout_s = 60.0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool failed = false;
    int failed_rank = -1;  // who poisoned the group first (-1: a timeout)
    std::string why;
    struct R {
        int device = -1;
        bool joined = false;
        void *pub[2] = {nullptr, nullptr};
        size_t cap = 0;
        hipEvent_t e_pub[2] = {nullptr, nullptr}, e_done[2] = {nullptr, nullptr};
        bool have_done[2] = {false, false};
    } r[kLocalMaxWorld];
    int poisoned_locked() const {
        return fail(VQHIP_ERR_RUNTIME, "in-process group of %d ranks is poisoned: %s", world, why.empty() ? "a rank failed" : why.c_str());
    }
    // (also the happens-before edge between the ranks' plain fields above)
    int barrier(int rank, const char *where) {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return poisoned_locked();
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return VQHIP_OK;
        }
        const bool woke = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || failed; });
        if (!woke) {  // nobody came: this rank poisons the group so that late arrivals do not wait in turn
            char buf[256];
            snprintf(buf, sizeof buf, "rank %d waited %.0f s at %s with %d of %d ranks present (VQHIP_COMM_TIMEOUT_S)", rank, timeout_s,
                     where, arrived, world);
            failed = true, failed_rank = -1, why = buf;
            cv.notify_all();
            return poisoned_locked();
        }
        if (gen != g) return VQHIP_OK;  // (the barrier completed before anything failed)
        return poisoned_locked();
    }
    void fail_all(int rank, const char *text) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) {
            char buf[512];
            snprintf(buf, sizeof buf, "rank %d (device %d) failed: %s", rank, rank >= 0 && rank < world ? r[rank].device : -1,
                     text && text[0] ? text : "(no error text)");
            failed = true, failed_rank = rank, why = buf;
        }
        cv.notify_all();
    }
};

struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
    LocalGroup *local = nullptr;  // a rank of an in-process group (not owned)
    uint64_t seq = 0;             // collectives this rank has enqueued on the group
    std::atomic<bool> aborted{false};  // comm_abort_rccl took the ncclComm_t away
};

struct LocalSrcs {
    const void *p[kLocalMaxWorld];
};
template <typename T>
__global__ __launch_bounds__(256) void k_local_sum(LocalSrcs src, int world, T *__restrict__ dst, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        T acc = static_cast<const T *>(src.p[0])[i];
        for (int q = 1; q < world; ++q) acc += static_cast<const T *>(src.p[q])[i];  // rank order: every rank forms the same sum
        dst[i] = acc;
    }
}

int local_group_create(int world, LocalGroup **out) {
    if (world < 1 || world > kLocalMaxWorld) return fail(VQHIP_ERR_INVALID_INPUT, "an in-process group takes 1..%d ranks, not %d", kLocalMaxWorld, world);
    LocalGroup *g = new LocalGroup();
    g->world = world;
    g->timeout_s = comm_timeout_s();
    *out = g;
    return VQHIP_OK;
}

// after every rank's communicator is gone (each rank frees its own buffers and events)
void local_group_destroy(LocalGroup *g) {
    if (!g) return;
    for (int q = 0; q < g->world; ++q) {  // what a rank that failed inside comm_create_local left behind
        LocalGroup::R &r = g->r[q];
        for (int par = 0; par < 2; ++par) {
            if (r.pub[par]) (void)hipFree(r.pub[par]);
            if (r.e_pub[par]) (void)hipEventDestroy(r.e_pub[par]);
            if (r.e_done[par]) (void)hipEventDestroy(r.e_done[par]);
        }
    }
    delete g;
}

static int local_allreduce(Comm *c, void *buf, size_t count, bool f64, hipStream_t stream);

// the pattern rank q publishes in the exchange self-test: exact in f64 and distinct per (rank, word)
static inline double selftest_word(int q, size_t i) { return (double)((q + 1) * 4099 + (int)i * 7 + 1); }
constexpr size_t kSelfTestWords = 512;

// Exchange self-test, the last step of comm_create_local (peer access between two physical devices
// has never executed on this pool -- the first 8-GPU user must not be the test).  Every rank publishes a rank-dependent
// pattern; then (1) this rank reads EVERY peer's published buffer on its own with the kernel the exchange uses and
// compares on the host -- a wrong word names the device pair; (2) one real local_allreduce of the pattern (f64) and one
// of a bit pattern (u32) must give the closed-form sums.  ~10 launches and two small copies per rank, once per group.
static int local_selftest(Comm *c, hipStream_t stream) {
    LocalGroup *g = c->local;
    const int world = g->world, rank = c->rank;
    LocalGroup::R &me = g->r[rank];
    const size_t W = kSelfTestWords;
    double *dev = nullptr;
    std::vector<double> host(W * (size_t)(world + 1));
    auto bail = [&](int rc) {
        const std::string keep = tls().last_error;
        if (dev) (void)hipFree(dev);
        g->fail_all(rank, keep.c_str());
        return fail(rc, "%s", keep.c_str());
    };
    if (hipMalloc(reinterpret_cast<void **>(&dev), W * 8 * (size_t)(world + 1)) != hipSuccess)
        return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d: hipMalloc failed", rank));
    for (size_t i = 0; i < W; ++i) host[i] = selftest_word(rank, i);
    if (const char *bad = getenv("VQHIP_TEST_SELFTEST_CORRUPT"); bad && bad[0] && atoi(bad) == rank) host[3] += 1.0;  // (tests: a rank whose memory reads wrong)
    hipError_t e = hipMemcpyAsync(dev, host.data(), W * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // (host.data() is reused below)
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d (device %d): %s", rank, me.device, hipGetErrorString(e)));
    // (2a) the real exchange, f64; it also leaves every rank's pattern in its pub[par] buffer for (1)
    const int par = (int)(c->seq & 1u);
    {
        const int rc = local_allreduce(c, dev, W, true, stream);
        if (rc != VQHIP_OK) {
            (void)hipFree(dev);
            return rc;  // (local_allreduce has poisoned the group)
        }
    }
    // (1) every peer's published buffer, one at a time
    for (int q = 0; q < world; ++q) {
        LocalSrcs src;
        for (int t = 0; t < kLocalMaxWorld; ++t) src.p[t] = nullptr;
        src.p[0] = g->r[q].pub[par];
        hipLaunchKernelGGL(k_local_sum<double>, dim3(2), dim3(256), 0, stream, src, 1, dev + W * (size_t)(q + 1), W);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), dev, W * 8 * (size_t)(world + 1), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d (device %d): reading the peers' buffers failed: %s", rank, me.device,
                         hipGetErrorString(e)));
    for (int q = 0; q < world; ++q)
        for (size_t i = 0; i < W; ++i)
            if (host[W * (size_t)(q + 1) + i] != selftest_word(q, i))
                return bail(fail(VQHIP_ERR_RUNTIME,
                                 "exchange self-test: device %d (rank %d) read word %zu of device %d's (rank %d) published buffer as %.17g, "
                                 "expected %.17g -- peer access between this device pair does not deliver the data",
                                 me.device, rank, i, g->r[q].device, q, host[W * (size_t)(q + 1) + i], selftest_word(q, i)));
    for (size_t i = 0; i < W; ++i) {
        double want = 0;
        for (int q = 0; q < world; ++q) want += selftest_word(q, i);  // rank order, as the kernel (exact anyway)
        if (host[i] != want)
            return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d (device %d): all-reduced word %zu is %.17g, expected %.17g", rank,
                             me.device, i, host[i], want));
    }
    // (2b) the bit transport (global row ids, init rows): u32, one contributor per word
    std::vector<uint32_t> bits(W);
    for (size_t i = 0; i < W; ++i) bits[i] = ((int)(i % (size_t)world) == rank) ? 0x9e3779b9u * (uint32_t)(i + 1) : 0u;
    uint32_t *dev32 = reinterpret_cast<uint32_t *>(dev);
    e = hipMemcpyAsync(dev32, bits.data(), W * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d: %s", rank, hipGetErrorString(e)));
    {
        const int rc = local_allreduce(c, dev32, W, false, stream);
        if (rc != VQHIP_OK) {
            (void)hipFree(dev);
            return rc;
        }
    }
    e = hipMemcpyAsync(bits.data(), dev32, W * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d: %s", rank, hipGetErrorString(e)));
    for (size_t i = 0; i < W; ++i)
        if (bits[i] != 0x9e3779b9u * (uint32_t)(i + 1))
            return bail(fail(VQHIP_ERR_RUNTIME, "exchange self-test, rank %d (device %d): u32 word %zu is %08x, expected %08x (contributor: rank %d, device %d)",
                             rank, me.device, i, bits[i], 0x9e3779b9u * (uint32_t)(i + 1), (int)(i % (size_t)world), g->r[i % (size_t)world].device));
    // nobody frees the scratch while a peer could still... (peers read pub[], never `dev`): free at once
    (void)hipFree(dev);
    // every rank passed: one more meeting so that a rank's failure above reaches the others' return codes too
    return g->barrier(rank, "the end of the exchange self-test");
}

// collective over the group's ranks, each on its own thread with its device current
int comm_create_local(LocalGroup *g, int rank, Comm **out) {
    if (!g || rank < 0 || rank >= g->world) return fail(VQHIP_ERR_INVALID_INPUT, "rank %d of an in-process group of %d", rank, g ? g->world : 0);
    LocalGroup::R &me = g->r[rank];
    int rc = VQHIP_OK;
    hipError_t e = hipGetDevice(&me.device);
    for (int par = 0; par < 2 && e == hipSuccess; ++par) {
        e = hipEventCreateWithFlags(&me.e_pub[par], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&me.e_done[par], hipEventDisableTiming);
    }
    if (e != hipSuccess) rc = fail(VQHIP_ERR_RUNTIME, "in-process communicator, rank %d: %s", rank, hipGetErrorString(e));
    me.joined = true;
    // (a rank that leaves without its communicator leaves its events and buffers in the group's slot: peers may still
    // name them from their streams; local_group_destroy frees what comm_destroy did not)
    if (rc != VQHIP_OK) {
        g->fail_all(rank, tls().last_error.c_str());  // the peers' barrier below returns at once
        return rc;
    }
    VQ_TRY(g->barrier(rank, "the start of comm_create_local"));
    // every peer's published slab is read from this rank's kernels: peer access to the other devices
    for (int q = 0; q < g->world && rc == VQHIP_OK; ++q) {
        const int dq = g->r[q].device;
        if (dq == me.device) continue;
        int can = 0;
        e = hipDeviceCanAccessPeer(&can, me.device, dq);
        if (e != hipSuccess || !can) {
            rc = fail(VQHIP_ERR_RUNTIME, "device %d (rank %d) cannot access device %d's (rank %d) memory (peer access)", me.device, rank, dq, q);
            break;
        }
        e = hipDeviceEnablePeerAccess(dq, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            rc = fail(VQHIP_ERR_RUNTIME, "hipDeviceEnablePeerAccess(%d) from device %d (rank %d): %s", dq, me.device, rank, hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (rc != VQHIP_OK) {
        g->fail_all(rank, tls().last_error.c_str());
        return rc;
    }
    VQ_TRY(g->barrier(rank, "the peer-access step of comm_create_local"));
    Comm *c = new Comm();
    c->world = g->world;
    c->rank = rank;
    c->local = g;
    static const char *skip = getenv("VQHIP_COMM_SELFTEST");  // =0: skip (A/B of the constructor's cost)
    if (g->world > 1 && !(skip && skip[0] == '0')) {
        hipStream_t st = nullptr;
        rc = current_stream(&st);
        if (rc != VQHIP_OK) g->fail_all(rank, tls().last_error.c_str());
        if (rc == VQHIP_OK) rc = local_selftest(c, st);
        if (rc != VQHIP_OK) {
            const std::string keep = tls().last_error;
            (void)hipStreamSynchronize(st);
            delete c;
            return fail(rc, "%s", keep.c_str());
        }
    }
    *out = c;
    return VQHIP_OK;
}

static int local_allreduce(Comm *c, void *buf, size_t count, bool f64, hipStream_t stream) {
    LocalGroup *g = c->local;
    const int rank = c->rank;
    LocalGroup::R &me = g->r[rank];
    const size_t bytes = count * (f64 ? 8 : 4);
    const int par = (int)(c->seq & 1u);
    ++c->seq;
    auto bail = [&](int rc) {  // this rank's own failure: the peers must not wait for it
        const std::string keep = tls().last_error;
        g->fail_all(rank, keep.c_str());
        return fail(rc, "%s", keep.c_str());
    };
    if (bytes > me.cap) {
        // every rank passes the same sizes in the same order, so all grow at the same call: peers may still be reading the
        // old buffers from their streams -- all streams drained (barrier on either side) before anything is freed
        VQ_TRY(g->barrier(rank, "the all-reduce's buffer growth (1)"));
        if (hipStreamSynchronize(stream) != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce: stream synchronisation failed"));
        VQ_TRY(g->barrier(rank, "the all-reduce's buffer growth (2)"));
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        for (int q = 0; q < 2; ++q) {
            if (me.pub[q]) (void)hipFree(me.pub[q]);
            me.pub[q] = nullptr;
            if (hipMalloc(&me.pub[q], cap) != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce: hipMalloc(%zu) failed", cap));
            me.have_done[q] = false;
        }
        me.cap = cap;
        VQ_TRY(g->barrier(rank, "the all-reduce's buffer growth (3)"));
    }
    hipError_t e = hipSuccess;
    for (int q = 0; q < g->world && e == hipSuccess; ++q)  // the peers' reads of this buffer two calls ago
        if (q != rank && g->r[q].have_done[par]) e = hipStreamWaitEvent(stream, g->r[q].e_done[par], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(me.pub[par], buf, bytes, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(me.e_pub[par], stream);
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce (publish): %s", hipGetErrorString(e)));
    VQ_TRY(g->barrier(rank, "the all-reduce's publish step"));
    LocalSrcs src;
    for (int q = 0; q < kLocalMaxWorld; ++q) src.p[q] = nullptr;
    for (int q = 0; q < g->world && e == hipSuccess; ++q) {
        src.p[q] = g->r[q].pub[par];
        if (q != rank) e = hipStreamWaitEvent(stream, g->r[q].e_pub[par], 0);
    }
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce (wait): %s", hipGetErrorString(e)));
    const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, 1024);
    if (f64) hipLaunchKernelGGL(k_local_sum<double>, dim3(blocks), dim3(256), 0, stream, src, g->world, static_cast<double *>(buf), count);
    else hipLaunchKernelGGL(k_local_sum<uint32_t>, dim3(blocks), dim3(256), 0, stream, src, g->world, static_cast<uint32_t *>(buf), count);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(me.e_done[par], stream);
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce (sum): %s", hipGetErrorString(e)));
    me.have_done[par] = true;  // (read by the peers two calls on: behind the next call's barrier)
    return VQHIP_OK;
}

// A rank's call on a sharded entry point failed OUTSIDE the exchange (allocation, launch, fault injection): poison the
// in-process group so that peers blocked in -- or about to enter -- a barrier return at once instead of waiting for the
// timeout.  An RCCL communicator of another process cannot be reached from here (its peers run into RCCL's own
// watchdog); for the ranks of a one-process team multi.hip aborts every rank's communicator (comm_abort_rccl).
void comm_abort(Comm *c, const char *text) {
    if (c && c->local) c->local->fail_all(c->rank, text);
}

// ncclCommAbort on an owned communicator (failure path of a one-process RCCL team; never run on this pool: RCCL wants
// distinct devices per rank).  The handle is gone afterwards; comm_destroy then only frees the wrapper.
void comm_abort_rccl(Comm *c) {
    if (!c || !c->owned || !c->comm) return;
    const RcclApi *api;
    if (rccl(&api) != VQHIP_OK || !api->CommAbort) return;
    if (c->aborted.exchange(true)) return;
    (void)api->CommAbort(c->comm);  // (c->comm keeps its value: a peer thread may be inside ncclAllReduce with it)
}

int comm_unique_id(uint8_t *id128) {
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    ncclUniqueId id;
    VQ_NCCL(api, api->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return VQHIP_OK;
}

// The same check for an RCCL communicator, once behind ncclCommInitRank: an f64 all-reduce of the rank-dependent pattern
// must give the closed-form sum on every rank (a transport that drops or duplicates a rank's contribution shows here,
// not as slightly wrong centroids ten iterations into a fit).  Collective: every rank runs it or none.
static int rccl_selftest(Comm *c) {
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    hipStream_t st = nullptr;
    VQ_TRY(current_stream(&st));
    const size_t W = kSelfTestWords;
    std::vector<double> host(W);
    for (size_t i = 0; i < W; ++i) host[i] = selftest_word(c->rank, i);
    double *dev = nullptr;
    VQ_HIP(hipMalloc(reinterpret_cast<void **>(&dev), W * 8));
    hipError_t e = hipMemcpyAsync(dev, host.data(), W * 8, hipMemcpyHostToDevice, st);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = api->AllReduce(dev, dev, W, ncclFloat64, ncclSum, c->comm, st);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(host.data(), dev, W * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (r != ncclSuccess) return fail(VQHIP_ERR_RUNTIME, "RCCL self-test, rank %d of %d: ncclAllReduce failed: %s", c->rank, c->world, api->GetErrorString(r));
    if (e != hipSuccess) return fail(VQHIP_ERR_RUNTIME, "RCCL self-test, rank %d of %d: %s", c->rank, c->world, hipGetErrorString(e));
    for (size_t i = 0; i < W; ++i) {
        double want = 0;
        for (int q = 0; q < c->world; ++q) want += selftest_word(q, i);  // integers below 2^53: exact in any order
        if (host[i] != want)
            return fail(VQHIP_ERR_RUNTIME, "RCCL self-test, rank %d of %d: all-reduced word %zu is %.17g, expected %.17g", c->rank, c->world, i,
                        host[i], want);
    }
    return VQHIP_OK;
}

int comm_create(const uint8_t *id128, int world, int rank, Comm **out) {
    if (world < 1 || rank < 0 || rank >= world) return fail(VQHIP_ERR_INVALID_INPUT, "rank %d of %d", rank, world);
    Comm *c = new Comm();
    c->world = world;
    c->rank = rank;
    if (world > 1 || id128) {
        const RcclApi *api;
        int rc = rccl(&api);
        if (rc != VQHIP_OK) {
            delete c;
            return rc;
        }
        ncclUniqueId id;
        memcpy(&id, id128, 128);
        ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            delete c;
            return fail(VQHIP_ERR_RUNTIME, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
        }
        c->owned = true;
        // what vqhip_comm_info reports is what RCCL itself says about the communicator, not the arguments
        int w = 0, rk = -1;
        r = api->CommCount(c->comm, &w);
        if (r == ncclSuccess) r = api->CommUserRank(c->comm, &rk);
        if (r != ncclSuccess || w != world || rk != rank) {
            (void)api->CommDestroy(c->comm);
            delete c;
            return fail(VQHIP_ERR_RUNTIME, "communicator reports rank %d of %d, asked for rank %d of %d (%s)", rk, w, rank, world,
                        r == ncclSuccess ? "mismatch" : api->GetErrorString(r));
        }
        static const char *skip = getenv("VQHIP_COMM_SELFTEST");
        if (!(skip && skip[0] == '0')) {
            const int rc2 = rccl_selftest(c);
            if (rc2 != VQHIP_OK) {
                const std::string keep = tls().last_error;
                (void)api->CommDestroy(c->comm);
                delete c;
                return fail(rc2, "%s", keep.c_str());
            }
        }
    }
    *out = c;
    return VQHIP_OK;
}

int comm_adopt(void *nccl_comm, Comm **out) {
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    Comm *c = new Comm();
    c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    ncclResult_t r = api->CommCount(c->comm, &c->world);
    if (r == ncclSuccess) r = api->CommUserRank(c->comm, &c->rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(VQHIP_ERR_RUNTIME, "adopting the ncclComm_t failed: %s", api->GetErrorString(r));
    }
    *out = c;
    return VQHIP_OK;
}

int comm_kind(const Comm *c) { return !c ? 0 : (c->local ? 2 : (c->comm ? 1 : 0)); }

void comm_info(const Comm *c, int *world, int *rank) {
    if (world) *world = c ? c->world : 1;
    if (rank) *rank = c ? c->rank : 0;
}

int comm_destroy(Comm *c) {
    if (c && c->local) {  // this rank's share of the group; the caller has drained the streams that used it
        LocalGroup::R &me = c->local->r[c->rank];
        for (int q = 0; q < 2; ++q) {
            if (me.pub[q]) (void)hipFree(me.pub[q]);
            if (me.e_pub[q]) (void)hipEventDestroy(me.e_pub[q]);
            if (me.e_done[q]) (void)hipEventDestroy(me.e_done[q]);
            me.pub[q] = nullptr, me.e_pub[q] = me.e_done[q] = nullptr, me.have_done[q] = false;
        }
        me.cap = 0;
    }
    if (c && c->owned && c->comm && !c->aborted) {
        const RcclApi *api;
        if (rccl(&api) == VQHIP_OK) (void)api->CommDestroy(c->comm);
    }
    delete c;
    return VQHIP_OK;
}

// in-place sum over all ranks; a NULL or one-rank communicator without an RCCL handle is the identity
int comm_allreduce_f64(Comm *c, double *buf, size_t count, hipStream_t stream) {
    if (c && c->local) return c->world > 1 ? local_allreduce(c, buf, count, true, stream) : VQHIP_OK;
    if (!c || !c->comm) return VQHIP_OK;
    if (c->aborted) return fail(VQHIP_ERR_RUNTIME, "the RCCL communicator of rank %d was aborted", c->rank);
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

// bit transport: every word has exactly one non-zero contributor, so the u32 sum is that word
int comm_allreduce_u32(Comm *c, uint32_t *buf, size_t count, hipStream_t stream) {
    if (c && c->local) return c->world > 1 ? local_allreduce(c, buf, count, false, stream) : VQHIP_OK;
    if (!c || !c->comm) return VQHIP_OK;
    if (c->aborted) return fail(VQHIP_ERR_RUNTIME, "the RCCL communicator of rank %d was aborted", c->rank);
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclUint32, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

}  // namespace vqhip
