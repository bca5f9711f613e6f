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
#include <condition_variable>
#include <mutex>

#include "kernels.hpp"

namespace vqhip {
namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool failed = false;
    struct R {
        int device = -1;
        bool joined = false;
        void *pub[2] = {nullptr, nullptr};
        size_t cap = 0;
        hipEvent_t e_pub[2] = {nullptr, nullptr}, e_done[2] = {nullptr, nullptr};
        bool have_done[2] = {false, false};
    } r[kLocalMaxWorld];
    int barrier() {  // (also the happens-before edge between the ranks' plain fields above)
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return VQHIP_ERR_RUNTIME;
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g || failed; });
        }
        return failed ? VQHIP_ERR_RUNTIME : VQHIP_OK;
    }
    void fail_all() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
    LocalGroup *local = nullptr;  // a rank of an in-process group (not owned)
    uint64_t seq = 0;             // collectives this rank has enqueued on the group
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
    *out = g;
    return VQHIP_OK;
}

// after every rank's communicator is gone (each rank frees its own buffers and events)
void local_group_destroy(LocalGroup *g) { delete g; }

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
    if (rc != VQHIP_OK) g->fail_all();
    if (g->barrier() != VQHIP_OK) return rc != VQHIP_OK ? rc : fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
    // every peer's published slab is read from this rank's kernels: peer access to the other devices
    for (int q = 0; q < g->world && rc == VQHIP_OK; ++q) {
        const int dq = g->r[q].device;
        if (dq == me.device) continue;
        int can = 0;
        e = hipDeviceCanAccessPeer(&can, me.device, dq);
        if (e != hipSuccess || !can) {
            rc = fail(VQHIP_ERR_RUNTIME, "device %d cannot access device %d's memory (peer access)", me.device, dq);
            break;
        }
        e = hipDeviceEnablePeerAccess(dq, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            rc = fail(VQHIP_ERR_RUNTIME, "hipDeviceEnablePeerAccess(%d) from device %d: %s", dq, me.device, hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (rc != VQHIP_OK) g->fail_all();
    if (g->barrier() != VQHIP_OK) return rc != VQHIP_OK ? rc : fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
    Comm *c = new Comm();
    c->world = g->world;
    c->rank = rank;
    c->local = g;
    *out = c;
    return VQHIP_OK;
}

static int local_allreduce(Comm *c, void *buf, size_t count, bool f64, hipStream_t stream) {
    LocalGroup *g = c->local;
    LocalGroup::R &me = g->r[c->rank];
    const size_t bytes = count * (f64 ? 8 : 4);
    const int par = (int)(c->seq & 1u);
    ++c->seq;
    auto bail = [&](int rc) {
        g->fail_all();
        return rc;
    };
    if (bytes > me.cap) {
        // every rank passes the same sizes in the same order, so all grow at the same call: peers may still be reading the
        // old buffers from their streams -- all streams drained (barrier on either side) before anything is freed
        if (g->barrier() != VQHIP_OK) return fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
        if (hipStreamSynchronize(stream) != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce: stream synchronisation failed"));
        if (g->barrier() != VQHIP_OK) return fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        for (int q = 0; q < 2; ++q) {
            if (me.pub[q]) (void)hipFree(me.pub[q]);
            me.pub[q] = nullptr;
            if (hipMalloc(&me.pub[q], cap) != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce: hipMalloc(%zu) failed", cap));
            me.have_done[q] = false;
        }
        me.cap = cap;
        if (g->barrier() != VQHIP_OK) return fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
    }
    hipError_t e = hipSuccess;
    for (int q = 0; q < g->world && e == hipSuccess; ++q)  // the peers' reads of this buffer two calls ago
        if (q != c->rank && g->r[q].have_done[par]) e = hipStreamWaitEvent(stream, g->r[q].e_done[par], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(me.pub[par], buf, bytes, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(me.e_pub[par], stream);
    if (e != hipSuccess) return bail(fail(VQHIP_ERR_RUNTIME, "in-process all-reduce (publish): %s", hipGetErrorString(e)));
    if (g->barrier() != VQHIP_OK) return fail(VQHIP_ERR_RUNTIME, "another rank of the in-process group failed");
    LocalSrcs src;
    for (int q = 0; q < kLocalMaxWorld; ++q) src.p[q] = nullptr;
    for (int q = 0; q < g->world && e == hipSuccess; ++q) {
        src.p[q] = g->r[q].pub[par];
        if (q != c->rank) e = hipStreamWaitEvent(stream, g->r[q].e_pub[par], 0);
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

int comm_unique_id(uint8_t *id128) {
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    ncclUniqueId id;
    VQ_NCCL(api, api->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
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
    if (c && c->owned && c->comm) {
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
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

// bit transport: every word has exactly one non-zero contributor, so the u32 sum is that word
int comm_allreduce_u32(Comm *c, uint32_t *buf, size_t count, hipStream_t stream) {
    if (c && c->local) return c->world > 1 ? local_allreduce(c, buf, count, false, stream) : VQHIP_OK;
    if (!c || !c->comm) return VQHIP_OK;
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclUint32, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

}  // namespace vqhip
