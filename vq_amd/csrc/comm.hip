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

struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
};

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

void comm_info(const Comm *c, int *world, int *rank) {
    if (world) *world = c ? c->world : 1;
    if (rank) *rank = c ? c->rank : 0;
}

int comm_destroy(Comm *c) {
    if (c && c->owned && c->comm) {
        const RcclApi *api;
        if (rccl(&api) == VQHIP_OK) (void)api->CommDestroy(c->comm);
    }
    delete c;
    return VQHIP_OK;
}

// in-place sum over all ranks; a NULL or one-rank communicator without an RCCL handle is the identity
int comm_allreduce_f64(Comm *c, double *buf, size_t count, hipStream_t stream) {
    if (!c || !c->comm) return VQHIP_OK;
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

// bit transport: every word has exactly one non-zero contributor, so the u32 sum is that word
int comm_allreduce_u32(Comm *c, uint32_t *buf, size_t count, hipStream_t stream) {
    if (!c || !c->comm) return VQHIP_OK;
    const RcclApi *api;
    VQ_TRY(rccl(&api));
    VQ_NCCL(api, api->AllReduce(buf, buf, count, ncclUint32, ncclSum, c->comm, stream));
    return VQHIP_OK;
}

}  // namespace vqhip
