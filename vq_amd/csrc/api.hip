// api.hip -- extern "C" entry points of libvqhip (see include/vqhip.h for the contract and
// the reference file:line each entry replaces).  Host-side orchestration only; the kernels
// are in k_*.hip.  No CPU compute fallback exists anywhere in this file.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kernels.hpp"

namespace vqhip {

// ---------------------------------------------------------------- thread-local state ----
ThreadState &tls() {
    static thread_local ThreadState st;
    return st;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    tls().last_error = buf;
    return code;
}

// per-device facts, published once per device: std::call_once (threads of different handles may make their first call
// at the same time)
static int g_num_cus[64];
static int g_dev_ok[64];  // 0 unknown (probe failed), 1 gfx950, -1 other
static char g_dev_arch[64][64];
static std::once_flag g_dev_once[64];

int require_gfx950() {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(VQHIP_ERR_NO_DEVICE, "no HIP device available (%s); libvqhip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = 0;
    VQ_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(VQHIP_ERR_NO_DEVICE, "device index %d out of range", dev);
    std::call_once(g_dev_once[dev], [dev]() {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;  // g_dev_ok stays 0: reported below
        g_num_cus[dev] = prop.multiProcessorCount;
        strncpy(g_dev_arch[dev], prop.gcnArchName, sizeof(g_dev_arch[dev]) - 1);
        g_dev_ok[dev] = (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : -1;
    });
    if (g_dev_ok[dev] == 0) return fail(VQHIP_ERR_NO_DEVICE, "could not read the properties of HIP device %d", dev);
    if (g_dev_ok[dev] < 0)
        return fail(VQHIP_ERR_NO_DEVICE, "current HIP device is %s, not gfx950; libvqhip targets MI355X only", g_dev_arch[dev]);
    return VQHIP_OK;
}

int num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    return g_num_cus[dev] > 0 ? g_num_cus[dev] : 256;
}

int current_stream(hipStream_t *out) {
    ThreadState &st = tls();
    if (st.user_stream_set) {
        *out = st.user_stream;
        return VQHIP_OK;
    }
    int dev = 0;
    VQ_HIP(hipGetDevice(&dev));
    if (!st.own_stream || st.own_stream_device != dev) {
        // one lazily created non-blocking stream per thread and device
        hipStream_t s;
        VQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        st.own_stream = s;
        st.own_stream_device = dev;
    }
    *out = st.own_stream;
    return VQHIP_OK;
}

// ------------------------------------------------------------------ codebook state ----
struct CodebookState {
    uint32_t m = 0, k = 0, sd = 0, nt = 0, ks = 0;
    bool mfma_ok = false, bf16_ok = false;
    bool prepared = false;  // false = the centroids changed: every image is stale
    bool prepared_base = false, prepared_x32 = false;
    int metric = VQHIP_SQUARED_EUCLIDEAN;  // what the prepared images are for (cosine differs)
    DevBuf cb, prepA, prepCn, meta, cnsqrt, prepA32, cbc, cen, cn32;
    bool x32_ok = false;
    uint32_t x32_groups = 0;  // > 1: centroid groups (sub_dim 32 / 48), partial verdicts merged per row

    int init(uint32_t m_, uint32_t k_, uint32_t sd_) {
        m = m_;
        k = k_;
        sd = sd_;
        // the update / gather / prepare kernels index a codebook's m*k*(sub_dim+1) elements with 32 bits
        if ((uint64_t)m * k * ((uint64_t)sd + 1) >= (1ull << 32))
            return fail(VQHIP_ERR_UNSUPPORTED, "m*k*(sub_dim+1) = %llu exceeds 2^32-1 codebook elements",
                        (unsigned long long)((uint64_t)m * k * ((uint64_t)sd + 1)));
        mfma_ok = screen_supported(sd, k);
        VQ_TRY(cb.alloc((size_t)m * k * sd * 4));
        VQ_TRY(cnsqrt.alloc((size_t)m * k * 4));
        VQ_TRY(meta.alloc((size_t)m * 4 * 4));
        nt = ks = 0;
        if (mfma_ok) {
            screen_tiling(sd, k, &nt, &ks);
            VQ_TRY(prepA.alloc((size_t)m * nt * ks * 64 * 4));
            VQ_TRY(prepCn.alloc((size_t)m * nt * 16 * 4));
        }
        // bf16 engine (X32 kernels; their own tilings, e.g. sub_dim 24): only on a device whose bf16 MFMA
        // matches the accumulation model the margin is derived from (k_selftest.hip)
        x32_ok = screen_bf16_x32_supported(sd, k);
        if (x32_ok) {
            int trusted = 0;
            VQ_TRY(bf16_mfma_selftest(nullptr, nullptr, &trusted));
            if (!trusted) x32_ok = false;
        }
        bf16_ok = x32_ok;
        x32_groups = 0;
        if (x32_ok) {
            uint32_t per = 0;
            screen_bf16_x32_tiling(sd, k, &per, &x32_groups);
            const size_t tiles = (size_t)per * x32_groups;  // image padded to whole centroid groups
            VQ_TRY(prepA32.alloc((size_t)m * tiles * screen_bf16_x32_mfmas(sd) * 4 * 64 * 4));
            const uint32_t sdp = x32_padded_sd(sd);  // >= sd: the screen kernel's sub_dim (zero padding)
            VQ_TRY(cbc.alloc((size_t)m * k * sdp * 4));
            VQ_TRY(cen.alloc((size_t)m * (sdp + 4) * 4));
            VQ_TRY(cn32.alloc((size_t)m * tiles * 32 * 4));
        }
        prepared = false;
        return VQHIP_OK;
    }
    CodebookView view() const {
        CodebookView v;
        v.m = m;
        v.k = k;
        v.sd = sd;
        v.cb = cb.as<float>();
        v.nt = nt;
        v.ks = ks;
        v.prepA = mfma_ok ? prepA.as<float>() : nullptr;
        v.prepCn = mfma_ok ? prepCn.as<float>() : nullptr;
        v.meta = meta.as<float>();
        v.cnsqrt = cnsqrt.as<float>();
        v.prepA32 = x32_ok ? prepA32.as<uint32_t>() : nullptr;
        v.cen = x32_ok ? cen.as<float>() : nullptr;
        v.cn32 = x32_ok ? cn32.as<float>() : nullptr;
        return v;
    }
    // need_base = false: only the X32 images (squared-L2 on the single-pass bf16 screen reads nothing else: the
    // fp32 engine's image, the centroid norms and `meta` are for the other engines, cosine and the grouped merge)
    int prepare(hipStream_t stream, bool need_base = true) {
        CodebookView v = view();
        if (!prepared) prepared_base = prepared_x32 = false;
        if (need_base && !prepared_base) {
            VQ_TRY(launch_prepare_codebook(v, mfma_ok ? prepA.as<float>() : nullptr,
                                           mfma_ok ? prepCn.as<float>() : nullptr, meta.as<float>(),
                                           cnsqrt.as<float>(), stream));
            prepared_base = true;
        }
        if (x32_ok && !prepared_x32) {
            VQ_TRY(launch_prepare_bf16_x32(v, prepA32.as<uint32_t>(), metric == VQHIP_COSINE ? 1 : 0, cbc.as<float>(),
                                           cen.as<float>(), cn32.as<float>(), stream));
            prepared_x32 = true;
        }
        prepared = true;
        return VQHIP_OK;
    }
};

// --------------------------------------------------------------- assign workspace ----
struct AssignWorkspace {
    DevBuf wl_rows, wl_count, sub_list, sub_pos, wl_seg, part, codes_t;
    static constexpr uint32_t kSegCap = 4096;  // wave-private work-list segments per subspace
    uint32_t *seg_host = nullptr;               // pinned [m][kSegCap][2]
    uint32_t last_n_seg = 0;
    bool last_segmented = false;  // the last screened pass filled list segments (wl_seg), not the per-subspace counters
    uint64_t wl_stride = 0;
    uint32_t wl_m = 0;
    std::vector<uint32_t> sub_host;
    uint32_t *stats_host = nullptr;  // pinned [m]
    uint32_t stats_m = 0;
    bool stats_pending = false;
    int last_engine = 0;
    ~AssignWorkspace() {
        if (stats_host) (void)hipHostFree(stats_host);
        if (seg_host) (void)hipHostFree(seg_host);
    }
    int ensure(uint32_t m, uint64_t n, bool need_wl) {
        if (!sub_list.p || sub_list.bytes < (size_t)m * 4) VQ_TRY(sub_list.alloc((size_t)m * 4));
        if (!sub_pos.p || sub_pos.bytes < (size_t)m * 4) VQ_TRY(sub_pos.alloc((size_t)m * 4));
        if (!wl_count.p || wl_m < m) {
            VQ_TRY(wl_count.alloc((size_t)m * 4));
            if (stats_host) (void)hipHostFree(stats_host);
            stats_host = nullptr;
            VQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&stats_host), (size_t)m * 4));
            memset(stats_host, 0, (size_t)m * 4);
            stats_m = m;
            VQ_TRY(wl_seg.alloc((size_t)m * kSegCap * 8));
            if (seg_host) (void)hipHostFree(seg_host);
            seg_host = nullptr;
            VQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&seg_host), (size_t)m * kSegCap * 8));
        }
        if (need_wl && (wl_stride < n || wl_m < m)) {
            VQ_TRY(wl_rows.alloc((size_t)m * (size_t)n * 4));
            wl_stride = n;
        }
        if (wl_m < m) wl_m = m;
        return VQHIP_OK;
    }
    int set_sub_list(const std::vector<uint32_t> &subs, uint32_t m, hipStream_t stream) {
        if (subs != sub_host) {
            // the previous list may still be read by queued kernels: order on the stream
            std::vector<int32_t> pos(m, -1);  // subspace -> position in the list (the fused update's slab index)
            for (size_t i = 0; i < subs.size(); ++i)
                if (subs[i] < m) pos[subs[i]] = (int32_t)i;
            VQ_HIP(hipMemcpyAsync(sub_list.p, subs.data(), subs.size() * 4, hipMemcpyHostToDevice, stream));
            VQ_HIP(hipMemcpyAsync(sub_pos.p, pos.data(), (size_t)m * 4, hipMemcpyHostToDevice, stream));
            VQ_HIP(hipStreamSynchronize(stream));
            sub_host = subs;
        }
        return VQHIP_OK;
    }
};

// Optional per-call HIP-event timing of the two assignment stages, recorded on the stream
// the kernels are launched on (bench.py derives the roofline figure from it).
struct ProfileState {
    bool on = false;
    std::vector<hipEvent_t> ev;  // triples: start, after screen (or exact), after re-check
    std::vector<int> engines;
};
static thread_local ProfileState g_prof;

static int pick_engine(int requested, const CodebookState &cs, int metric, int *engine) {
    const bool l2_metric = (metric == VQHIP_SQUARED_EUCLIDEAN || metric == VQHIP_EUCLIDEAN);
    // cosine has a screen too (s = -x.c/|c| on the X32 bf16 engine); Manhattan has no contraction form
    const bool cos_ok = (metric == VQHIP_COSINE) && cs.x32_ok && cs.metric == VQHIP_COSINE;
    if (requested == VQHIP_ENGINE_EXACT) {
        *engine = VQHIP_ENGINE_EXACT;
    } else if (requested == VQHIP_ENGINE_MFMA) {
        if (!cs.mfma_ok || !l2_metric)
            return fail(VQHIP_ERR_UNSUPPORTED, "fp32 MFMA engine unavailable for sub_dim=%u k=%u metric=%d",
                        cs.sd, cs.k, metric);
        *engine = VQHIP_ENGINE_MFMA;
    } else if (requested == VQHIP_ENGINE_MFMA_BF16) {
        if (!((cs.bf16_ok && l2_metric) || cos_ok))
            return fail(VQHIP_ERR_UNSUPPORTED, "bf16 MFMA engine unavailable for sub_dim=%u k=%u metric=%d",
                        cs.sd, cs.k, metric);
        *engine = VQHIP_ENGINE_MFMA_BF16;
    } else {
        *engine = cos_ok                     ? VQHIP_ENGINE_MFMA_BF16
                  : !l2_metric               ? VQHIP_ENGINE_EXACT
                  : cs.bf16_ok               ? VQHIP_ENGINE_MFMA_BF16
                  : cs.mfma_ok               ? VQHIP_ENGINE_MFMA
                                             : VQHIP_ENGINE_EXACT;
    }
    return VQHIP_OK;
}

// Fused update of a training step: the screen adds the rows it proves into partial slabs, the rows it
// re-checks are added by a list-driven pass behind the re-check.  in: buffers + capacity in slabs; out: used / chunks.
struct FusedAcc {
    float *sums = nullptr;
    uint32_t *counts = nullptr;
    uint32_t slab_cap = 0;   // slabs of [k][sd] available (one per (chunk, active subspace))
    uint32_t n_patch = 16;   // list-driven chunks behind the screen's
    bool used = false;       // out: the launch sequence took the fused path
    uint32_t chunks = 0;     // out: chunks to reduce (screen + patch)
    const uint8_t *gate_active = nullptr;  // vqhip_kmeans_run: device-side gates handed to the screen
    const uint32_t *gate_halt = nullptr;
};

// assignment of every row of X [n][d] for the listed subspaces -> codes [n][m]
static int run_assign(CodebookState &cs, AssignWorkspace &ws, const float *X, uint64_t n, uint32_t d,
                      int metric, const std::vector<uint32_t> &subs, uint8_t *codes, int engine_req,
                      hipStream_t stream, FusedAcc *fused = nullptr) {
    if (n == 0 || subs.empty()) return VQHIP_OK;
    if (n >= (1ull << 32)) return fail(VQHIP_ERR_UNSUPPORTED, "more than 2^32-1 rows per device");
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0)
        return fail(VQHIP_ERR_INVALID_INPUT, "device row buffer must be 16-byte aligned");
    int engine = 0;
    VQ_TRY(pick_engine(engine_req, cs, metric, &engine));
    // k > 256 on the grouped X32 screen keeps a 16-byte partial verdict per (row, subspace, group): past 4 GiB
    // of them the exact scan is used instead (unless the caller asked for the screen by name)
    if (engine == VQHIP_ENGINE_MFMA_BF16 && engine_req == VQHIP_ENGINE_AUTO && cs.k > 256 &&
        (size_t)cs.m * cs.x32_groups * n * 16 > ((size_t)4 << 30))
        engine = VQHIP_ENGINE_EXACT;
    // small work (C1's encode: 10k rows x 4 subspaces x 16 centroids x 16 dimensions): the exact scan is one launch of a
    // few microseconds, the screen is two (+ its codebook images) -- launch latency is all there is at that size
    static const char *small_exact_env = getenv("VQHIP_SMALL_EXACT");  // =0: never (A/B)
    if (engine_req == VQHIP_ENGINE_AUTO && engine != VQHIP_ENGINE_EXACT && !(fused && fused->sums) &&
        (double)n * cs.m * cs.k * cs.sd <= 32.0e6 && !(small_exact_env && small_exact_env[0] == '0'))
        engine = VQHIP_ENGINE_EXACT;
    const bool x32_only = engine == VQHIP_ENGINE_MFMA_BF16 && cs.x32_groups == 1 && x32_padded_sd(cs.sd) <= 64 &&
                          (metric == VQHIP_SQUARED_EUCLIDEAN || metric == VQHIP_EUCLIDEAN) && cs.metric != VQHIP_COSINE;
    VQ_TRY(cs.prepare(stream, !x32_only));
    const bool screened = (engine == VQHIP_ENGINE_MFMA || engine == VQHIP_ENGINE_MFMA_BF16);
    VQ_TRY(ws.ensure(cs.m, n, screened));
    VQ_TRY(ws.set_sub_list(subs, cs.m, stream));
    AssignArgs a;
    a.X = X;
    a.n = n;
    a.d = d;
    a.metric = metric;
    a.sub_list = ws.sub_list.as<uint32_t>();
    a.n_sub = (uint32_t)subs.size();
    a.codes = codes;
    a.wl_rows = ws.wl_rows.as<uint32_t>();
    a.wl_count = ws.wl_count.as<uint32_t>();
    a.wl_stride = ws.wl_stride;
    a.wl_seg = ws.wl_seg.as<uint32_t>();
    a.wl_seg_cap = AssignWorkspace::kSegCap;
    a.n_seg = 0;
    if (engine == VQHIP_ENGINE_MFMA_BF16 && (cs.x32_groups > 1 || x32_padded_sd(cs.sd) > 64)) {  // (the wide kernel goes through the partial verdicts even with one group)
        VQ_TRY(ws.part.ensure((size_t)cs.m * cs.x32_groups * n * 16));
        a.part = ws.part.p;
    }
    // many subspaces: the single-pass bf16 screen writes its codes subspace-major into a scratch and a transposition
    // forms [n][m] (k_screen_bf16.hip): at m = 96 the byte stores m apart were 1.46 GB of write traffic for 96 MB of codes
    static const char *ct_env = getenv("VQHIP_CODES_TRANSPOSE");  // =0: never, =1: whenever the shape allows (A/B)
    const bool ct_shape = engine == VQHIP_ENGINE_MFMA_BF16 && cs.x32_groups == 1 && x32_padded_sd(cs.sd) <= 64 && cs.k <= 256 &&
                          cs.m % 4 == 0 && (size_t)256 * codes_transpose_pitch(cs.m) + cs.m <= 60 * 1024 && (reinterpret_cast<uintptr_t>(codes) & 3) == 0;
    if (ct_shape && !(ct_env && ct_env[0] == '0') && (cs.m >= 16 || (ct_env && ct_env[0] == '1'))) {
        const uint64_t pitch = (n + 255) / 256 * 256;
        VQ_TRY(ws.codes_t.ensure((size_t)cs.m * pitch));
        a.codes_t = ws.codes_t.as<uint8_t>();
        a.codes_t_pitch = pitch;
    }
    if (fused) fused->used = false;
    if (fused && fused->sums && engine == VQHIP_ENGINE_MFMA_BF16 && metric != VQHIP_COSINE && cs.x32_groups == 1 &&
        screen_bf16_fused_update_supported(cs.sd, cs.k) && fused->slab_cap / a.n_sub > fused->n_patch) {
        a.acc_sums = fused->sums;
        a.acc_counts = fused->counts;
        a.acc_chunk_cap = fused->slab_cap / a.n_sub - fused->n_patch;
        a.gate_active = fused->gate_active;
        a.gate_halt = fused->gate_halt;
    }
    CodebookView v = cs.view();
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (g_prof.on) {
        VQ_HIP(hipEventCreate(&e0));
        VQ_HIP(hipEventCreate(&e1));
        VQ_HIP(hipEventCreate(&e2));
        g_prof.ev.push_back(e0);
        g_prof.ev.push_back(e1);
        g_prof.ev.push_back(e2);
        g_prof.engines.push_back(engine);
    }
    if (screened) {
        // the single-pass bf16 screen fills wave-private list SEGMENTS (wl_seg); only the other screens append to the
        // per-subspace counters, which must start at zero
        const bool segmented = engine == VQHIP_ENGINE_MFMA_BF16 && cs.x32_groups == 1 && x32_padded_sd(cs.sd) <= 64;
        if (!segmented) VQ_HIP(hipMemsetAsync(ws.wl_count.p, 0, (size_t)cs.m * 4, stream));
        if (e0) VQ_HIP(hipEventRecord(e0, stream));
        if (engine == VQHIP_ENGINE_MFMA_BF16) VQ_TRY(launch_assign_screen_bf16(v, a, stream));
        else VQ_TRY(launch_assign_screen(v, a, stream));
        if (e1) VQ_HIP(hipEventRecord(e1, stream));
        VQ_TRY(launch_assign_exact(v, a, true, stream));
        if (a.acc_sums) {  // the re-checked rows, now that their codes are final
            VQ_TRY(launch_accumulate_listed(cs.m, cs.k, cs.sd, X, d, codes, a.sub_list, a.n_sub, a.wl_rows, a.wl_stride, a.wl_seg,
                                            a.n_seg, a.acc_chunks, fused->n_patch, a.acc_sums, a.acc_counts, stream));
            fused->used = true;
            fused->chunks = a.acc_chunks + fused->n_patch;
        }
        if (e2) VQ_HIP(hipEventRecord(e2, stream));
        // the re-check counts stay on the device: vqhip_last_assign_stats fetches them when somebody asks (a copy queued
        // behind every pass was one of five stream operations of a 10k-row encode)
        ws.last_n_seg = a.n_seg;
        ws.last_segmented = segmented;
        ws.stats_pending = true;
    } else {
        if (e0) VQ_HIP(hipEventRecord(e0, stream));
        VQ_TRY(launch_assign_exact(v, a, false, stream));
        if (e1) VQ_HIP(hipEventRecord(e1, stream));
        if (e2) VQ_HIP(hipEventRecord(e2, stream));
        ws.stats_pending = false;
    }
    ws.last_engine = engine;
    ThreadState &st = tls();
    st.last_engine = engine;
    st.last_rechecked = 0;
    return VQHIP_OK;
}

// ----------------------------------------------------------- handles under threads ----
// The reference's quantizers are plain data (src/pq.rs:39-45, src/tsvq.rs:186-191): `Send + Sync`, and
// `quantize(&self)` may run on many threads at once.  Here a handle owns device workspaces and work is queued on the
// CALLING thread's stream, so two things are needed for the same guarantee:
//   * one recursive mutex per handle around every entry point that touches the handle's state (entry points call each
//     other: vqhip_pq_encode -> _device, vqhip_kmeans_run -> _step -> _set_active);
//   * stream order across threads: a call that returns with work still queued leaves a tail (its stream); the next
//     call on ANOTHER stream first waits for that tail through an event.  The event is recorded when the other
//     stream shows up if the tail is one of the library's own per-thread streams (never destroyed); for a caller's
//     stream (vqhip_set_stream) it is recorded when the call leaves, because the caller may destroy that stream later.
struct HandleSync {
    std::recursive_mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t tail = nullptr;  // stream holding queued work of this handle (nullptr: the last call synchronised)
    bool tail_lazy = false;      // the event for `tail` has not been recorded yet (library-owned stream)
    int depth = 0;
    ~HandleSync() {
        if (ev) (void)hipEventDestroy(ev);
    }
    int event() {
        if (!ev) VQ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        return VQHIP_OK;
    }
    // order stream s behind whatever this handle still has queued elsewhere
    int order(hipStream_t s) {
        if (tail && tail != s) {
            VQ_TRY(event());
            if (tail_lazy) VQ_HIP(hipEventRecord(ev, tail));
            VQ_HIP(hipStreamWaitEvent(s, ev, 0));
            tail = nullptr;
        }
        return VQHIP_OK;
    }
    void leave(hipStream_t s, bool synced) {
        if (synced || !s) {
            if (synced) tail = nullptr;
            return;
        }
        tail = s;
        tail_lazy = true;
        if (tls().user_stream_set && tls().user_stream == s) {
            if (event() == VQHIP_OK && hipEventRecord(ev, s) == hipSuccess) tail_lazy = false;
        }
    }
};

// RAII for one entry point: lock, order the calling thread's stream behind the handle's tail, and on the way out
// note whether the call left work queued (the default) or synchronised its stream (`synced()`).
class Entry {
   public:
    explicit Entry(HandleSync &h) : h_(&h) {
        h_->mu.lock();
        ++h_->depth;
    }
    ~Entry() { release(); }
    Entry(const Entry &) = delete;
    Entry &operator=(const Entry &) = delete;
    int stream(hipStream_t *out) {
        VQ_TRY(current_stream(&s_));
        VQ_TRY(h_->order(s_));
        *out = s_;
        return VQHIP_OK;
    }
    void synced() { synced_ = true; }
    // give the handle back before the call ends (the rest of the call touches nothing the handle owns mutably)
    void release() {
        if (!h_) return;
        if (--h_->depth == 0) h_->leave(s_, synced_);
        h_->mu.unlock();
        h_ = nullptr;
    }

   private:
    HandleSync *h_;
    hipStream_t s_ = nullptr;
    bool synced_ = false;
};

// vqhip_last_assign_stats names "the most recent pass of this thread" by workspace; the workspace may belong to a
// handle another thread destroys, so the thread-local note is an id looked up in a registry of live workspaces.
struct WsEntry {
    AssignWorkspace *ws;
    HandleSync *sync;
    int pins = 0;  // vqhip_last_assign_stats queries in flight: the handle's destructor waits for them
};
static std::mutex g_ws_mu;
static std::condition_variable g_ws_cv;
static std::map<uint64_t, WsEntry> g_ws_live;
static std::atomic<uint64_t> g_ws_next{1};
static thread_local uint64_t g_last_ws = 0;  // registry id, 0 = none

static uint64_t ws_register(AssignWorkspace *ws, HandleSync *sync) {
    const uint64_t id = g_ws_next.fetch_add(1);
    std::lock_guard<std::mutex> lk(g_ws_mu);
    g_ws_live[id] = WsEntry{ws, sync, 0};
    return id;
}
static void ws_unregister(uint64_t id) {
    std::unique_lock<std::mutex> lk(g_ws_mu);
    auto it = g_ws_live.find(id);
    if (it == g_ws_live.end()) return;
    g_ws_cv.wait(lk, [&] { return it->second.pins == 0; });  // only THIS handle's destructor waits for a query on it
    g_ws_live.erase(it);
}
// a live workspace pinned for one query: the registry lock is held only to look it up and to let it go, so a query that
// blocks behind a long call on its handle stalls no other handle's constructor / destructor
struct WsPin {
    uint64_t id = 0;
    AssignWorkspace *ws = nullptr;
    HandleSync *sync = nullptr;
    explicit WsPin(uint64_t want) {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws_live.find(want);
        if (it == g_ws_live.end()) return;
        ++it->second.pins;
        id = want;
        ws = it->second.ws;
        sync = it->second.sync;
    }
    ~WsPin() {
        if (!id) return;
        std::lock_guard<std::mutex> lk(g_ws_mu);
        auto it = g_ws_live.find(id);
        if (it != g_ws_live.end() && --it->second.pins == 0) g_ws_cv.notify_all();
    }
    WsPin(const WsPin &) = delete;
    WsPin &operator=(const WsPin &) = delete;
};


}  // namespace vqhip

using namespace vqhip;

// ------------------------------------------------------------------------ handles ----
struct vqhip_dataset {
    DevBuf own;
    const float *X = nullptr;
    uint64_t n = 0;
    uint32_t d = 0;
    mutable TsvqPolicyCache tsvq_policy;  // (library-owned rows only: a borrowed buffer may change between builds)
};

struct vqhip_kmeans {
    HandleSync sync;
    uint64_t ws_id = 0;
    const vqhip_dataset *ds = nullptr;
    CodebookState cs;
    AssignWorkspace ws;
    UpdatePlan plan;
    DevBuf codes, partial_sums, partial_counts, slab, counts, changed, active_dev, rows_tmp, xs_ws, gather_ws;
    DevBuf agree;      // two words the ranks of a sharded run all-reduce before anything is queued
    DevBuf sm_psum, sm_pcnt, sm_tick;  // k_lloyd_small.hip: partials per 64-row range, the two flag sets of a device-driven run
    DevBuf run_state;  // vqhip_kmeans_run: [0] halt flag, [1 .. m] iterations executed per subspace, [m+1], [m+2] k_finalize's own
    std::vector<uint8_t> active;
    bool all_active = true;
    int engine = VQHIP_ENGINE_AUTO;
    int exact_update = 0;
    uint32_t fused_slabs = 0;     // > 0: partial buffers sized for the fused update (screen + list-driven slabs)
    bool sums_by_chains = false;  // sub_dim beyond the LDS update kernels: cluster sums always through launch_exact_sums
    bool accumulated = false;
    uint32_t *counts_host = nullptr;   // pinned [m*k]
    uint32_t *changed_host = nullptr;  // pinned [m]
    // pinned, vqhip_kmeans_run's read-back: {paused, iterations [m]} | small-problem flag sets [6 m] | active set [m bytes].
    // (Read into vectors on the stack, every one of those copies was a staged pageable copy with a wait of its own: 85 us per
    // call, whatever the number of iterations -- a third of a ten-iteration run at 10k rows.)
    uint32_t *run_host = nullptr;
    // launch-bound regime (small n): one Lloyd step = ~14 stream operations, replayed as a hipGraph
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;
    uint64_t graph_key = 0, warm_key = 0;
    bool graph_failed = false;
    void drop_graph() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
    }
    vqhip_kmeans() { ws_id = ws_register(&ws, &sync); }
    ~vqhip_kmeans() {
        ws_unregister(ws_id);
        drop_graph();
        if (counts_host) (void)hipHostFree(counts_host);
        if (changed_host) (void)hipHostFree(changed_host);
        if (run_host) (void)hipHostFree(run_host);
    }
};

struct vqhip_comm {
    std::recursive_mutex mu;  // an ncclComm_t takes one enqueueing thread at a time
    Comm *c = nullptr;
};

// Pinned, device-mapped staging for the per-vector latency path (k_small.hip)
struct PinnedStage {
    void *host = nullptr, *dev = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return VQHIP_OK;
        if (host) (void)hipHostFree(host);
        host = dev = nullptr;
        bytes = 0;
        const size_t cap = need < 65536 ? 65536 : need;
        VQ_HIP(hipHostMalloc(&host, cap, hipHostMallocMapped));
        VQ_HIP(hipHostGetDevicePointer(&dev, host, 0));
        bytes = cap;
        return VQHIP_OK;
    }
    ~PinnedStage() {
        if (host) (void)hipHostFree(host);
    }
};
constexpr uint64_t kSmallRows = 8;  // calls with at most this many host rows take the one-kernel path

// ------------------------------------------------- host <-> device streaming for large batches ----
// What this pool's boxes deliver (profiles/ubench/pcie_duplex.hip, tools/host_xfer.py): a copy between device memory
// and host pages that have been touched runs at 55-57 GB/s each way whether the pages are pinned or not, both ways at
// once if TWO host threads issue them (a pageable copy occupies its thread); staging through pinned buffers with host
// memcpys (29 GB/s per thread) only loses (measured: 4 lanes of pinned staging 6.1e7 vectors/s against 7.1e7 for the
// plain copies in turn, rows in / f16 out at 1M x 128).  So a large host batch whose results are a sizeable share of
// its input goes through LANES: a few host threads (three), each with its own stream and device buffers, take every
// n-th chunk end to end -- H2D straight from the caller's rows, the kernels (the handle orders the lanes' launches on the
// device: HandleSync), D2H straight into the caller's buffers -- so one lane's results travel while the other lane's rows do.
struct XferLane {
    int device = -1;
    hipStream_t stream = nullptr;
    DevBuf dev_in, dev_out, dev_out2;
    int ensure() {
        if (!stream) VQ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return VQHIP_OK;
    }
};
constexpr int kXferLanesMax = 8;
static int xfer_lanes() {  // host threads per large transfer (VQHIP_XFER_LANES, 1..8; A/B)
    static const int v = [] {
        const char *e = getenv("VQHIP_XFER_LANES");
        const int x = e ? atoi(e) : 3;  // (two suffice when they stay out of phase, 9.7 ms at 1M x 128 rows in / f16 out, but now and then a call falls back to 14 ms; three: 10.2-10.5 ms every time)
        return x < 1 ? 1 : (x > kXferLanesMax ? kXferLanesMax : x);
    }();
    return v;
}
static size_t xfer_chunk_bytes() {  // bytes of input per chunk (VQHIP_XFER_CHUNK_MB; A/B)
    static const size_t v = [] {
        const char *e = getenv("VQHIP_XFER_CHUNK_MB");
        const int x = e ? atoi(e) : 32;
        return (size_t)(x < 1 ? 1 : (x > 1024 ? 1024 : x)) << 20;
    }();
    return v;
}
constexpr size_t kXferMinBytes = (size_t)96 << 20;  // smaller batches keep the one-stream path (two host threads to start)
// lanes pay when the results are at least a quarter of the rows in bytes (f16 reconstructions); codes alone are 1-6 %
static bool xfer_lanes_pay(size_t in_b, size_t out_b) {
    static const char *no_lanes = getenv("VQHIP_NO_XFER_LANES");  // =1: one stream, copies and kernels in turn (A/B)
    return in_b >= kXferMinBytes && 4 * out_b >= in_b && !(no_lanes && no_lanes[0] == '1');
}
struct XferPool {  // leaked on purpose, like StagePool
    std::mutex mu;
    std::vector<XferLane *> idle;
};
static XferPool &xfer_pool() {
    static XferPool *p = new XferPool();
    return *p;
}
// fn(lane index, lane, lanes) on that many host threads (each bound to the caller's device); the first failure's status
// and text come back on the calling thread
static std::atomic<uint64_t> g_xfer_lane_calls{0};
template <class Fn>
static int run_lanes(Fn fn) {
    int dev = 0;
    VQ_HIP(hipGetDevice(&dev));
    g_xfer_lane_calls.fetch_add(1);
    const int n_lanes = xfer_lanes();
    XferLane *lanes[kXferLanesMax] = {};
    {
        XferPool &p = xfer_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        int got = 0;
        for (size_t i = 0; i < p.idle.size() && got < n_lanes;)
            if (p.idle[i]->device == dev) {
                lanes[got++] = p.idle[i];
                p.idle.erase(p.idle.begin() + (long)i);
            } else {
                ++i;
            }
        for (; got < n_lanes; ++got) {
            lanes[got] = new XferLane();
            lanes[got]->device = dev;
        }
    }
    int rcs[kXferLanesMax];
    std::string errs[kXferLanesMax];
    std::thread th[kXferLanesMax];
    for (int t = 0; t < n_lanes; ++t)
        th[t] = std::thread([&, t] {
            rcs[t] = VQHIP_OK;
            if (hipSetDevice(dev) != hipSuccess) {
                rcs[t] = VQHIP_ERR_RUNTIME;
                errs[t] = "hipSetDevice failed on a transfer lane";
                return;
            }
            rcs[t] = lanes[t]->ensure();
            if (rcs[t] == VQHIP_OK) {
                ThreadState &st = tls();
                st.user_stream = lanes[t]->stream, st.user_stream_set = true;  // this thread's launches go to the lane's stream
                rcs[t] = fn(t, *lanes[t], n_lanes);
            }
            if (rcs[t] != VQHIP_OK) errs[t] = tls().last_error;
        });
    for (int t = 0; t < n_lanes; ++t) th[t].join();
    {
        XferPool &p = xfer_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        for (int t = 0; t < n_lanes; ++t) p.idle.push_back(lanes[t]);
    }
    for (int t = 0; t < n_lanes; ++t)
        if (rcs[t] != VQHIP_OK) return fail(rcs[t], "%s", errs[t].c_str());
    return VQHIP_OK;
}

// The per-vector path stages through a buffer that belongs to the CALL, not to the handle: `quantize(&self)` from many
// threads on one quantizer then overlaps on the device (each thread its own stream and staging) instead of queueing on
// the handle's lock.  Buffers are pooled per device and never returned to the runtime (64 KB each, as many as there
// were concurrent calls); the pool itself is leaked on purpose -- static destructors run after the HIP runtime's.
struct StagePool {
    std::mutex mu;
    std::vector<std::pair<int, PinnedStage *>> idle;
};
static StagePool &stage_pool() {
    static StagePool *p = new StagePool();
    return *p;
}
class StageLease {
   public:
    StageLease() = default;
    StageLease(const StageLease &) = delete;
    StageLease &operator=(const StageLease &) = delete;
    ~StageLease() {
        if (!st_) return;
        std::lock_guard<std::mutex> lk(stage_pool().mu);
        stage_pool().idle.emplace_back(dev_, st_);
    }
    int acquire(size_t need) {
        VQ_HIP(hipGetDevice(&dev_));
        {
            StagePool &p = stage_pool();
            std::lock_guard<std::mutex> lk(p.mu);
            for (size_t i = 0; i < p.idle.size(); ++i)
                if (p.idle[i].first == dev_) {
                    st_ = p.idle[i].second;
                    p.idle.erase(p.idle.begin() + (long)i);
                    break;
                }
        }
        if (!st_) st_ = new PinnedStage();
        return st_->ensure(need);
    }
    char *host() const { return static_cast<char *>(st_->host); }
    char *dev() const { return static_cast<char *>(st_->dev); }

   private:
    PinnedStage *st_ = nullptr;
    int dev_ = 0;
};

// wait for a ~10-40 us kernel by polling: hipStreamSynchronize parks the thread after a short spin and
// then costs ~100 us to wake up (measured 134 us per TSVQ quantize call against a 14 us kernel)
static int spin_wait(hipStream_t s) {
    for (int i = 0; i < 200000; ++i) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return VQHIP_OK;
        if (e != hipErrorNotReady) return fail(VQHIP_ERR_RUNTIME, "stream query failed: %s", hipGetErrorString(e));
    }
    VQ_HIP(hipStreamSynchronize(s));
    return VQHIP_OK;
}

struct vqhip_pq_encoder {
    HandleSync sync;
    uint64_t ws_id = 0;
    vqhip_pq_encoder() { ws_id = ws_register(&ws, &sync); }
    ~vqhip_pq_encoder() { ws_unregister(ws_id); }
    CodebookState cs;
    AssignWorkspace ws;
    int metric = VQHIP_EUCLIDEAN;
    int engine = VQHIP_ENGINE_AUTO;
    DevBuf xbuf, codes, f16buf, f32buf, adc_q, adc_lut, adc_dist, adc_idx, adc_out, adc_codes, adc_state, adc_cand, adc_redo, adc_resident;
    uint64_t adc_resident_n = 0;   // rows of the code store vqhip_pq_adc_set_codes uploaded
    uint32_t adc_last_redone = 0;  // queries of the last ADC call that went through the full pass
    std::vector<uint32_t> all_subs;
    // the per-vector path's images (codebooks, cosine norms) are complete on the DEVICE -- set only after a host wait behind
    // their prepare kernels; `cs.prepared` alone says they were ENQUEUED, possibly on another thread's stream
    bool small_ready = false;
};

struct vqhip_tsvq {
    HandleSync sync;
    uint32_t n_nodes = 0, d = 0;
    int metric = VQHIP_EUCLIDEAN;
    DevBuf centroids, cnorm, left, right, xbuf, leafbuf, f16buf, table16;
    // screened descent (k_tsvq_screen.hip); use_screen = false -> exact walk only
    bool use_screen = false, last_screened = false;
    TsvqScreen scr;
    DevBuf scr_w, scr_info, scr_mu, scr_wl, scr_count, scr_slot_node, scr_node_slot;
};

// Per-node data of the screened descent, from the host copy of the tree (f64, rounded once).
static int tsvq_prepare_screen(vqhip_tsvq *t, const float *centroids, const int32_t *left, const int32_t *right,
                               hipStream_t s) {
    const uint32_t n_nodes = t->n_nodes, d = t->d;
    uint32_t n_int = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) n_int += (left[i] >= 0 && right[i] >= 0) ? 1u : 0u;
    const char *off = getenv("VQHIP_TSVQ_EXACT");
    if ((off && off[0] == '1') || n_int == 0 || !tsvq_screen_supported(1, n_nodes, d, t->metric)) return VQHIP_OK;
    // A tree too large for one CU's LDS: the two-child nodes are given slots in breadth-first order, the first
    // n_lds of them (the levels nearest the root) live in LDS, the kernel reads the others' w / record from L2.
    uint32_t n_lds = n_int;
    while (n_lds > 1 && !tsvq_screen_supported(n_lds, n_nodes, d, t->metric)) --n_lds;
    const uint32_t dp = tsvq_screen_width(d);  // >= d: width of the kernel's w / mu rows (zeros behind d)
    const bool cosine = t->metric == VQHIP_COSINE, manhattan = t->metric == VQHIP_MANHATTAN;
    const uint32_t nv = (cosine || manhattan) ? 2u : 1u;  // vectors per slot
    std::vector<float> cn32;               // cosine: the reference's f32 centroid norms, as the device computed them
    if (cosine) {
        cn32.resize(n_nodes);
        VQ_HIP(hipMemcpyAsync(cn32.data(), t->cnorm.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipStreamSynchronize(s));
    }
    std::vector<float> w((size_t)n_int * nv * dp, 0.0f);
    std::vector<int32_t> info((size_t)n_int * 4), slot_node(n_int), slot_of(n_nodes, -1);
    const float *mu = centroids;  // root
    double r2max = 0.0;
    bool finite = true;
    std::vector<double> a2(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        double acc = 0.0;
        for (uint32_t q = 0; q < d; ++q) {
            const double a = (double)centroids[(size_t)i * d + q] - (double)mu[q];
            acc += a * a;
        }
        a2[i] = acc;
        if (!(acc <= 1e300)) finite = false;
        if (acc > r2max) r2max = acc;
    }
    uint32_t slot = 0;
    {
        std::vector<int32_t> queue;  // breadth-first: the levels nearest the root get the LDS slots
        queue.reserve(n_nodes);
        queue.push_back(0);
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const int32_t i = queue[qi];
            if (left[i] >= 0 && right[i] >= 0) {
                slot_of[i] = (int32_t)slot;
                slot_node[slot++] = i;
            }
            if (left[i] >= 0) queue.push_back(left[i]);
            if (right[i] >= 0) queue.push_back(right[i]);
        }
    }
    // where a walk entering node i first has to decide (slot >= 0) or ends (leaf: -1-node); one-child
    // nodes just pass the row on (tsvq.rs:124-129)
    auto resolve = [&](int32_t i) {
        for (;;) {
            const int32_t l = left[i], r = right[i];
            if (l >= 0 && r >= 0) return slot_of[i];
            if (l < 0 && r < 0) return -1 - i;
            i = (l >= 0) ? l : r;
        }
    };
    for (uint32_t sl = 0; sl < n_int; ++sl) {
        const int32_t i = slot_node[sl], l = left[i], r = right[i];
        int32_t *rec = info.data() + (size_t)sl * 4;
        if (manhattan) {
            // the children's centroids as they are; both L1 sums are within a relative gamma of the same exact sum of
            // the reference's own terms: m = 2.01 (d + D/32 + 6) u (+1 % for the threshold's own arithmetic)
            for (uint32_t side = 0; side < 2; ++side)
                memcpy(&w[((size_t)sl * 2 + side) * dp], centroids + (size_t)(side == 0 ? l : r) * d, (size_t)d * 4);
            const float mf = (float)(2.01 * ((double)d + (double)dp / 32.0 + 6.0) * 5.9604644775390625e-08 * 1.01);
            rec[0] = resolve(l);
            rec[1] = resolve(r);
            memcpy(&rec[2], &mf, 4);
            rec[3] = 0;
            continue;
        }
        if (cosine) {
            // per child: c / nb with nb the REFERENCE's f32 norm of the centroid (cnorm, as the exact walk uses it; f64
            // quotient rounded once), so the screen and the reference divide by the same number.  The slot's margin
            // M = 2.0002 e_p + a_l + a_r + u  (DESIGN.md 4.4 "cosine descent"): e_p = (D/32 + 6.1) u for the screen's
            // summation tree, a = (2.02 + 1.0003 |w o c| / nb) u for the reference's sequential dot product, w_i the
            // number of roundings term i goes through.  A child whose norm the EPSILON rule or an overflowing squared
            // norm could touch, or that is not finite, makes the slot exact-only (NaN margin).
            bool usable = true;
            double a_sum = 0.0;
            for (uint32_t side = 0; side < 2; ++side) {
                const int32_t ch = side == 0 ? l : r;
                const float *c = centroids + (size_t)ch * d;
                const double nb = (double)cn32[ch];
                double c2 = 0.0, wc2 = 0.0;
                for (uint32_t q = 0; q < d; ++q) {
                    const double wq = (q == 0) ? (double)d : (double)(d - q + 1);  // additions behind term q, plus its product
                    c2 += (double)c[q] * (double)c[q];
                    wc2 += wq * wq * (double)c[q] * (double)c[q];
                }
                const double cn = std::sqrt(c2);
                if (!(cn >= 1e-9 && cn <= 1e18) || !(nb >= 1e-9 && nb <= 1e18)) {
                    usable = false;
                    continue;
                }
                for (uint32_t q = 0; q < d; ++q) w[((size_t)sl * 2 + side) * dp + q] = (float)((double)c[q] / nb);
                a_sum += 2.02 + 1.0003 * std::sqrt(wc2) / nb;
            }
            const double e_p = (double)dp / 32.0 + 6.1;
            const double margin = (2.0002 * e_p + a_sum + 1.0) * 5.9604644775390625e-08 * 1.01;  // 1 %: the threshold's own f32 arithmetic
            const float mf = usable ? (float)margin : std::numeric_limits<float>::quiet_NaN();
            rec[0] = resolve(l);
            rec[1] = resolve(r);
            memcpy(&rec[2], &mf, 4);
            rec[3] = 0;
            continue;
        }
        double w2 = 0.0;
        for (uint32_t q = 0; q < d; ++q) {
            const float wv = centroids[(size_t)l * d + q] - centroids[(size_t)r * d + q];
            w[(size_t)sl * dp + q] = wv;
            w2 += (double)wv * (double)wv;
        }
        const float b = (float)(a2[l] - a2[r]);
        const float wn = (float)(std::sqrt(w2) * 1.000001);
        if (!(w2 <= 1e300)) finite = false;
        rec[0] = resolve(l);
        rec[1] = resolve(r);
        memcpy(&rec[2], &b, 4);
        memcpy(&rec[3], &wn, 4);
    }
    const int32_t start = resolve(0);
    if (start < 0) return VQHIP_OK;  // the root leads to one leaf without any decision: exact walk
    t->scr.start_slot = start;
    VQ_TRY(t->scr_slot_node.alloc((size_t)n_int * 4));
    VQ_HIP(hipMemcpyAsync(t->scr_slot_node.p, slot_node.data(), (size_t)n_int * 4, hipMemcpyHostToDevice, s));
    t->scr.slot_node = t->scr_slot_node.as<int32_t>();
    VQ_TRY(t->scr_node_slot.alloc((size_t)n_nodes * 4));
    VQ_HIP(hipMemcpyAsync(t->scr_node_slot.p, slot_of.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
    t->scr.node_slot = t->scr_node_slot.as<int32_t>();
    VQ_TRY(t->scr_w.alloc(w.size() * 4));
    VQ_TRY(t->scr_info.alloc(info.size() * 4));
    VQ_TRY(t->scr_mu.alloc((size_t)dp * 4));
    VQ_HIP(hipMemsetAsync(t->scr_mu.p, 0, (size_t)dp * 4, s));
    VQ_TRY(t->scr_count.alloc(8));  // two counters used in turn (TsvqScreen::turn)
    VQ_HIP(hipMemsetAsync(t->scr_count.p, 0, 8, s));
    VQ_HIP(hipMemcpyAsync(t->scr_w.p, w.data(), w.size() * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipMemcpyAsync(t->scr_info.p, info.data(), info.size() * 4, hipMemcpyHostToDevice, s));
    if (!cosine && !manhattan) VQ_HIP(hipMemcpyAsync(t->scr_mu.p, mu, (size_t)d * 4, hipMemcpyHostToDevice, s));  // else y = x
    VQ_HIP(hipStreamSynchronize(s));  // w / info are stack-owned
    t->scr.w = t->scr_w.as<float>();
    t->scr.info = t->scr_info.as<int4>();
    t->scr.mu = t->scr_mu.as<float>();
    t->scr.wl_count = t->scr_count.as<uint32_t>();
    t->scr.n_int = n_lds;  // slots resident in LDS; w / info / slot_node hold all of the tree's slots
    t->scr.n_slots = n_int;
    t->scr.n_nodes = n_nodes;
    // a non-finite tree sends every row to the exact continuation (T = NaN never passes)
    t->scr.R = finite ? (float)(std::sqrt(r2max) * 1.000001) : std::numeric_limits<float>::quiet_NaN();
    // T = u * base * (coef_a * base + coef_b * |w|), base >= |x - mu| + R
    t->scr.coef_a = 1.1f * (2.0f * d + 20.0f);
    t->scr.coef_b = 1.1f * (2.0f * d + 16.0f);
    if (cosine || manhattan) {
        // DESIGN.md 4.4 "cosine descent" / "Manhattan descent": the margin travels in the slot's record
        t->scr.R = 0.0f;
        t->scr.coef_a = 0.0f;
        t->scr.coef_b = 0.0f;
    }
    t->use_screen = true;
    return VQHIP_OK;
}

#define VQ_API_BEGIN try {
#define VQ_API_END                                                                  \
    }                                                                               \
    catch (const std::bad_alloc &) { return fail(VQHIP_ERR_FAILURE, "host allocation failed"); } \
    catch (...) { return fail(VQHIP_ERR_FAILURE, "unexpected C++ exception"); }

extern "C" {

// ------------------------------------------------------------------ library/device ----
const char *vqhip_backend(void) { return "libvqhip 0.1 (HIP, gfx950 / MI355X: bf16-split MFMA screen + exact VALU re-check)"; }

const char *vqhip_last_error(void) { return tls().last_error.c_str(); }

int vqhip_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    int good = 0;
    for (int i = 0; i < ndev; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            ++good;
    }
    return good;
}

int vqhip_set_device(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(VQHIP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(VQHIP_ERR_INVALID_INPUT, "device %d out of range [0,%d)", device, ndev);
    VQ_HIP(hipSetDevice(device));
    return require_gfx950();
}

int vqhip_get_device(int *device) {
    if (!device) return fail(VQHIP_ERR_NULL_PTR, "device is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VQHIP_ERR_NO_DEVICE, "no HIP device available");
    VQ_HIP(hipGetDevice(device));
    return VQHIP_OK;
}

int vqhip_set_stream(void *hip_stream) {
    ThreadState &st = tls();
    st.user_stream = reinterpret_cast<hipStream_t>(hip_stream);
    st.user_stream_set = (hip_stream != nullptr);
    return VQHIP_OK;
}

int vqhip_selftest(float *bf16_32x32x16_ratio, float *bf16_16x16x32_ratio, int *bf16_engine_trusted) {
    VQ_API_BEGIN
    VQ_TRY(require_gfx950());
    return bf16_mfma_selftest(bf16_32x32x16_ratio, bf16_16x16x32_ratio, bf16_engine_trusted);
    VQ_API_END
}

int vqhip_mfma_bf16_probe(const uint16_t *a, const uint16_t *b, const float *c, uint64_t trials, float *d) {
    VQ_API_BEGIN
    if (trials && (!a || !b || !c || !d)) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    return mfma_bf16_probe(a, b, c, trials, d);
    VQ_API_END
}

int vqhip_mfma_bf16_model(const uint16_t *a, const uint16_t *b, const float *c, uint64_t trials, float *d) {
    VQ_API_BEGIN
    if (trials && (!a || !b || !c || !d)) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    mfma_bf16_model_host(a, b, c, trials, d);  // pure integer model of one instruction: needs no device
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_mfma_bf16_model_check(uint64_t trials, uint64_t seed, uint64_t *mismatches, uint64_t *first_bad_trial) {
    VQ_API_BEGIN
    VQ_TRY(require_gfx950());
    return mfma_bf16_model_check(trials, seed, mismatches, first_bad_trial, nullptr, 0);
    VQ_API_END
}

int vqhip_mfma_bf16_model_failures(uint64_t trials, uint64_t seed, uint64_t *trial_ids, uint32_t cap, uint64_t *n_failures) {
    VQ_API_BEGIN
    if (!trial_ids || !n_failures) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    return mfma_bf16_model_check(trials, seed, n_failures, nullptr, trial_ids, cap);
    VQ_API_END
}

int vqhip_mfma_bf16_model_case(uint64_t seed, uint64_t trial, uint16_t *a, uint16_t *b, float *c) {
    if (!a || !b || !c) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    mfma_bf16_model_case(seed, trial, a, b, c);
    return VQHIP_OK;
}

int vqhip_synchronize(void) {
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    VQ_HIP(hipStreamSynchronize(s));
    return VQHIP_OK;
}

int vqhip_last_assign_stats(uint64_t *rechecked, int *engine) {
    VQ_API_BEGIN
    ThreadState &st = tls();
    if (engine) *engine = st.last_engine;
    if (rechecked) {
        *rechecked = 0;
        if (!g_last_ws) return VQHIP_OK;
        // the workspace is pinned for the query: its handle's destructor unregisters first and waits for the pin
        WsPin pin(g_last_ws);
        if (!pin.ws) return VQHIP_OK;  // the handle is gone
        AssignWorkspace *ws = pin.ws;
        Entry in(*pin.sync);
        if (ws->stats_pending) {
            // the copy is ordered behind the pass whatever stream that ran on (Entry::stream waits for the handle's
            // tail); a handle that several threads use reports its most recent pass, whoever queued it
            hipStream_t s;
            VQ_TRY(in.stream(&s));
            uint64_t tot = 0;
            if (ws->last_segmented) {
                if (ws->last_n_seg > 0) {
                    VQ_HIP(hipMemcpyAsync(ws->seg_host, ws->wl_seg.p, (size_t)ws->stats_m * ws->last_n_seg * 8, hipMemcpyDeviceToHost, s));
                    VQ_HIP(hipStreamSynchronize(s));
                    for (size_t i = 0; i < (size_t)ws->stats_m * ws->last_n_seg; ++i) tot += ws->seg_host[2 * i + 1];
                }
            } else {
                VQ_HIP(hipMemcpyAsync(ws->stats_host, ws->wl_count.p, (size_t)ws->stats_m * 4, hipMemcpyDeviceToHost, s));
                VQ_HIP(hipStreamSynchronize(s));
                for (uint32_t i = 0; i < ws->stats_m; ++i) tot += ws->stats_host[i];
            }
            in.synced();
            *rechecked = tot;
        }
    }
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_xfer_lane_calls(uint64_t *calls) {
    if (!calls) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    *calls = g_xfer_lane_calls.load();
    return VQHIP_OK;
}

int vqhip_set_profiling(int on) {
    g_prof.on = on != 0;
    return VQHIP_OK;
}

int vqhip_profile_collect(uint32_t *n_calls, double *primary_ms, double *recheck_ms) {
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    VQ_HIP(hipStreamSynchronize(s));
    double a = 0.0, b = 0.0;
    const size_t calls = g_prof.ev.size() / 3;
    for (size_t i = 0; i < calls; ++i) {
        float t01 = 0.f, t12 = 0.f;
        VQ_HIP(hipEventElapsedTime(&t01, g_prof.ev[3 * i], g_prof.ev[3 * i + 1]));
        VQ_HIP(hipEventElapsedTime(&t12, g_prof.ev[3 * i + 1], g_prof.ev[3 * i + 2]));
        a += t01;
        b += t12;
    }
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.engines.clear();
    if (n_calls) *n_calls = (uint32_t)calls;
    if (primary_ms) *primary_ms = a;
    if (recheck_ms) *recheck_ms = b;
    return VQHIP_OK;
}

int vqhip_memcpy_device(void *dst, const void *src, uint64_t bytes) {
    if (bytes == 0) return VQHIP_OK;
    if (!dst || !src) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    VQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    return VQHIP_OK;
}

// ------------------------------------------------------------------------ datasets ----
int vqhip_dataset_from_host(const float *rows, uint64_t n, uint32_t d, vqhip_dataset **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (n == 0 || d == 0) return fail(VQHIP_ERR_INVALID_INPUT, "empty dataset (n=%llu, d=%u)", (unsigned long long)n, d);
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    vqhip_dataset *ds = new vqhip_dataset();
    int rc = ds->own.alloc((size_t)n * d * 4);
    if (rc != VQHIP_OK) {
        delete ds;
        return rc;
    }
    // ONE copy straight from the caller's pages: 54.7 GB/s here (tools/host_xfer.py), 0.96 of what pinned memory gives;
    // staging through pinned buffers on several host threads was measured slower (45.9 GB/s: the host memcpys)
    hipError_t e = hipMemcpyAsync(ds->own.p, rows, (size_t)n * d * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        delete ds;
        return fail(VQHIP_ERR_RUNTIME, "H2D copy of the dataset failed: %s", hipGetErrorString(e));
    }
    ds->X = ds->own.as<float>();
    ds->n = n;
    ds->d = d;
    *out = ds;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_dataset_from_device(const void *dev_rows, uint64_t n, uint32_t d, vqhip_dataset **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (n == 0 || d == 0) return fail(VQHIP_ERR_INVALID_INPUT, "empty dataset");
    if (!dev_rows) return fail(VQHIP_ERR_NULL_PTR, "dev_rows is NULL");
    if ((reinterpret_cast<uintptr_t>(dev_rows) & 15) != 0)
        return fail(VQHIP_ERR_INVALID_INPUT, "device row buffer must be 16-byte aligned");
    VQ_TRY(require_gfx950());
    vqhip_dataset *ds = new vqhip_dataset();
    ds->X = reinterpret_cast<const float *>(dev_rows);
    ds->n = n;
    ds->d = d;
    *out = ds;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_dataset_synthetic(uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset, vqhip_dataset **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (n == 0 || d == 0) return fail(VQHIP_ERR_INVALID_INPUT, "empty dataset");
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    vqhip_dataset *ds = new vqhip_dataset();
    int rc = ds->own.alloc((size_t)n * d * 4);
    if (rc == VQHIP_OK) rc = launch_synth_uniform(ds->own.as<float>(), n, d, seed, row_offset, s);
    // a data set is immutable and shared freely between threads / streams: it is complete when the handle exists
    if (rc == VQHIP_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(VQHIP_ERR_RUNTIME, "generating the synthetic rows failed");
    if (rc != VQHIP_OK) {
        delete ds;
        return rc;
    }
    ds->X = ds->own.as<float>();
    ds->n = n;
    ds->d = d;
    *out = ds;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_dataset_info(const vqhip_dataset *ds, uint64_t *n, uint32_t *d, const void **dev_rows) {
    if (!ds) return fail(VQHIP_ERR_NULL_PTR, "dataset is NULL");
    if (n) *n = ds->n;
    if (d) *d = ds->d;
    if (dev_rows) *dev_rows = ds->X;
    return VQHIP_OK;
}

int vqhip_dataset_read(const vqhip_dataset *ds, uint64_t row0, uint64_t nrows, float *out) {
    if (!ds || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (row0 > ds->n || nrows > ds->n - row0) return fail(VQHIP_ERR_INVALID_INPUT, "row range out of bounds");
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    VQ_HIP(hipMemcpyAsync(out, ds->X + row0 * ds->d, (size_t)nrows * ds->d * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    return VQHIP_OK;
}

int vqhip_dataset_destroy(vqhip_dataset *ds) {
    delete ds;
    return VQHIP_OK;
}

int vqhip_synth_uniform_host(float *out, uint64_t n, uint32_t d, uint64_t seed, uint64_t row_offset) {
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    synth_uniform_host(out, n, d, seed, row_offset);
    return VQHIP_OK;
}

// ------------------------------------------------------------------------- k-means ----
int vqhip_kmeans_create(const vqhip_dataset *ds, uint32_t m, uint32_t k, vqhip_kmeans **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (!ds) return fail(VQHIP_ERR_NULL_PTR, "dataset is NULL");
    if (m == 0 || ds->d < m || ds->d % m != 0)
        return fail(VQHIP_ERR_INVALID_INPUT, "dimension (%u) must be divisible by m (%u)", ds->d, m);
    if (k == 0) return fail(VQHIP_ERR_INVALID_INPUT, "k must be greater than 0");
    if (k > kMaxCentroids) return fail(VQHIP_ERR_UNSUPPORTED, "k=%u > 65536: codes are at most two bytes per subspace", k);
    VQ_TRY(require_gfx950());
    const uint32_t sd = ds->d / m;
    std::unique_ptr<vqhip_kmeans> km(new vqhip_kmeans());
    km->ds = ds;
    VQ_TRY(km->cs.init(m, k, sd));
    // sub_dim > 1024 (lbg_quantize on long whole vectors): no LDS accumulator layout; the bucket-and-chain sums of
    // the reference-order update serve any length
    km->sums_by_chains = sd > 1024;
    if (!km->sums_by_chains) VQ_TRY(plan_update(m, k, sd, ds->n, &km->plan));
    VQ_TRY(km->codes.alloc((size_t)ds->n * m * code_bytes(k)));
    if (!km->sums_by_chains) {
        size_t sums_b = km->plan.partial_floats * km->plan.n_row_chunks * 4, cnt_b = km->plan.partial_counts * km->plan.n_row_chunks * 4;
        static const char *no_fused = getenv("VQHIP_FUSED_UPDATE");  // =0: always the separate accumulate pass (A/B)
        if (km->cs.x32_ok && screen_bf16_fused_update_supported(sd, k) && !(no_fused && no_fused[0] == '0')) {
            // fused update: one slab per (screen wave chunk or patch chunk, active subspace); <= 8 waves per CU
            km->fused_slabs = (uint32_t)num_cus() * 8 + 16 * m;
            sums_b = std::max(sums_b, (size_t)km->fused_slabs * k * sd * 4);
            cnt_b = std::max(cnt_b, (size_t)km->fused_slabs * k * 4);
        }
        VQ_TRY(km->partial_sums.alloc(sums_b));
        VQ_TRY(km->partial_counts.alloc(cnt_b));
    }
    VQ_TRY(km->slab.alloc((size_t)m * k * (sd + 1) * 8));
    VQ_TRY(km->counts.alloc((size_t)m * k * 4));
    VQ_TRY(km->changed.alloc((size_t)m * 4));
    VQ_TRY(km->active_dev.alloc(m));
    VQ_TRY(km->rows_tmp.alloc((size_t)m * k * 8));
    VQ_TRY(km->agree.alloc(16));
    VQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&km->counts_host), (size_t)m * k * 4));
    VQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&km->changed_host), (size_t)m * 4));
    VQ_HIP(hipHostMalloc(reinterpret_cast<void **>(&km->run_host), (size_t)(m + 1 + 6 * m) * 4 + ((size_t)m + 3) / 4 * 4));
    km->active.assign(m, 1);
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    VQ_HIP(hipMemsetAsync(km->active_dev.p, 1, m, s));
    VQ_HIP(hipMemsetAsync(km->cs.cb.p, 0, km->cs.cb.bytes, s));
    VQ_HIP(hipMemsetAsync(km->slab.p, 0, km->slab.bytes, s));
    VQ_HIP(hipStreamSynchronize(s));
    *out = km.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_kmeans_destroy(vqhip_kmeans *km) {
    delete km;  // (unregisters its workspace: a later vqhip_last_assign_stats of any thread finds it gone)
    return VQHIP_OK;
}

int vqhip_kmeans_set_centroids(vqhip_kmeans *km, const float *centroids) {
    if (!km || !centroids) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_HIP(hipMemcpyAsync(km->cs.cb.p, centroids, (size_t)km->cs.m * km->cs.k * km->cs.sd * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    km->cs.prepared = false;
    return VQHIP_OK;
}

int vqhip_kmeans_init_from_rows(vqhip_kmeans *km, const uint64_t *init_rows) {
    if (!km || !init_rows) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    const size_t cnt = (size_t)km->cs.m * km->cs.k;
    for (size_t i = 0; i < cnt; ++i)
        if (init_rows[i] >= km->ds->n)
            return fail(VQHIP_ERR_INVALID_INPUT, "init row %llu out of range", (unsigned long long)init_rows[i]);
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_HIP(hipMemcpyAsync(km->rows_tmp.p, init_rows, cnt * 8, hipMemcpyHostToDevice, s));
    VQ_TRY(launch_gather_rows(km->ds->X, km->ds->d, km->cs.m, km->cs.k, km->cs.sd, km->rows_tmp.as<uint64_t>(),
                              km->cs.cb.as<float>(), s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    km->cs.prepared = false;
    return VQHIP_OK;
}

int vqhip_kmeans_get_centroids(vqhip_kmeans *km, float *centroids) {
    if (!km || !centroids) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_HIP(hipMemcpyAsync(centroids, km->cs.cb.p, (size_t)km->cs.m * km->cs.k * km->cs.sd * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
}

int vqhip_kmeans_set_active(vqhip_kmeans *km, const uint8_t *active) {
    if (!km || !active) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    bool all = true;
    for (uint32_t s = 0; s < km->cs.m; ++s) {
        km->active[s] = active[s] ? 1 : 0;
        all = all && km->active[s];
    }
    km->all_active = all;
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_HIP(hipMemcpyAsync(km->active_dev.p, km->active.data(), km->cs.m, hipMemcpyHostToDevice, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
}

int vqhip_kmeans_get_active(const vqhip_kmeans *km, uint8_t *active) {
    if (!km || !active) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(const_cast<vqhip_kmeans *>(km)->sync.mu);
    for (uint32_t s = 0; s < km->cs.m; ++s) active[s] = km->active[s] ? 1 : 0;
    return VQHIP_OK;
}

int vqhip_kmeans_set_engine(vqhip_kmeans *km, int engine) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (engine < VQHIP_ENGINE_AUTO || engine > VQHIP_ENGINE_MFMA_BF16) return fail(VQHIP_ERR_INVALID_INPUT, "unknown engine %d", engine);
    std::lock_guard<std::recursive_mutex> lk(km->sync.mu);
    km->engine = engine;
    return VQHIP_OK;
}

uint32_t vqhip_code_bytes(uint32_t k) { return code_bytes(k); }

int vqhip_kmeans_set_exact_update(vqhip_kmeans *km, int exact_update) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(km->sync.mu);
    km->exact_update = exact_update ? 1 : 0;
    return VQHIP_OK;
}

// A small problem's iteration is TWO launches (k_lloyd_small.hip) wherever the whole loop body runs on this device: the
// single-GPU vqhip_kmeans_step / vqhip_kmeans_run; the split accumulate / all-reduce / finalize forms run the same
// kernels up to the f64 slab and k_finalize behind it, with the same bits.  Against the general path (VQHIP_SMALL_LLOYD=0)
// codes, counts and flags are equal bit for bit; the means differ in the last bits (other row ranges behind the f64
// combination), both inside the tolerance DESIGN.md 2 states.
static bool kmeans_small_eligible(const vqhip_kmeans *km) {
    static const char *env = getenv("VQHIP_SMALL_LLOYD");  // =0: never (A/B)
    if (env && env[0] == '0') return false;
    return km->engine == VQHIP_ENGINE_AUTO && !km->exact_update && !km->sums_by_chains && !g_prof.on &&
           lloyd_small_supported(km->ds->n, km->cs.m, km->cs.k, km->cs.sd);
}
static int kmeans_small_workspace(vqhip_kmeans *km, hipStream_t s) {
    const uint32_t m = km->cs.m, k = km->cs.k, sd = km->cs.sd;
    if (!km->sm_tick.p) {
        size_t cnt_b = 0, tick_b = 0;
        const size_t sum_b = lloyd_small_workspace(km->ds->n, m, k, sd, &cnt_b, &tick_b);
        VQ_TRY(km->sm_psum.alloc(sum_b));
        VQ_TRY(km->sm_pcnt.alloc(cnt_b));
        VQ_TRY(km->sm_tick.alloc(tick_b));
        VQ_HIP(hipMemsetAsync(km->sm_tick.p, 0, tick_b, s));
    }
    return VQHIP_OK;
}
static void kmeans_small_note(vqhip_kmeans *km) {  // host-side state behind a small-path launch
    km->ws.stats_pending = false;
    km->ws.last_engine = VQHIP_ENGINE_EXACT;
    ThreadState &st = tls();
    st.last_engine = VQHIP_ENGINE_EXACT;
    st.last_rechecked = 0;
    g_last_ws = km->ws_id;
}
// run: iteration `it` of a device-driven run (its flags behind sm_tick, the iterations executed in run_state[1 .. m])
static int kmeans_small_enqueue(vqhip_kmeans *km, hipStream_t s, bool run, uint32_t it = 0) {
    const uint32_t m = km->cs.m, k = km->cs.k, sd = km->cs.sd;
    VQ_TRY(kmeans_small_workspace(km, s));
    const uint8_t *act = run ? km->active_dev.as<uint8_t>() : (km->all_active ? nullptr : km->active_dev.as<uint8_t>());
    VQ_TRY(launch_lloyd_small(km->ds->X, km->ds->n, km->ds->d, m, k, sd, km->cs.cb.as<float>(), km->codes.as<uint8_t>(),
                              km->sm_psum.as<float>(), km->sm_pcnt.as<uint32_t>(), km->counts.as<uint32_t>(), km->changed.as<uint32_t>(),
                              act, run ? km->sm_tick.as<uint32_t>() : nullptr, run ? km->run_state.as<uint32_t>() + 1 : nullptr, it, s));
    km->cs.prepared = false;
    km->accumulated = false;
    kmeans_small_note(km);
    return VQHIP_OK;
}

// queue assign + update of one Lloyd iteration on `s` (no host synchronisation: capturable); gated: inside a
// device-driven run (vqhip_kmeans_run) every kernel that changes state checks the run's device flags
static int kmeans_accumulate_enqueue(vqhip_kmeans *km, hipStream_t s, bool gated = false, bool fuse_finalize = false) {
    std::vector<uint32_t> subs;
    for (uint32_t i = 0; i < km->cs.m; ++i)
        if (km->active[i]) subs.push_back(i);
    const vqhip_dataset *ds = km->ds;
    g_last_ws = km->ws_id;
    if (kmeans_small_eligible(km)) {  // small problem: assignment + this rank's f64 slab in one launch (k_lloyd_small.hip)
        VQ_TRY(kmeans_small_workspace(km, s));
        const uint8_t *act = gated ? km->active_dev.as<uint8_t>() : (km->all_active ? nullptr : km->active_dev.as<uint8_t>());
        VQ_TRY(launch_lloyd_small_slab(ds->X, ds->n, ds->d, km->cs.m, km->cs.k, km->cs.sd, km->cs.cb.as<float>(), km->codes.as<uint8_t>(),
                                       km->sm_psum.as<float>(), km->sm_pcnt.as<uint32_t>(), act,
                                       gated ? km->run_state.as<uint32_t>() : nullptr, km->changed.as<uint32_t>(), km->slab.as<double>(), s));
        kmeans_small_note(km);
        km->accumulated = true;
        return VQHIP_OK;
    }
    // assignment: always squared L2 (src/core/vector.rs:352-363); with the update fused in where the shape allows
    FusedAcc fused;
    if (km->fused_slabs && !km->exact_update && !km->sums_by_chains) {
        fused.sums = km->partial_sums.as<float>();
        fused.counts = km->partial_counts.as<uint32_t>();
        fused.slab_cap = km->fused_slabs;
        if (gated) {
            fused.gate_active = km->active_dev.as<uint8_t>();
            fused.gate_halt = km->run_state.as<uint32_t>();
        }
    }
    VQ_TRY(run_assign(km->cs, km->ws, ds->X, ds->n, ds->d, VQHIP_SQUARED_EUCLIDEAN, subs,
                      km->codes.as<uint8_t>(), km->engine, s, &fused));
    if (gated && !fused.used) return fail(VQHIP_ERR_FAILURE, "device-driven run without the fused update");
    const uint8_t *act = km->all_active ? nullptr : km->active_dev.as<uint8_t>();
    if (fused.used && gated && fuse_finalize) {
        // one rank, device-driven: sums and means in one launch, the iteration's decisions behind it (k_reduce_finalize_run)
        uint32_t *rs = km->run_state.as<uint32_t>();  // [0] halt, [1..m] iterations, [m+1] unused, [m+2] empty-cluster flag, [m+3..] `changed` scratch [m]
        const uint32_t m = km->cs.m;
        VQ_TRY(launch_reduce_finalize_run(m, km->cs.k, km->cs.sd, km->partial_sums.as<float>(), km->partial_counts.as<uint32_t>(),
                                          fused.chunks, (uint32_t)subs.size(), km->ws.sub_pos.as<int32_t>(), km->slab.as<double>(),
                                          km->active_dev.as<uint8_t>(), km->cs.cb.as<float>(), km->counts.as<uint32_t>(),
                                          km->changed.as<uint32_t>(), rs, rs + 1, rs + 1 + m, rs + 3 + m, s));
    } else if (fused.used) {
        VQ_TRY(launch_reduce_partials_pos(km->cs.m, km->cs.k, km->cs.sd, km->partial_sums.as<float>(),
                                          km->partial_counts.as<uint32_t>(), fused.chunks, (uint32_t)subs.size(),
                                          km->ws.sub_pos.as<int32_t>(), km->slab.as<double>(), s, fused.gate_active,
                                          fused.gate_halt, gated ? km->changed.as<uint32_t>() : nullptr));
    } else if (km->exact_update || km->sums_by_chains) {
        size_t need = exact_sums_workspace_bytes(km->cs.m, km->cs.k, ds->n);
        VQ_TRY(km->xs_ws.ensure(need));
        VQ_TRY(launch_exact_sums(km->cs.m, km->cs.k, km->cs.sd, ds->X, ds->n, ds->d, km->codes.as<uint8_t>(), act,
                                 km->xs_ws.p, km->xs_ws.bytes, km->slab.as<double>(), s));
    } else {
        VQ_TRY(launch_accumulate(km->plan, ds->X, ds->n, ds->d, km->codes.as<uint8_t>(), act,
                                 km->partial_sums.as<float>(), km->partial_counts.as<uint32_t>(), s));
        VQ_TRY(launch_reduce_partials(km->plan, km->partial_sums.as<float>(), km->partial_counts.as<uint32_t>(), act,
                                      km->slab.as<double>(), s));
    }
    km->accumulated = true;
    return VQHIP_OK;
}

int vqhip_kmeans_accumulate(vqhip_kmeans *km) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    return kmeans_accumulate_enqueue(km, s);
    VQ_API_END
}

int vqhip_kmeans_partials(vqhip_kmeans *km, void **dev_slab, uint64_t *n_doubles) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(km->sync.mu);
    if (dev_slab) *dev_slab = km->slab.p;
    if (n_doubles) *n_doubles = (uint64_t)km->cs.m * km->cs.k * (km->cs.sd + 1);
    return VQHIP_OK;
}

// queue mean / convergence test + the read-back of counts and flags (capturable); gated (device-driven run): the same
// kernel also ends the iteration (retire converged subspaces, count it, halt on an empty cluster)
static int kmeans_finalize_enqueue(vqhip_kmeans *km, hipStream_t s, bool gated = false, bool read_back = true, bool done_already = false) {
    const uint32_t m = km->cs.m, k = km->cs.k;
    if (done_already) {
        // (k_reduce_finalize_run has formed the means behind the sums)
    } else if (gated) {
        uint32_t *rs = km->run_state.as<uint32_t>();  // [0] halt, [1..m] iterations, [m+1] finished-workgroup counter, [m+2] empty-cluster flag of the iteration
        VQ_TRY(launch_finalize_run(m, k, km->cs.sd, km->slab.as<double>(), km->active_dev.as<uint8_t>(), km->cs.cb.as<float>(),
                                   km->counts.as<uint32_t>(), km->changed.as<uint32_t>(), rs, rs + 1, rs + 1 + m, s));
    } else {
        const uint8_t *act = km->all_active ? nullptr : km->active_dev.as<uint8_t>();
        VQ_TRY(launch_finalize(m, k, km->cs.sd, km->slab.as<double>(), act, km->cs.cb.as<float>(),
                               km->counts.as<uint32_t>(), km->changed.as<uint32_t>(), km->exact_update, s));
    }
    km->cs.prepared = false;
    km->accumulated = false;
    if (read_back) {
        VQ_HIP(hipMemcpyAsync(km->counts_host, km->counts.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipMemcpyAsync(km->changed_host, km->changed.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    }
    return VQHIP_OK;
}
static void kmeans_finalize_collect(vqhip_kmeans *km, uint32_t *counts, uint8_t *changed) {
    const uint32_t m = km->cs.m, k = km->cs.k;
    if (counts) memcpy(counts, km->counts_host, (size_t)m * k * 4);
    if (changed)
        for (uint32_t i = 0; i < m; ++i) changed[i] = (km->active[i] && km->changed_host[i]) ? 1 : 0;
}

int vqhip_kmeans_finalize(vqhip_kmeans *km, uint32_t *counts, uint8_t *changed) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    if (!km->accumulated) return fail(VQHIP_ERR_INVALID_INPUT, "finalize without a preceding accumulate");
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_TRY(kmeans_finalize_enqueue(km, s));
    VQ_TRY(spin_wait(s));  // a parked thread wakes ~100 us late; an iteration is 0.1-10 ms
    in.synced();
    kmeans_finalize_collect(km, counts, changed);
    return VQHIP_OK;
    VQ_API_END
}

// Small data sets are launch-bound (C1, 10k x 64: ~60 us of kernels behind ~14 stream operations),
// so the whole step is captured once per (active set, engine) and replayed as a hipGraph; the first
// step with a given key runs eagerly (allocations, code objects, sub-list upload), the second is
// captured, later ones are replays.  VQHIP_GRAPH=0 disables, =1 forces it for any size.
static bool kmeans_graph_eligible(const vqhip_kmeans *km) {
    static const char *env = getenv("VQHIP_GRAPH");
    if (env && env[0] == '0') return false;
    if (km->graph_failed || km->exact_update || km->sums_by_chains || g_prof.on) return false;
    if (env && env[0] == '1') return true;
    return (uint64_t)km->ds->n * km->cs.m <= (4ull << 20);
}

int vqhip_kmeans_step(vqhip_kmeans *km, uint32_t *counts, uint8_t *changed) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    if (kmeans_small_eligible(km)) {
        const uint32_t m = km->cs.m, k = km->cs.k;
        VQ_TRY(kmeans_small_enqueue(km, s, false));
        VQ_HIP(hipMemcpyAsync(km->counts_host, km->counts.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipMemcpyAsync(km->changed_host, km->changed.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        VQ_TRY(spin_wait(s));
        in.synced();
        kmeans_finalize_collect(km, counts, changed);
        return VQHIP_OK;
    }
    if (!kmeans_graph_eligible(km)) {
        VQ_TRY(kmeans_accumulate_enqueue(km, s));
        VQ_TRY(kmeans_finalize_enqueue(km, s));
        VQ_TRY(spin_wait(s));
        in.synced();
        kmeans_finalize_collect(km, counts, changed);
        return VQHIP_OK;
    }
    uint64_t key = 1469598103934665603ull;  // FNV-1a over what shapes the launch sequence
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    for (uint32_t i = 0; i < km->cs.m; ++i) mix(km->active[i]);
    mix((uint64_t)km->engine + 17);
    mix(reinterpret_cast<uintptr_t>(s));
    if (km->graph_exec && km->graph_key == key) {
        VQ_HIP(hipGraphLaunch(km->graph_exec, s));
        VQ_TRY(spin_wait(s));
    } else if (km->warm_key == key) {
        km->drop_graph();
        km->cs.prepared = false;  // the codebook images must be part of the captured sequence
        hipGraph_t graph = nullptr;
        VQ_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = kmeans_accumulate_enqueue(km, s);
        if (rc == VQHIP_OK) rc = kmeans_finalize_enqueue(km, s);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc == VQHIP_OK && e == hipSuccess && graph &&
            hipGraphInstantiate(&km->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            (void)hipGraphDestroy(graph);
            km->graph_key = key;
            VQ_HIP(hipGraphLaunch(km->graph_exec, s));
            VQ_TRY(spin_wait(s));
        } else {  // capture is an optimisation only: fall back to the plain sequence for this handle
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            km->graph_exec = nullptr;
            km->graph_failed = true;
            km->cs.prepared = false;
            VQ_TRY(kmeans_accumulate_enqueue(km, s));
            VQ_TRY(kmeans_finalize_enqueue(km, s));
            VQ_TRY(spin_wait(s));
        }
    } else {
        VQ_TRY(kmeans_accumulate_enqueue(km, s));
        VQ_TRY(kmeans_finalize_enqueue(km, s));
        VQ_TRY(spin_wait(s));
        km->warm_key = key;
    }
    in.synced();  // every branch above ended in spin_wait
    // host-side state the enqueue functions leave behind (a replay does not run them)
    km->cs.prepared = false;
    km->accumulated = false;
    km->ws.stats_pending = (km->ws.last_engine == VQHIP_ENGINE_MFMA || km->ws.last_engine == VQHIP_ENGINE_MFMA_BF16);
    g_last_ws = km->ws_id;
    tls().last_engine = km->ws.last_engine;
    kmeans_finalize_collect(km, counts, changed);
    return VQHIP_OK;
    VQ_API_END
}

// ---- failure containment of the sharded entry points -------------------------------------------------------------------
// A rank that leaves a collective call early (allocation / launch failure on its device only) must not leave its peers
// waiting: every sharded entry point goes out through sharded_exit, which poisons an in-process group (comm.hip:
// comm_abort; the peers' barriers return VQHIP_ERR_RUNTIME at once, and their waits are bounded anyway).  Argument errors
// every rank sees alike (NULL, INVALID_INPUT, UNSUPPORTED) do not poison.
static int sharded_exit(vqhip_comm *comm, int rc) {
    if ((rc == VQHIP_ERR_RUNTIME || rc == VQHIP_ERR_FAILURE || rc == VQHIP_ERR_NO_DEVICE) && comm && comm->c) {
        const char *silent = getenv("VQHIP_TEST_FAIL_SILENT");  // tests: a rank that dies without telling (peers time out)
        if (!(silent && silent[0] == '1')) comm_abort(comm->c, tls().last_error.c_str());
    }
    return rc;
}

// Fault injection for tests/test_gpu_multi.py (compiled in, off unless BOTH variables are set; read per call):
// rank VQHIP_TEST_FAIL_RANK of a sharded run fails in front of iteration VQHIP_TEST_FAIL_ITER with VQHIP_ERR_RUNTIME, as a
// launch failure on that rank's device would.
static int fault_injected(Comm *comm, uint32_t it) {
    const char *fr = getenv("VQHIP_TEST_FAIL_RANK"), *fi = getenv("VQHIP_TEST_FAIL_ITER");
    if (!fr || !fi || !fr[0] || !fi[0]) return VQHIP_OK;
    int world = 1, rank = 0;
    comm_info(comm, &world, &rank);
    if (rank != atoi(fr) || it != (uint32_t)atoi(fi)) return VQHIP_OK;
    return fail(VQHIP_ERR_RUNTIME, "fault injection: rank %d of %d fails in front of iteration %u (VQHIP_TEST_FAIL_RANK / _ITER)", rank, world, it);
}

// Up to max_iters Lloyd iterations (src/core/vector.rs:415-458) with the loop's decisions taken on the device: the
// iterations are queued back to back; a converged subspace stops being processed, an empty cluster in an active
// subspace pauses the run after that iteration (every later queued kernel becomes a no-op) so that the caller can
// draw the reseed rows.  Shapes without the fused update take the same decisions on the host, one step at a time.
static int kmeans_run_impl(vqhip_kmeans *km, Comm *comm, uint32_t max_iters, uint32_t *iters_done, uint32_t *counts,
                           uint8_t *changed, int *paused) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    int world = 1;
    comm_info(comm, &world, nullptr);
    Entry in(km->sync);
    // (arguments every rank passes alike -- the host program's contract -- may end the call before the agreement below;
    // anything that depends on THIS rank's device, environment or memory may not: a rank that left early would leave
    // its peers blocked in the collective)
    if (world > 1 && km->exact_update)
        return fail(VQHIP_ERR_UNSUPPORTED, "exact_update sums rows in one sequential chain: single GPU only");
    const uint32_t m = km->cs.m, k = km->cs.k;
    if (iters_done) memset(iters_done, 0, (size_t)m * 4);
    if (paused) *paused = 0;
    bool any_active = false;
    for (uint32_t i = 0; i < m; ++i) any_active = any_active || km->active[i];
    if (max_iters == 0 || !any_active) {
        if (changed) memset(changed, 0, m);
        return VQHIP_OK;
    }
    hipStream_t s = nullptr;
    int engine = 0;
    int local_rc = require_gfx950();
    if (local_rc == VQHIP_OK) local_rc = in.stream(&s);
    if (local_rc == VQHIP_OK) local_rc = pick_engine(km->engine, km->cs, VQHIP_SQUARED_EUCLIDEAN, &engine);
    static const char *host_loop_env = getenv("VQHIP_RUN_ON_HOST");  // =1: decisions on the host (A/B)
    uint32_t n_active = 0;
    for (uint32_t i = 0; i < m; ++i) n_active += km->active[i] ? 1u : 0u;
    // the device-driven loop needs the fused update: exactly run_assign's predicate (engine, one centroid group, a slab
    // per (chunk, active subspace) beyond the list-driven ones), evaluated BEFORE anything is queued -- otherwise the
    // host-driven loop below serves the shape
    bool device_loop = local_rc == VQHIP_OK && km->fused_slabs && !km->exact_update && !km->sums_by_chains &&
                       engine == VQHIP_ENGINE_MFMA_BF16 && km->cs.x32_groups == 1 && screen_bf16_fused_update_supported(km->cs.sd, k) &&
                       km->fused_slabs / n_active > FusedAcc().n_patch && km->ds->n != 0 && km->ds->n < (1ull << 32) &&
                       !g_prof.on && !(host_loop_env && host_loop_env[0] == '1');
    if (local_rc == VQHIP_OK && kmeans_small_eligible(km) && !(host_loop_env && host_loop_env[0] == '1')) device_loop = true;
    if (world > 1) {
        // every rank must queue the same number of all-reduces: the decision depends on per-rank state (environment,
        // profiling hooks, the local row count), so the ranks agree on it -- device loop only if ALL of them can -- and
        // in the same collective on whether every rank got this far: {ranks that are ready, ranks that can loop on
        // the device}.  `agree` was allocated with the handle, so a rank short of memory still takes part.
        const std::string my_error = local_rc == VQHIP_OK ? std::string() : tls().last_error;
        if (!s) VQ_TRY(current_stream(&s));  // (no stream at all: nothing can be queued, the peers time out in RCCL)
        const uint32_t mine[2] = {local_rc == VQHIP_OK ? 1u : 0u, device_loop ? 1u : 0u};
        uint32_t all[2] = {0, 0};
        VQ_HIP(hipMemcpyAsync(km->agree.p, mine, 8, hipMemcpyHostToDevice, s));
        VQ_TRY(comm_allreduce_u32(comm, km->agree.as<uint32_t>(), 2, s));
        VQ_HIP(hipMemcpyAsync(all, km->agree.p, 8, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipStreamSynchronize(s));
        if (local_rc != VQHIP_OK) return fail(local_rc, "%s", my_error.c_str());
        if (all[0] != (uint32_t)world)
            return fail(VQHIP_ERR_FAILURE, "%u of %d ranks could not start the run (their own error text says why)",
                        (uint32_t)world - all[0], world);
        device_loop = all[1] == (uint32_t)world;
    }
    VQ_TRY(local_rc);
    // small problems: one launch per iteration on a single GPU (small_loop); row-sharded, the slab form of the same kernel
    // + all-reduce + k_finalize<true>, still without the host
    const bool small_ok = local_rc == VQHIP_OK && kmeans_small_eligible(km) && !(host_loop_env && host_loop_env[0] == '1');
    const bool small_loop = world == 1 && small_ok;
    if (!device_loop) {
        std::vector<uint32_t> cnt((size_t)m * k);
        std::vector<uint8_t> chg(m);
        for (uint32_t it = 0; it < max_iters; ++it) {
            bool any = false;
            for (uint32_t i = 0; i < m; ++i) any = any || km->active[i];
            if (!any) break;
            VQ_TRY(fault_injected(comm, it));
            if (world > 1) {
                VQ_TRY(kmeans_accumulate_enqueue(km, s));
                VQ_TRY(comm_allreduce_f64(comm, km->slab.as<double>(), (size_t)m * k * (km->cs.sd + 1), s));
                VQ_TRY(kmeans_finalize_enqueue(km, s));
                VQ_TRY(spin_wait(s));
                kmeans_finalize_collect(km, cnt.data(), chg.data());
            } else {
                VQ_TRY(vqhip_kmeans_step(km, cnt.data(), chg.data()));
            }
            bool empty = false;
            for (uint32_t i = 0; i < m; ++i) {
                if (!km->active[i]) continue;
                if (iters_done) iters_done[i] += 1;
                for (uint32_t j = 0; j < k; ++j) empty = empty || cnt[(size_t)i * k + j] == 0;
            }
            if (counts) memcpy(counts, cnt.data(), cnt.size() * 4);
            if (changed) memcpy(changed, chg.data(), m);
            if (empty) {
                if (paused) *paused = 1;
                break;
            }
            std::vector<uint8_t> act(km->active.begin(), km->active.end());
            bool flip = false;
            for (uint32_t i = 0; i < m; ++i)
                if (act[i] && !chg[i]) {
                    act[i] = 0;
                    flip = true;
                }
            if (flip) VQ_TRY(vqhip_kmeans_set_active(km, act.data()));
        }
        in.synced();  // every step ended in a wait
        return VQHIP_OK;
    }
    VQ_TRY(km->run_state.ensure((size_t)(2 * m + 3) * 4));
    VQ_HIP(hipMemsetAsync(km->run_state.p, 0, (size_t)(2 * m + 3) * 4, s));
    // one rank: nothing is exchanged between the sums and the means, so both are one launch (VQHIP_FUSED_FINALIZE=0: two, for A/B)
    static const char *ff_env = getenv("VQHIP_FUSED_FINALIZE");
    const bool fuse_finalize = world == 1 && !small_loop && !(ff_env && ff_env[0] == '0');
    if (small_loop) {  // the small form keeps the loop's decisions in two flag sets (k_lloyd_small.hip)
        VQ_TRY(kmeans_small_workspace(km, s));
        VQ_HIP(hipMemsetAsync(km->sm_tick.p, 0, (size_t)6 * m * 4, s));
    }
    for (uint32_t it = 0; it < max_iters; ++it) {
        if (small_loop) {  // two launches per iteration, the loop's decisions read from the previous iteration's flags
            VQ_TRY(kmeans_small_enqueue(km, s, true, it));
            continue;
        }
        VQ_TRY(fault_injected(comm, it));
        VQ_TRY(kmeans_accumulate_enqueue(km, s, true, fuse_finalize));
        // row-sharded: the one exchange of the iteration (a paused run re-sums a slab nobody reads: all ranks pause alike)
        if (!fuse_finalize) VQ_TRY(comm_allreduce_f64(comm, km->slab.as<double>(), (size_t)m * k * (km->cs.sd + 1), s));
        VQ_TRY(kmeans_finalize_enqueue(km, s, true, false, fuse_finalize));  // + the iteration's decisions (k_finalize<true> + k_run_decide)
    }
    // everything the host reads goes to pinned memory, queued back to back, ONE wait
    uint32_t *const st = km->run_host, *const sm_flags = st + (m + 1);
    uint8_t *const act = reinterpret_cast<uint8_t *>(sm_flags + (size_t)6 * m);
    VQ_HIP(hipMemcpyAsync(km->counts_host, km->counts.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipMemcpyAsync(km->changed_host, km->changed.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipMemcpyAsync(st, km->run_state.p, (size_t)(m + 1) * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipMemcpyAsync(act, km->active_dev.p, m, hipMemcpyDeviceToHost, s));
    if (small_loop) VQ_HIP(hipMemcpyAsync(sm_flags, km->sm_tick.p, (size_t)6 * m * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    if (small_loop) {  // the loop's decisions from the flags the last executed iteration left (the kernels never write the set itself)
        bool pause = false;
        std::vector<uint8_t> start(act, act + m);  // active_dev still holds the set the run started from
        lloyd_small_run_result(m, start.data(), sm_flags, st + 1, &pause, act, km->changed_host);
        st[0] = pause ? 1u : 0u;
        // the set as the run left it goes back behind everything queued so far; pinned source, nothing else writes it before
        // the next run's own read-back has been waited for: no wait here (the handle's stream tail stays set)
        VQ_HIP(hipMemcpyAsync(km->active_dev.p, act, m, hipMemcpyHostToDevice, s));
    } else {
        in.synced();
    }
    kmeans_finalize_collect(km, counts, changed);  // flags of the last executed iteration, for the subspaces active in it
    if (iters_done) memcpy(iters_done, st + 1, (size_t)m * 4);
    if (paused) *paused = st[0] ? 1 : 0;
    bool all = true;
    for (uint32_t i = 0; i < m; ++i) {
        km->active[i] = act[i] ? 1 : 0;  // converged subspaces retired on the device (none on a pause: the caller decides)
        all = all && km->active[i];
    }
    km->all_active = all;
    km->ws.stats_pending = !small_loop;
    g_last_ws = km->ws_id;
    tls().last_engine = km->ws.last_engine;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_kmeans_run(vqhip_kmeans *km, uint32_t max_iters, uint32_t *iters_done, uint32_t *counts, uint8_t *changed,
                     int *paused) {
    return kmeans_run_impl(km, nullptr, max_iters, iters_done, counts, changed, paused);
}

int vqhip_kmeans_run_sharded(vqhip_kmeans *km, vqhip_comm *comm, uint32_t max_iters, uint32_t *iters_done, uint32_t *counts,
                             uint8_t *changed, int *paused) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk_km(km->sync.mu);  // lock order: k-means handle, then communicator
    std::unique_lock<std::recursive_mutex> lk_comm;
    if (comm) lk_comm = std::unique_lock<std::recursive_mutex>(comm->mu);
    return sharded_exit(comm, kmeans_run_impl(km, comm ? comm->c : nullptr, max_iters, iters_done, counts, changed, paused));
}

int vqhip_kmeans_patch_centroid(vqhip_kmeans *km, uint32_t s, uint32_t j, const float *sub_row) {
    if (!km || !sub_row) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (s >= km->cs.m || j >= km->cs.k) return fail(VQHIP_ERR_INVALID_INPUT, "centroid (%u,%u) out of range", s, j);
    Entry in(km->sync);
    hipStream_t st;
    VQ_TRY(in.stream(&st));
    float *dst = km->cs.cb.as<float>() + ((size_t)s * km->cs.k + j) * km->cs.sd;
    VQ_HIP(hipMemcpyAsync(dst, sub_row, (size_t)km->cs.sd * 4, hipMemcpyHostToDevice, st));
    VQ_HIP(hipStreamSynchronize(st));
    in.synced();
    km->cs.prepared = false;
    return VQHIP_OK;
}

int vqhip_kmeans_patch_from_row(vqhip_kmeans *km, uint32_t s, uint32_t j, uint64_t row) {
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (s >= km->cs.m || j >= km->cs.k) return fail(VQHIP_ERR_INVALID_INPUT, "centroid (%u,%u) out of range", s, j);
    if (row >= km->ds->n) return fail(VQHIP_ERR_INVALID_INPUT, "row %llu out of range", (unsigned long long)row);
    Entry in(km->sync);
    hipStream_t st;
    VQ_TRY(in.stream(&st));
    float *dst = km->cs.cb.as<float>() + ((size_t)s * km->cs.k + j) * km->cs.sd;
    const float *src = km->ds->X + row * km->ds->d + (size_t)s * km->cs.sd;
    VQ_HIP(hipMemcpyAsync(dst, src, (size_t)km->cs.sd * 4, hipMemcpyDeviceToDevice, st));
    km->cs.prepared = false;
    return VQHIP_OK;
}

int vqhip_kmeans_get_assignments(vqhip_kmeans *km, uint8_t *codes) {
    if (!km || !codes) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_HIP(hipMemcpyAsync(codes, km->codes.p, (size_t)km->ds->n * km->cs.m * code_bytes(km->cs.k), hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
}

// ------------------------------------------------- row-sharded training (RCCL below the ABI) ----
int vqhip_comm_unique_id(uint8_t *id) {
    VQ_API_BEGIN
    if (!id) return fail(VQHIP_ERR_NULL_PTR, "id is NULL");
    return comm_unique_id(id);
    VQ_API_END
}

int vqhip_comm_create(const uint8_t *id, int world, int rank, vqhip_comm **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (world > 1 && !id) return fail(VQHIP_ERR_NULL_PTR, "a communicator of %d ranks needs the unique id", world);
    if (id) VQ_TRY(require_gfx950());  // ncclCommInitRank binds the calling thread's current device
    std::unique_ptr<vqhip_comm> h(new vqhip_comm());
    VQ_TRY(comm_create(id, world, rank, &h->c));
    *out = h.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_comm_adopt(void *nccl_comm, vqhip_comm **out) {
    VQ_API_BEGIN
    if (!out || !nccl_comm) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    *out = nullptr;
    std::unique_ptr<vqhip_comm> h(new vqhip_comm());
    VQ_TRY(comm_adopt(nccl_comm, &h->c));
    *out = h.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_comm_info(const vqhip_comm *comm, int *world, int *rank) {
    comm_info(comm ? comm->c : nullptr, world, rank);
    return VQHIP_OK;
}

int vqhip_comm_destroy(vqhip_comm *comm) {
    if (comm) {
        (void)comm_destroy(comm->c);
        delete comm;
    }
    return VQHIP_OK;
}

struct vqhip_comm_group {
    LocalGroup *g = nullptr;
};

int vqhip_comm_group_create(int world, vqhip_comm_group **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    std::unique_ptr<vqhip_comm_group> h(new vqhip_comm_group());
    VQ_TRY(local_group_create(world, &h->g));
    *out = h.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_comm_create_local(vqhip_comm_group *group, int rank, vqhip_comm **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (!group) return fail(VQHIP_ERR_NULL_PTR, "group is NULL");
    VQ_TRY(require_gfx950());
    std::unique_ptr<vqhip_comm> h(new vqhip_comm());
    VQ_TRY(comm_create_local(group->g, rank, &h->c));
    *out = h.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_comm_group_destroy(vqhip_comm_group *group) {
    if (group) {
        local_group_destroy(group->g);
        delete group;
    }
    return VQHIP_OK;
}

// from ANY thread, no lock taken (the thread that owns the communicator may be blocked inside a collective with it)
int vqhip_comm_abort(vqhip_comm *comm) {
    if (!comm || !comm->c) return VQHIP_OK;
    comm_abort(comm->c, "aborted by the caller (vqhip_comm_abort)");
    comm_abort_rccl(comm->c);
    return VQHIP_OK;
}

int vqhip_comm_kind(const vqhip_comm *comm, int *kind) {
    if (!kind) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    *kind = comm_kind(comm ? comm->c : nullptr);
    return VQHIP_OK;
}

static int kmeans_allreduce_impl(vqhip_kmeans *km, vqhip_comm *comm) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(km->sync);
    std::unique_lock<std::recursive_mutex> lk_comm;
    if (comm) lk_comm = std::unique_lock<std::recursive_mutex>(comm->mu);
    if (!km->accumulated) return fail(VQHIP_ERR_INVALID_INPUT, "allreduce without a preceding accumulate");
    int world = 1;
    comm_info(comm ? comm->c : nullptr, &world, nullptr);
    if (world > 1 && km->exact_update)
        return fail(VQHIP_ERR_UNSUPPORTED, "exact_update sums rows in one sequential chain: single GPU only");
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    return comm_allreduce_f64(comm ? comm->c : nullptr, km->slab.as<double>(),
                              (size_t)km->cs.m * km->cs.k * (km->cs.sd + 1), s);
    VQ_API_END
}

int vqhip_kmeans_allreduce(vqhip_kmeans *km, vqhip_comm *comm) { return sharded_exit(comm, kmeans_allreduce_impl(km, comm)); }

static int kmeans_step_sharded_impl(vqhip_kmeans *km, vqhip_comm *comm, uint32_t *counts, uint8_t *changed) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    std::unique_lock<std::recursive_mutex> lk_comm;
    if (comm) lk_comm = std::unique_lock<std::recursive_mutex>(comm->mu);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_TRY(kmeans_accumulate_enqueue(km, s));
    VQ_TRY(kmeans_allreduce_impl(km, comm));
    VQ_TRY(kmeans_finalize_enqueue(km, s));
    VQ_TRY(spin_wait(s));
    in.synced();
    kmeans_finalize_collect(km, counts, changed);
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_kmeans_step_sharded(vqhip_kmeans *km, vqhip_comm *comm, uint32_t *counts, uint8_t *changed) { return sharded_exit(comm, kmeans_step_sharded_impl(km, comm, counts, changed)); }

// bits of the owned rows among `global_rows` [m][k] into xs_ws (device u32 [m][k][sd]); zeros elsewhere
static int gather_owned_enqueue(vqhip_kmeans *km, const uint64_t *global_rows, uint64_t row_offset, hipStream_t s) {
    const size_t cnt = (size_t)km->cs.m * km->cs.k;
    VQ_TRY(km->gather_ws.ensure(cnt * km->cs.sd * 4));
    VQ_HIP(hipMemcpyAsync(km->rows_tmp.p, global_rows, cnt * 8, hipMemcpyHostToDevice, s));
    return launch_gather_rows_owned(km->ds->X, km->ds->d, km->cs.m, km->cs.k, km->cs.sd, km->rows_tmp.as<uint64_t>(),
                                    row_offset, km->ds->n, km->gather_ws.as<uint32_t>(), s);
}

int vqhip_kmeans_gather_owned_rows(vqhip_kmeans *km, const uint64_t *global_rows, uint64_t row_offset, uint32_t *bits_out) {
    VQ_API_BEGIN
    if (!km || !global_rows || !bits_out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    VQ_TRY(gather_owned_enqueue(km, global_rows, row_offset, s));
    VQ_HIP(hipMemcpyAsync(bits_out, km->gather_ws.p, (size_t)km->cs.m * km->cs.k * km->cs.sd * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
    VQ_API_END
}

static int kmeans_init_from_global_rows_impl(vqhip_kmeans *km, vqhip_comm *comm, const uint64_t *global_rows, uint64_t row_offset) {
    VQ_API_BEGIN
    if (!km || !global_rows) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    std::unique_lock<std::recursive_mutex> lk_comm;
    if (comm) lk_comm = std::unique_lock<std::recursive_mutex>(comm->mu);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    const size_t words = (size_t)km->cs.m * km->cs.k * km->cs.sd;
    VQ_TRY(gather_owned_enqueue(km, global_rows, row_offset, s));
    VQ_TRY(comm_allreduce_u32(comm ? comm->c : nullptr, km->gather_ws.as<uint32_t>(), words, s));
    VQ_HIP(hipMemcpyAsync(km->cs.cb.p, km->gather_ws.p, words * 4, hipMemcpyDeviceToDevice, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    km->cs.prepared = false;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_kmeans_init_from_global_rows(vqhip_kmeans *km, vqhip_comm *comm, const uint64_t *global_rows, uint64_t row_offset) { return sharded_exit(comm, kmeans_init_from_global_rows_impl(km, comm, global_rows, row_offset)); }

static int kmeans_patch_from_global_row_impl(vqhip_kmeans *km, vqhip_comm *comm, uint32_t sub, uint32_t j, uint64_t global_row, uint64_t row_offset) {
    VQ_API_BEGIN
    if (!km) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (sub >= km->cs.m || j >= km->cs.k) return fail(VQHIP_ERR_INVALID_INPUT, "centroid (%u,%u) out of range", sub, j);
    VQ_TRY(require_gfx950());
    Entry in(km->sync);
    std::unique_lock<std::recursive_mutex> lk_comm;
    if (comm) lk_comm = std::unique_lock<std::recursive_mutex>(comm->mu);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    const uint32_t sd = km->cs.sd;
    VQ_TRY(km->gather_ws.ensure((size_t)km->cs.m * km->cs.k * sd * 4));
    VQ_HIP(hipMemcpyAsync(km->rows_tmp.p, &global_row, 8, hipMemcpyHostToDevice, s));
    // one (subspace, cluster) pair: the kernel's subspace 0 is column block `sub` of the rows
    VQ_TRY(launch_gather_rows_owned(km->ds->X + (size_t)sub * sd, km->ds->d, 1, 1, sd, km->rows_tmp.as<uint64_t>(),
                                    row_offset, km->ds->n, km->gather_ws.as<uint32_t>(), s));
    VQ_TRY(comm_allreduce_u32(comm ? comm->c : nullptr, km->gather_ws.as<uint32_t>(), sd, s));
    VQ_HIP(hipMemcpyAsync(km->cs.cb.as<float>() + ((size_t)sub * km->cs.k + j) * sd, km->gather_ws.p, (size_t)sd * 4,
                          hipMemcpyDeviceToDevice, s));
    VQ_HIP(hipStreamSynchronize(s));  // &global_row is a stack address
    in.synced();
    km->cs.prepared = false;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_kmeans_patch_from_global_row(vqhip_kmeans *km, vqhip_comm *comm, uint32_t sub, uint32_t j, uint64_t global_row, uint64_t row_offset) { return sharded_exit(comm, kmeans_patch_from_global_row_impl(km, comm, sub, j, global_row, row_offset)); }

// ----------------------------------------------------------------------- PQ encode ----
int vqhip_pq_encoder_create(const float *codebooks, uint32_t m, uint32_t k, uint32_t sub_dim, int metric,
                            vqhip_pq_encoder **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (!codebooks) return fail(VQHIP_ERR_NULL_PTR, "codebooks is NULL");
    if (m == 0 || k == 0 || sub_dim == 0) return fail(VQHIP_ERR_INVALID_INPUT, "m, k and sub_dim must be positive");
    if (k > kMaxCentroids) return fail(VQHIP_ERR_UNSUPPORTED, "k=%u > 65536: codes are at most two bytes per subspace", k);
    if (metric < VQHIP_SQUARED_EUCLIDEAN || metric > VQHIP_COSINE_UNCLAMPED) return fail(VQHIP_ERR_INVALID_INPUT, "unknown metric %d", metric);
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    std::unique_ptr<vqhip_pq_encoder> enc(new vqhip_pq_encoder());
    enc->metric = metric;
    VQ_TRY(enc->cs.init(m, k, sub_dim));
    enc->cs.metric = (metric == VQHIP_COSINE) ? VQHIP_COSINE : VQHIP_SQUARED_EUCLIDEAN;
    VQ_HIP(hipMemcpyAsync(enc->cs.cb.p, codebooks, (size_t)m * k * sub_dim * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < m; ++i) enc->all_subs.push_back(i);
    *out = enc.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_pq_encoder_destroy(vqhip_pq_encoder *enc) {
    delete enc;
    return VQHIP_OK;
}

int vqhip_pq_encoder_set_engine(vqhip_pq_encoder *enc, int engine) {
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (engine < VQHIP_ENGINE_AUTO || engine > VQHIP_ENGINE_MFMA_BF16) return fail(VQHIP_ERR_INVALID_INPUT, "unknown engine %d", engine);
    std::lock_guard<std::recursive_mutex> lk(enc->sync.mu);
    enc->engine = engine;
    return VQHIP_OK;
}

int vqhip_pq_encode_device(vqhip_pq_encoder *enc, const void *dev_rows, uint64_t n, void *dev_codes, void *dev_f16_out) {
    VQ_API_BEGIN
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!dev_rows) return fail(VQHIP_ERR_NULL_PTR, "dev_rows is NULL");
    VQ_TRY(require_gfx950());
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    uint8_t *codes = reinterpret_cast<uint8_t *>(dev_codes);
    if (!codes) {  // f16-only output still needs the codes internally
        VQ_TRY(enc->codes.ensure((size_t)n * enc->cs.m * code_bytes(enc->cs.k)));
        codes = enc->codes.as<uint8_t>();
    }
    g_last_ws = enc->ws_id;
    VQ_TRY(run_assign(enc->cs, enc->ws, reinterpret_cast<const float *>(dev_rows), n, enc->cs.m * enc->cs.sd,
                      enc->metric, enc->all_subs, codes, enc->engine, s));
    if (dev_f16_out) VQ_TRY(launch_gather_f16(enc->cs.view(), codes, n, reinterpret_cast<uint16_t *>(dev_f16_out), s));
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_pq_encode(vqhip_pq_encoder *enc, const float *rows, uint64_t n, uint8_t *codes, uint16_t *f16_out) {
    VQ_API_BEGIN
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    VQ_TRY(require_gfx950());
    Entry in(enc->sync);
    hipStream_t s;
    const uint32_t m = enc->cs.m, d = enc->cs.m * enc->cs.sd;
    const size_t cw = code_bytes(enc->cs.k);  // bytes per code: 1 (k <= 256) or 2
    static const char *no_small = getenv("VQHIP_NO_SMALL_PATH");
    const bool small = n <= kSmallRows && !(no_small && no_small[0] == '1');
    // (the per-vector path needs no order behind the handle's queued batch work once its images are known to be complete:
    // it shares nothing mutable with it)
    if (small && enc->small_ready) VQ_TRY(current_stream(&s));
    else VQ_TRY(in.stream(&s));
    if (small) {
        // per-vector calls (Quantizer::quantize): one kernel over mapped pinned memory, exact arithmetic.  The kernel
        // reads only what an encoder never changes after its first use (codebooks, centroid norms), and the staging
        // belongs to the call: the handle is given back before the launch, so calls from many threads on one encoder
        // (`quantize(&self)` on a Sync type, src/pq.rs:39-45) run side by side on their threads' streams.
        const size_t in_b = (size_t)n * d * 4, code_b = ((size_t)n * m * cw + 15) & ~(size_t)15, f16_b = (size_t)n * d * 2;
        if (!enc->small_ready) {
            // once per encoder: s is ordered behind the handle's tail here (in.stream), so the wait also covers a prepare
            // another thread's asynchronous batch call enqueued on ITS stream; other threads' streams read the images
            // without any order from now on
            if (!(enc->cs.prepared && enc->cs.prepared_base)) VQ_TRY(enc->cs.prepare(s));  // centroid norms (cosine)
            VQ_HIP(hipStreamSynchronize(s));
            in.synced();
            enc->small_ready = true;
        }
        const int metric = enc->metric;
        const uint32_t k = enc->cs.k, sd = enc->cs.sd;
        const float *cb = enc->cs.cb.as<float>(), *cnsqrt = enc->cs.cnsqrt.as<float>();
        enc->ws.stats_pending = false;
        enc->ws.last_engine = VQHIP_ENGINE_EXACT;
        in.release();
        StageLease stage;
        VQ_TRY(stage.acquire(in_b + code_b + f16_b));
        char *hb = stage.host(), *db = stage.dev();
        memcpy(hb, rows, in_b);
        VQ_TRY(launch_pq_encode_small(reinterpret_cast<const float *>(db), (uint32_t)n, d, m, k, sd, metric, cb, cnsqrt,
                                      reinterpret_cast<uint8_t *>(db + in_b),
                                      f16_out ? reinterpret_cast<uint16_t *>(db + in_b + code_b) : nullptr, s));
        VQ_TRY(spin_wait(s));
        if (codes) memcpy(codes, hb + in_b, (size_t)n * m * cw);
        if (f16_out) memcpy(f16_out, hb + in_b + code_b, f16_b);
        ThreadState &st = tls();
        st.last_engine = VQHIP_ENGINE_EXACT;
        st.last_rechecked = 0;
        g_last_ws = 0;
        return VQHIP_OK;
    }
    if (xfer_lanes_pay((size_t)n * d * 4, (codes ? (size_t)n * m * cw : 0) + (f16_out ? (size_t)n * d * 2 : 0))) {
        // large host batch with the f16 reconstruction coming back: alternate chunks on two lanes (see XferLane), so the
        // results of one chunk travel while the rows of the next do.  The lanes call vqhip_pq_encode_device themselves,
        // from their own threads: this call gives the handle back first (its lock is per thread; whatever it still has
        // queued stays its tail) and the handle orders the lanes' launches.
        in.release();
        const size_t row_b = (size_t)d * 4;
        const uint64_t per = std::max<uint64_t>(1, xfer_chunk_bytes() / row_b), chunks = (n + per - 1) / per;
        std::mutex h2d_turn;  // one lane's rows on the bus at a time: the lanes stay out of phase (both copying in, then both out, overlaps nothing: 13.3 against 10.6 ms)
        const int rc_lanes = run_lanes([&](int t, XferLane &ln, int n_lanes) -> int {
            VQ_TRY(ln.dev_in.ensure((size_t)per * row_b));
            VQ_TRY(ln.dev_out.ensure((size_t)per * m * cw));
            if (f16_out) VQ_TRY(ln.dev_out2.ensure((size_t)per * d * 2));
            for (uint64_t c = (uint64_t)t; c < chunks; c += (uint64_t)n_lanes) {
                const uint64_t r0 = c * per, nr = std::min(per, n - r0);
                {
                    std::lock_guard<std::mutex> turn(h2d_turn);
                    VQ_HIP(hipMemcpyAsync(ln.dev_in.p, rows + r0 * d, (size_t)nr * row_b, hipMemcpyHostToDevice, ln.stream));
                    VQ_HIP(hipStreamSynchronize(ln.stream));
                }
                VQ_TRY(vqhip_pq_encode_device(enc, ln.dev_in.p, nr, ln.dev_out.p, f16_out ? ln.dev_out2.p : nullptr));
                if (codes) VQ_HIP(hipMemcpyAsync(codes + r0 * m * cw, ln.dev_out.p, (size_t)nr * m * cw, hipMemcpyDeviceToHost, ln.stream));
                if (f16_out) VQ_HIP(hipMemcpyAsync(f16_out + r0 * d, ln.dev_out2.p, (size_t)nr * d * 2, hipMemcpyDeviceToHost, ln.stream));
                VQ_HIP(hipStreamSynchronize(ln.stream));
            }
            return VQHIP_OK;
        });
        // "the most recent pass of this thread" for vqhip_last_assign_stats: the lanes' passes were this call's
        g_last_ws = enc->ws_id;
        {
            std::lock_guard<std::recursive_mutex> lk(enc->sync.mu);
            tls().last_engine = enc->ws.last_engine;
        }
        return rc_lanes;
    }
    // bounded staging: at most ~1 GiB of rows per pass
    uint64_t chunk = std::max<uint64_t>(1, (1ull << 30) / ((uint64_t)d * 4));
    if (chunk > n) chunk = n;
    VQ_TRY(enc->xbuf.ensure((size_t)chunk * d * 4));
    VQ_TRY(enc->codes.ensure((size_t)chunk * m * cw));
    if (f16_out) VQ_TRY(enc->f16buf.ensure((size_t)chunk * d * 2));
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint64_t nr = std::min(chunk, n - r0);
        VQ_HIP(hipMemcpyAsync(enc->xbuf.p, rows + r0 * d, (size_t)nr * d * 4, hipMemcpyHostToDevice, s));
        VQ_TRY(vqhip_pq_encode_device(enc, enc->xbuf.p, nr, enc->codes.p, f16_out ? enc->f16buf.p : nullptr));
        if (codes) VQ_HIP(hipMemcpyAsync(codes + r0 * m * cw, enc->codes.p, (size_t)nr * m * cw, hipMemcpyDeviceToHost, s));
        if (f16_out) VQ_HIP(hipMemcpyAsync(f16_out + r0 * d, enc->f16buf.p, (size_t)nr * d * 2, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipStreamSynchronize(s));
    }
    in.synced();
    return VQHIP_OK;
    VQ_API_END
}

// ------------------------------------------------------------------------ ADC search ----
int vqhip_pq_adc_search_device(vqhip_pq_encoder *enc, const void *dev_codes, uint64_t n, const float *queries,
                               uint32_t nq, uint32_t topk, uint32_t *idx_out, float *dist_out) {
    VQ_API_BEGIN
    if (!enc || !dev_codes || !queries || !idx_out || !dist_out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (nq == 0) return VQHIP_OK;
    if (n == 0 || n >= (1ull << 32)) return fail(VQHIP_ERR_INVALID_INPUT, "n must be in [1, 2^32)");
    VQ_TRY(require_gfx950());
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    const uint32_t m = enc->cs.m, k = enc->cs.k, sd = enc->cs.sd, dim = m * sd;
    constexpr uint32_t kAdcCallQueries = 4096;  // per set of launches: the candidate lists are 64 KB per query
    if (nq > kAdcCallQueries) {
        uint32_t redone = 0;
        for (uint32_t q0 = 0; q0 < nq; q0 += kAdcCallQueries) {
            const uint32_t cn = std::min(kAdcCallQueries, nq - q0);
            VQ_TRY(vqhip_pq_adc_search_device(enc, dev_codes, n, queries + (size_t)q0 * dim, cn, topk, idx_out + (size_t)q0 * topk,
                                              dist_out + (size_t)q0 * topk));
            redone += enc->adc_last_redone;
        }
        enc->adc_last_redone = redone;
        return VQHIP_OK;
    }
    VQ_TRY(enc->adc_q.ensure((size_t)nq * dim * 4));
    VQ_TRY(enc->adc_idx.ensure((size_t)nq * topk * 4));
    VQ_TRY(enc->adc_out.ensure((size_t)nq * topk * 4));
    // Small calls stage through pinned, device-mapped memory: the queries leave as one asynchronous DMA and the results
    // are WRITTEN into host memory by the last kernel -- a copy from or into pageable memory is a staged transfer inside
    // the runtime that holds the calling thread (~12 us each; an 8-query call spent more time in its four copies than in
    // its kernels)
    const size_t q_b = (size_t)nq * dim * 4, nres = (size_t)nq * topk, pack_b = (2 * nres + 2 * (size_t)nq) * 4;
    const bool fast = adc_fast_eligible(m, k, n, topk), staged = fast && q_b + pack_b <= (1u << 20);
    StageLease stage;
    if (staged) {
        VQ_TRY(stage.acquire(q_b + pack_b));
        memcpy(stage.host(), queries, q_b);  // (the table kernel reads them from there: no copy engine in front of the kernels)
    } else {
        VQ_HIP(hipMemcpyAsync(enc->adc_q.p, queries, q_b, hipMemcpyHostToDevice, s));
    }
    // the full pass (every distance written, histogram cut, candidates collected) for the queries [q0, q0 + cnt)
    auto full_pass = [&](uint32_t q0, uint32_t cnt) -> int {
        const uint32_t qg = adc_query_group(n, cnt);
        VQ_TRY(enc->adc_lut.ensure((size_t)qg * m * k * 4));
        VQ_TRY(enc->adc_dist.ensure((size_t)qg * n * 4));
        VQ_TRY(enc->adc_state.ensure(adc_state_bytes(qg)));
        VQ_TRY(enc->adc_cand.ensure(adc_cand_bytes(qg)));
        return launch_adc_search(enc->cs.cb.as<float>(), m, k, sd, enc->metric, reinterpret_cast<const uint8_t *>(dev_codes), n,
                                 enc->adc_q.as<float>() + (size_t)q0 * dim, cnt, topk, enc->adc_lut.as<float>(), enc->adc_dist.as<float>(),
                                 enc->adc_state.p, enc->adc_cand.as<unsigned long long>(), enc->adc_idx.as<uint32_t>() + (size_t)q0 * topk,
                                 enc->adc_out.as<float>() + (size_t)q0 * topk, s, qg);
    };
    if (fast) {
        // one scan of the codes per batch of queries against a sampled threshold, only candidates written (k_adc.hip);
        // a query whose threshold let too few or too many rows pass is repeated through the full pass
        VQ_TRY(enc->adc_lut.ensure(adc_fast_lut_bytes(m, k, nq)));
        VQ_TRY(enc->adc_state.ensure(adc_fast_state_bytes(m, k, nq)));
        VQ_TRY(enc->adc_cand.ensure(adc_fast_cand_bytes(m, k, nq)));
        // results, flags and candidate counts in ONE buffer: [nq][topk] idx | [nq][topk] dist | [nq] redo | [nq] count -- one
        // copy back instead of three (each a staged transfer of its own into pageable memory, ~12 us)
        if (!staged) VQ_TRY(enc->adc_redo.ensure(pack_b));
        uint32_t *pack_dev = staged ? reinterpret_cast<uint32_t *>(stage.dev() + q_b) : enc->adc_redo.as<uint32_t>();
        VQ_TRY(launch_adc_search_fast(enc->cs.cb.as<float>(), m, k, sd, enc->metric, reinterpret_cast<const uint8_t *>(dev_codes), n,
                                      staged ? reinterpret_cast<const float *>(stage.dev()) : enc->adc_q.as<float>(), nq, topk, enc->adc_lut.as<float>(), enc->adc_state.p,
                                      enc->adc_cand.as<unsigned long long>(), pack_dev, reinterpret_cast<float *>(pack_dev + nres),
                                      pack_dev + 2 * nres, s));
        std::vector<uint32_t> pack_copy;
        const uint32_t *pack = nullptr;
        if (staged) {
            VQ_TRY(spin_wait(s));
            pack = reinterpret_cast<const uint32_t *>(stage.host() + q_b);
        } else {
            pack_copy.resize(2 * nres + 2 * (size_t)nq);
            VQ_HIP(hipMemcpyAsync(pack_copy.data(), pack_dev, pack_b, hipMemcpyDeviceToHost, s));
            VQ_HIP(hipStreamSynchronize(s));
            pack = pack_copy.data();
        }
        memcpy(idx_out, pack, nres * 4);
        memcpy(dist_out, pack + nres, nres * 4);
        const uint32_t *redo = pack + 2 * nres;
        uint32_t redone = 0;
        for (uint32_t q0 = 0; q0 < nq;) {
            if (!redo[q0]) {
                ++q0;
                continue;
            }
            uint32_t q1 = q0;
            while (q1 < nq && redo[q1]) ++q1;  // a run of flagged queries goes through together
            if (staged && !redone) VQ_HIP(hipMemcpyAsync(enc->adc_q.p, queries, q_b, hipMemcpyHostToDevice, s));  // (the full pass reads them from device memory)
            VQ_TRY(full_pass(q0, q1 - q0));
            VQ_HIP(hipMemcpyAsync(idx_out + (size_t)q0 * topk, enc->adc_idx.as<uint32_t>() + (size_t)q0 * topk, (size_t)(q1 - q0) * topk * 4, hipMemcpyDeviceToHost, s));
            VQ_HIP(hipMemcpyAsync(dist_out + (size_t)q0 * topk, enc->adc_out.as<float>() + (size_t)q0 * topk, (size_t)(q1 - q0) * topk * 4, hipMemcpyDeviceToHost, s));
            VQ_HIP(hipStreamSynchronize(s));
            redone += q1 - q0;
            q0 = q1;
        }
        enc->adc_last_redone = redone;
        in.synced();
        return VQHIP_OK;
    }
    VQ_TRY(full_pass(0, nq));
    enc->adc_last_redone = nq;
    VQ_HIP(hipMemcpyAsync(idx_out, enc->adc_idx.p, (size_t)nq * topk * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipMemcpyAsync(dist_out, enc->adc_out.p, (size_t)nq * topk * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
    VQ_API_END
}

// A code store that is searched again and again (vq_amd.store.PQIndex): its codes uploaded ONCE into the encoder's own
// buffer, every later search is the device form over them (an 8 MB upload per call is 0.15 ms in front of a 0.06 ms search).
int vqhip_pq_adc_set_codes(vqhip_pq_encoder *enc, const uint8_t *codes, uint64_t n) {
    VQ_API_BEGIN
    if (!enc || (!codes && n)) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (n >= (1ull << 32)) return fail(VQHIP_ERR_INVALID_INPUT, "n must be below 2^32");
    VQ_TRY(require_gfx950());
    const uint32_t m = enc->cs.m, k = enc->cs.k, cw = code_bytes(k);
    // the scan indexes its LDS tables by code: checked here, once, not per search
    for (uint64_t i = 0; i < n * m; ++i) {
        const uint32_t c = cw == 1 ? codes[i] : reinterpret_cast<const uint16_t *>(codes)[i];
        if (c >= k) return fail(VQHIP_ERR_INVALID_INPUT, "code %u at element %llu is outside [0, %u)", c, (unsigned long long)i, k);
    }
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    enc->adc_resident_n = 0;
    if (n) {
        VQ_TRY(enc->adc_resident.ensure((size_t)n * m * cw));
        VQ_HIP(hipMemcpyAsync(enc->adc_resident.p, codes, (size_t)n * m * cw, hipMemcpyHostToDevice, s));
        VQ_HIP(hipStreamSynchronize(s));
        in.synced();
    }
    enc->adc_resident_n = n;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_pq_adc_search_resident(vqhip_pq_encoder *enc, const float *queries, uint32_t nq, uint32_t topk, uint32_t *idx_out,
                                 float *dist_out) {
    VQ_API_BEGIN
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(enc->sync);
    if (!enc->adc_resident_n) return fail(VQHIP_ERR_INVALID_INPUT, "no codes loaded: call vqhip_pq_adc_set_codes first");
    return vqhip_pq_adc_search_device(enc, enc->adc_resident.p, enc->adc_resident_n, queries, nq, topk, idx_out, dist_out);
    VQ_API_END
}

int vqhip_pq_adc_last_redone(vqhip_pq_encoder *enc, uint32_t *queries_out) {
    VQ_API_BEGIN
    if (!enc || !queries_out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(enc->sync);
    *queries_out = enc->adc_last_redone;
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_pq_adc_search(vqhip_pq_encoder *enc, const uint8_t *codes, uint64_t n, const float *queries, uint32_t nq,
                        uint32_t topk, uint32_t *idx_out, float *dist_out) {
    VQ_API_BEGIN
    if (!enc || !codes) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    const size_t code_b = (size_t)n * enc->cs.m * code_bytes(enc->cs.k);
    VQ_TRY(enc->adc_codes.ensure(code_b));
    VQ_HIP(hipMemcpyAsync(enc->adc_codes.p, codes, code_b, hipMemcpyHostToDevice, s));
    return vqhip_pq_adc_search_device(enc, enc->adc_codes.p, n, queries, nq, topk, idx_out, dist_out);
    VQ_API_END
}

int vqhip_dequantize_f16(const uint16_t *f16_in, uint64_t count, float *out) {
    VQ_API_BEGIN
    if (count == 0) return VQHIP_OK;
    if (!f16_in || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    DevBuf in, o;
    VQ_TRY(in.alloc((size_t)count * 2));
    VQ_TRY(o.alloc((size_t)count * 4));
    VQ_HIP(hipMemcpyAsync(in.p, f16_in, (size_t)count * 2, hipMemcpyHostToDevice, s));
    VQ_TRY(launch_dequant_f16(in.as<uint16_t>(), count, o.as<float>(), s));
    VQ_HIP(hipMemcpyAsync(out, o.p, (size_t)count * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_pq_decode(vqhip_pq_encoder *enc, const uint8_t *codes, uint64_t n, float *out) {
    VQ_API_BEGIN
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!codes || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    const uint32_t m = enc->cs.m, d = enc->cs.m * enc->cs.sd;
    const size_t cw = code_bytes(enc->cs.k);
    for (uint64_t i = 0; i < n * m; ++i) {
        uint32_t c = codes[i];
        if (cw == 2) {
            uint16_t w;
            memcpy(&w, codes + 2 * i, 2);
            c = w;
        }
        if (c >= enc->cs.k) return fail(VQHIP_ERR_INVALID_INPUT, "code %u >= k=%u", c, enc->cs.k);
    }
    VQ_TRY(enc->codes.ensure((size_t)n * m * cw));
    VQ_TRY(enc->f32buf.ensure((size_t)n * d * 4));
    VQ_HIP(hipMemcpyAsync(enc->codes.p, codes, (size_t)n * m * cw, hipMemcpyHostToDevice, s));
    VQ_TRY(launch_decode_f32(enc->cs.view(), enc->codes.as<uint8_t>(), n, enc->f32buf.as<float>(), s));
    VQ_HIP(hipMemcpyAsync(out, enc->f32buf.p, (size_t)n * d * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    in.synced();
    return VQHIP_OK;
    VQ_API_END
}

// device forms (asynchronous on the current stream, like vqhip_pq_encode_device): what a host that keeps codes / f16 rows
// resident calls, and what bench.py times.  Codes are the caller's to keep inside [0, k): an out-of-range code reads a
// wrong (in-bounds of the allocation or not) codebook row -- the host form checks, this one cannot without a read-back.
int vqhip_pq_decode_device(vqhip_pq_encoder *enc, const void *dev_codes, uint64_t n, void *dev_out) {
    VQ_API_BEGIN
    if (!enc) return fail(VQHIP_ERR_NULL_PTR, "encoder is NULL");
    if (n == 0) return VQHIP_OK;
    if (!dev_codes || !dev_out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    Entry in(enc->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    return launch_decode_f32(enc->cs.view(), static_cast<const uint8_t *>(dev_codes), n, static_cast<float *>(dev_out), s);
    VQ_API_END
}

int vqhip_dequantize_f16_device(const void *dev_f16_in, uint64_t count, void *dev_out) {
    VQ_API_BEGIN
    if (count == 0) return VQHIP_OK;
    if (!dev_f16_in || !dev_out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    return launch_dequant_f16(static_cast<const uint16_t *>(dev_f16_in), count, static_cast<float *>(dev_out), s);
    VQ_API_END
}

int vqhip_distance_batch(int metric, const float *a, const float *b, uint64_t n, uint32_t d, float *out) {
    VQ_API_BEGIN
    if (n == 0) return VQHIP_OK;
    if (!a || !b || !out) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (metric < VQHIP_SQUARED_EUCLIDEAN || metric > VQHIP_COSINE_UNCLAMPED) return fail(VQHIP_ERR_INVALID_INPUT, "unknown metric %d", metric);
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    DevBuf da, db, dout;
    VQ_TRY(da.alloc((size_t)n * d * 4));
    VQ_TRY(db.alloc((size_t)n * d * 4));
    VQ_TRY(dout.alloc((size_t)n * 4));
    VQ_HIP(hipMemcpyAsync(da.p, a, (size_t)n * d * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipMemcpyAsync(db.p, b, (size_t)n * d * 4, hipMemcpyHostToDevice, s));
    VQ_TRY(launch_distance_batch(metric, da.as<float>(), db.as<float>(), n, d, dout.as<float>(), s));
    VQ_HIP(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    VQ_HIP(hipStreamSynchronize(s));
    return VQHIP_OK;
    VQ_API_END
}

// ---------------------------------------------------------------------------- TSVQ ----
int vqhip_tsvq_build(const vqhip_dataset *ds, uint32_t max_depth, uint32_t cap, float *centroids, int32_t *left,
                     int32_t *right, int32_t *n_nodes) {
    VQ_API_BEGIN
    if (!ds || !centroids || !left || !right || !n_nodes) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    return tsvq_build_device(ds->X, ds->n, ds->d, max_depth, cap, centroids, left, right, n_nodes, s, ds->own.p ? &ds->tsvq_policy : nullptr);
    VQ_API_END
}

int vqhip_tsvq_create(const float *centroids, const int32_t *left, const int32_t *right, uint32_t n_nodes, uint32_t d,
                      int metric, vqhip_tsvq **out) {
    VQ_API_BEGIN
    if (!out) return fail(VQHIP_ERR_NULL_PTR, "out is NULL");
    *out = nullptr;
    if (!centroids || !left || !right) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    if (n_nodes == 0 || d == 0) return fail(VQHIP_ERR_INVALID_INPUT, "empty tree");
    if (metric < VQHIP_SQUARED_EUCLIDEAN || metric > VQHIP_COSINE_UNCLAMPED) return fail(VQHIP_ERR_INVALID_INPUT, "unknown metric %d", metric);
    {
        // the arrays must describe ONE tree rooted at node 0: children point forward (pre-order, so the
        // descent terminates), a node's two children differ, no node has two parents and every node but
        // the root has one (the encoder's breadth-first images are sized by the node count)
        std::vector<uint8_t> has_parent(n_nodes, 0);
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const int32_t ch[2] = {left[i], right[i]};
            if (ch[0] >= 0 && ch[0] == ch[1]) return fail(VQHIP_ERR_INVALID_INPUT, "node %u names node %d as both children", i, ch[0]);
            for (int c = 0; c < 2; ++c) {
                if (ch[c] < 0) continue;
                if (ch[c] <= (int32_t)i || ch[c] >= (int32_t)n_nodes)
                    return fail(VQHIP_ERR_INVALID_INPUT, "node %u has an out-of-order child index", i);
                if (has_parent[ch[c]]) return fail(VQHIP_ERR_INVALID_INPUT, "node %d has two parents", ch[c]);
                has_parent[ch[c]] = 1;
            }
        }
        for (uint32_t i = 1; i < n_nodes; ++i)
            if (!has_parent[i]) return fail(VQHIP_ERR_INVALID_INPUT, "node %u is not reachable from the root", i);
    }
    VQ_TRY(require_gfx950());
    hipStream_t s;
    VQ_TRY(current_stream(&s));
    std::unique_ptr<vqhip_tsvq> t(new vqhip_tsvq());
    t->n_nodes = n_nodes;
    t->d = d;
    t->metric = metric;
    VQ_TRY(t->centroids.alloc((size_t)n_nodes * d * 4));
    VQ_TRY(t->left.alloc((size_t)n_nodes * 4));
    VQ_TRY(t->right.alloc((size_t)n_nodes * 4));
    VQ_HIP(hipMemcpyAsync(t->centroids.p, centroids, (size_t)n_nodes * d * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipMemcpyAsync(t->left.p, left, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
    VQ_HIP(hipMemcpyAsync(t->right.p, right, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
    VQ_TRY(t->cnorm.alloc((size_t)n_nodes * 4));
    VQ_TRY(launch_tsvq_node_norms(t->centroids.as<float>(), n_nodes, d, t->cnorm.as<float>(), s));
    VQ_TRY(t->table16.alloc((size_t)n_nodes * d * 2));
    VQ_TRY(launch_tsvq_table_f16(t->centroids.as<float>(), n_nodes, d, t->table16.as<uint16_t>(), s));
    VQ_TRY(tsvq_prepare_screen(t.get(), centroids, left, right, s));
    VQ_HIP(hipStreamSynchronize(s));
    *out = t.release();
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_tsvq_destroy(vqhip_tsvq *t) {
    delete t;
    return VQHIP_OK;
}

int vqhip_tsvq_encode_device(vqhip_tsvq *t, const void *dev_rows, uint64_t n, void *dev_leaf, void *dev_f16_out) {
    VQ_API_BEGIN
    if (!t) return fail(VQHIP_ERR_NULL_PTR, "tree is NULL");
    if (n == 0) return VQHIP_OK;
    if (!dev_rows) return fail(VQHIP_ERR_NULL_PTR, "dev_rows is NULL");
    VQ_TRY(require_gfx950());
    Entry in(t->sync);
    hipStream_t s;
    VQ_TRY(in.stream(&s));
    int32_t *leaf = reinterpret_cast<int32_t *>(dev_leaf);
    if (!leaf) {
        VQ_TRY(t->leafbuf.ensure((size_t)n * 4));
        leaf = t->leafbuf.as<int32_t>();
    }
    t->last_screened = false;
    const float *X = reinterpret_cast<const float *>(dev_rows);
    uint16_t *out = reinterpret_cast<uint16_t *>(dev_f16_out);
    const bool out_vec = out && t->d % 8 == 0 && (reinterpret_cast<uintptr_t>(dev_f16_out) & 15) == 0;
    bool out_done = false;
    if (t->use_screen && n <= 0xFFFFFFFFull && (reinterpret_cast<uintptr_t>(dev_rows) & 15) == 0) {
        t->last_screened = true;
        VQ_TRY(t->scr_wl.ensure((size_t)n * 8));
        t->scr.wl = t->scr_wl.as<uint2>();
        // the reconstruction rides on the descent when it can be written in 16-byte pieces
        VQ_TRY(launch_tsvq_screen_encode(X, n, t->d, t->centroids.as<float>(), t->cnorm.as<float>(), t->left.as<int32_t>(),
                                         t->right.as<int32_t>(), t->metric, t->scr, leaf, s,
                                         out_vec ? t->table16.as<uint16_t>() : nullptr, out_vec ? out : nullptr));
        out_done = out_vec;
    } else {
        VQ_TRY(launch_tsvq_encode(X, n, t->d, t->centroids.as<float>(), t->cnorm.as<float>(), t->left.as<int32_t>(),
                                  t->right.as<int32_t>(), t->metric, leaf, s));
    }
    if (out && !out_done) {
        if (out_vec)
            VQ_TRY(launch_tsvq_gather_table(t->table16.as<uint16_t>(), t->d, leaf, n, out, s));
        else
            VQ_TRY(launch_tsvq_gather_f16(t->centroids.as<float>(), t->d, leaf, n, out, s));
    }
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_tsvq_last_stats(vqhip_tsvq *t, int *screened, uint64_t *undecided) {
    VQ_API_BEGIN
    if (!t || !screened || !undecided) return fail(VQHIP_ERR_NULL_PTR, "NULL argument");
    Entry in(t->sync);
    *screened = t->last_screened ? 1 : 0;
    *undecided = 0;
    if (t->last_screened) {
        hipStream_t s;
        VQ_TRY(in.stream(&s));
        uint32_t c = 0;
        VQ_HIP(hipMemcpyAsync(&c, t->scr_count.as<uint32_t>() + (t->scr.turn ^ 1u), 4, hipMemcpyDeviceToHost, s));  // the last call's counter
        VQ_HIP(hipStreamSynchronize(s));
        in.synced();
        *undecided = c;
    }
    return VQHIP_OK;
    VQ_API_END
}

int vqhip_tsvq_encode(vqhip_tsvq *t, const float *rows, uint64_t n, int32_t *leaf, uint16_t *f16_out) {
    VQ_API_BEGIN
    if (!t) return fail(VQHIP_ERR_NULL_PTR, "tree is NULL");
    if (n == 0) return VQHIP_OK;
    if (!rows) return fail(VQHIP_ERR_NULL_PTR, "rows is NULL");
    VQ_TRY(require_gfx950());
    Entry in(t->sync);
    hipStream_t s;
    const uint32_t d = t->d;
    static const char *no_small = getenv("VQHIP_NO_SMALL_PATH");
    const bool small = n <= kSmallRows && tsvq_small_supported(d) && !(no_small && no_small[0] == '1');
    if (small) VQ_TRY(current_stream(&s));  // nothing mutable shared with the handle's queued batch work: no order needed
    else VQ_TRY(in.stream(&s));
    if (small) {
        // per-vector calls: the kernel reads the tree only (fixed at creation, complete before the handle existed) and
        // stages through a buffer of its own, so the handle is given back before the launch (see vqhip_pq_encode)
        const size_t in_b = (size_t)n * d * 4, leaf_b = ((size_t)n * 4 + 15) & ~(size_t)15, f16_b = (size_t)n * d * 2;
        t->last_screened = false;
        in.release();
        StageLease stage;
        VQ_TRY(stage.acquire(in_b + leaf_b + f16_b));
        char *hb = stage.host(), *db = stage.dev();
        memcpy(hb, rows, in_b);
        VQ_TRY(launch_tsvq_encode_small(reinterpret_cast<const float *>(db), (uint32_t)n, d, t->metric,
                                        t->centroids.as<float>(), t->cnorm.as<float>(), t->left.as<int32_t>(),
                                        t->right.as<int32_t>(), reinterpret_cast<int32_t *>(db + in_b),
                                        f16_out ? reinterpret_cast<uint16_t *>(db + in_b + leaf_b) : nullptr, s));
        VQ_TRY(spin_wait(s));
        if (leaf) memcpy(leaf, hb + in_b, (size_t)n * 4);
        if (f16_out) memcpy(f16_out, hb + in_b + leaf_b, f16_b);
        return VQHIP_OK;
    }
    if (xfer_lanes_pay((size_t)n * d * 4, (leaf ? (size_t)n * 4 : 0) + (f16_out ? (size_t)n * d * 2 : 0))) {  // two lanes, as vqhip_pq_encode
        in.release();
        const size_t row_b = (size_t)d * 4;
        const uint64_t per = std::max<uint64_t>(1, xfer_chunk_bytes() / row_b), chunks = (n + per - 1) / per;
        std::mutex h2d_turn;
        return run_lanes([&](int tl, XferLane &ln, int n_lanes) -> int {
            VQ_TRY(ln.dev_in.ensure((size_t)per * row_b));
            VQ_TRY(ln.dev_out.ensure((size_t)per * 4));
            if (f16_out) VQ_TRY(ln.dev_out2.ensure((size_t)per * d * 2));
            for (uint64_t c = (uint64_t)tl; c < chunks; c += (uint64_t)n_lanes) {
                const uint64_t r0 = c * per, nr = std::min(per, n - r0);
                {
                    std::lock_guard<std::mutex> turn(h2d_turn);
                    VQ_HIP(hipMemcpyAsync(ln.dev_in.p, rows + r0 * d, (size_t)nr * row_b, hipMemcpyHostToDevice, ln.stream));
                    VQ_HIP(hipStreamSynchronize(ln.stream));
                }
                VQ_TRY(vqhip_tsvq_encode_device(t, ln.dev_in.p, nr, ln.dev_out.p, f16_out ? ln.dev_out2.p : nullptr));
                if (leaf) VQ_HIP(hipMemcpyAsync(leaf + r0, ln.dev_out.p, (size_t)nr * 4, hipMemcpyDeviceToHost, ln.stream));
                if (f16_out) VQ_HIP(hipMemcpyAsync(f16_out + r0 * d, ln.dev_out2.p, (size_t)nr * d * 2, hipMemcpyDeviceToHost, ln.stream));
                VQ_HIP(hipStreamSynchronize(ln.stream));
            }
            return VQHIP_OK;
        });
    }
    uint64_t chunk = std::max<uint64_t>(1, (1ull << 30) / ((uint64_t)d * 4));
    if (chunk > n) chunk = n;
    VQ_TRY(t->xbuf.ensure((size_t)chunk * d * 4));
    DevBuf leafdev;
    VQ_TRY(leafdev.alloc((size_t)chunk * 4));
    if (f16_out) VQ_TRY(t->f16buf.ensure((size_t)chunk * d * 2));
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint64_t nr = std::min(chunk, n - r0);
        VQ_HIP(hipMemcpyAsync(t->xbuf.p, rows + r0 * d, (size_t)nr * d * 4, hipMemcpyHostToDevice, s));
        VQ_TRY(vqhip_tsvq_encode_device(t, t->xbuf.p, nr, leafdev.p, f16_out ? t->f16buf.p : nullptr));
        if (leaf) VQ_HIP(hipMemcpyAsync(leaf + r0, leafdev.p, (size_t)nr * 4, hipMemcpyDeviceToHost, s));
        if (f16_out) VQ_HIP(hipMemcpyAsync(f16_out + r0 * d, t->f16buf.p, (size_t)nr * d * 2, hipMemcpyDeviceToHost, s));
        VQ_HIP(hipStreamSynchronize(s));
    }
    in.synced();
    return VQHIP_OK;
    VQ_API_END
}

}  // extern "C"
