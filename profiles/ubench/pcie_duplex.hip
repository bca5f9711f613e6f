// pcie_duplex.hip -- what the host <-> device paths of this pool's boxes deliver, to size the host-buffer entry points
// (vqhip_dataset_from_host, vqhip_pq_encode with host rows in / f16 out):
//   pageable vs pinned host memory, one direction at a time and both at once (two host threads, two streams).
// build: hipcc --offload-arch=gfx950 -O2 -o pcie_duplex pcie_duplex.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));               \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t in_b = (size_t)512 << 20, out_b = (size_t)256 << 20;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, in_b));
    CK(hipMalloc(&d_out, out_b));
    char *h_in = (char *)malloc(in_b), *h_out = (char *)malloc(out_b);
    memset(h_in, 1, in_b);
    memset(h_out, 2, out_b);
    char *p_in, *p_out;
    CK(hipHostMalloc((void **)&p_in, in_b));
    CK(hipHostMalloc((void **)&p_out, out_b));
    memset(p_in, 1, in_b);
    memset(p_out, 2, out_b);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto h2d = [&](const void *src, hipStream_t s) {
        CK(hipMemcpyAsync(d_in, src, in_b, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
    };
    auto d2h = [&](void *dst, hipStream_t s) {
        CK(hipMemcpyAsync(dst, d_out, out_b, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
    };
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        h2d(h_in, s1);
        double t1 = now();
        d2h(h_out, s2);
        double t2 = now();
        h2d(p_in, s1);
        double t3 = now();
        d2h(p_out, s2);
        double t4 = now();
        printf("rep %d: pageable H2D %.1f GB/s, pageable D2H %.1f GB/s; pinned H2D %.1f GB/s, pinned D2H %.1f GB/s\n", rep, in_b / (t1 - t0) / 1e9,
               out_b / (t2 - t1) / 1e9, in_b / (t3 - t2) / 1e9, out_b / (t4 - t3) / 1e9);
        // both directions at once, two host threads
        t0 = now();
        {
            std::thread a([&] { h2d(h_in, s1); }), b([&] { d2h(h_out, s2); });
            a.join();
            b.join();
        }
        t1 = now();
        {
            std::thread a([&] { h2d(p_in, s1); }), b([&] { d2h(p_out, s2); });
            a.join();
            b.join();
        }
        t2 = now();
        // one thread, both async (pinned): does a single thread overlap them?
        CK(hipMemcpyAsync(d_in, p_in, in_b, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(p_out, d_out, out_b, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        t3 = now();
        // one thread, pageable, chunked 32 MB alternating
        {
            const size_t ch = (size_t)32 << 20;
            for (size_t o = 0; o < in_b; o += ch) {
                CK(hipMemcpyAsync((char *)d_in + o, h_in + o, ch, hipMemcpyHostToDevice, s1));
                if (o / 2 < out_b) CK(hipMemcpyAsync(h_out + o / 2, (char *)d_out + o / 2, ch / 2, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
        }
        t4 = now();
        printf("        duplex (512 MB in + 256 MB out): pageable 2 threads %.1f ms, pinned 2 threads %.1f ms, pinned 1 thread async %.1f ms, pageable 1 thread chunked %.1f ms"
               "   (sequential pinned would be %.1f ms)\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, 0.0);
        // host memcpy rate (staging through pinned buffers costs this)
        t0 = now();
        memcpy(p_in, h_in, in_b);
        t1 = now();
        printf("        host memcpy pageable -> pinned, one thread: %.1f GB/s\n", in_b / (t1 - t0) / 1e9);
        // pin on the fly
        t0 = now();
        CK(hipHostRegister(h_in, in_b, hipHostRegisterDefault));
        t1 = now();
        h2d(h_in, s1);
        t2 = now();
        CK(hipHostUnregister(h_in));
        t3 = now();
        printf("        hipHostRegister 512 MB %.1f ms, H2D from it %.1f GB/s, unregister %.1f ms\n", (t1 - t0) * 1e3, in_b / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    return 0;
}
