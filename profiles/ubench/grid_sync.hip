// Micro-benchmark (round 6): what does a grid-wide barrier cost at the sizes of the small Lloyd path (C1: 628 workgroups of
// 256 threads)?  cooperative_groups::this_grid().sync() under hipLaunchCooperativeKernel, 200 barriers in one launch.
//   hipcc --offload-arch=gfx950 -O3 grid_sync.hip -o bin/grid_sync && ./bin/grid_sync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k(unsigned *buf, int rounds) {
    cg::grid_group grid = cg::this_grid();
    const unsigned gid = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    unsigned v = 0;
    for (int r = 0; r < rounds; ++r) {
        buf[gid] = r + 1;                 // every workgroup writes ...
        grid.sync();
        v += buf[(gid + 4099u) % n];       // ... and reads another workgroup's word behind the barrier
        grid.sync();
    }
    if (v != (unsigned)rounds * (rounds + 1) / 2) buf[n] = 1;  // a stale read shows here
}

int main() {
    for (int blocks : {64, 160, 628, 1024, 2048}) {
        unsigned *buf;
        hipMalloc(&buf, (size_t)(blocks * 256 + 1) * 4);
        hipMemset(buf, 0, (size_t)(blocks * 256 + 1) * 4);
        int rounds = 100;
        void *args[] = {&buf, &rounds};
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipError_t err = hipLaunchCooperativeKernel(reinterpret_cast<void *>(k), dim3(blocks), dim3(256), args, 0, nullptr);
        if (err != hipSuccess) {
            printf("%d workgroups: cooperative launch refused: %s\n", blocks, hipGetErrorString(err));
            (void)hipGetLastError();
            continue;
        }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchCooperativeKernel(reinterpret_cast<void *>(k), dim3(blocks), dim3(256), args, 0, nullptr);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned bad = 0;
        hipMemcpy(&bad, buf + blocks * 256, 4, hipMemcpyDeviceToHost);
        printf("%4d workgroups x 256: %.2f us per grid barrier (200 in %.3f ms), stale reads: %u\n", blocks, ms * 1e3 / 200, ms, bad);
        hipFree(buf);
    }
    return 0;
}
