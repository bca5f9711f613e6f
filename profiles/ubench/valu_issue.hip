// Micro-benchmark: what does ONE wave per SIMD pay per vector instruction in the shapes the screen's epilogue uses?
//   T0  independent v_min3_f32 (4 rotating destinations)
//   T1  one dependent chain of v_min3_f32
//   T2  the epilogue as the compiler emits it: and_or, and_or, med3, min3 (each right behind its producers), min
//   T3  the same multiset, producers at least 4 instructions ahead of their consumers
//   T4  v_min3_f32 with three sources in one register bank (v4, v8, v12) / T5 in three banks (v4, v5, v6)
//   T6  T2 with one v_mfma_f32_32x32x16_bf16 per 9 instructions, T7: T3 likewise
//   T8  v_and_or_b32 alone (sgpr mask + inline constant), T9: v_med3 alone, independent
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int T>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ float pad[];
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f, v8 = 8.f, v9 = 9.f, v10 = 10.f, v11 = 11.f;
    float q10 = 1e30f, q11 = 1e30f, q12 = 1e30f, q13 = 1e30f, q20 = 1e30f, q21 = 1e30f, q22 = 1e30f, q23 = 1e30f;
    float a0 = v0 + 1, a1 = v0 + 2, a2 = v0 + 3, a3 = v0 + 4, a4 = v0 + 5, a5 = v0 + 6, a6 = v0 + 7, a7 = v0 + 8;
    float p0, p1, p2, p3, p4, p5, p6, p7, t0, t1, t2, t3;
    unsigned mask = 0xFFFFFFC0u;
    asm volatile("" : "+s"(mask));
    f32x16 acc = {0};
    bf16x8 am = {1, 2, 3, 4, 5, 6, 7, 8}, bm = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (T == 0) {
            REP16(asm volatile("v_min3_f32 %0, %4, %5, %6\n v_min3_f32 %1, %4, %5, %6\n v_min3_f32 %2, %4, %5, %6\n v_min3_f32 %3, %4, %5, %6"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "v"(v6));)
        } else if constexpr (T == 1) {
            REP16(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %1, %2"
                               : "+v"(v0) : "v"(v4), "v"(v5));)
        } else if constexpr (T == 2 || T == 6) {
            // 4 chains x (2 and_or, med3, min3, min) = 20 instructions; REP4 -> 80 (+ 8 MFMAs for T6)
#define CH(q1, q2, xa, xb, pa, pb, t, ia, ib)                                                         \
    "v_and_or_b32 %[" #pa "], %[" #xa "], %[m], " #ia "\n v_and_or_b32 %[" #pb "], %[" #xb "], %[m], " #ib \
    "\n v_med3_f32 %[" #t "], %[" #q1 "], %[" #pa "], %[" #pb "]\n v_min3_f32 %[" #q1 "], %[" #q1 "], %[" #pa "], %[" #pb "]\n v_min_f32 %[" #q2 "], %[" #q2 "], %[" #t "]\n"
#define MF "v_mfma_f32_32x32x16_bf16 %[acc], %[am], %[bm], %[acc]\n"
            REP4(asm volatile(CH(q10, q20, a0, a1, p0, p1, t0, 1, 2) CH(q11, q21, a2, a3, p2, p3, t1, 3, 4)
                              CH(q12, q22, a4, a5, p4, p5, t2, 5, 6) CH(q13, q23, a6, a7, p6, p7, t3, 7, 8)
                              : [q10] "+v"(q10), [q11] "+v"(q11), [q12] "+v"(q12), [q13] "+v"(q13), [q20] "+v"(q20), [q21] "+v"(q21),
                                [q22] "+v"(q22), [q23] "+v"(q23), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
                                [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7), [t0] "=&v"(t0), [t1] "=&v"(t1),
                                [t2] "=&v"(t2), [t3] "=&v"(t3)
                              : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6),
                                [a7] "v"(a7), [m] "s"(mask));
                 if constexpr (T == 6) { asm volatile(MF : [acc] "+v"(acc) : [am] "v"(am), [bm] "v"(bm)); asm volatile(MF : [acc] "+v"(acc) : [am] "v"(am), [bm] "v"(bm)); })
        } else if constexpr (T == 3 || T == 7) {
            // same 20 instructions, producers >= 4 ahead: 8 and_or, 4 med3, 4 min3, 4 min
#define AO(pa, xa, ia) "v_and_or_b32 %[" #pa "], %[" #xa "], %[m], " #ia "\n"
#define MD(t, q1, pa, pb) "v_med3_f32 %[" #t "], %[" #q1 "], %[" #pa "], %[" #pb "]\n"
#define M3(q1, pa, pb) "v_min3_f32 %[" #q1 "], %[" #q1 "], %[" #pa "], %[" #pb "]\n"
#define MN(q2, t) "v_min_f32 %[" #q2 "], %[" #q2 "], %[" #t "]\n"
            REP4(asm volatile(AO(p0, a0, 1) AO(p1, a1, 2) AO(p2, a2, 3) AO(p3, a3, 4) AO(p4, a4, 5) AO(p5, a5, 6) AO(p6, a6, 7) AO(p7, a7, 8)
                              MD(t0, q10, p0, p1) MD(t1, q11, p2, p3) MD(t2, q12, p4, p5) MD(t3, q13, p6, p7)
                              M3(q10, p0, p1) M3(q11, p2, p3) M3(q12, p4, p5) M3(q13, p6, p7)
                              MN(q20, t0) MN(q21, t1) MN(q22, t2) MN(q23, t3)
                              : [q10] "+v"(q10), [q11] "+v"(q11), [q12] "+v"(q12), [q13] "+v"(q13), [q20] "+v"(q20), [q21] "+v"(q21),
                                [q22] "+v"(q22), [q23] "+v"(q23), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
                                [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7), [t0] "=&v"(t0), [t1] "=&v"(t1),
                                [t2] "=&v"(t2), [t3] "=&v"(t3)
                              : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6),
                                [a7] "v"(a7), [m] "s"(mask));
                 if constexpr (T == 7) { asm volatile(MF : [acc] "+v"(acc) : [am] "v"(am), [bm] "v"(bm)); asm volatile(MF : [acc] "+v"(acc) : [am] "v"(am), [bm] "v"(bm)); })
        } else if constexpr (T == 4) {
            REP16(asm volatile("v_min3_f32 v0, v4, v8, v12\n v_min3_f32 v1, v4, v8, v12\n v_min3_f32 v2, v4, v8, v12\n v_min3_f32 v3, v4, v8, v12" ::: "v0", "v1", "v2", "v3");)
        } else if constexpr (T == 5) {
            REP16(asm volatile("v_min3_f32 v0, v4, v5, v6\n v_min3_f32 v1, v4, v5, v6\n v_min3_f32 v2, v4, v5, v6\n v_min3_f32 v3, v4, v5, v6" ::: "v0", "v1", "v2", "v3");)
        } else if constexpr (T == 8) {
            REP16(asm volatile("v_and_or_b32 %0, %4, %5, 1\n v_and_or_b32 %1, %4, %5, 2\n v_and_or_b32 %2, %4, %5, 3\n v_and_or_b32 %3, %4, %5, 4"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "s"(mask));)
        } else if constexpr (T == 9) {
            REP16(asm volatile("v_med3_f32 %0, %4, %5, %6\n v_med3_f32 %1, %4, %5, %6\n v_med3_f32 %2, %4, %5, %6\n v_med3_f32 %3, %4, %5, %6"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "v"(v6));)
        } else if constexpr (T == 10) {  // MFMA alone, one chain
            REP16(asm volatile(MF : [acc] "+v"(acc) : [am] "v"(am), [bm] "v"(bm));)
        } else if constexpr (T == 12) {  // min3, 16 rotating destinations, sources rotate too
            REP4(asm volatile("v_min3_f32 v0, v16, v17, v18\n v_min3_f32 v1, v17, v18, v19\n v_min3_f32 v2, v18, v19, v20\n v_min3_f32 v3, v19, v20, v21\n"
                              "v_min3_f32 v4, v20, v21, v22\n v_min3_f32 v5, v21, v22, v23\n v_min3_f32 v6, v22, v23, v24\n v_min3_f32 v7, v23, v24, v25\n"
                              "v_min3_f32 v8, v24, v25, v26\n v_min3_f32 v9, v25, v26, v27\n v_min3_f32 v10, v26, v27, v28\n v_min3_f32 v11, v27, v28, v29\n"
                              "v_min3_f32 v12, v28, v29, v30\n v_min3_f32 v13, v29, v30, v31\n v_min3_f32 v14, v30, v31, v16\n v_min3_f32 v15, v31, v16, v17"
                              ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");)
        } else if constexpr (T == 13) {  // alternate VOP3 (min3) and VOP2 e32 (min), 4 dests
            REP16(asm volatile("v_min3_f32 %0, %4, %5, %6\n v_min_f32 %1, %4, %5\n v_min3_f32 %2, %4, %5, %6\n v_min_f32 %3, %4, %5"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "v"(v6));)
        } else if constexpr (T == 14) {  // s_nop 0 only
            REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
        } else if constexpr (T == 15) {  // salu only
            REP16(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23");)
        } else if constexpr (T == 16) {  // min3 with 2 waves per SIMD (launched with 512 threads)
            REP16(asm volatile("v_min3_f32 %0, %4, %5, %6\n v_min3_f32 %1, %4, %5, %6\n v_min3_f32 %2, %4, %5, %6\n v_min3_f32 %3, %4, %5, %6"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "v"(v6));)
        } else if constexpr (T == 17) {  // ds_read_b128 from a fixed address, no wait in the loop
            REP16(asm volatile("ds_read_b128 v[16:19], %0\n ds_read_b128 v[20:23], %0\n ds_read_b128 v[24:27], %0\n ds_read_b128 v[28:31], %0" :: "v"(0) : "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 11) {  // v_min_f32 (VOP2) independent
            REP16(asm volatile("v_min_f32 %0, %4, %5\n v_min_f32 %1, %4, %5\n v_min_f32 %2, %4, %5\n v_min_f32 %3, %4, %5"
                               : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5));)
        }
    }
    const unsigned long long c1 = clock64();
    float s = v0 + v1 + v2 + v3 + q10 + q11 + q12 + q13 + q20 + q21 + q22 + q23 + acc[0] + acc[5];
    if (s == 1234.5f) sink[0] = s + v7 + v8 + v9 + v10 + v11 + pad[0];
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int T>
void run(const char* name, int instr_per_iter, int mfma_per_iter, int threads = 256) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount, iters = 2000;
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, grid * 8 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T><<<grid, threads, 100 * 1024>>>(out, sink, 10);
    hipEventRecord(e0);
    k<T><<<grid, threads, 100 * 1024>>>(out, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);  // first 4 waves of each block
    hipMemcpy(h.data(), out, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("%-44s %6.2f clock64 ticks/instr (median wave), kernel %.3f ms -> %.2f ns per instr, %d instr + %d mfma per iter\n", name,
           med / iters / instr_per_iter, ms, ms * 1e6 / iters / instr_per_iter, instr_per_iter, mfma_per_iter);
    hipFree(out);
    hipFree(sink);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<0>("T0 min3 independent", 64, 0);
    run<1>("T1 min3 dependent chain", 64, 0);
    run<2>("T2 epilogue, compiler order", 80, 0);
    run<3>("T3 epilogue, producers 4 ahead", 80, 0);
    run<4>("T4 min3 sources in one bank", 64, 0);
    run<5>("T5 min3 sources in three banks", 64, 0);
    run<6>("T6 T2 + 1 mfma per 10 instr", 80, 8);
    run<7>("T7 T3 + 1 mfma per 10 instr", 80, 8);
    run<8>("T8 and_or independent", 64, 0);
    run<9>("T9 med3 independent", 64, 0);
    run<10>("T10 mfma chain alone (per mfma)", 16, 16);
    run<11>("T11 v_min_f32 independent", 64, 0);
    run<12>("T12 min3, 16 dests, rotating sources", 64, 0);
    run<13>("T13 alternate min3 / min e32", 64, 0);
    run<14>("T14 s_nop 0", 64, 0);
    run<15>("T15 s_add_u32", 64, 0);
    run<16>("T16 min3 independent, TWO waves per SIMD", 64, 0, 512);
    run<17>("T17 ds_read_b128 (+1 wait per 64)", 64, 0);
    return 0;
}
