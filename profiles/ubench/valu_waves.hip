// Micro-benchmark (round 6): how close do W = 1..4 waves per SIMD get to the VALU pipe's one instruction per 4 cycles
// in the shapes the sub_dim-8 screen runs (two waves per SIMD; 395 VALU + 24 MFMA + 32 ds_read_b128 per 32-row step)?
//   P0  independent v_min3_f32
//   P1  the screen's reduce: and_or, and_or, med3, min3 per pair of values, one min3 per two pairs (9 per 4 values)
//   P2  P1 with one v_mfma_f32_32x32x16_bf16 per 18 VALU instructions (chains of three on a ring of four accumulators)
//   P3  P2 + 4 ds_read_b128 and an s_waitcnt lgkmcnt(4) per three MFMAs (the |c|^2 re-reads of a tile)
// Reported: ns per VALU instruction and SIMD (all waves of the SIMD together) -- at one instruction per 4 cycles and
// 2.4 GHz that is 1.67 ns.
//   hipcc --offload-arch=gfx950 -O3 valu_waves.hip -o valu_waves && ./valu_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define RED4(q1, q2, a, b, c, d)                                                                      \
    "v_and_or_b32 v100, " a ", %[m], 1\n v_and_or_b32 v101, " b ", %[m], 2\n"                          \
    "v_med3_f32 v104, " q1 ", v100, v101\n v_min3_f32 " q1 ", " q1 ", v100, v101\n"                     \
    "v_and_or_b32 v102, " c ", %[m], 3\n v_and_or_b32 v103, " d ", %[m], 4\n"                          \
    "v_med3_f32 v105, " q1 ", v102, v103\n v_min3_f32 " q1 ", " q1 ", v102, v103\n v_min3_f32 " q2 ", " q2 ", v104, v105\n"
#define RED8(b0, b1, b2, b3, b4, b5, b6, b7) RED4("v110", "v111", b0, b1, b2, b3) RED4("v112", "v113", b4, b5, b6, b7)
#define MF_ON(acc, areg) "v_mfma_f32_32x32x16_bf16 " acc ", " areg ", v[96:99], " acc "\n"
#define MF_OFF(acc, areg) ""
#define DS_ON(a, b, c, d) "ds_read_b128 " a ", %[la]\n ds_read_b128 " b ", %[la] offset:32\n ds_read_b128 " c ", %[la] offset:64\n ds_read_b128 " d ", %[la] offset:96\n s_waitcnt lgkmcnt(4)\n"
#define DS_OFF(a, b, c, d) ""
// four phases = the ring of four accumulators once around: 144 VALU, 12 MFMA, 16 ds_read_b128
#define BODY(MF, DS)                                                                                                \
    DS("v[48:51]", "v[52:55]", "v[56:59]", "v[60:63]")                                                              \
    MF("v[32:47]", "v[64:67]") RED8("v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7")                                  \
    MF("v[32:47]", "v[68:71]") RED8("v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")                            \
    MF("v[32:47]", "v[72:75]")                                                                                      \
    DS("v[0:3]", "v[4:7]", "v[8:11]", "v[12:15]")                                                                   \
    MF("v[48:63]", "v[76:79]") RED8("v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23")                          \
    MF("v[48:63]", "v[80:83]") RED8("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31")                          \
    MF("v[48:63]", "v[84:87]")                                                                                      \
    DS("v[16:19]", "v[20:23]", "v[24:27]", "v[28:31]")                                                              \
    MF("v[0:15]", "v[64:67]") RED8("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39")                           \
    MF("v[0:15]", "v[68:71]") RED8("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")                           \
    MF("v[0:15]", "v[72:75]")                                                                                       \
    DS("v[32:35]", "v[36:39]", "v[40:43]", "v[44:47]")                                                              \
    MF("v[16:31]", "v[76:79]") RED8("v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")                          \
    MF("v[16:31]", "v[80:83]") RED8("v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")                          \
    MF("v[16:31]", "v[84:87]")

// P4: the three MFMAs of a chain back to back, then the 36 VALU of the phase
#define BODY_B2B(MF)                                                                                                \
    MF("v[32:47]", "v[64:67]") MF("v[32:47]", "v[68:71]") MF("v[32:47]", "v[72:75]")                                  \
    RED8("v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7") RED8("v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")  \
    MF("v[48:63]", "v[76:79]") MF("v[48:63]", "v[80:83]") MF("v[48:63]", "v[84:87]")                                  \
    RED8("v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23") RED8("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31") \
    MF("v[0:15]", "v[64:67]") MF("v[0:15]", "v[68:71]") MF("v[0:15]", "v[72:75]")                                     \
    RED8("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39") RED8("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47") \
    MF("v[16:31]", "v[76:79]") MF("v[16:31]", "v[80:83]") MF("v[16:31]", "v[84:87]")                                  \
    RED8("v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55") RED8("v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")
// P5: first MFMA of a chain with C = 0 (no accumulator read)
#define MF_C0(acc, areg) "v_mfma_f32_32x32x16_bf16 " acc ", " areg ", v[96:99], 0\n"
#define BODY_C0                                                                                                      \
    MF_C0("v[32:47]", "v[64:67]") RED8("v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7")                                \
    MF_ON("v[32:47]", "v[68:71]") RED8("v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")                          \
    MF_ON("v[32:47]", "v[72:75]")                                                                                    \
    MF_C0("v[48:63]", "v[76:79]") RED8("v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23")                        \
    MF_ON("v[48:63]", "v[80:83]") RED8("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31")                        \
    MF_ON("v[48:63]", "v[84:87]")                                                                                    \
    MF_C0("v[0:15]", "v[64:67]") RED8("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39")                         \
    MF_ON("v[0:15]", "v[68:71]") RED8("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")                         \
    MF_ON("v[0:15]", "v[72:75]")                                                                                     \
    MF_C0("v[16:31]", "v[76:79]") RED8("v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")                        \
    MF_ON("v[16:31]", "v[80:83]") RED8("v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")                        \
    MF_ON("v[16:31]", "v[84:87]")
// P6: the reduce with no instruction depending on the one in front of it (two chains interleaved)
#define REDI(a0, a1, a2, a3, b0, b1, b2, b3)                                                                          \
    "v_and_or_b32 v100, " a0 ", %[m], 1\n v_and_or_b32 v101, " a1 ", %[m], 2\n v_and_or_b32 v106, " b0 ", %[m], 1\n v_and_or_b32 v107, " b1 ", %[m], 2\n" \
    "v_med3_f32 v104, v110, v100, v101\n v_med3_f32 v114, v112, v106, v107\n v_min3_f32 v110, v110, v100, v101\n v_min3_f32 v112, v112, v106, v107\n"       \
    "v_and_or_b32 v102, " a2 ", %[m], 3\n v_and_or_b32 v103, " a3 ", %[m], 4\n v_and_or_b32 v108, " b2 ", %[m], 3\n v_and_or_b32 v109, " b3 ", %[m], 4\n" \
    "v_med3_f32 v105, v110, v102, v103\n v_med3_f32 v115, v112, v108, v109\n v_min3_f32 v110, v110, v102, v103\n v_min3_f32 v112, v112, v108, v109\n"       \
    "v_min3_f32 v111, v111, v104, v105\n v_min3_f32 v113, v113, v114, v115\n"
#define BODY_I(MF)                                                                                                    \
    MF("v[32:47]", "v[64:67]") REDI("v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7")                                   \
    MF("v[32:47]", "v[68:71]") REDI("v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")                             \
    MF("v[32:47]", "v[72:75]")                                                                                       \
    MF("v[48:63]", "v[76:79]") REDI("v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23")                           \
    MF("v[48:63]", "v[80:83]") REDI("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31")                           \
    MF("v[48:63]", "v[84:87]")                                                                                       \
    MF("v[0:15]", "v[64:67]") REDI("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39")                            \
    MF("v[0:15]", "v[68:71]") REDI("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")                            \
    MF("v[0:15]", "v[72:75]")                                                                                        \
    MF("v[16:31]", "v[76:79]") REDI("v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")                           \
    MF("v[16:31]", "v[80:83]") REDI("v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")                           \
    MF("v[16:31]", "v[84:87]")
// P8: A operands in AGPRs
#define MF_AG(acc, areg) "v_mfma_f32_32x32x16_bf16 " acc ", a[0:3], v[96:99], " acc "\n"
// P9: 16x16x32 MFMAs (half the passes each, two per 32x32x16's work)
#define MF_16(acc, areg) "v_mfma_f32_16x16x32_bf16 v[88:91], " areg ", v[96:99], v[88:91]\n v_mfma_f32_16x16x32_bf16 v[92:95], " areg ", v[96:99], v[92:95]\n"
#define CLOB                                                                                                                                  \
    "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",   \
        "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",  \
        "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",  \
        "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",  \
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v96", "v97", "v98", "v99", "v100", "v101", "v102",      \
        "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v88", "v89", "v90", "v91", "v92",      \
        "v93", "v94", "v95"

template <int P, int LB = 1024>
__global__ __launch_bounds__(LB) void k(float* sink, int iters) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = (float)threadIdx.x;
    lds[threadIdx.x + 1024] = 1.0f;
    __syncthreads();
    unsigned mask = 0xFFFFFFC0u;
    asm volatile("" : "+s"(mask));
    const unsigned la = ((threadIdx.x >> 5) & 1) * 16;
    asm volatile("v_mov_b32 v110, 0x7f800000\n v_mov_b32 v111, 0x7f800000\n v_mov_b32 v112, 0x7f800000\n v_mov_b32 v113, 0x7f800000\n"
                 "v_mov_b32 v96, 0\n v_mov_b32 v97, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n " ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if constexpr (P == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_min3_f32 v100, v0, v1, v2\n v_min3_f32 v101, v3, v4, v5\n v_min3_f32 v102, v6, v7, v8\n v_min3_f32 v103, v9, v10, v11\n"
                             "v_min3_f32 v104, v12, v13, v14\n v_min3_f32 v105, v15, v16, v17\n v_min3_f32 v100, v18, v19, v20\n v_min3_f32 v101, v21, v22, v23\n"
                             "v_min3_f32 v102, v24, v25, v26\n v_min3_f32 v103, v27, v28, v29\n v_min3_f32 v104, v30, v31, v32\n v_min3_f32 v105, v33, v34, v35\n"
                             "v_min3_f32 v100, v36, v37, v38\n v_min3_f32 v101, v39, v40, v41\n v_min3_f32 v102, v42, v43, v44\n v_min3_f32 v103, v45, v46, v47\n"
                             "v_min3_f32 v104, v48, v49, v50\n v_min3_f32 v105, v51, v52, v53\n" ::: CLOB);
        } else if constexpr (P == 1) {
            asm volatile(BODY(MF_OFF, DS_OFF)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 2) {
            asm volatile(BODY(MF_ON, DS_OFF)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 3) {
            asm volatile(BODY(MF_ON, DS_ON)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 4) {
            asm volatile(BODY_B2B(MF_ON)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 5) {
            asm volatile(BODY_C0::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 6) {
            asm volatile(BODY_I(MF_OFF)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 7) {
            asm volatile(BODY_I(MF_ON)::[m] "s"(mask), [la] "v"(la) : CLOB);
        } else if constexpr (P == 10 || P == 11) {
            // P3 + LDS float atomics without return at pseudo-random cluster ids (dimension-major planes: bank = id mod 32):
            // P10: 2 per four phases (= 4 per step per lane, the paired-tail update of sub_dim 8), P11: 5
            asm volatile(BODY(MF_ON, DS_ON)::[m] "s"(mask), [la] "v"(la) : CLOB);
            const unsigned j = ((threadIdx.x * 2654435761u + (unsigned)it * 40503u) >> 13) & 255u;
            const unsigned ad = 8192u + ((threadIdx.x >> 5) & 1u) * 64u + j * 4u;
            float one = 1.0f;
            asm volatile("ds_add_f32 %0, %1 offset:0\n ds_add_f32 %0, %1 offset:1088\n" ::"v"(ad), "v"(one) : "memory");
            if constexpr (P == 11) asm volatile("ds_add_f32 %0, %1 offset:2176\n ds_add_f32 %0, %1 offset:3264\n ds_add_u32 %0, %1 offset:4352\n" ::"v"(ad), "v"(one) : "memory");
        } else if constexpr (P >= 12 && P <= 16) {
            // which LDS atomics are cheap?  P3 + TWO per four phases: P12 ds_add_u32, P13 ds_add_u64, P14 ds_add_rtn_u32 (waited
            // for a phase later), P15 ds_add_f64, P16 ds_max_u32
            asm volatile(BODY(MF_ON, DS_ON)::[m] "s"(mask), [la] "v"(la) : CLOB);
            const unsigned j = ((threadIdx.x * 2654435761u + (unsigned)it * 40503u) >> 13) & 255u;
            const unsigned ad = 8192u + ((threadIdx.x >> 5) & 1u) * 64u + j * (P == 13 || P == 15 ? 8u : 4u);
            unsigned one = 1u;
            unsigned long long one64 = 1ull;
            double oned = 1.0;
            if constexpr (P == 12) asm volatile("ds_add_u32 %0, %1 offset:0\n ds_add_u32 %0, %1 offset:1088\n" ::"v"(ad), "v"(one) : "memory");
            if constexpr (P == 13) asm volatile("ds_add_u64 %0, %1 offset:0\n ds_add_u64 %0, %1 offset:2176\n" ::"v"(ad), "v"(one64) : "memory");
            if constexpr (P == 14) { unsigned r0, r1; asm volatile("ds_add_rtn_u32 %0, %2, %3 offset:0\n ds_add_rtn_u32 %1, %2, %3 offset:1088\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(ad), "v"(one) : "memory"); }
            if constexpr (P == 15) asm volatile("ds_add_f64 %0, %1 offset:0\n ds_add_f64 %0, %1 offset:2176\n" ::"v"(ad), "v"(oned) : "memory");
            if constexpr (P == 16) asm volatile("ds_max_u32 %0, %1 offset:0\n ds_max_u32 %0, %1 offset:1088\n" ::"v"(ad), "v"(one) : "memory");
        } else if constexpr (P == 8) {
            asm volatile(BODY(MF_AG, DS_OFF)::[m] "s"(mask), [la] "v"(la) : CLOB, "a0", "a1", "a2", "a3");
        } else if constexpr (P == 9) {
            asm volatile(BODY(MF_16, DS_OFF)::[m] "s"(mask), [la] "v"(la) : CLOB);
        }
    }
    float s;
    asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, v110, v111\n v_add_f32 %0, %0, v112\n v_add_f32 %0, %0, v113\n v_add_f32 %0, %0, v100" : "=v"(s));
    if (s == 1234.5f) sink[0] = s;
}

template <int P, int LB = 1024>
void run(const char* name, int valu_per_iter, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount, iters = 4000;
    float* sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    for (int w = 0; w < 3; ++w) k<P, LB><<<grid, threads>>>(sink, iters);  // clocks up
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<P, LB><<<grid, threads>>>(sink, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // per SIMD: waves_per_simd waves x iters x valu_per_iter instructions in `best` ms
    const double ns = best * 1e6 / ((double)iters * valu_per_iter * waves_per_simd);
    printf("%-34s W=%d  kernel %.3f ms  %.3f ns per VALU instruction and SIMD  (x 2.4 GHz = %.2f cycles)\n", name, waves_per_simd, best, ns, ns * 2.4);
    hipFree(sink);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int w = 1; w <= 4; ++w) run<0>("P0 min3 independent", 144, w);
    for (int w = 1; w <= 4; ++w) run<1>("P1 reduce pattern", 144, w);
    for (int w = 1; w <= 4; ++w) run<2>("P2 reduce + mfma/18", 144, w);
    for (int w = 1; w <= 4; ++w) run<3>("P3 reduce + mfma + lds", 144, w);
    for (int w = 1; w <= 4; ++w) run<4>("P4 mfma chains back to back", 144, w);
    for (int w = 1; w <= 4; ++w) run<5>("P5 chain's first mfma with C=0", 144, w);
    for (int w = 1; w <= 4; ++w) run<6>("P6 reduce interleaved (no mfma)", 144, w);
    for (int w = 1; w <= 4; ++w) run<7>("P7 reduce interleaved + mfma/18", 144, w);
    for (int w = 1; w <= 3; ++w) run<10>("P10 P3 + 2 ds_add_f32 / 4 phases", 144, w);
    for (int w = 1; w <= 3; ++w) run<11>("P11 P3 + 5 lds atomics / 4 phases", 144, w);
    for (int w = 2; w <= 2; ++w) run<12>("P12 P3 + 2 ds_add_u32", 144, w);
    for (int w = 2; w <= 2; ++w) run<13>("P13 P3 + 2 ds_add_u64", 144, w);
    for (int w = 2; w <= 2; ++w) run<14>("P14 P3 + 2 ds_add_rtn_u32 + wait", 144, w);
    for (int w = 2; w <= 2; ++w) run<15>("P15 P3 + 2 ds_add_f64", 144, w);
    for (int w = 2; w <= 2; ++w) run<16>("P16 P3 + 2 ds_max_u32", 144, w);
    for (int w = 1; w <= 2; ++w) run<2, 512>("P2 again, launch bounds 512", 144, w);
    for (int w = 1; w <= 2; ++w) run<8, 512>("P8 P2 with A in AGPRs", 144, w);
    for (int w = 1; w <= 4; ++w) run<9>("P9 P2 with 2 x 16x16x32 per mfma", 144, w);
    return 0;
}
