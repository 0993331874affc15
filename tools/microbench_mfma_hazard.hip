// How many wait states does gfx950 really need between a v_mfma_i32_32x32x32_i8 and a VALU instruction that READS (RAW) or
// WRITES (WAW) a register of its destination tile?  There is no hardware interlock for either: the compiler pads with s_nop
// from a table (GCNHazardRecognizer: passes + 3 = 11 wait states for this 8-pass instruction, the same for both), and a VALU
// write that lands before the matrix pipe's own write-back of that register is silently overwritten - which is what a register
// allocator does when it parks another value in a DEAD register of a tile whose MFMA is still in flight (round 4:
// profiles/r04_mfma_hazard.txt; round 3's "undefined operand" incident was the same thing).
//
// Each kernel runs, per wave and iteration, with explicit registers:
//     v[16:31] <- marker;  (optionally another MFMA into v[32:47] first, so that the pipe is busy);
//     v_mfma v[16:31], A = 1s, B = 1s, 0      -> every register of the tile becomes 32
//     K wait states: s_nop K-1 (fill=s_nop), or K independent VALU instructions (fill=valu) - what compiled code has there
//     RAW: x <- v[16+Q]            expected 32;  stale = the marker
//     WAW: v[16+Q] <- magic; long wait; x <- v[16+Q]     expected magic;  clobbered = 32
// and counts lanes that saw the wrong value.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_mfma_hazard.hip -o tools/bin/mbhaz && tools/bin/mbhaz
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

#define TILE_CLOBBER "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", \
                     "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48"
#define STR2(x) #x
#define STR(x) STR2(x)
#define MARK_TILE                                                                                                                  \
    "v_mov_b32 v16, %[mark]\nv_mov_b32 v17, %[mark]\nv_mov_b32 v18, %[mark]\nv_mov_b32 v19, %[mark]\nv_mov_b32 v20, %[mark]\n"      \
    "v_mov_b32 v21, %[mark]\nv_mov_b32 v22, %[mark]\nv_mov_b32 v23, %[mark]\nv_mov_b32 v24, %[mark]\nv_mov_b32 v25, %[mark]\n"      \
    "v_mov_b32 v26, %[mark]\nv_mov_b32 v27, %[mark]\nv_mov_b32 v28, %[mark]\nv_mov_b32 v29, %[mark]\nv_mov_b32 v30, %[mark]\n"      \
    "v_mov_b32 v31, %[mark]\ns_nop 7\n"
#define LONG_WAIT "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"

// MODE 0: RAW, 1: WAW.  BUSY 1: another MFMA (other tile) issued right before; BUSY 2: the MFMA is the third of a dependent chain
// into the same tile (expected value 96).  K: wait states.  Q: register of the tile.
template <int MODE, int BUSY, int K, int Q, int FILL>
__global__ __launch_bounds__(256) void k_hazard(u32* __restrict__ bad, int iters) {
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const u32 mark = 0x55555555u, magic = 0x12345u;
    u32 wrong = 0;
    for (int it = 0; it < iters; it++) {
        u32 x;
        if (MODE == 0) {
            asm volatile(MARK_TILE
                         ".if %[busy] == 1\nv_mfma_i32_32x32x32_i8 v[32:47], %[a], %[b], 0\n.endif\n"
                         ".if %[busy] == 2\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], 0\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], v[16:31]\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], v[16:31]\n.else\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], 0\n.endif\n"
                         ".if %[fill]\n.rept %[k]\nv_mov_b32 v48, %[mark]\n.endr\n.else\n.if %[k] > 0\ns_nop %[k] - 1\n.endif\n.endif\n"
                         "v_mov_b32 %[x], v[%[q]]\n" LONG_WAIT
                         : [x] "=&v"(x)
                         : [a] "v"(ones), [b] "v"(ones), [mark] "v"(mark), [k] "n"(K), [busy] "n"(BUSY), [q] "n"(Q), [fill] "n"(FILL)
                         : TILE_CLOBBER);
            wrong += x != (BUSY == 2 ? 96u : 32u);
        } else {
            asm volatile(MARK_TILE
                         ".if %[busy] == 1\nv_mfma_i32_32x32x32_i8 v[32:47], %[a], %[b], 0\n.endif\n"
                         ".if %[busy] == 2\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], 0\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], v[16:31]\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], v[16:31]\n.else\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], %[b], 0\n.endif\n"
                         ".if %[fill]\n.rept %[k]\nv_mov_b32 v48, %[mark]\n.endr\n.else\n.if %[k] > 0\ns_nop %[k] - 1\n.endif\n.endif\n"
                         "v_mov_b32 v[%[q]], %[magic]\n" LONG_WAIT
                         "v_mov_b32 %[x], v[%[q]]\n"
                         : [x] "=&v"(x)
                         : [a] "v"(ones), [b] "v"(ones), [mark] "v"(mark), [magic] "v"(magic), [k] "n"(K), [busy] "n"(BUSY), [q] "n"(Q), [fill] "n"(FILL)
                         : TILE_CLOBBER);
            wrong += x != magic;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int MODE, int BUSY, int K, int Q, int FILL>
static void run(u32* dbad, int blocks, int iters, unsigned long long* out) {
    CHECK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL((k_hazard<MODE, BUSY, K, Q, FILL>), dim3(blocks), dim3(256), 0, 0, dbad, iters);
    CHECK(hipDeviceSynchronize());
    u32 h;
    CHECK(hipMemcpy(&h, dbad, 4, hipMemcpyDeviceToHost));
    *out = h;
}

template <int MODE, int BUSY, int Q, int FILL>
static void sweep(u32* dbad, int blocks, int iters) {
    unsigned long long r[20];
    run<MODE, BUSY, 0, Q, FILL>(dbad, blocks, iters, r + 0);
    run<MODE, BUSY, 2, Q, FILL>(dbad, blocks, iters, r + 1);
    run<MODE, BUSY, 4, Q, FILL>(dbad, blocks, iters, r + 2);
    run<MODE, BUSY, 6, Q, FILL>(dbad, blocks, iters, r + 3);
    run<MODE, BUSY, 8, Q, FILL>(dbad, blocks, iters, r + 4);
    run<MODE, BUSY, 9, Q, FILL>(dbad, blocks, iters, r + 5);
    run<MODE, BUSY, 10, Q, FILL>(dbad, blocks, iters, r + 6);
    run<MODE, BUSY, 11, Q, FILL>(dbad, blocks, iters, r + 7);
    run<MODE, BUSY, 12, Q, FILL>(dbad, blocks, iters, r + 8);
    run<MODE, BUSY, 13, Q, FILL>(dbad, blocks, iters, r + 9);
    run<MODE, BUSY, 14, Q, FILL>(dbad, blocks, iters, r + 10);
    run<MODE, BUSY, 15, Q, FILL>(dbad, blocks, iters, r + 11);
    run<MODE, BUSY, 16, Q, FILL>(dbad, blocks, iters, r + 12);
    const int ks[13] = {0, 2, 4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    printf("%s busy=%d fill=%s reg %2d :", MODE ? "WAW" : "RAW", BUSY, FILL ? "valu " : "s_nop", Q - 16);
    for (int i = 0; i < 13; i++) printf(" K=%d:%llu", ks[i], r[i]);
    printf("\n");
}

// WAR on a SOURCE operand: the MFMA's B (or A) tuple is v[32:35]; K wait states after the MFMA a VALU instruction overwrites
// register v[32 + Q] of it.  If the matrix pipe had not read that register yet, the product is wrong (expected 32 everywhere).
// PRE = 1: an independent MFMA is issued right before, so the tested one may wait for the pipe.
template <int WHICH_A, int PRE, int K, int Q>
__global__ __launch_bounds__(256) void k_war(u32* __restrict__ bad, int iters) {
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const u32 garbage = 0x7F7F7F7Fu;
    u32 wrong = 0;
    for (int it = 0; it < iters; it++) {
        u32 x0, x15;
        asm volatile("v_mov_b32 v32, %[one]\nv_mov_b32 v33, %[one]\nv_mov_b32 v34, %[one]\nv_mov_b32 v35, %[one]\ns_nop 7\n"
                     ".if %[pre]\nv_mfma_i32_32x32x32_i8 v[36:51], %[a], %[a], 0\n.endif\n"
                     ".if %[wa]\nv_mfma_i32_32x32x32_i8 v[16:31], v[32:35], %[a], 0\n.else\nv_mfma_i32_32x32x32_i8 v[16:31], %[a], v[32:35], 0\n.endif\n"
                     ".rept %[k]\nv_mov_b32 v52, %[one]\n.endr\n"
                     "v_mov_b32 v[%[q]], %[g]\n" LONG_WAIT
                     "v_mov_b32 %[x0], v16\nv_mov_b32 %[x15], v31\n"
                     : [x0] "=&v"(x0), [x15] "=&v"(x15)
                     : [a] "v"(ones), [one] "v"(0x01010101u), [g] "v"(garbage), [k] "n"(K), [q] "n"(32 + Q), [wa] "n"(WHICH_A), [pre] "n"(PRE)
                     : TILE_CLOBBER, "v48", "v49", "v50", "v51", "v52");
        wrong += (x0 != 32u) | (x15 != 32u);
    }
    if (wrong) atomicAdd(bad, wrong);
}
template <int WHICH_A, int PRE, int K, int Q>
static unsigned long long run_war(u32* dbad, int blocks, int iters) {
    CHECK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL((k_war<WHICH_A, PRE, K, Q>), dim3(blocks), dim3(256), 0, 0, dbad, iters);
    CHECK(hipDeviceSynchronize());
    u32 h;
    CHECK(hipMemcpy(&h, dbad, 4, hipMemcpyDeviceToHost));
    return h;
}
template <int WHICH_A, int PRE, int Q>
static void sweep_war(u32* dbad, int blocks, int iters) {
    printf("WAR on %s register %d%s :", WHICH_A ? "A" : "B", Q, PRE ? " (pipe busy)" : "            ");
    printf(" K=0:%llu", run_war<WHICH_A, PRE, 0, Q>(dbad, blocks, iters));
    printf(" K=1:%llu", run_war<WHICH_A, PRE, 1, Q>(dbad, blocks, iters));
    printf(" K=2:%llu", run_war<WHICH_A, PRE, 2, Q>(dbad, blocks, iters));
    printf(" K=3:%llu", run_war<WHICH_A, PRE, 3, Q>(dbad, blocks, iters));
    printf(" K=4:%llu", run_war<WHICH_A, PRE, 4, Q>(dbad, blocks, iters));
    printf(" K=6:%llu", run_war<WHICH_A, PRE, 6, Q>(dbad, blocks, iters));
    printf(" K=8:%llu", run_war<WHICH_A, PRE, 8, Q>(dbad, blocks, iters));
    printf(" K=12:%llu\n", run_war<WHICH_A, PRE, 12, Q>(dbad, blocks, iters));
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048;   // 256 CUs x 4 SIMDs: 2048 blocks of 4 waves = 2 waves per SIMD at a time and more
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    u32* dbad;
    CHECK(hipMalloc(&dbad, 4));
    printf("wrong lane-results out of %llu per cell (v_mfma_i32_32x32x32_i8, wait states K between the MFMA and the VALU access)\n",
           (unsigned long long)blocks * 256 * iters);
    sweep<0, 0, 16, 0>(dbad, blocks, iters);
    sweep<0, 0, 31, 0>(dbad, blocks, iters);
    sweep<0, 1, 31, 0>(dbad, blocks, iters);
    sweep<0, 2, 16, 0>(dbad, blocks, iters);
    sweep<0, 2, 31, 0>(dbad, blocks, iters);
    sweep<0, 0, 16, 1>(dbad, blocks, iters);
    sweep<0, 0, 31, 1>(dbad, blocks, iters);
    sweep<0, 1, 31, 1>(dbad, blocks, iters);
    sweep<0, 2, 16, 1>(dbad, blocks, iters);
    sweep<0, 2, 31, 1>(dbad, blocks, iters);
    sweep<1, 0, 16, 0>(dbad, blocks, iters);
    sweep<1, 0, 24, 0>(dbad, blocks, iters);
    sweep<1, 0, 31, 0>(dbad, blocks, iters);
    sweep<1, 1, 31, 0>(dbad, blocks, iters);
    sweep<1, 2, 16, 0>(dbad, blocks, iters);
    sweep<1, 2, 31, 0>(dbad, blocks, iters);
    sweep<1, 0, 16, 1>(dbad, blocks, iters);
    sweep<1, 0, 24, 1>(dbad, blocks, iters);
    sweep<1, 0, 31, 1>(dbad, blocks, iters);
    sweep<1, 1, 31, 1>(dbad, blocks, iters);
    sweep<1, 2, 16, 1>(dbad, blocks, iters);
    sweep<1, 2, 18, 1>(dbad, blocks, iters);
    sweep<1, 2, 31, 1>(dbad, blocks, iters);
    printf("VALU write to a source register K instructions after the MFMA that reads it (wrong lane-results out of %llu)\n", (unsigned long long)blocks * 256 * iters);
    sweep_war<0, 0, 0>(dbad, blocks, iters);
    sweep_war<0, 0, 3>(dbad, blocks, iters);
    sweep_war<0, 1, 0>(dbad, blocks, iters);
    sweep_war<0, 1, 3>(dbad, blocks, iters);
    sweep_war<1, 0, 0>(dbad, blocks, iters);
    sweep_war<1, 0, 3>(dbad, blocks, iters);
    sweep_war<1, 1, 0>(dbad, blocks, iters);
    sweep_war<1, 1, 3>(dbad, blocks, iters);
    return 0;
}
