// Microbenchmark 2: issue cost of the carry / select / shift ops that make up the non-multiply share of the Goldilocks kernels,
// by waves per SIMD (gfx950).   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu2.hip -o /tmp/mb2 && /tmp/mb2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

#define ITER 2048
#define REP8(x) x x x x x x x x
#define ALL8(M) REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u32 a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    u32 m = seed | 1;
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {
#define M(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 1) {  // carry out to VCC (VOP2)
#define M(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(m) : "vcc");
            ALL8(M)
#undef M
        } else if (OP == 2) {  // carry in and out through VCC
#define M(x) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
            ALL8(M)
#undef M
        } else if (OP == 3) {  // carry out to an SGPR pair (VOP3 encoding)
#define M(x) asm volatile("v_add_co_u32 %0, s[20:21], %0, %1" : "+v"(x) : "v"(m) : "s20", "s21");
            ALL8(M)
#undef M
        } else if (OP == 4) {  // carry in and out through SGPR pairs (VOP3)
#define M(x) asm volatile("v_addc_co_u32 %0, s[20:21], %0, %1, s[20:21]" : "+v"(x) : "v"(m) : "s20", "s21");
            ALL8(M)
#undef M
        } else if (OP == 5) {
#define M(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
            ALL8(M)
#undef M
        } else if (OP == 6) {
#define M(x) asm volatile("v_alignbit_b32 %0, %0, %1, 10" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 7) {
#define M(x) asm volatile("v_lshl_or_b32 %0, %0, 12, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 8) {
#define M(x) asm volatile("v_lshlrev_b32 %0, 12, %0" : "+v"(x));
            ALL8(M)
#undef M
        } else if (OP == 9) {
#define M(x) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
            ALL8(M)
#undef M
        } else if (OP == 10) {  // mad with carry-out consumed by an addc: the 96-bit accumulate step
#define M(x) asm volatile("v_mad_u64_u32 v[40:41], vcc, %0, %1, v[40:41]\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x) : "v"(m) : "vcc", "v40", "v41");
            ALL8(M)
#undef M
        } else if (OP == 11) {
#define M(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 12) {  // two independent instruction kinds interleaved: mad + add (does a 32-bit op hide beside a mad?)
#define M(x) asm volatile("v_mad_u64_u32 v[40:41], vcc, %1, %1, v[40:41]\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(m) : "vcc", "v40", "v41");
            ALL8(M)
#undef M
        } else if (OP == 13) {
#define M(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 14) {
#define M(x) asm volatile("v_mad_u64_u32 v[40:41], vcc, %0, %1, v[40:41]" : "+v"(x) : "v"(m) : "vcc", "v40", "v41");
            ALL8(M)
#undef M
        } else if (OP == 15) {  // SDWA / DPP forms are not used; v_mov as the baseline of a plain VOP1
#define M(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 16) {  // v_pk_add_u16 stands for the packed pipe
#define M(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(m));
            ALL8(M)
#undef M
        } else if (OP == 17) {  // literal constant operand (64-bit encoding)
#define M(x) asm volatile("v_add_u32 %0, 0x12345678, %0" : "+v"(x));
            ALL8(M)
#undef M
        } else if (OP == 18) {  // SGPR operand
#define M(x) asm volatile("v_add_u32 %0, s20, %0" : "+v"(x) : : "s20");
            ALL8(M)
#undef M
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char* name, int instr_per_m) {
    u32* d;
    hipMalloc(&d, 256 * 16 * 256 * 4);
    printf("%-34s", name);
    for (int bpc : {1, 2, 4, 8}) {  // blocks per CU x 4 waves -> 1, 2, 4, 8 waves per SIMD
        int blocks = 256 * bpc;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * ITER * 64 * instr_per_m;
        double per_simd_cycle = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
        printf("  w%d: %5.2f cyc", bpc, 1.0 / per_simd_cycle);
    }
    printf("   (cycles per wave-instr per SIMD @2.4 GHz)\n");
    hipFree(d);
}

int main() {
    run<15>("v_mov_b32", 1);
    run<0>("v_add_u32", 1);
    run<17>("v_add_u32 literal", 1);
    run<18>("v_add_u32 sgpr", 1);
    run<11>("v_add3_u32", 1);
    run<1>("v_add_co_u32 (vcc)", 1);
    run<2>("v_addc_co_u32 (vcc in/out)", 1);
    run<3>("v_add_co_u32 (sgpr pair)", 1);
    run<4>("v_addc_co_u32 (sgpr in/out)", 1);
    run<9>("v_sub_co+v_subb_co", 2);
    run<5>("v_cndmask_b32", 1);
    run<6>("v_alignbit_b32", 1);
    run<7>("v_lshl_or_b32", 1);
    run<8>("v_lshlrev_b32", 1);
    run<16>("v_pk_add_u16", 1);
    run<13>("v_mul_lo_u32", 1);
    run<14>("v_mad_u64_u32", 1);
    run<10>("v_mad_u64_u32 + v_addc", 2);
    run<12>("v_mad_u64_u32 + v_add_u32", 2);
    return 0;
}
