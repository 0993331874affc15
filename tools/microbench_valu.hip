// Microbenchmark: issue rate of the integer VALU ops the Goldilocks kernels are made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define ITER 4096
#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u64 seed) {
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u64 a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    u32 m = (u32)seed | 1;
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {  // v_mad_u64_u32
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"((u32)x), "v"(m) : "vcc");
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 1) {  // v_mul_lo_u32
#define M(x) { u32 t = (u32)x; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 2) {  // v_mul_hi_u32
#define M(x) { u32 t = (u32)x; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 3) {  // v_lshl_add_u64
#define M(x) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(x) : "v"(a0));
            REP8(M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7) M(a1))
#undef M
        } else if (OP == 4) {  // 32-bit add
#define M(x) { u32 t = (u32)x; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 5) {  // add_co + addc (64-bit add)
#define M(x) { u32 lo = (u32)x, hi = (u32)(x >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(m), "v"(m) : "vcc"); x = ((u64)hi << 32) | lo; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 6) {  // v_mul_u32_u24
#define M(x) { u32 t = (u32)x; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 7) {  // v_cmp_lt_u64 + v_cndmask
#define M(x) { u32 lo = (u32)x; asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(lo) : "v"(x), "v"(a0), "v"(m) : "vcc"); x = (x & 0xffffffff00000000ull) | lo; }
            REP8(M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7) M(a1))
#undef M
        } else if (OP == 8) {  // v_mad_u32_u24
#define M(x) { u32 t = (u32)x; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 9) {  // v_add3_u32
#define M(x) { u32 t = (u32)x; asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(t) : "v"(m)); x = t; }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 10) {  // v_mul_f64 (dp rate reference)
#define M(x) { double t = __longlong_as_double(x); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(t)); x = __double_as_longlong(t); }
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        } else if (OP == 11) {  // v_lshlrev_b64
#define M(x) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x));
            REP8(M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7))
#undef M
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
double run(const char* name, int instr_per_m) {
    u64* d;
    int blocks = 256 * 8;  // 8 blocks/CU x 4 waves = 8 waves/SIMD
    hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITER * 64 * instr_per_m;  // wave-instructions
    double per_simd_cycle = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s %8.3f ms  %7.3f wave-instr/cycle/SIMD  (= %5.2f cycles per wave-instr @2.4GHz)\n", name, ms, per_simd_cycle,
           1.0 / per_simd_cycle);
    hipFree(d);
    return ms;
}

int main() {
    run<4>("v_add_u32", 1);
    run<9>("v_add3_u32", 1);
    run<5>("v_add_co+v_addc (pair)", 2);
    run<3>("v_lshl_add_u64", 1);
    run<11>("v_lshlrev_b64", 1);
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<6>("v_mul_u32_u24", 1);
    run<8>("v_mad_u32_u24", 1);
    run<7>("v_cmp_lt_u64+v_cndmask", 2);
    run<10>("v_fma_f64", 1);
    return 0;
}
