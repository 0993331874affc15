// Microbenchmark + self-check of canonical Goldilocks add/sub formulations (butterfly a+b, a-b) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
static constexpr u64 P = 0xFFFFFFFF00000001ULL, EPS = 0xFFFFFFFFULL;

__device__ __forceinline__ u64 add0(u64 a, u64 b) { u64 s = a + b; if (s < a) s += EPS; return s >= P ? s - P : s; }
__device__ __forceinline__ u64 sub0(u64 a, u64 b) { u64 d = a - b; return a >= b ? d : d + P; }
__device__ __forceinline__ u64 add1(u64 a, u64 b) {
    u64 s, u; bool c = __builtin_uaddll_overflow(a, b, &s); bool c2 = __builtin_uaddll_overflow(s, EPS, &u);
    return (c | c2) ? u : s;
}
__device__ __forceinline__ u64 sub1(u64 a, u64 b) { u64 d; bool br = __builtin_usubll_overflow(a, b, &d); return d - (br ? EPS : 0); }
// carry-flag versions
__device__ __forceinline__ u64 add2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), s0, s1, u0, u1, r0, r1;
    u64 c2;
    asm volatile(
        "v_add_co_u32 %0, vcc, %7, %9\n\t"
        "v_addc_co_u32 %1, vcc, %8, %10, vcc\n\t"
        "v_add_co_u32 %2, %6, %0, -1\n\t"
        "v_addc_co_u32 %3, %6, %1, 0, %6\n\t"
        "s_or_b64 vcc, vcc, %6\n\t"
        "v_cndmask_b32 %4, %0, %2, vcc\n\t"
        "v_cndmask_b32 %5, %1, %3, vcc"
        : "=&v"(s0), "=&v"(s1), "=&v"(u0), "=&v"(u1), "=&v"(r0), "=&v"(r1), "=&s"(c2)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    return ((u64)r1 << 32) | r0;
}
__device__ __forceinline__ u64 sub2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), d0, d1, e;
    asm volatile(
        "v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(d0), "=&v"(d1), "=&v"(e)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    return ((u64)d1 << 32) | d0;
}
// 32-bit limb versions with the carry builtins (what gl_field.hpp uses on the device)
__device__ __forceinline__ u64 add3(u64 a, u64 b) {
    u32 c0, c1, d0, d1;
    u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
    u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
    u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &d0);
    u32 t1 = __builtin_addc(s1, 0u, d0, &d1);
    const bool sel = (c1 | d1) != 0;
    return sel ? ((u64)t0 | ((u64)t1 << 32)) : ((u64)s0 | ((u64)s1 << 32));
}
__device__ __forceinline__ u64 sub3(u64 a, u64 b) {
    u32 b0, b1, k;
    u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &b0);
    u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b0, &b1);
    u32 m = 0u - b1;
    u32 e0 = __builtin_subc(d0, m, 0u, &k);
    u32 e1 = d1 - k;
    return (u64)e0 | ((u64)e1 << 32);
}
template <int V> __device__ __forceinline__ u64 addv(u64 a, u64 b) { return V == 0 ? add0(a, b) : (V == 1 ? add1(a, b) : (V == 2 ? add2(a, b) : add3(a, b))); }
template <int V> __device__ __forceinline__ u64 subv(u64 a, u64 b) { return V == 0 ? sub0(a, b) : (V == 1 ? sub1(a, b) : (V == 2 ? sub2(a, b) : sub3(a, b))); }

#define ITER 4096
template <int V>
__global__ __launch_bounds__(256) void kbench(u64* out, u64 seed) {
    u64 x[8];
    for (int i = 0; i < 8; i++) x[i] = (seed * (threadIdx.x + 1) * (2 * i + 3)) % P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) { u64 a = x[i], b = x[i + 1]; x[i] = addv<V>(a, b); x[i + 1] = subv<V>(a, b); }
#pragma unroll
        for (int i = 0; i < 4; i++) { u64 a = x[i], b = x[i + 4]; x[i] = addv<V>(a, b); x[i + 4] = subv<V>(a, b); }
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int V>
__global__ void kcheck(const u64* a, const u64* b, u64* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { o[2 * i] = addv<V>(a[i], b[i]); o[2 * i + 1] = subv<V>(a[i], b[i]); }
}
template <int V>
void run(const char* name) {
    std::vector<u64> a, b;
    u64 edge[] = {0, 1, 2, EPS, EPS + 1, P - 1, P - 2, 1ULL << 32, (1ULL << 63), P - EPS, P - EPS - 1, 0xFFFFFFFE00000001ULL};
    for (u64 x : edge) for (u64 y : edge) { a.push_back(x); b.push_back(y); }
    u64 s = 88172645463325252ULL;
    for (int i = 0; i < 100000; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a.push_back(s % P); s ^= s << 13; s ^= s >> 7; s ^= s << 17; b.push_back(s % P); }
    int n = a.size();
    u64 *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, n * 16);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kcheck<V>, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<u64> o(2 * n);
    hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        u64 wa = (u64)(((unsigned __int128)a[i] + b[i]) % P), ws = (u64)(((unsigned __int128)a[i] + P - b[i]) % P);
        if (o[2 * i] != wa || o[2 * i + 1] != ws) { if (bad < 3) printf("  MISMATCH %s a=%llx b=%llx\n", name, a[i], b[i]); bad++; }
    }
    int blocks = 256 * 8;
    u64* d; hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kbench<V>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kbench<V>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bf = (double)blocks * 256 * ITER * 8;  // butterflies
    printf("%-12s bad=%d %8.3f ms  %6.1f cycles per wave-butterfly (add+sub) per SIMD @2.4GHz\n", name, bad, ms, ms * 1e-3 * 2.4e9 / (bf / 64 / 1024));
    hipFree(d); hipFree(da); hipFree(db); hipFree(dout);
}
int main() { run<0>("compare"); run<1>("builtins u64"); run<2>("carry-asm"); run<3>("builtins u32 limbs"); return 0; }
