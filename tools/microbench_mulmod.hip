// Microbenchmark + self-check of Goldilocks mul-mod formulations on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
static constexpr u64 P = 0xFFFFFFFF00000001ULL, EPS = 0xFFFFFFFFULL;

__host__ __device__ inline u64 canon(u64 x) { return x >= P ? x - P : x; }

// M0: canonical in/out (current product code)
__device__ __forceinline__ u64 mul0(u64 a, u64 b) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = (hl << 32) - hl;
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += EPS;
    return canon(t2);
}
// M1: lazy (any u64 in, any u64 out congruent mod p), overflow builtins
__device__ __forceinline__ u64 mul1(u64 a, u64 b) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0, t2;
    bool br = __builtin_usubll_overflow(lo, (u64)hh, &t0);
    t0 -= br ? EPS : 0;
    bool cy = __builtin_uaddll_overflow(t0, (u64)hl * EPS, &t2);
    t2 += cy ? EPS : 0;
    return t2;
}
// M2: lazy, explicit 32x32 products (4 mads) + fold
__device__ __forceinline__ u64 mul2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + ((p01 >> 32) + (p10 >> 32));
    u64 lo = (p10 << 32) | (u32)p00;
    u32 hl = (u32)p11, hh = (u32)(p11 >> 32);
    u64 t0, t2;
    bool br = __builtin_usubll_overflow(lo, (u64)hh, &t0);
    t0 -= br ? EPS : 0;
    bool cy = __builtin_uaddll_overflow(t0, (u64)hl * EPS, &t2);
    t2 += cy ? EPS : 0;
    return t2;
}
// M3: lazy, base-phi limbs: x = x0 + x1 f + x2 f^2 + x3 f^3, f^2 = f - 1, f^3 = -1
//   x = (x0 - x2 - x3) + (x1 + x2) f
__device__ __forceinline__ u64 mul3(u64 a, u64 b) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u32 x0 = (u32)lo, x1 = (u32)(lo >> 32), x2 = (u32)hi, x3 = (u32)(hi >> 32);
    // r = lo + x2*(f-1) - x3 : compute as 64-bit with a signed small correction word
    long long c = 0;  // multiples of 2^64 to fold: c * EPS
    u64 r = lo;
    u64 add = ((u64)x2 << 32) - x2;  // x2 * (f - 1) < 2^64
    u64 r2 = r + add;
    c += r2 < r;
    u64 r3 = r2 - x3;
    c -= r2 < x3;
    (void)x0; (void)x1;
    // c in {-1,0,1}
    u64 corr = c == 0 ? 0 : (c > 0 ? EPS : (u64)0 - EPS);
    return r3 + corr;
}

// M4: lazy, 4 mads, fold written on 32-bit limbs with carry builtins (clean v_add_co / v_subb chains, no 64-bit compares):
//   x = lo + hl 2^32 - (hl + hh) + (carry - borrow) EPS
__device__ __forceinline__ u64 fold4(u32 r0, u32 r1, u32 hl, u32 hh) {
    u32 cs, c1, bw, B, k1, k2;
    u32 s0 = __builtin_addc(hl, hh, 0u, &cs);     // hl + hh (33 bits)
    u32 a1w = __builtin_addc(r1, hl, 0u, &c1);    // high word of lo + hl 2^32, carry c1 = one 2^64
    u32 d0 = __builtin_subc(r0, s0, 0u, &bw);
    u32 d1 = __builtin_subc(a1w, cs, bw, &B);     // borrow B = minus one 2^64
    u32 mC = 0u - c1, mB = 0u - B;                // (c1 - B) EPS as a 64-bit two's complement value
    u32 cl = __builtin_subc(mC, mB, 0u, &k1);
    u32 ch = 0u - k1;
    u32 f0 = __builtin_addc(d0, cl, 0u, &k2);
    u32 f1 = d1 + ch + k2;
    return (u64)f0 | ((u64)f1 << 32);
}
__device__ __forceinline__ u64 mul4(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + ((p01 >> 32) + (p10 >> 32));
    return fold4((u32)p00, (u32)p10, (u32)p11, (u32)(p11 >> 32));
}
// M5: as M4, the last partial product's second addend enters through the carry chain instead of a 64-bit add
__device__ __forceinline__ u64 mul5(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32);
    u32 c;
    u32 hl = __builtin_addc((u32)p11, (u32)(p10 >> 32), 0u, &c);
    u32 hh = (u32)(p11 >> 32) + c;
    return fold4((u32)p00, (u32)p10, hl, hh);
}

// M6: 4 mads; fold = one v_mad_u64_u32 for lo + hl (2^32 - 1) with its carry in an SGPR pair, a 64-bit subtract of hh with
// its borrow in another, the (carry - borrow) (2^32 - 1) correction selected through scalar mask logic: 7 VALU ops instead of 11.
// gfx950 needs two wait states between a VALU that writes an SGPR and a VALU that reads it (carry-in / select): s_nop 1 inside.
template <bool NOPS>
__device__ __forceinline__ u64 fold6(u32 r0, u32 r1, u32 hl, u32 hh) {
    const u64 lo = (u64)r0 | ((u64)r1 << 32);
    u64 t, c, b0, B;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
    u32 d0, d1;
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(d0), "=s"(b0) : "v"((u32)t), "v"(hh));
    if (NOPS) asm("s_nop 1\n\tv_subb_co_u32 %0, %1, %2, 0, %3" : "=v"(d1), "=s"(B) : "v"((u32)(t >> 32)), "s"(b0));
    else asm("v_subb_co_u32 %0, %1, %2, 0, %3" : "=v"(d1), "=s"(B) : "v"((u32)(t >> 32)), "s"(b0));
    const u64 oc = c & ~B, ob = B & ~c;   // scalar: carry only / borrow only
    u32 x, cl, ch;
    if (NOPS) {
        asm("s_nop 1\n\tv_cndmask_b32 %0, 0, 1, %1" : "=v"(x) : "s"(ob));
        asm("s_nop 1\n\tv_cndmask_b32 %0, %1, -1, %2" : "=v"(cl) : "v"(x), "s"(oc));
        asm("s_nop 1\n\tv_cndmask_b32 %0, 0, -1, %1" : "=v"(ch) : "s"(ob));
    } else {
        asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(x) : "s"(ob));
        asm("v_cndmask_b32 %0, %1, -1, %2" : "=v"(cl) : "v"(x), "s"(oc));
        asm("v_cndmask_b32 %0, 0, -1, %1" : "=v"(ch) : "s"(ob));
    }
    const u64 d = (u64)d0 | ((u64)d1 << 32), k = (u64)cl | ((u64)ch << 32);
    return d + k;   // (cl, ch) = +EPS, -EPS or 0 as a 64-bit two's complement value; no second overflow (see fold4)
}
template <bool NOPS>
__device__ __forceinline__ u64 mul6(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + ((p01 >> 32) + (p10 >> 32));
    return fold6<NOPS>((u32)p00, (u32)p10, (u32)p11, (u32)(p11 >> 32));
}
// M8: as M6, the correction applied with a mad (d + cl, zero extended) and a masked subtract on the high word
__device__ __forceinline__ u64 mul8(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + ((p01 >> 32) + (p10 >> 32));
    const u32 r0 = (u32)p00, r1 = (u32)p10, hl = (u32)p11, hh = (u32)(p11 >> 32);
    const u64 lo = (u64)r0 | ((u64)r1 << 32);
    u64 t, c, b0m, B, dummy;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=s"(c) : "v"(hl), "v"(lo));
    u32 d0, d1;
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(d0), "=s"(b0m) : "v"((u32)t), "v"(hh));
    asm("s_nop 1\n\tv_subb_co_u32 %0, %1, %2, 0, %3" : "=v"(d1), "=s"(B) : "v"((u32)(t >> 32)), "s"(b0m));
    const u64 oc = c & ~B, ob = B & ~c;
    u32 x, cl;
    asm("s_nop 1\n\tv_cndmask_b32 %0, 0, 1, %1" : "=v"(x) : "s"(ob));
    asm("s_nop 1\n\tv_cndmask_b32 %0, %1, -1, %2" : "=v"(cl) : "v"(x), "s"(oc));
    const u64 d = (u64)d0 | ((u64)d1 << 32);
    u64 f;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=&v"(f), "=s"(dummy) : "v"(cl), "v"(d));
    u32 f1;
    asm("s_nop 1\n\tv_subb_co_u32 %0, %1, %2, 0, %3" : "=v"(f1), "=s"(dummy) : "v"((u32)(f >> 32)), "s"(ob));
    return (u64)(u32)f | ((u64)f1 << 32);
}

template <int V>
__device__ __forceinline__ u64 mulv(u64 a, u64 b) {
    if (V == 6) return mul6<true>(a, b);
    if (V == 7) return mul6<false>(a, b);
    if (V == 8) return mul8(a, b);
    if (V == 4) return mul4(a, b);
    if (V == 5) return mul5(a, b);
    if (V == 0) return mul0(a, b);
    if (V == 1) return mul1(a, b);
    if (V == 2) return mul2(a, b);
    return mul3(a, b);
}

#define ITER 2048
template <int V>
__global__ __launch_bounds__(256) void kbench(u64* out, u64 seed) {
    u64 x[8];
    for (int i = 0; i < 8; i++) x[i] = canon(seed * (threadIdx.x + 1) * (2 * i + 3));
    u64 m = canon(seed ^ 0x123456789abcdefULL);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = mulv<V>(x[i], m);
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = mulv<V>(x[i], x[(i + 1) & 7]);
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
__global__ void kcheck(const u64* a, const u64* b, u64* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = canon(canon(mulv<V>(a[i], b[i])));
}

static u64 host_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % P); }

template <int V>
void run(const char* name) {
    // check
    std::vector<u64> a, b;
    u64 edge[] = {0, 1, 2, EPS, EPS + 1, P - 1, P, P + 1, ~0ULL, ~0ULL - 1, 1ULL << 32, (1ULL << 32) - 2, 0xFFFFFFFE00000001ULL, 0x8000000000000000ULL};
    for (u64 x : edge) for (u64 y : edge) { a.push_back(x); b.push_back(y); }
    u64 s = 88172645463325252ULL;
    for (int i = 0; i < 100000; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a.push_back(s); s ^= s << 13; s ^= s >> 7; s ^= s << 17; b.push_back(V == 0 ? s % P : s); }
    if (V == 0) for (auto& x : a) x %= P, (void)0;
    if (V == 0) for (auto& x : b) x %= P;
    int n = a.size();
    u64 *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kcheck<V>, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<u64> o(n);
    hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (o[i] != host_mul(a[i], b[i])) { if (bad < 3) printf("  MISMATCH %s a=%llx b=%llx got %llx want %llx\n", name, a[i], b[i], o[i], host_mul(a[i], b[i])); bad++; }
    // bench
    int blocks = 256 * 8;
    u64* d; hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kbench<V>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kbench<V>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double muls = (double)blocks * 256 * ITER * 16;
    double wave_muls_per_simd = muls / 64 / 1024;
    printf("%-10s bad=%d  %8.3f ms  %7.2f Gmul/s  %6.1f cycles/wave-mul/SIMD @2.4GHz\n", name, bad, ms, muls / ms / 1e6, ms * 1e-3 * 2.4e9 / wave_muls_per_simd);
    hipFree(d); hipFree(da); hipFree(db); hipFree(dout);
}

int main() {
    run<0>("M0 canon");
    run<1>("M1 lazy");
    run<2>("M2 lazy4m");
    run<3>("M3 phi");
    run<4>("M4 limb32");
    run<5>("M5 limb32b");
    run<6>("M6 madfold");
    run<7>("M7 nonops");
    run<8>("M8 madfold2");
    return 0;
}
