// BabyBear field p = 2^31 - 2^27 + 1 in 32-bit Montgomery form (R = 2^32) for gfx950 device code
// and the host-side table builders.
//
// Replaces what the reference takes from p3-baby-bear / p3-monty-31 (Cargo.toml:17-24, unpinned):
// generator 31, two_adic_generator(27) = 0x1a427a41 - recalled from upstream, self-consistent, NOT pinned
// by any fixture in the reference.  Device-resident BabyBear data (coefficients, LDE)
// is kept in Montgomery form; digests, caps and everything that crosses the C ABI is canonical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bb {

typedef unsigned int u32;
typedef unsigned long long u64;

static constexpr u32 P = 0x78000001u;          // 2013265921
static constexpr u32 PINV = 0x88000001u;       // P^-1 mod 2^32
static constexpr u32 R1 = 0x0ffffffeu;         // 2^32 mod P  (Montgomery form of 1)
static constexpr u32 R2 = 0x45dddde3u;         // 2^64 mod P
static constexpr u32 GENERATOR = 31;           // F::generator(), the LDE coset shift
static constexpr u32 TWO_ADIC_GEN_27 = 0x1a427a41u;

__host__ __device__ __forceinline__ u32 add(u32 a, u32 b) {  // a, b < P
    u32 s = a + b;
    u32 t = s - P;
    return t < s ? t : s;  // min(s, s - P) as unsigned: s - P wraps high when s < P
}
// a, b < P.  a >= b: d in [0, P) and d + P is larger; a < b: d wraps to >= 2^32 - P and d + P wraps to the
// canonical value, which is then the smaller one - so min() selects without a compare (v_sub, v_add, v_min).
__host__ __device__ __forceinline__ u32 sub(u32 a, u32 b) {
    u32 d = a - b;
    u32 t = d + P;
    return t < d ? t : d;
}
__host__ __device__ __forceinline__ u32 neg(u32 a) { return a ? P - a : 0; }

// t < P * 2^32  ->  t * 2^-32 mod P, canonical.  m = -lo(t) / P mod 2^32 makes t + m P a multiple of 2^32 below 2 P 2^32, so the
// whole reduction step is ONE multiply-add (v_mad_u64_u32 on the device) whose high word lies in [0, 2P); min() with the word less
// P selects the canonical one.  (A v_mul_hi_u32 and a subtraction more in the t - m P form with m = lo(t) / P.)
static constexpr u32 NPINV = 0u - PINV;        // -P^-1 mod 2^32
__host__ __device__ __forceinline__ u32 reduce(u64 t) {
    const u32 m = (u32)t * NPINV;
    const u32 r = (u32)((t + (u64)m * P) >> 32);
    const u32 e = r - P;
    return e < r ? e : r;
}
// t < P * 2^32 -> a word in [0, 2P) congruent to t * 2^-32: the reduction without its final selection;
// good wherever the result only feeds another Montgomery product with a canonical partner or an unreduced sum
__host__ __device__ __forceinline__ u32 reduce_lazy(u64 t) {
    const u32 m = (u32)t * NPINV;
    return (u32)((t + (u64)m * P) >> 32);
}
__host__ __device__ __forceinline__ u32 mul_lazy(u32 a, u32 b) { return reduce_lazy((u64)a * b); }  // a * b < P * 2^32
// SIGNED Montgomery product: a, b signed words with |a|, |b| <= 1.03 P -> a word r congruent to a b 2^-32 with |r| < 0.97 P + 1.
// t = a b exactly; m = lo(t) / P mod 2^32 taken as SIGNED makes t - m P a multiple of 2^32 of magnitude < P^2 + 2^31 P, so its
// high word needs no selection at all and the representation is closed under itself (0.97 P < 1.03 P): a product is two
// multiplies and one multiply-add (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32), with nothing after them - the unsigned forms
// above end in [0, 2P), which a second product with a lazy partner would overflow.  `bias` (0 or P) is added for a consumer that
// wants a non-negative word: r + P lies in (0, 2 P).
__host__ __device__ __forceinline__ int mul_signed(int a, int b, u32 bias = 0) {
    const long long t = (long long)a * b;
    const int m = (int)((u32)t * PINV);
    const long long d = t - (long long)m * (int)P;   // one v_mad_i64_i32 on the device: the low word comes out zero
    return (int)((u32)(d >> 32) + bias);
}
// the same reduction for a small unsigned 64-bit sum: hi(t) - P/2 < r <= hi(t) + P/2, i.e. within +-1.03 P for t < 2^38
__host__ __device__ __forceinline__ int reduce_signed(u64 t) {
    const int m = (int)((u32)t * PINV);
    const long long d = (long long)t - (long long)m * (int)P;
    return (int)(d >> 32);
}
// Montgomery product: (a R)(b R) -> (ab R)
__host__ __device__ __forceinline__ u32 mul(u32 a, u32 b) { return reduce((u64)a * b); }
__host__ __device__ __forceinline__ u32 sqr(u32 a) { return mul(a, a); }
__host__ __device__ __forceinline__ u32 to_mont(u32 x) { return mul(x, R2); }  // canonical -> Montgomery
__host__ __device__ __forceinline__ u32 from_mont(u32 x) { return reduce((u64)x); }

__host__ __device__ inline u32 pow(u32 b, u64 e) {  // Montgomery in/out
    u32 r = R1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
__host__ __device__ inline u32 inv(u32 a) { return pow(a, P - 2); }
// F::two_adic_generator(bits) in Montgomery form
__host__ __device__ inline u32 two_adic_generator(unsigned bits) {
    u32 g = to_mont(TWO_ADIC_GEN_27);
    for (unsigned i = bits; i < 27; i++) g = sqr(g);
    return g;
}

}  // namespace bb
