// Goldilocks field p = 2^64 - 2^32 + 1 for gfx950 device code (and the host-side table builders).
//
// Replaces the arithmetic the reference takes from p3-goldilocks (Cargo.toml:17-24); call sites on
// the hot path: field/src/fft.rs:141-159 (butterflies), hash/poseidon_goldilocks.rs:840-846 (x^7),
// fri/oracle.rs:141 (F::generator() = 7).  All kernels keep elements canonical (< p) in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef unsigned long long u64;
typedef unsigned int u32;

static constexpr u64 P = 0xFFFFFFFF00000001ULL;
static constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
static constexpr u64 GENERATOR = 7;        // F::generator(), the LDE coset shift
static constexpr u64 TWO_ADIC_GEN_32 = 1753635133440165772ULL;  // order 2^32

__host__ __device__ __forceinline__ u64 canon(u64 x) { return x >= P ? x - P : x; }

// The ONE host / device split of this header: the forms that differ between the device pass (32-bit limbs, the instructions the
// issue-cost model of DESIGN.md 4 prices) and the host pass (the table builders, the verifier; plain 64 / 128-bit C).
//   add / sub: a, b canonical -> canonical.  s - p == s + EPS (mod 2^64), so both the wrapped case (a + b >= 2^64) and the s >= p
//     case select s + EPS.  Device form: four v_add(c)_co + two v_cndmask, all 32-bit-rate ops (the u64 form lowers to two
//     v_lshl_add_u64 and two v_cmp_u64, ~40 % more issue cycles per butterfly).
//   mad_carry / addc_carry: a1 b0 + p01 as ONE v_mad_u64_u32 whose carry (weight 2^96) comes out in the scalar carry operand and
//     enters the top limb through a v_addc (mul_limbs below); mad_one: x * 1 + acc, the mad as an adder.
// Under hipcc both sets exist in both passes as __device__ / __host__ OVERLOADS (a __host__ __device__ caller binds to its own
// side's); a plain C++ compiler (tests/host_shim) sees the host set only.
#if defined(__HIPCC__)
__device__ __forceinline__ u64 add(u64 a, u64 b) {
    u32 c0, c1, d0, d1;
    u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
    u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
    u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &d0);
    u32 t1 = __builtin_addc(s1, 0u, d0, &d1);
    const bool sel = (c1 | d1) != 0;
    return sel ? ((u64)t0 | ((u64)t1 << 32)) : ((u64)s0 | ((u64)s1 << 32));
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u32 b0, b1, k;
    u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &b0);
    u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b0, &b1);
    u32 m = 0u - b1;  // EPS when the difference wrapped: d - EPS == d + p (mod 2^64)
    u32 e0 = __builtin_subc(d0, m, 0u, &k);
    u32 e1 = d1 - k;
    return (u64)e0 | ((u64)e1 << 32);
}
__device__ __forceinline__ u64 mulhi(u64 a, u64 b) { return __umul64hi(a, b); }
__device__ __forceinline__ u64 mad_carry(u32 a, u32 b, u64 acc, u64& carry) {   // a b + acc = result + 2^64 [carry]
    u64 m;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m), "=s"(carry) : "v"(a), "v"(b), "v"(acc));
    return m;
}
__device__ __forceinline__ u32 addc_carry(u32 x, u64 carry) {                   // x + [carry]
    u32 top;
    u64 unused;
    asm("v_addc_co_u32 %0, %1, 0, %2, %3" : "=v"(top), "=s"(unused) : "v"(x), "s"(carry));
    return top;
}
__device__ __forceinline__ u64 mad_one(u32 x, u64 acc) {                        // x * 1 + acc (< 2^64 where it is used)
    u64 unused;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0" : "+v"(acc), "=s"(unused) : "v"(x));
    return acc;
}
#endif
__host__ inline u64 add(u64 a, u64 b) {
    u64 s, u;
    bool c = __builtin_uaddll_overflow(a, b, &s);
    bool c2 = __builtin_uaddll_overflow(s, EPS, &u);
    return (c | c2) ? u : s;
}
__host__ inline u64 sub(u64 a, u64 b) {
    u64 d;
    bool br = __builtin_usubll_overflow(a, b, &d);
    return d - (br ? EPS : 0);  // + p
}
__host__ inline u64 mulhi(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) >> 64); }
__host__ inline u64 mad_carry(u32 a, u32 b, u64 acc, u64& carry) {
    const unsigned __int128 t = (unsigned __int128)a * b + acc;
    carry = (u64)(t >> 64);
    return (u64)t;
}
__host__ inline u32 addc_carry(u32 x, u64 carry) { return x + (u32)(carry & 1); }
__host__ inline u64 mad_one(u32 x, u64 acc) { return acc + x; }
__host__ __device__ __forceinline__ u64 neg(u64 a) { return a ? P - a : 0; }

// (lo + 2^64 hi) mod p, any 128-bit input, canonical output.
// 2^64 = 2^32 - 1 and 2^96 = -1 (mod p):  x = lo - hi_hi + hi_lo * (2^32 - 1)
__host__ __device__ __forceinline__ u64 reduce128(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;           // borrow: add p (mod 2^64)
    u64 t1 = (hi_lo << 32) - hi_lo;      // hi_lo * EPS, < 2^64
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += EPS;              // carry: 2^64 = EPS
    return canon(t2);
}
// (r0 + 2^32 r1 + 2^64 hl + 2^96 hh) mod p as SOME u64 congruent to it (not canonical; < 2^64 < 2p).
// 2^64 = 2^32 - 1 and 2^96 = -1:  x = lo + hl 2^32 - (hl + hh), then every carry out of bit 64 is worth +EPS and every
// borrow -EPS.  Written on 32-bit limbs with the carry builtins so that the compiler emits plain v_add_co / v_subb chains
// (11 32-bit VALU ops) instead of 64-bit adds and compares: 64.4 vs 70.5 cycles per wave-multiply
// (tools/microbench_mulmod.hip, M4 vs M2).  No second overflow is possible: after a carry the sum is < 2^64 - 2^32,
// after a borrow it is > 2^64 - 2^33.
__host__ __device__ __forceinline__ u64 fold128(u32 r0, u32 r1, u32 hl, u32 hh) {
    u32 cs, c1, bw, B, k1, k2;
    u32 s0 = __builtin_addc(hl, hh, 0u, &cs);   // hl + hh (33 bits: s0, cs)
    u32 a1 = __builtin_addc(r1, hl, 0u, &c1);   // high word of lo + hl 2^32; c1 = one 2^64
    u32 d0 = __builtin_subc(r0, s0, 0u, &bw);
    u32 d1 = __builtin_subc(a1, cs, bw, &B);    // B = minus one 2^64
    u32 mC = 0u - c1, mB = 0u - B;              // (c1 - B) EPS as a 64-bit two's-complement value (cl, ch)
    u32 cl = __builtin_subc(mC, mB, 0u, &k1);
    u32 ch = 0u - k1;
    u32 f0 = __builtin_addc(d0, cl, 0u, &k2);
    u32 f1 = d1 + ch + k2;
    return (u64)f0 | ((u64)f1 << 32);
}
// The same for a five-limb value r0 + 2^32 r1 + 2^64 hl + 2^96 hh + 2^128 r4 with a small r4 (a sum of a few 128-bit
// products): 2^128 = -2^32, so r4 2^32 joins the subtrahend hl + hh, whose high word becomes cs + r4 (r4 <= 2^31).
__device__ __forceinline__ u64 fold160(u32 r0, u32 r1, u32 hl, u32 hh, u32 r4) {
    u32 cs, c1, bw, B, k1, k2;
    u32 s0 = __builtin_addc(hl, hh, 0u, &cs);
    u32 s1 = cs + r4;
    u32 a1 = __builtin_addc(r1, hl, 0u, &c1);
    u32 d0 = __builtin_subc(r0, s0, 0u, &bw);
    u32 d1 = __builtin_subc(a1, s1, bw, &B);
    u32 mC = 0u - c1, mB = 0u - B;
    u32 cl = __builtin_subc(mC, mB, 0u, &k1);
    u32 ch = 0u - k1;
    u32 f0 = __builtin_addc(d0, cl, 0u, &k2);
    u32 f1 = d1 + ch + k2;
    return (u64)f0 | ((u64)f1 << 32);
}
// a * b as four 32-bit limbs (gfx950 issue costs, measured: v_mad_u64_u32 4.5 cycles per wave, carry / select / 64-bit-add ops
// ~2.9 in a mixed stream, v_mov 2.4; tools/microbench_*.hip).
// FIVE = false (round 4, the default): FOUR multiply-adds.  The second cross product takes the whole of p01 as its addend - a 64-bit
//   register pair that is already in place - and the one carry that sum can produce (weight 2^96) comes out in the mad's scalar carry
//   operand and enters the top limb through a v_addc: 4 mads + 3 plain.
// FIVE = true (rounds 2-3): four mads for the partial products and a fifth as an ADDER - the second carry word enters the top
//   product as x * 1 + acc.  The strided LDE pass (k_gl_lde_pa16x2, k_gl_lde_pa32) measured 3 % faster with this form: its
//   140-register body loses more to the extra scalar carry pair than it gains from the mad.
template <bool FIVE = false>
__host__ __device__ __forceinline__ void mul_limbs(u64 a, u64 b, u32& r0, u32& r1, u32& hl, u32& hh) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    if constexpr (!FIVE) {
        u64 carry;
        const u64 m2 = mad_carry(a1, b0, p01, carry);   // a1 b0 + p01 = m2 + 2^64 [carry]
        const u64 p11 = (u64)a1 * b1 + (m2 >> 32);      // < 2^64 - 2^32: the carry still fits
        r0 = (u32)p00; r1 = (u32)m2; hl = (u32)p11; hh = addc_carry((u32)(p11 >> 32), carry);
    } else {
        const u64 p10 = (u64)a1 * b0 + (u32)p01;
        const u64 p11 = mad_one((u32)(p10 >> 32), (u64)a1 * b1 + (p01 >> 32));   // < 2^64: the product is < 2^128
        r0 = (u32)p00; r1 = (u32)p10; hl = (u32)p11; hh = (u32)(p11 >> 32);
    }
}
// (x0 + 2^32 x1 + 2^64 x2 + 2^96 x3) / 2^64 mod p as SOME u64 congruent to it: Montgomery reduction with R = 2^64, for which
// p = 2^64 - 2^32 + 1 needs no multiplication (-1/p = -(1 + 2^32) mod 2^64):  a = lo + (lo << 32), b = a - (a >> 32) - carry,
// r = hi - b, minus EPS when that borrows.  b <= p - 1 for every 128-bit input, so the last step cannot borrow twice.
// 8 carry ops against fold128's 11.
__host__ __device__ __forceinline__ u64 mont_fold(u32 x0, u32 x1, u32 x2, u32 x3) {
    u32 e, bw, k0, c0, c, k;
    const u32 a1 = __builtin_addc(x1, x0, 0u, &e);   // a = (x0, a1), carry e
    const u32 b0 = __builtin_subc(x0, a1, e, &bw);
    const u32 b1 = __builtin_subc(a1, 0u, bw, &k0);
    const u32 r0 = __builtin_subc(x2, b0, 0u, &c0);
    const u32 r1 = __builtin_subc(x3, b1, c0, &c);
    const u32 m = 0u - c;                            // EPS when hi < b: r + p = r - EPS (mod 2^64)
    const u32 f0 = __builtin_subc(r0, m, 0u, &k);
    const u32 f1 = r1 - k;
    return (u64)f0 | ((u64)f1 << 32);
}
// a * t for a table value stored in MONTGOMERY form (t R mod p, R = 2^64): a (t R) / R = a t (and Montgomery times Montgomery
// stays Montgomery).  Any u64 operands give some residue (mul_mont_lazy).  CANONICAL operands give the CANONICAL product with no
// further step (mul_mont): a, t R <= p - 1 makes the product's high word xh <= (p - 1)^2 / 2^64 < p, and mont_fold returns
// xh - b without a borrow (<= xh < p) or xh - b + p with one (in [p - b, p - 1], b <= p - 1).  So a multiplication by a table
// value costs 5 mads + 8 carry ops where gl::mul takes 5 + 11 + 4 (fold, then canonicalise).
template <bool FIVE = false>
__host__ __device__ __forceinline__ u64 mul_mont_lazy(u64 a, u64 t_mont) {
    u32 r0, r1, hl, hh;
    mul_limbs<FIVE>(a, t_mont, r0, r1, hl, hh);
    return mont_fold(r0, r1, hl, hh);
}
// mul_mont is mul_mont_lazy under the name that states its contract: CANONICAL operands in, canonical product out.  The NTT kernels
// rely on it for what they store: their data operands are canonical (loaded from memory, or outputs of add / sub / mul_pow2 /
// mul_mont), every table entry is canonical, and the running twiddles `f <- f * ratio` (computed with mul_mont_lazy) start from
// canonical table entries and therefore stay canonical by induction.  A lazy operand (some residue >= p) would make the product a
// non-canonical word in memory - and a Merkle leaf that differs from the reference's; keep add_lazy-style values out of these chains.
template <bool FIVE = false>
__host__ __device__ __forceinline__ u64 mul_mont(u64 a_canonical, u64 t_mont_canonical) { return mul_mont_lazy<FIVE>(a_canonical, t_mont_canonical); }
// x R mod p on the host (table builders)
__host__ __device__ inline u64 to_mont_slow(u64 x);

// a * b mod p, any u64 in, canonical out.
#if defined(__HIPCC__)   // (device overload on limbs; the host's is one 64 x 64 -> 128 multiplication: the transcript hashes with it)
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u32 r0, r1, hl, hh;
    mul_limbs(a, b, r0, r1, hl, hh);
    return canon(fold128(r0, r1, hl, hh));
}
#endif
__host__ inline u64 mul(u64 a, u64 b) {
    const unsigned __int128 t = (unsigned __int128)a * b;
    return reduce128((u64)t, (u64)(t >> 64));
}
__host__ __device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }
__host__ __device__ inline u64 to_mont_slow(u64 x) { return mul(x, EPS); }   // R = 2^64 mod p = 2^32 - 1

__host__ __device__ inline u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
__host__ __device__ inline u64 inv(u64 a) { return pow(a, P - 2); }

// F::two_adic_generator(bits) (call sites field/src/fft.rs:16, circuit_data.rs:666-668)
__host__ __device__ inline u64 two_adic_generator(unsigned bits) {
    u64 g = TWO_ADIC_GEN_32;
    for (unsigned i = bits; i < 32; i++) g = sqr(g);
    return g;
}

// ---- quadratic extension F[x]/(x^2 - 7): BinomialExtensionField<Goldilocks, 2> (field/src/types.rs:25-29)
static constexpr u64 EXT_W = 7;
struct ext2 {
    u64 c0, c1;
};
__host__ __device__ __forceinline__ ext2 e2(u64 a, u64 b = 0) { return ext2{a, b}; }
__host__ __device__ __forceinline__ ext2 add(ext2 a, ext2 b) { return ext2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
__host__ __device__ __forceinline__ ext2 sub(ext2 a, ext2 b) { return ext2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
__host__ __device__ __forceinline__ ext2 mul(ext2 a, ext2 b) {
    return ext2{add(mul(a.c0, b.c0), mul(EXT_W, mul(a.c1, b.c1))), add(mul(a.c0, b.c1), mul(a.c1, b.c0))};
}
__host__ __device__ __forceinline__ ext2 scale(ext2 a, u64 s) { return ext2{mul(a.c0, s), mul(a.c1, s)}; }
__host__ __device__ inline ext2 inv(ext2 a) {
    u64 nrm = sub(sqr(a.c0), mul(EXT_W, sqr(a.c1)));
    u64 ni = inv(nrm);
    return ext2{mul(a.c0, ni), mul(neg(a.c1), ni)};
}
__host__ __device__ inline ext2 pow(ext2 b, u64 e) {
    ext2 r = e2(1);
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}

}  // namespace gl
