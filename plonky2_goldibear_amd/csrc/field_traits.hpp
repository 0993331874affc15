// Field traits the prover kernels (kernels_prover.hip) and the prove() orchestration (prover_host.inc) are
// written against, so that one source serves both of the reference's configurations:
//   GlF  Goldilocks, D = 2 (x^2 - 7),  H = 4, Poseidon-12   (PoseidonGoldilocksConfig, plonk/config.rs:119-133)
//   BbF  BabyBear,   D = 4 (x^4 - 11), H = 8, Poseidon2-16  (Poseidon2BabyBearConfig,  plonk/config.rs:135-150)
// T is the DEVICE representation of a base element: canonical u64 for Goldilocks, 32-bit Montgomery for
// BabyBear (bb_field.hpp); enc()/dec() convert from/to the canonical values the transcript and the proof
// bytes use.  The BabyBear extension non-residue W = 11 is, like the other BabyBear field constants, recalled
// from upstream Plonky3 and unpinned.
#pragma once
#include "bb_field.hpp"
#include "gl_field.hpp"

#define GB_HD __host__ __device__ __forceinline__
// a table that exists twice (a __device__ copy and a host copy), named by the pass that compiles the expression
#if defined(__HIP_DEVICE_COMPILE__)
#define GB_DEV_OR_HOST(dev, host) dev
#else
#define GB_DEV_OR_HOST(dev, host) host
#endif

namespace gbk {

typedef unsigned long long u64;
typedef unsigned int u32;

struct GlF {
    typedef u64 T;
    typedef gl::ext2 E;
    static constexpr u32 D = 2, H = 4, SPONGE_W = 12, ORDER_BITS = 64, TWO_ADICITY = 32, TAG = 0;
    static constexpr u64 ORDER = gl::P;
    static GB_HD T zero() { return 0; }
    static GB_HD T one() { return 1; }
    static GB_HD T enc(u64 canonical) { return canonical; }
    static GB_HD u64 dec(T x) { return x; }
    static GB_HD T add(T a, T b) { return gl::add(a, b); }
    static GB_HD T sub(T a, T b) { return gl::sub(a, b); }
    static GB_HD T mul(T a, T b) { return gl::mul(a, b); }
    // a * b as ANY u64 residue: for product chains whose result is only multiplied again (mul() accepts any u64 and
    // canonicalises its own output; add / sub need canonical operands)
    static GB_HD T mul_lazy(T a, T b) {
        u32 r0, r1, hl, hh;
        gl::mul_limbs(a, b, r0, r1, hl, hh);
        return gl::fold128(r0, r1, hl, hh);
    }
    // a + b for a result that is only multiplied (by a canonical partner): canonical here
    static GB_HD T add_lazy(T a, T b) { return gl::add(a, b); }
    // Multiplication by a per-proof CONSTANT (a challenge, a table value) that the host stores in "constant form": for Goldilocks
    // c R (Montgomery), so that x c costs 5 mads + 8 carry ops and comes out canonical for canonical x (gl::mul_mont) instead of
    // 5 + 11 + 4; for BabyBear the device form is Montgomery already.
    static GB_HD T cform(T c) { return gl::to_mont_slow(c); }
    // Product chains np <- np * f that are only multiplied again: Montgomery steps (8-op fold instead of 11); each step divides
    // by R, so a chain of `len` factors starts from R^len instead of 1 and ends on the plain product (some residue).
    static GB_HD T chain_one(u32 len) {
        constexpr unsigned __int128 R = gl::EPS;
        struct Tab { u64 v[17]; };
        constexpr Tab t = [] {
            Tab x{};
            unsigned __int128 a = 1;
            for (int i = 0; i < 17; i++) { x.v[i] = (u64)a; a = a * R % gl::P; }
            return x;
        }();
        return t.v[len];
    }
    static GB_HD T mul_chain(T acc, T f) { return gl::mul_mont_lazy(acc, f); }
    static GB_HD T mulc(T x, T c_form) { return gl::mul_mont(x, c_form); }
    // sum_t term_t * c_t with wave-uniform constants c_t in constant form (the quotient kernel's alpha fold): two sums side by
    // side, because BabyBear's form pairs them (below); here simply add(acc, mulc(term, c))
    typedef T Acc;
    static GB_HD Acc acc_from(T x) { return x; }
    static GB_HD void acc_mac2(Acc& a0, Acc& a1, T term, T c0, T c1) {
        a0 = add(a0, mulc(term, c0));
        a1 = add(a1, mulc(term, c1));
    }
    static GB_HD void acc_mac(Acc& a, T term, T c) { a = add(a, mulc(term, c)); }
    static GB_HD T acc_finish(Acc a) { return a; }
    static GB_HD T inv(T a) { return gl::inv(a); }
    static GB_HD T pow(T a, u64 e) { return gl::pow(a, e); }
    static GB_HD T generator() { return gl::GENERATOR; }
    static GB_HD T two_adic_generator(u32 bits) { return gl::two_adic_generator(bits); }
    static GB_HD T ext_w() { return 7; }  // the extension is F[x]/(x^2 - 7)
    static GB_HD E ezero() { return gl::e2(0); }
    static GB_HD E efrom(T x) { return gl::e2(x); }
    static GB_HD E eadd(E a, E b) { return gl::add(a, b); }
    static GB_HD E esub(E a, E b) { return gl::sub(a, b); }
    static GB_HD E emul(E a, E b) { return gl::mul(a, b); }
    static GB_HD E escale(E a, T s) { return gl::scale(a, s); }
    static GB_HD E einv(E a) { return gl::inv(a); }
    static GB_HD T coord(const E& e, u32 k) { return k ? e.c1 : e.c0; }
    static GB_HD void set_coord(E& e, u32 k, T v) { if (k) e.c1 = v; else e.c0 = v; }
};

struct bb_ext4 {
    u32 c[4];
};

struct BbF {
    typedef u32 T;
    typedef bb_ext4 E;
    static constexpr u32 D = 4, H = 8, SPONGE_W = 16, ORDER_BITS = 31, TWO_ADICITY = 27, TAG = 1;
    static constexpr u64 ORDER = bb::P;
    static constexpr u32 W_MONT = (u32)((11ull << 32) % bb::P);  // the non-residue 11 in Montgomery form
    static GB_HD T zero() { return 0; }
    static GB_HD T one() { return bb::R1; }
    static GB_HD T enc(u64 canonical) { return bb::to_mont((u32)canonical); }
    static GB_HD u64 dec(T x) { return bb::from_mont(x); }
    static GB_HD T add(T a, T b) { return bb::add(a, b); }
    static GB_HD T sub(T a, T b) { return bb::sub(a, b); }
    static GB_HD T mul(T a, T b) { return bb::mul(a, b); }
    static GB_HD T mul_lazy(T a, T b) { return bb::mul(a, b); }
    // a + b < 2p unreduced: fine as ONE operand of a Montgomery product whose other operand is canonical (2 p^2 < p 2^32)
    static GB_HD T add_lazy(T a, T b) { return a + b; }
    static GB_HD T cform(T c) { return c; }
    static GB_HD T mulc(T x, T c_form) { return bb::mul(x, c_form); }
    static GB_HD T chain_one(u32) { return bb::R1; }
    static GB_HD T mul_chain(T acc, T f) { return bb::mul(acc, f); }
    // sum_t term_t * c_t (term Montgomery form, c_t wave-uniform Montgomery constants) kept UNREDUCED as a 96-bit integer: a term
    // costs one v_mad_u64_u32 and one add-with-carry per sum instead of a Montgomery product and a canonical addition (8
    // instructions); one reduction per sum at the end.  Two sums at a time so that each carry-out has its own scalar register and
    // an instruction between its write and its read (VALU-written carry -> v_addc needs two wait states on this chip).
    struct Acc {
        u64 lo;
        u32 hi;
    };
    static GB_HD Acc acc_from(T x) { return Acc{(u64)x << 32, 0u}; }   // x 2^32: the final division by 2^32 returns x
    static GB_HD void acc_mac2(Acc& a0, Acc& a1, T term, T c0, T c1) {
#if defined(__HIP_DEVICE_COMPILE__)
        u64 k0, k1;
        asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\t"
            "v_mad_u64_u32 %2, %5, %6, %8, %2\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %1, %4, 0, %1, %4\n\t"
            "v_addc_co_u32_e64 %3, %5, 0, %3, %5"
            : "+v"(a0.lo), "+v"(a0.hi), "+v"(a1.lo), "+v"(a1.hi), "=&s"(k0), "=&s"(k1)
            : "v"(term), "s"(c0), "s"(c1));
#else
        acc_mac(a0, term, c0);
        acc_mac(a1, term, c1);
#endif
    }
    static GB_HD void acc_mac(Acc& a, T term, T c) {
        const u64 p = (u64)term * c, s = a.lo + p;
        a.hi += s < p;
        a.lo = s;
    }
    // (hi 2^64 + lo) / 2^32 mod p = hi 2^32 + (lo >> 32) + mont_reduce(lo & 0xffffffff), every piece canonical
    static GB_HD T acc_finish(Acc a) {
        const T top = bb::mul(a.hi, bb::R2);                      // hi R^2 / R = hi 2^32 mod p   (hi < 2^31: a few hundred terms)
        const T mid = bb::mul((u32)(a.lo >> 32), bb::R1);         // x R / R = x mod p for any 32-bit x (x R1 < p 2^32)
        const T low = bb::reduce((u64)(u32)a.lo);
        return bb::add(bb::add(top, mid), low);
    }
    static GB_HD T inv(T a) { return bb::inv(a); }
    static GB_HD T pow(T a, u64 e) { return bb::pow(a, e); }
    static GB_HD T generator() { return bb::to_mont(bb::GENERATOR); }
    static GB_HD T two_adic_generator(u32 bits) { return bb::two_adic_generator(bits); }
    static GB_HD T ext_w() { return W_MONT; }
    static GB_HD E ezero() { return E{{0, 0, 0, 0}}; }
    static GB_HD E efrom(T x) { return E{{x, 0, 0, 0}}; }
    static GB_HD E eadd(E a, E b) { return E{{bb::add(a.c[0], b.c[0]), bb::add(a.c[1], b.c[1]), bb::add(a.c[2], b.c[2]), bb::add(a.c[3], b.c[3])}}; }
    static GB_HD E esub(E a, E b) { return E{{bb::sub(a.c[0], b.c[0]), bb::sub(a.c[1], b.c[1]), bb::sub(a.c[2], b.c[2]), bb::sub(a.c[3], b.c[3])}}; }
    // BinomialExtensionField<BabyBear, 4>: c_k = sum_{i+j=k} a_i b_j + W sum_{i+j=k+4} a_i b_j
    static GB_HD E emul(E a, E b) {
        using bb::add; using bb::mul;
        u32 lo0 = mul(a.c[0], b.c[0]);
        u32 lo1 = add(mul(a.c[0], b.c[1]), mul(a.c[1], b.c[0]));
        u32 lo2 = add(add(mul(a.c[0], b.c[2]), mul(a.c[1], b.c[1])), mul(a.c[2], b.c[0]));
        u32 lo3 = add(add(mul(a.c[0], b.c[3]), mul(a.c[1], b.c[2])), add(mul(a.c[2], b.c[1]), mul(a.c[3], b.c[0])));
        u32 hi0 = add(add(mul(a.c[1], b.c[3]), mul(a.c[2], b.c[2])), mul(a.c[3], b.c[1]));
        u32 hi1 = add(mul(a.c[2], b.c[3]), mul(a.c[3], b.c[2]));
        u32 hi2 = mul(a.c[3], b.c[3]);
        return E{{add(lo0, mul(W_MONT, hi0)), add(lo1, mul(W_MONT, hi1)), add(lo2, mul(W_MONT, hi2)), lo3}};
    }
    static GB_HD E escale(E a, T s) { return E{{bb::mul(a.c[0], s), bb::mul(a.c[1], s), bb::mul(a.c[2], s), bb::mul(a.c[3], s)}}; }
    // a = A + x B with A = a0 + a2 y, B = a1 + a3 y in F[y]/(y^2 - W), y = x^2:
    // 1/a = (A - x B) / (A^2 - y B^2); the denominator c0 + c1 y is inverted through its norm c0^2 - W c1^2.
    static __host__ __device__ inline E einv(E a) {
        using bb::add; using bb::sub; using bb::mul;
        const u32 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
        // A^2 = (a0^2 + W a2^2) + (2 a0 a2) y ; B^2 = (a1^2 + W a3^2) + (2 a1 a3) y ; y B^2 = W (2 a1 a3) + (a1^2 + W a3^2) y
        u32 b2_0 = add(mul(a1, a1), mul(W_MONT, mul(a3, a3))), b2_1 = mul(add(a1, a1), a3);
        u32 c0 = sub(add(mul(a0, a0), mul(W_MONT, mul(a2, a2))), mul(W_MONT, b2_1));
        u32 c1 = sub(mul(add(a0, a0), a2), b2_0);
        u32 ninv = bb::inv(sub(mul(c0, c0), mul(W_MONT, mul(c1, c1))));
        u32 d0 = mul(c0, ninv), d1 = mul(bb::neg(c1), ninv);  // 1 / (c0 + c1 y) = d0 + d1 y
        // (A - x B)(d0 + d1 y): A d = (a0 d0 + W a2 d1) + (a0 d1 + a2 d0) y ; B d likewise
        u32 ad0 = add(mul(a0, d0), mul(W_MONT, mul(a2, d1))), ad1 = add(mul(a0, d1), mul(a2, d0));
        u32 bd0 = add(mul(a1, d0), mul(W_MONT, mul(a3, d1))), bd1 = add(mul(a1, d1), mul(a3, d0));
        return E{{ad0, bb::neg(bd0), ad1, bb::neg(bd1)}};
    }
    static GB_HD T coord(const E& e, u32 k) { return e.c[k]; }
    static GB_HD void set_coord(E& e, u32 k, T v) { e.c[k] = v; }
};

template <class F>
__host__ __device__ inline typename F::E epow(typename F::E b, u64 e) {
    typename F::E r = F::efrom(F::one());
    while (e) {
        if (e & 1) r = F::emul(r, b);
        b = F::emul(b, b);
        e >>= 1;
    }
    return r;
}
template <class F>
__host__ __device__ inline bool eis_one(const typename F::E& e) {
    bool ok = F::coord(e, 0) == F::one();
    for (u32 k = 1; k < F::D; k++) ok = ok && F::coord(e, k) == F::zero();
    return ok;
}

}  // namespace gbk
