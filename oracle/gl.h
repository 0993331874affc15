/* TEST ORACLE (CPU restatement) - Goldilocks field p = 2^64 - 2^32 + 1 and its quadratic extension.
 *
 * This directory is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.  The product (plonky2_goldibear_amd/) never includes or links it.
 *
 * The reference takes its field arithmetic from the un-vendored Plonky3 fork (p3-goldilocks,
 * branch goldilocks_improvements, no pinned rev; /root/reference/Cargo.toml:17-24).  Restated here
 * from the published definition; the constants are pinned by the reference's own fixtures
 * (see tests/test_oracle_fixture.py): generator 7, two-adic generator 1753635133440165772 (order
 * 2^32), extension non-residue W = 7 (x^2 - 7).
 */
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 7ULL
#define GL_TWO_ADIC_GENERATOR_32 1753635133440165772ULL
#define GL_TWO_ADICITY 32
#define GL_EXT_W 7ULL

/* All selections below are written as masks: on random field elements every carry is a coin flip and a
 * mispredicted branch costs more than the arithmetic. */
static inline gl_t gl_canon(uint64_t x) { return x - (-(uint64_t)(x >= GL_P) & GL_P); }

static inline gl_t gl_add(gl_t a, gl_t b) {
    uint64_t s = a + b;
    s += -(uint64_t)(s < a) & GL_EPS; /* wrapped: 2^64 = 2^32 - 1 (mod p) */
    return gl_canon(s);
}
static inline gl_t gl_sub(gl_t a, gl_t b) { return a - b + (-(uint64_t)(a < b) & GL_P); }
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

/* 128-bit -> canonical. Follows the same split the reference uses for its Poseidon
 * (hash/poseidon_goldilocks.rs:254-267): x = lo + 2^64*hi, 2^64 = 2^32-1, 2^96 = -1. */
static inline gl_t gl_reduce128(unsigned __int128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= -(uint64_t)(lo < hi_hi) & GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t t2 = t0 + t1;
    t2 += -(uint64_t)(t2 < t0) & GL_EPS;
    return gl_canon(t2);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((unsigned __int128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }

static inline gl_t gl_pow(gl_t b, uint64_t e) {
    gl_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }

/* F::two_adic_generator(bits): g^(2^(32-bits))  (p3-goldilocks TwoAdicField; call sites
 * field/src/fft.rs:16, field/src/types.rs:14-17) */
static inline gl_t gl_two_adic_generator(unsigned bits) {
    gl_t g = GL_TWO_ADIC_GENERATOR_32;
    for (unsigned i = bits; i < GL_TWO_ADICITY; i++) g = gl_sqr(g);
    return g;
}

/* quadratic extension F[x]/(x^2 - 7) (BinomialExtensionField<Goldilocks,2>; field/src/types.rs:25-29) */
typedef struct { gl_t c[2]; } gl2_t;
static inline gl2_t gl2_from(gl_t a) { gl2_t r = {{a, 0}}; return r; }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { gl2_t r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])}}; return r; }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { gl2_t r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])}}; return r; }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl2_t r;
    r.c[0] = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
    r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return r;
}
static inline gl2_t gl2_scale(gl2_t a, gl_t s) { gl2_t r = {{gl_mul(a.c[0], s), gl_mul(a.c[1], s)}}; return r; }
static inline gl2_t gl2_inv(gl2_t a) {
    /* 1/(a0 + a1 x) = (a0 - a1 x)/(a0^2 - 7 a1^2) */
    gl_t n = gl_sub(gl_sqr(a.c[0]), gl_mul(GL_EXT_W, gl_sqr(a.c[1])));
    gl_t ni = gl_inv(n);
    gl2_t r = {{gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni)}};
    return r;
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r = gl2_from(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_mul(b, b);
        e >>= 1;
    }
    return r;
}
