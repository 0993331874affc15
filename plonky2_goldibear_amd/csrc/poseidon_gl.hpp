// Poseidon width-12 permutation over Goldilocks, one permutation per lane (gfx950).
//
// Same function as the reference's PoseidonGoldilocks::poseidon (hash/poseidon_goldilocks.rs:912-922):
// 4 full rounds, 22 partial rounds in the "fast" (v, w_hat, M_init) form (:899-909, :718-744),
// 4 full rounds; x^7 s-box (:840-846); MDS = circulant(MDS_CIRC) + diag(MDS_DIAG) (:301-302,:547-557).
//
// gfx950 cost model (tools/microbench_valu.hip): every 64-bit-result VALU op - v_mad_u64_u32,
// v_lshl_add_u64, v_mul_* - issues at ~4.2 cycles per wave, plain 32-bit VOP2 ops at ~2.6, so the
// work is counted in instructions, and a v_mad_u64_u32 (32x32+64) is the densest one.  Hence:
//  * LAZY residues: a state word is any u64 congruent to the value; canonical form only at the end
//    (tools/microbench_mulmod.hip: 70 vs 108 cycles per wave-mul);
//  * dot products (w_hat row, M_init columns) are accumulated UNREDUCED as six carry-free 64-bit
//    sums of 32-bit x 22-bit partial products (12 terms < 2^58) and reduced once;
//  * the MDS layer is 24 mads per output on 32-bit halves, one fold per output.
// Round loops stay rolled (uniform index -> scalar loads of the constants).
#pragma once
#include "gl_field.hpp"
#include "poseidon_constants.h"
#include "poseidon_gl_lab.hpp"   // GB_PROBE_AT / GB_LAB_*: empty in the product

namespace poseidon_gl {

using gl::u32;
using gl::u64;
typedef unsigned __int128 u128;

static constexpr int WIDTH = 12, RATE = 8, HOUT = 4, N_PARTIAL = 22, HALF_FULL = 4;
static constexpr u64 EPS = gl::EPS;

// ------------------------------------------------------------------ constants
struct Limb3 {
    u32 l[3];  // b = l0 + l1 * 2^22 + l2 * 2^44
};
template <int N>
struct LimbTable {
    Limb3 v[N];
};
template <int N>
constexpr LimbTable<N> split22(const u64 (&src)[N]) {
    LimbTable<N> t{};
    for (int i = 0; i < N; i++) {
        t.v[i].l[0] = (u32)(src[i] & 0x3FFFFF);
        t.v[i].l[1] = (u32)((src[i] >> 22) & 0x3FFFFF);
        t.v[i].l[2] = (u32)(src[i] >> 44);
    }
    return t;
}
namespace raw {
constexpr u64 WHATS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST};
constexpr u64 INIT[11 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST};
}  // namespace raw

// The state lives in MONTGOMERY form (x R, R = 2^64 mod p) inside the permutation: a state-by-state product (the s-box) is then
// reduced by the Montgomery fold (8 carry ops, mont_fold) instead of the plain one (11, gl::fold128), while everything linear -
// small-integer MDS sums, products with the PLAIN constants w_hat / v / M_init (x R times c is (x c) R under the plain fold) -
// is unchanged.  Only the constants that are ADDED to the state are stored times R.
template <int N>
struct U64Table {
    u64 v[N];
};
namespace raw {
constexpr u64 cmulmod_r(u64 a) { return (u64)(((u128)(a % gl::P) * gl::EPS) % gl::P); }  // a R mod p (R = 2^64 mod p = 2^32 - 1)
template <int N>
constexpr U64Table<N> times_r(const u64 (&src)[N]) {
    U64Table<N> t{};
    for (int i = 0; i < N; i++) t.v[i] = cmulmod_r(src[i]);
    return t;
}
constexpr u64 ALL_RC[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN] = {GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST};
constexpr u64 FIRST_RC[12] = {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST};
constexpr u64 PARTIAL_RC[22] = {GL_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS_LIST};
}  // namespace raw
__device__ static const U64Table<GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN> RC_T = raw::times_r(raw::ALL_RC);
// A permutation that starts from a ZERO capacity (two_to_one: hash/hashing.rs:76-96; the first absorption of a sponge, :100-123):
// words 8..11 after the first constant layer ARE the round constants, so their first s-boxes are constants too -
// (rc^7) R, Montgomery form like the state - and 4 of the permutation's 118 s-boxes are not computed (round 5).
namespace raw {
constexpr u64 cpow7(u64 c) {
    const u128 p = gl::P;
    const u64 c2 = (u64)((u128)c * c % p), c4 = (u64)((u128)c2 * c2 % p), c3 = (u64)((u128)c2 * c % p);
    return (u64)((u128)c3 * c4 % p);
}
constexpr U64Table<4> zero_capacity_sboxes() {
    U64Table<4> t{};
    for (int i = 0; i < 4; i++) t.v[i] = cmulmod_r(cpow7(ALL_RC[8 + i] % gl::P));
    return t;
}
}  // namespace raw
__device__ static const U64Table<4> ZERO_CAP_SBOX_T = raw::zero_capacity_sboxes();
__device__ static const U64Table<12> FP_FIRST_T = raw::times_r(raw::FIRST_RC);
__device__ static const U64Table<22> FP_RC_T = raw::times_r(raw::PARTIAL_RC);
#define GB_RC (poseidon_gl::RC_T.v)
#define GB_FP_FIRST (poseidon_gl::FP_FIRST_T.v)
#define GB_FP_RC (poseidon_gl::FP_RC_T.v)
__device__ static const u64 FP_VS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST};
// Two partial rounds at a time (see partial_rounds): the second round's w_hat row meets the first round's pending update
// u_k v_k of the state, sum_i v_{k,i} w_hat_{k+1,i} = PAIR_C[k/2], a per-pair constant (compile-time, mod p).
namespace raw {
constexpr u64 VS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST};
constexpr u64 cmulmod(u64 a, u64 b) { return (u64)(((u128)(a % gl::P) * (b % gl::P)) % gl::P); }
struct PairTable {
    u64 v[11];
};
constexpr PairTable pair_constants() {
    PairTable t{};
    for (int kk = 0; kk < 11; kk++) {
        u64 acc = 0;
        for (int i = 0; i < 11; i++) acc = (u64)(((u128)acc + cmulmod(VS[(2 * kk) * 11 + i], WHATS[(2 * kk + 1) * 11 + i])) % gl::P);
        t.v[kk] = acc;
    }
    return t;
}
constexpr PairTable PAIR = pair_constants();
}  // namespace raw
// The last full round of the first half, its MDS, the first partial constant layer and mds_partial_layer_init are one
// linear map of the s-box outputs y: s_c = sum_j y_j MI[j][c-1] + MI_K[c-1] (c >= 1), MI = MDS rows 1..11 times M_init.
namespace raw {
constexpr u64 CIRC[12] = {GL_POSEIDON_MDS_CIRC_LIST};
constexpr u64 FIRST[12] = {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST};
struct MergedInit {
    u64 m[12 * 11];  // [j][c]
    u64 k[11];
};
constexpr MergedInit merged_init() {
    MergedInit t{};
    for (int c = 0; c < 11; c++) {
        for (int j = 0; j < 12; j++) {
            u64 acc = 0;
            for (int r = 1; r < 12; r++)  // MDS[r][j] = CIRC[(j - r) mod 12] for r >= 1 (DIAG is non-zero at r = 0 only)
                acc = (u64)(((u128)acc + cmulmod(CIRC[(j - r + 12) % 12], INIT[(r - 1) * 11 + c])) % gl::P);
            t.m[j * 11 + c] = acc;
        }
        u64 acc = 0;
        for (int r = 1; r < 12; r++) acc = (u64)(((u128)acc + cmulmod(FIRST[r], INIT[(r - 1) * 11 + c])) % gl::P);
        t.k[c] = (u64)(((u128)acc * gl::EPS) % gl::P);   // added to the state: times R (Montgomery form)
    }
    return t;
}
constexpr MergedInit MERGED = merged_init();
}  // namespace raw
__device__ static const LimbTable<12 * 11> MI_L = split22(raw::MERGED.m);
__device__ static const u64 MI_K[11] = {raw::MERGED.k[0], raw::MERGED.k[1], raw::MERGED.k[2], raw::MERGED.k[3], raw::MERGED.k[4], raw::MERGED.k[5],
                                        raw::MERGED.k[6], raw::MERGED.k[7], raw::MERGED.k[8], raw::MERGED.k[9], raw::MERGED.k[10]};
__device__ static const LimbTable<22 * 11> WHATS_L = split22(raw::WHATS);
__device__ static const LimbTable<11> PAIR_L = split22(raw::PAIR.v);
__device__ static const LimbTable<11 * 11> INIT_L = split22(raw::INIT);

// MDS_MATRIX_CIRC / MDS_MATRIX_DIAG (hash/poseidon_goldilocks.rs:301-302) as immediates
__device__ __forceinline__ constexpr u32 mds_circ(int i) {
    constexpr u32 c[12] = {GL_POSEIDON_MDS_CIRC_LIST};
    return c[i];
}
static constexpr u32 MDS_DIAG0 = 8;
namespace raw {
constexpr u64 DIAG[12] = {GL_POSEIDON_MDS_DIAG_LIST};
static_assert(DIAG[0] == MDS_DIAG0 && DIAG[1] == 0 && DIAG[2] == 0 && DIAG[3] == 0 && DIAG[4] == 0 && DIAG[5] == 0 && DIAG[6] == 0 &&
                  DIAG[7] == 0 && DIAG[8] == 0 && DIAG[9] == 0 && DIAG[10] == 0 && DIAG[11] == 0,
              "MDS_MATRIX_DIAG must be [8, 0, ...]");
}  // namespace raw

// ------------------------------------------------------------------ lazy arithmetic (any u64 in, any u64 out)

// (lo + 2^64 hi) mod p, not canonical: gl::fold128 on the four limbs.
__device__ __forceinline__ u64 reduce128_lazy(u64 lo, u64 hi) {
    return gl::fold128((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
}
__device__ __forceinline__ u64 reduce128_lazy(u128 x) { return reduce128_lazy((u64)x, (u64)(x >> 64)); }

__device__ __forceinline__ u64 mul_lazy(u64 a, u64 b) {
    u32 r0, r1, hl, hh;
    gl::mul_limbs(a, b, r0, r1, hl, hh);
    return gl::fold128(r0, r1, hl, hh);
}
using gl::mont_fold;   // the Montgomery reduction (gl_field.hpp): 8 carry ops against fold128's 11
// a b / R for two Montgomery-form residues: any u64 in, any u64 out
__device__ __forceinline__ u64 mul_mont(u64 a, u64 b) {
    u32 r0, r1, hl, hh;
    gl::mul_limbs(a, b, r0, r1, hl, hh);
    return mont_fold(r0, r1, hl, hh);
}
// x -> x R = x (2^32 - 1) = (x0 << 32) - (x0 + x1) for x = x0 + 2^32 x1 (2^64 = 2^32 - 1 mod p), any u64 in, any u64 out
__device__ __forceinline__ u64 to_mont(u64 x) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    u32 tc, b0, B, k;
    const u32 t0 = __builtin_addc(x0, x1, 0u, &tc);
    const u32 r0 = __builtin_subc(0u, t0, 0u, &b0);
    const u32 r1 = __builtin_subc(x0, tc, b0, &B);
    const u32 m = 0u - B;                            // negative (x0 = 0): + p
    const u32 f0 = __builtin_subc(r0, m, 0u, &k);
    const u32 f1 = r1 - k;
    return (u64)f0 | ((u64)f1 << 32);
}
// x R -> x, canonical
__device__ __forceinline__ u64 from_mont(u64 x) { return gl::canon(mont_fold((u32)x, (u32)(x >> 32), 0u, 0u)); }

// a * b + c
__device__ __forceinline__ u64 mul_add_lazy(u64 a, u64 b, u64 c) {
    u32 r0, r1, hl, hh, k0, k1, k2;
    gl::mul_limbs(a, b, r0, r1, hl, hh);
    r0 = __builtin_addc(r0, (u32)c, 0u, &k0);
    r1 = __builtin_addc(r1, (u32)(c >> 32), k0, &k1);
    hl = __builtin_addc(hl, 0u, k1, &k2);
    hh += k2;  // a b + c < 2^128: cannot overflow
    return gl::fold128(r0, r1, hl, hh);
}
// x lazy, rc canonical (< p): one carry fix is enough
__device__ __forceinline__ u64 add_rc(u64 x, u64 rc) {
    u64 s;
    bool cy = __builtin_uaddll_overflow(x, rc, &s);
    s += cy ? EPS : 0;
    return s;
}
__device__ __forceinline__ u64 to_canonical(u64 x) { return gl::canon(x); }

__device__ __forceinline__ u64 sbox(u64 x) {  // Montgomery form in and out
    u64 x2 = mul_mont(x, x), x4 = mul_mont(x2, x2), x3 = mul_mont(x, x2);
    return mul_mont(x3, x4);
}

// Unreduced dot product sum_i a_i * b_i for up to 16 terms: six carry-free 64-bit sums.
struct Dot {
    u64 t0[3], t1[3];
    __device__ __forceinline__ Dot() : t0{0, 0, 0}, t1{0, 0, 0} {}
    __device__ __forceinline__ void acc(u64 a, const Limb3& b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            t0[k] += (u64)a0 * b.l[k];
            t1[k] += (u64)a1 * b.l[k];
        }
    }
    __device__ __forceinline__ void acc_small(u64 a, u32 c) {  // c < 2^22
        t0[0] += (u64)(u32)a * c;
        t1[0] += (u64)(u32)(a >> 32) * c;
    }
    // value = sum_k (t0[k] + 2^32 t1[k]) 2^(22 k), every t < 2^58: assembled as five 32-bit limbs (each term is three limbs,
    // shifted by 0 / 22 / 44 bits) and folded once (gl::fold160) - 18 instructions fewer than two 128-bit folds
    __device__ __forceinline__ u64 finish() const {
        u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            u32 c, c2;
            const u32 a0 = (u32)t0[k];
            const u32 a1 = __builtin_addc((u32)(t0[k] >> 32), (u32)t1[k], 0u, &c);
            const u32 a2 = (u32)(t1[k] >> 32) + c;  // < 2^27
            u32 b0, b1, b2, b3;
            if (k == 0) {
                b0 = a0; b1 = a1; b2 = a2; b3 = 0;
            } else if (k == 1) {
                b0 = a0 << 22; b1 = (a1 << 22) | (a0 >> 10); b2 = (a2 << 22) | (a1 >> 10); b3 = a2 >> 10;
            } else {  // 44 = 32 + 12
                b0 = 0; b1 = a0 << 12; b2 = (a1 << 12) | (a0 >> 20); b3 = (a2 << 12) | (a1 >> 20);
                r4 += a2 >> 20;
            }
            r0 = __builtin_addc(r0, b0, 0u, &c);
            r1 = __builtin_addc(r1, b1, c, &c2);
            r2 = __builtin_addc(r2, b2, c2, &c);
            r3 = __builtin_addc(r3, b3, c, &c2);
            r4 += c2;
        }
        return gl::fold160(r0, r1, r2, r3, r4);
    }
};

// ------------------------------------------------------------------ layers

// res[r] = sum_i s[(i+r)%12] * CIRC[i] + s[r]*DIAG[r]; entries < 2^6, so the two 32-bit halves
// accumulate in 64 bits (< 2^41) and fold once: value = sl + 2^32 sh  (the decomposition of the
// reference's mds_layer :497-528, without its FFT form).
// `rc`: the constants of the layer that FOLLOWS (next round's constant layer), added into the unreduced sums as two
// 32-bit halves - two 64-bit adds per word instead of a separate add-with-carry-fix after the fold.
// sl + 2^32 sh for two unreduced sums of 32-bit halves (sl, sh < 2^63, sh >> 32 < 2^31) as a lazy u64 residue: 2^64 = EPS moves
// sh's high word onto sl (one mad), sh's low word goes onto the high word, and the one possible carry is worth EPS again.
__device__ __forceinline__ u64 fold_halves(u64 sl, u64 sh) {
    const u64 t = sl + (u64)(u32)(sh >> 32) * EPS;
    u32 cy, k;
    u32 r1 = __builtin_addc((u32)(t >> 32), (u32)sh, 0u, &cy);
    const u32 m = 0u - cy;
    const u32 r0 = __builtin_addc((u32)t, m, 0u, &k);
    r1 += k;
    return (u64)r0 | ((u64)r1 << 32);
}

__device__ __forceinline__ void mds_layer(u64 (&s)[12], const u64* __restrict__ rc) {
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const u64 c = rc[r];
        u64 sl = (u32)c, sh = c >> 32;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            sl += (u64)lo[(i + r) % 12] * mds_circ(i);
            sh += (u64)hi[(i + r) % 12] * mds_circ(i);
        }
        if (r == 0) {
            sl += (u64)lo[0] * MDS_DIAG0;
            sh += (u64)hi[0] * MDS_DIAG0;
        }
        // sl + 2^32 (sh_lo + 2^32 sh_hi) = sl + sh_hi * EPS + (sh_lo << 32), sh_hi < 2^10
        u64 t = sl + (sh >> 32) * EPS;  // < 2^44
        u64 r2;
        bool cy = __builtin_uaddll_overflow(t, sh << 32, &r2);
        r2 += cy ? EPS : 0;
        s[r] = r2;
    }
}

__device__ static const u64 ZERO_RC[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// ------------------------------------------------------------------ the MDS layer on the matrix pipe
// The hash kernels saturate VALU issue while the MFMA pipe idles, and the MDS matrix is CONSTANT with entries < 2^6.  Cut into
// byte planes, s_i = sum_p b_{i,p} 2^(8p), the layer is eight 12 x 12 integer matrix-vector products per state, one per plane:
// C_p[q] = sum_i M[q][i] b_{i,p} < 2^17, and  out[q] = sum_p 2^(8p) C_p[q].  One v_mfma_i32_32x32x32_i8 does a plane for the 64
// states of a wave (one state per lane) with NO cross-lane traffic, because the constant operand A is block-diagonal:
//   B (data):   lane l = (r = l & 31, h = l >> 5) supplies k = 16 h + j, j < 16, of column r: byte p of word j of ITS OWN state
//   A (matrix): row m = (q & 3) + 8 (q >> 2) + 4 h' holds M[q][.] in the k-block h' only (q < 12 the output word, h' in {0, 1})
//   D:          lane l, register q = D[(q & 3) + 8 (q >> 2) + 4 h][r] = sum_i M[q][i] byte_p(word i of state l)
// (C/D map of the 32 x 32 shapes, /opt/skills/guides/cdna_hip_programming.md section 3; A and B use the same lane -> k rule, so
// only "same h, same j meet" is relied on.)  The i8 operands are signed: bytes go in as b ^ 0x80 = b - 128, which shifts every
// plane sum by the per-row constant 128 rowsum(q); that shift, the biases that keep the signed accumulators non-negative and the
// round constants of the layer that follows are ONE constant per (layer, word), folded into the accumulators' start values
// (MFMA_INIT).  Per layer: 24 v_xor + 48 v_perm (six 4 x 4 byte transposes) to build the planes, 8 MFMAs, then per word four
// v_lshl_add_u32 (planes pairwise, 2^8 apart), four v_mad_i64_i32 (2^16 apart, low and high 32-bit halves) and the 5-instruction
// fold_halves: ~250 VALU instructions where the VALU form takes ~480 (24 mads per word plus folds and moves).
// tools/microbench_mds_mfma.hip: bit-identical on 12.6 M words, layer alone 2.6x faster, a full round (12 s-boxes + layer) 1.36x.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
static constexpr long long MFMA_BIAS = 1ll << 41;   // > |sum_p<4 2^(8p) C'_p| (< 2^39.2): the accumulators stay positive
static constexpr int MFMA_NO_RC = 30;               // index of "no constants follow" in MFMA_INIT
namespace raw {
constexpr u64 caddmod(u64 a, u64 b) { return (u64)(((u128)a + b) % gl::P); }
constexpr u64 csubmod(u64 a, u64 b) { return (u64)(((u128)a + gl::P - b % gl::P) % gl::P); }
constexpr u32 mds_entry(int q, int i) { return (u32)CIRC[((i - q) % 12 + 12) % 12] + (q == 0 && i == 0 ? MDS_DIAG0 : 0u); }
struct MfmaInitTable {
    u64 lo[31 * 12], hi[31 * 12];   // [next round | MFMA_NO_RC][word]: start values of the low / high half accumulators
};
constexpr MfmaInitTable mfma_init_table() {
    MfmaInitTable t{};
    for (int q = 0; q < 12; q++) {
        u64 rowsum = 0;
        for (int i = 0; i < 12; i++) rowsum += mds_entry(q, i);
        // true value = acc_lo + 2^32 acc_hi + 128 rowsum 0x0101..01 - BIAS - 2^32 BIAS: what has to be added, mod p
        u64 k = cmulmod(128 * rowsum, 0x0101010101010101ULL % gl::P);
        k = csubmod(k, (u64)MFMA_BIAS);
        k = csubmod(k, cmulmod((u64)MFMA_BIAS, 1ULL << 32));
        for (int rn = 0; rn <= 30; rn++) {
            const u64 c = rn < 30 ? caddmod(k, cmulmod_r(ALL_RC[12 * rn + q])) : k;   // + the next round's constant (times R)
            t.lo[12 * rn + q] = (u64)MFMA_BIAS + (c & 0xFFFFFFFFULL);
            t.hi[12 * rn + q] = (u64)MFMA_BIAS + (c >> 32);
        }
    }
    return t;
}
}  // namespace raw
__device__ static const raw::MfmaInitTable MFMA_INIT = raw::mfma_init_table();

// The compiler pads MFMA -> VALU hazards with s_nop (gfx950 has no interlock: a read of a tile register is stale for 11 wait
// states after the MFMA, a write to one is overwritten by the pipe's write-back for 8; tools/microbench_mfma_hazard.hip), but its
// hazard recognizer does not look inside INLINE ASM: the recombination below is therefore written in C (round 3's asm form parked
// results in tile registers the kernel never reads and hashed wrongly, HISTORY.md round 4); tests/test_mfma_guard.py checks the
// assembly of every kernel for the pattern (tools/mfma_guard.py).
// a * b + c, signed 32 x 32 + 64, as ONE v_mad_i64_i32.  Written in C so that the hazard recognizer sees the instruction; the
// multipliers 1 and 2^16 are OPAQUE register constants (MdsOperand::one / ::k16: a volatile asm v_mov at the top of the kernel,
// before any MFMA exists, that the optimizer cannot look into), or the compiler turns the products into sign-extend + shift + add.
// With `one` in a VGPR the start value stays the instruction's one scalar source: v_mad_i64_i32 d, t, v_one, s[c:c+1].
struct MdsOperand {
    v4i a;          // this lane's share of the constant A operand (mds_mfma_matrix)
    int one, k16;   // 1 and 65536
    GB_LAB_PROBE_FIELDS
};
__device__ __forceinline__ long long mad_i64(int a, const MdsOperand& m, long long c) { return (long long)a * (long long)m.k16 + c; }
// a + c with c a UNIFORM 64-bit value: the accumulator's start value is the addend of a multiplication by the opaque 1
__device__ __forceinline__ long long mad_i64_start(int a, const MdsOperand& m, u64 c_uniform) {
    return (long long)a * (long long)m.one + (long long)c_uniform;
}
// 4 x 4 byte transpose: t[p] = [w0.b_p, w1.b_p, w2.b_p, w3.b_p] (8 v_perm_b32)
__device__ __forceinline__ void byte_transpose4(u32 w0, u32 w1, u32 w2, u32 w3, u32 (&t)[4]) {
    const u32 a_lo = __builtin_amdgcn_perm(w1, w0, 0x05010400u);  // [w0.b0, w1.b0, w0.b1, w1.b1]
    const u32 a_hi = __builtin_amdgcn_perm(w1, w0, 0x07030602u);  // [w0.b2, w1.b2, w0.b3, w1.b3]
    const u32 b_lo = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
    const u32 b_hi = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    t[0] = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);
    t[1] = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
    t[2] = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
    t[3] = __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
}
// This lane's share of the constant A operand (see above); call with all 64 lanes of the wave active, blockDim.x a multiple of 64.
__device__ __forceinline__ MdsOperand mds_mfma_matrix() {
    const u32 l = threadIdx.x & 63, r = l & 31, h = l >> 5;
    const u32 hp = (r >> 2) & 1, q = (r & 3) + 4 * (r >> 3);
    v4i a = {0, 0, 0, 0};
    if (h == hp && q < 12) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            u32 v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const u32 i = 4 * g + e;
                u32 m = mds_circ((int)((i + 12 - q) % 12));
                if (q == 0 && i == 0) m += MDS_DIAG0;
                v |= m << (8 * e);
            }
            a[g] = (int)v;
        }
    }
    MdsOperand m;
    m.a = a;
    asm volatile("v_mov_b32 %0, 1\n\tv_mov_b32 %1, 0x10000" : "=v"(m.one), "=v"(m.k16));
    return m;
}
// s <- MDS s + constants of round `rnext` (MFMA_NO_RC: none); every lane of the wave must execute this (MFMA), whatever its data.
// Q0: the first word that is produced (words below it are left undefined).  An overwrite-mode sponge discards words 0..7 of a
// permutation's output whenever a full absorption follows (hash/hashing.rs:100-123), so the LAST layer of such a permutation only
// has to recombine and fold the capacity words: Q0 = 8 saves 80 of the layer's 192 instructions.
template <int Q0 = 0>
__device__ __forceinline__ void mds_layer_mfma(u64 (&s)[12], const MdsOperand& amat, int rnext) {
    u32 w[24];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        w[i] = (u32)s[i] ^ 0x80808080u;
        w[12 + i] = (u32)(s[i] >> 32) ^ 0x80808080u;
    }
    u32 pl[8][3];   // pl[p][g]: byte p of words 4g .. 4g + 3
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int g = 0; g < 3; g++) {
            u32 t[4];
            byte_transpose4(w[12 * half + 4 * g], w[12 * half + 4 * g + 1], w[12 * half + 4 * g + 2], w[12 * half + 4 * g + 3], t);
#pragma unroll
            for (int p = 0; p < 4; p++) pl[4 * half + p][g] = t[p];
        }
    GB_PROBE_AT(amat, 20, pl);   // layer: byte planes cut
    long long lo[12], hi[12];
    const u64* ilo = MFMA_INIT.lo + 12 * rnext;   // uniform index: scalar loads
    const u64* ihi = MFMA_INIT.hi + 12 * rnext;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int pad = amat.a[3];   // the fourth dword of the B operands (k = 12..15, which meet zeros in A): A's own fourth dword, zero in every lane
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {   // planes two at a time: d_p + 2^8 d_(p+1) fits 32 bits (|.| < 2^25)
        v4i b0, b1;
        b0[0] = (int)pl[2 * pp][0]; b0[1] = (int)pl[2 * pp][1]; b0[2] = (int)pl[2 * pp][2]; b0[3] = pad;
        b1[0] = (int)pl[2 * pp + 1][0]; b1[1] = (int)pl[2 * pp + 1][1]; b1[2] = (int)pl[2 * pp + 1][2]; b1[3] = pad;
        const v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(amat.a, b0, zero, 0, 0, 0);
        const v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(amat.a, b1, zero, 0, 0, 0);
#pragma unroll
        for (int q = Q0; q < 12; q++) {
            const int t = (int)(((u32)d1[q] << 8) + (u32)d0[q]);
            if (pp == 0) lo[q] = mad_i64_start(t, amat, ilo[q]);
            else if (pp == 1) lo[q] = mad_i64(t, amat, lo[q]);
            else if (pp == 2) hi[q] = mad_i64_start(t, amat, ihi[q]);
            else hi[q] = mad_i64(t, amat, hi[q]);
        }
    }
    if constexpr (Q0 == 0) GB_PROBE_AT(amat, 21, lo, hi);   // layer: 8 MFMAs + recombination
    // fold_halves with its carry fix on a rare path: value = lo + 2^32 hi = (lo + (hi >> 32) EPS) + 2^32 (u32)hi, and the last
    // addition wraps only when (u32)hi lies within 2^12 of 2^32 - about 4e-4 of the wave-layers have such a lane.  The fast path
    // is one mad and one add per word; the wave-wide OR of the carries is scalar work, and a wave in which any lane wrapped
    // takes the branch and adds EPS (= 2^64 mod p) in those lanes (no second wrap: a wrapped high word is < 2^12).
    u64 any_carry = 0;
#pragma unroll
    for (int q = Q0; q < 12; q++) {
        const u64 sl = (u64)lo[q], sh = (u64)hi[q];
        const u64 t = sl + (u64)(u32)(sh >> 32) * EPS;
        u32 r1;
        u64 carry_lanes;   // the add's carry-out as it comes: a lane mask in a scalar register pair
        asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r1), "=s"(carry_lanes) : "v"((u32)(t >> 32)), "v"((u32)sh));
        any_carry |= carry_lanes;
        s[q] = (u64)(u32)t | ((u64)r1 << 32);
    }
    if (__builtin_expect(any_carry != 0, 0)) {
#pragma unroll
        for (int q = Q0; q < 12; q++) {
            const bool wrapped = (u32)(s[q] >> 32) < (u32)hi[q];   // r1 = t_hi + (u32)hi wrapped  <=>  r1 < (u32)hi
            s[q] += wrapped ? EPS : 0;
        }
    }
}

}  // namespace poseidon_gl
