// Poseidon2 width-16 permutation over BabyBear, one permutation per lane, state in Montgomery form.
//
// Same function as the reference's Permuter31 for BabyBear (hash/poseidon2_babybear.rs:150-159), in the
// order the reference itself restates it in gates/poseidon2_babybear.rs:609-672: initial M_E; 4 x (add
// EXTERNAL_CONSTANTS[r], x^7, M_E); 13 x (s0 += INTERNAL_CONSTANTS[r], s0^7, M_I); 4 x (rc 4..7, x^7, M_E).
// M_E: apply_mat4 per 4-lane group + column-class sums (:804-832, :903-917).
// M_I: s *= 2^-32; part = sum_{i>=1} s_i; s0 <- part - s0; s_{i+1} <- (part + s0) + s_{i+1} 2^shift_i (:787-802).
// In Montgomery form "multiply by 2^-32" is one Montgomery reduction, and "(s 2^-32) 2^k" is one
// reduction of the shifted word - no general multiplication in the internal layer.
#pragma once
#include "bb_field.hpp"
#include "poseidon_constants.h"

namespace poseidon2_bb {

using bb::u32;
using bb::u64;

static constexpr int WIDTH = 16, RATE = 8, HOUT = 8;

template <int N>
struct MontTable {
    u32 v[N];
};
template <int N>
constexpr MontTable<N> to_mont_table(const u32 (&src)[N]) {
    MontTable<N> t{};
    for (int i = 0; i < N; i++) t.v[i] = (u32)((((u64)src[i]) << 32) % bb::P);
    return t;
}
namespace raw {
constexpr u32 EXT[128] = {BB_POSEIDON2_EXTERNAL_CONSTANTS_LIST};
constexpr u32 INT[13] = {BB_POSEIDON2_INTERNAL_CONSTANTS_LIST};
}  // namespace raw
// ------------------------------------------------------------------ scale tracking
// The external linear layer is computed on UNREDUCED 64-bit sums (every word of a layer is a sum of at most 35 inputs
// < p, plus the next round's constant) and brought back with ONE Montgomery reduction per word.  That reduction
// multiplies by R^-1, so a word holds kappa * R * x for a scale kappa that is the same for all 16 words and is known at
// compile time: the layer turns kappa into kappa / R, the s-box into kappa^7.  Round constants are pre-multiplied by
// the scale in force, the single s-box of an internal round is followed by one multiplication by kappa^-6 to bring
// word 0 back to the common scale, and the last step of the permutation multiplies by 1 / kappa_final.
// Canonical add = 3 instructions, so the 68 additions + 16 constant additions of a layer (252) become 76 one-instruction
// 64-bit adds / mads + 16 reductions of 5 (156).
namespace plan {
constexpr u32 cmul(u32 a, u32 b) { return (u32)((u64)a * b % bb::P); }
constexpr u32 cpow(u32 b, u64 e) {
    u32 r = 1;
    while (e) {
        if (e & 1) r = cmul(r, b);
        b = cmul(b, b);
        e >>= 1;
    }
    return r;
}
constexpr u32 cinv(u32 a) { return cpow(a, bb::P - 2); }
constexpr u32 R = (u32)(((u64)1 << 32) % bb::P);
constexpr u32 RINV = cinv(R);
struct Plan {
    u32 ext[8][16];  // Montgomery form of kappa * EXTERNAL_CONSTANTS[r], kappa = the scale when round r's constants are added
    u32 in[14];      // Montgomery form of kappa * INTERNAL_CONSTANTS[r]; in[13] = ext4[0], what word 0 meets after the last internal round
    u32 sumc[13];    // internal round j: minus the sum of the offsets the lazy words 1..15 carry at that point (see internal_round)
    u32 ext4[16];    // round 4's constants (ext[4]) minus the offsets left by the thirteenth internal round
    u32 fix6;        // kappa^-6 (plain) during the internal rounds: mont(s-box output, fix6) = (word 0 at the common scale) * 2^-32
    u32 out;         // Montgomery form of 1 / kappa_final: mont(word, out) = Montgomery form at scale 1
    u32 out_canon;   // 1 / kappa_final (plain): mont(word, out_canon) = the canonical value
};
constexpr Plan make_plan() {
    Plan p{};
    u32 k = 1;
    auto fill_ext = [&](int r, u32 kappa) {
        for (int i = 0; i < 16; i++) p.ext[r][i] = cmul(cmul(kappa, raw::EXT[16 * r + i] % bb::P), R);
    };
    // initial layer adds round 0's constants; it is followed by rounds 0..3, each: s-box, layer (+ next constants)
    fill_ext(0, k);
    k = cmul(k, RINV);
    for (int r = 0; r < 4; r++) {
        k = cpow(k, 7);
        if (r < 3) fill_ext(r + 1, k);
        k = cmul(k, RINV);
    }
    for (int r = 0; r < 13; r++) p.in[r] = cmul(cmul(k, raw::INT[r] % bb::P), R);
    p.fix6 = cinv(cpow(k, 6));
    fill_ext(4, k);  // added before round 4's s-box
    {   // offsets of the lazy words through the internal rounds: word i enters round j as (true word) + e[i]; the round stores
        // full + (word 2^sh_i / 2^32) + ceil(p/2) (internal_round), so e[i] <- e[i] 2^sh_i / 2^32 + ceil(p/2), and the sum the
        // round takes over the words is corrected by -sum e[i]
        constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
        u32 e[15] = {};
        for (int r = 0; r < 13; r++) {
            u32 tot = 0;
            for (int i = 0; i < 15; i++) tot = (u32)(((u64)tot + e[i]) % bb::P);
            p.sumc[r] = (bb::P - tot) % bb::P;
            for (int i = 0; i < 15; i++) e[i] = (u32)(((u64)cmul(cmul(e[i], (u32)(((u64)1 << SH[i]) % bb::P)), RINV) + (bb::P + 1) / 2) % bb::P);
        }
        p.ext4[0] = p.ext[4][0];
        p.in[13] = p.ext[4][0];
        for (int i = 0; i < 15; i++) p.ext4[i + 1] = (u32)(((u64)p.ext[4][i + 1] + bb::P - e[i]) % bb::P);
    }
    for (int r = 4; r < 8; r++) {
        k = cpow(k, 7);
        if (r < 7) fill_ext(r + 1, k);
        k = cmul(k, RINV);
    }
    p.out = cmul(cinv(k), R);
    p.out_canon = cinv(k);
    return p;
}
constexpr Plan PLAN = make_plan();
}  // namespace plan
__device__ static const plan::Plan PLAN = plan::PLAN;
__device__ static const u32 ZERO16[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// x^7 for a word x with |x| <= 1.03 p as a signed number (canonical, or what bb::reduce_signed leaves); the result is LAZY
// (in (0, 2p)).  The three inner products are SIGNED Montgomery products (bb::mul_signed: no final selection, the word just stays
// within +-0.97 p), the last one adds p to come out non-negative: 12 multiplies + 5 plain operations, where two canonical and two
// lazy unsigned products took 12 + 10.  Every consumer here takes a lazy word: the unreduced sums of the external layer and the
// multiplication by kappa^-6 of the internal rounds.
__device__ __forceinline__ u32 sbox7(u32 x) {
    const int x1 = (int)x;
    const int x2 = bb::mul_signed(x1, x1), x4 = bb::mul_signed(x2, x2), x3 = bb::mul_signed(x1, x2);
    return (u32)bb::mul_signed(x3, x4, bb::P);
}

// M_E (gates/poseidon2_babybear.rs:804-832, 903-917): per 4-block [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], i.e.
// n0 = T + a + 2b, n1 = T + b + 2c, n2 = T + c + 2d, n3 = T + d + 2a with T = a + b + c + d, then every word gets the sum
// of its column class.  All in 64 bits, `c` (the next constants, scaled) added, one reduction per word (scale / R).
// SIGNED: the words go straight into s-boxes, which take signed words within +-1.03 p (bb::reduce_signed: two operations fewer per
// word); otherwise canonical.
template <bool SIGNED = false>
__device__ __forceinline__ void external_layer(u32 (&s)[16], const u32* __restrict__ c) {
    u64 n[16];
#pragma unroll
    for (int b = 0; b < 16; b += 4) {
        const u64 x0 = s[b], x1 = s[b + 1], x2 = s[b + 2], x3 = s[b + 3];
        const u64 t01 = x0 + x1, t23 = x2 + x3, t = t01 + t23;
        const u64 ta = t + x1, tb = t + x3;          // nine additions for the four rows
        n[b] = ta + t01;                             // 2 x0 + 3 x1 + x2 + x3
        n[b + 1] = ta + 2 * x2;                      // x0 + 2 x1 + 3 x2 + x3
        n[b + 2] = tb + t23;                         // x0 + x1 + 2 x2 + 3 x3
        n[b + 3] = tb + 2 * x0;                      // 3 x0 + x1 + x2 + 2 x3
    }
    u64 sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = n[k] + n[4 + k] + n[8 + k] + n[12 + k];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 t = n[i] + sums[i & 3] + c[i];  // < 71 p: 35 lazy words and a constant
        s[i] = SIGNED ? (u32)bb::reduce_signed(t) : bb::reduce(t);
    }
}

// One internal round: s0 <- (s0 + rc)^7, then M_I (gates/poseidon2_babybear.rs:787-802).  `rc` is scaled (PLAN.in);
// the s-box leaves word 0 at scale kappa^7, and y0 = (word 0 at the common scale) * 2^-32 comes out of ONE Montgomery
// multiplication by kappa^-6.
// Words 1..15 are LAZY inside the internal rounds (any word below LAZY_MAX): they only feed the 64-bit sum and a Montgomery
// reduction of the shifted word, which takes any 32-bit word.  That reduction is the signed one (bb::mul_signed's): with
// t = s 2^k and m = lo(t) / P as a signed word, hi(t - m P) lies in (hi(t) - p/2, hi(t) + p/2]; adding `full` + ceil(p/2) first
// makes the new word non-negative, and it all fits THREE multiply-adds per word - m = s (2^k / P) as one v_mul_lo,
// d = s 2^k + 2^32 (full + ceil(p/2)) as one v_mad_u64_u32, d - m P as one v_mad_i64_i32, whose high word is the new s_{i+1} -
// where shift, reduction, selection and addition took seven.  ceil(p/2) is not a multiple of p: every lazy word carries an OFFSET
// that is a compile-time constant per (round, word); `sumc` (PLAN.sumc) takes the offsets out of the sum, and the constants of the
// external round that follows the internal ones take out the last ones (PLAN.ext4).
// a * b + c with a wave-uniform b (a power of two here): one v_mad_u64_u32 where the compiler builds a zero-extended pair, shifts
// it and adds (v_mov + v_lshlrev_b64 + v_lshl_add_u64)
__device__ __forceinline__ u64 mad_wide(u32 a, u32 b_uniform, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, carry_unused;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_unused) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
#else
    return (u64)a * b_uniform + c;
#endif
}
static constexpr u32 HALF_P = (bb::P + 1) / 2;
static constexpr u32 LAZY_MAX = 2 * bb::P + (1u << 15) + 1;   // full < p, + ceil(p/2), + hi(t) < 2^15, + p/2
static_assert((u64)LAZY_MAX < ((u64)1 << 32), "lazy words must fit 32 bits");
// Word 0 comes in as the s-box INPUT of this round (constant added: a signed word within +-p) and leaves as the next round's:
// (part + rc_next) - y0 is one canonical addition off the s-box's dependency chain and one subtraction, where part - y0 and
// + rc_next took two canonical operations.
__device__ __forceinline__ void internal_round(u32 (&s)[16], u32 rc_next, u32 sumc) {
    constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};  // gates/poseidon2_babybear.rs:41-42
    // part = sum_{i>=1} s_i 2^-32: the reduction is linear, so reduce the 36-bit sum once instead of 15 words
    u64 sum = (u64)s[1] + sumc;
#pragma unroll
    for (int i = 2; i < 16; i++) sum = mad_wide(s[i], 1u, sum);
    const u32 part = bb::reduce(sum);
    const u32 y0 = bb::mul(sbox7(s[0]), PLAN.fix6);
    const u32 full = bb::add(part, y0);
    s[0] = bb::add(part, rc_next) - y0;                           // in (-p, p) as a signed word
    const u64 base = (u64)(full + HALF_P) << 32;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        const u32 w = s[i + 1];
        const int m = (int)(w * (bb::PINV << SH[i]));            // lo(w 2^k) / P mod 2^32
        u64 d = SH[i] ? mad_wide(w, 1u << SH[i], base) : (base | w);
        d -= (u64)((long long)m * (int)bb::P);                    // low word: zero
        s[i + 1] = (u32)(d >> 32);
    }
}

// The permutation up to the final common scale kappa_final: finish every word that is used afterwards with
// renorm() (-> Montgomery form at scale 1, e.g. the capacity words of a sponge) or canonical_out() (-> canonical value).
__device__ __forceinline__ void permute_scaled(u32 (&s)[16]) {
    external_layer<true>(s, PLAN.ext[0]);
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i]);
        external_layer<true>(s, PLAN.ext[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox7(s[i]);
    external_layer(s, ZERO16);
    s[0] = bb::add(s[0], PLAN.in[0]);
    for (int r = 0; r < 13; r++) internal_round(s, PLAN.in[r + 1], PLAN.sumc[r]);
    // round 4's constants, for the signed s-box: word 0 has its own already (internal_round), the lazy words come below p + 2^15 with
    // one selection, and adding (constant - p) leaves every word within (-p, p + 2^15)
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const u32 t = s[i] - bb::P;
        s[i] = (t < s[i] ? t : s[i]) + (PLAN.ext4[i] - bb::P);
    }
    for (int r = 4; r < 7; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i]);
        external_layer<true>(s, PLAN.ext[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox7(s[i]);
    external_layer(s, ZERO16);
}
__device__ __forceinline__ u32 renorm(u32 x) { return bb::mul(x, PLAN.out); }
// the same, LAZY (a word in [0, 2p)): enough for words that go straight into the next permutation, whose first layer sums lazy words
__device__ __forceinline__ u32 renorm_lazy(u32 x) { return bb::mul_lazy(x, PLAN.out); }
__device__ __forceinline__ u32 canonical_out(u32 x) { return bb::mul(x, PLAN.out_canon); }

// state: Montgomery form in, Montgomery form out (scale 1 on both sides)
__device__ __forceinline__ void permute(u32 (&s)[16]) {
    permute_scaled(s);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = renorm(s[i]);
}

}  // namespace poseidon2_bb
