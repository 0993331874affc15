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
__device__ static const MontTable<128> EXT_RC = to_mont_table(raw::EXT);
__device__ static const MontTable<13> INT_RC = to_mont_table(raw::INT);

__device__ __forceinline__ u32 sbox7(u32 x) {
    u32 x2 = bb::sqr(x), x4 = bb::sqr(x2), x3 = bb::mul(x, x2);
    return bb::mul(x3, x4);
}

__device__ __forceinline__ void apply_mat4(u32& a, u32& b, u32& c, u32& d) {
    u32 t01 = bb::add(a, b), t23 = bb::add(c, d), t0123 = bb::add(t01, t23);
    u32 t01123 = bb::add(t0123, b), t01233 = bb::add(t0123, d);
    u32 n3 = bb::add(t01233, bb::add(a, a));
    u32 n1 = bb::add(t01123, bb::add(c, c));
    u32 n0 = bb::add(t01123, t01);
    u32 n2 = bb::add(t01233, t23);
    a = n0; b = n1; c = n2; d = n3;
}

__device__ __forceinline__ void external_layer(u32 (&s)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i += 4) apply_mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
    u32 sums[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sums[k] = bb::add(bb::add(s[k], s[4 + k]), bb::add(s[8 + k], s[12 + k]));
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = bb::add(s[i], sums[i & 3]);
}

__device__ __forceinline__ void internal_layer(u32 (&s)[16]) {
    constexpr int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};  // gates/poseidon2_babybear.rs:41-42
    // part = sum_{i>=1} s_i 2^-32: the reduction is linear, so reduce the 35-bit sum once instead of 15 words
    const u64 sum = (u64)(s[1] + s[2]) + (u64)(s[3] + s[4]) + (u64)(s[5] + s[6]) + (u64)(s[7] + s[8]) + (u64)(s[9] + s[10]) +
                    (u64)(s[11] + s[12]) + (u64)(s[13] + s[14]) + (u64)s[15];  // words < 2^31: the pair sums fit 32 bits
    const u32 part = bb::reduce(sum);
    const u32 y0 = bb::reduce((u64)s[0]);
    const u32 full = bb::add(part, y0);
    s[0] = bb::sub(part, y0);
#pragma unroll
    for (int i = 0; i < 15; i++) s[i + 1] = bb::add(full, bb::reduce((u64)s[i + 1] << SH[i]));  // (s_{i+1} 2^-32) 2^k
}

// state: Montgomery form in, Montgomery form out
__device__ __forceinline__ void permute(u32 (&s)[16]) {
    external_layer(s);
    for (int r = 0; r < 4; r++) {
        const u32* rc = EXT_RC.v + 16 * r;
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(bb::add(s[i], rc[i]));
        external_layer(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7(bb::add(s[0], INT_RC.v[r]));
        internal_layer(s);
    }
    for (int r = 4; r < 8; r++) {
        const u32* rc = EXT_RC.v + 16 * r;
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(bb::add(s[i], rc[i]));
        external_layer(s);
    }
}

}  // namespace poseidon2_bb
