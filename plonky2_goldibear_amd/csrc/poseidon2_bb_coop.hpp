// Poseidon2 width-16 over BabyBear with ONE STATE PER 16-LANE ROW (lane l holds word l), for trees too small to fill the
// machine with one permutation per lane - the BabyBear counterpart of poseidon_gl_coop.hpp.  Width 16 is exactly a DPP row, so
// no LDS is involved: the 4x4 blocks of the external layer M_E read their quad with quad_perm, its column sums and the
// internal layer's full sum are rotate-and-add all-reduces (row_ror).  Plain canonical Montgomery arithmetic (this path is
// latency-bound, not issue-bound): ~1.4 k dependent instructions per permutation against ~7 k in the lane-per-state form.
// Same function as gates/poseidon2_babybear.rs:609-672 (M_E :804-832, apply_mat4 :903-917, M_I :787-802), bit-exact.
#pragma once
#include "poseidon2_bb.hpp"

namespace poseidon2_bb_coop {

using bb::u32;
using bb::u64;

__device__ static const poseidon2_bb::MontTable<128> EXT_M = poseidon2_bb::to_mont_table(poseidon2_bb::raw::EXT);
__device__ static const poseidon2_bb::MontTable<13> INT_M = poseidon2_bb::to_mont_table(poseidon2_bb::raw::INT);
namespace raw {
constexpr u32 SHIFTS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};  // gates/poseidon2_babybear.rs:41-42
struct Diag {
    u32 v[16];
};
// Montgomery form of 2^-32 * (lane 0: 1; lane i >= 1: 2^SHIFTS[i-1]) : M_I scales every word by 2^-32 = 943718400 first
constexpr Diag diag() {
    Diag d{};
    for (int i = 0; i < 16; i++) {
        const u64 c = (u64)943718400u * (i == 0 ? 1u : ((u64)1 << SHIFTS[i - 1])) % bb::P;
        d.v[i] = (u32)((c << 32) % bb::P);
    }
    return d;
}
}  // namespace raw
__device__ static const raw::Diag DIAG_M = raw::diag();

template <int CTRL>
__device__ __forceinline__ u32 dpp(u32 x) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 sbox7(u32 x) {
    const u32 x2 = bb::mul(x, x), x3 = bb::mul(x2, x), x4 = bb::mul(x2, x2);
    return bb::mul(x3, x4);
}
// permute_external_mut: row r of [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] is [2,3,1,1] rotated by r, so the lane at position
// r of its quad computes 2 x_r + 3 x_{r+1} + x_{r+2} + x_{r+3}; then every word gets the sum of its column class (l mod 4)
__device__ __forceinline__ u32 external_layer(u32 x) {
    const u32 v = dpp<0x39>(x), w = dpp<0x4E>(x), z = dpp<0x93>(x);  // quad_perm [1,2,3,0], [2,3,0,1], [3,0,1,2]
    const u32 t = bb::add(x, v);
    const u32 n = bb::add(bb::add(bb::add(t, t), v), bb::add(w, z));
    u32 s = bb::add(n, dpp<0x124>(n));   // row_ror:4
    s = bb::add(s, dpp<0x128>(s));       // row_ror:8
    return bb::add(n, s);
}
__device__ __forceinline__ u32 row_sum(u32 x) {
    x = bb::add(x, dpp<0x128>(x));
    x = bb::add(x, dpp<0x124>(x));
    x = bb::add(x, dpp<0x122>(x));
    return bb::add(x, dpp<0x121>(x));
}

// x: this lane's word, Montgomery form, canonical (< P); returns the same
__device__ __forceinline__ u32 permute(u32 x, u32 l) {
    x = external_layer(x);
    u32 rc = EXT_M.v[l];
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const u32 rc_r = rc;
        rc = EXT_M.v[16 * (r + 1) + l];  // the next external round's constant (round 4 after the internal rounds)
        x = external_layer(sbox7(bb::add(x, rc_r)));
    }
    const u32 diag = DIAG_M.v[l], k0 = DIAG_M.v[0];
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        const u32 y = sbox7(bb::add(x, INT_M.v[r]));
        x = l == 0 ? y : x;
        // M_I: s <- 2^-32 s; full = sum; s_0 <- full - 2 s_0; s_i <- full + 2^shift s_i
        const u32 full = bb::mul(row_sum(x), k0);
        const u32 d = bb::mul(x, diag);
        x = l == 0 ? bb::sub(full, bb::add(d, d)) : bb::add(full, d);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
        const u32 rc_r = rc;
        if (r + 1 < 8) rc = EXT_M.v[16 * (r + 1) + l];
        x = external_layer(sbox7(bb::add(x, rc_r)));
    }
    return x;
}

}  // namespace poseidon2_bb_coop
