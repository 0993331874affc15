// Poseidon width-12 over Goldilocks with ONE STATE PER 16-LANE ROW (lane l < 12 holds word l), for the parts of a proof
// where the number of independent permutations is small: the upper levels of every Merkle tree and all of a
// recursion-sized proof's trees.  The lane-per-state form (poseidon_gl.hpp) runs ~22 k dependent-ish instructions per
// permutation in each lane; a wave with nothing to interleave with takes ~56 us for that, whatever the number of states it
// carries, and a tree below ~2^13 nodes cannot fill the machine.  Here the twelve s-boxes of a full round, the twelve products
// of a partial round's dot product and the eleven updates run in different lanes, so the dependent chain of one permutation
// is ~5 k instructions.  Same function as hash/poseidon_goldilocks.rs:912-922 (fast partial rounds :632-770), bit-exact.
//
// Data movement: a full round's MDS reads the twelve words through LDS (one b64 write, twelve b64 reads per lane); a partial
// round's dot product is a rotate-and-add all-reduce over the row with DPP (row_ror 8, 4, 2, 1), which leaves the sum - the
// next s_0 - in every lane, where the next round's s-box is evaluated redundantly; per-lane constants (w_hat, v) are fetched
// one round ahead.  Blocks are one wave (64 threads = 4 states), so __syncthreads() only orders the LDS traffic.
#pragma once
#include "poseidon_gl.hpp"

namespace poseidon_gl_coop {

using gl::u32;
using gl::u64;
using poseidon_gl::FP_VS;
using poseidon_gl::mul_add_lazy;
using poseidon_gl::mul_lazy;
using poseidon_gl::sbox;

__device__ static const u64 WHATS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST};
__device__ static const u64 INIT[11 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST};

// a + b for ANY two u64 residues -> some u64 residue (a carry out of bit 64 is worth EPS; the corrected sum can carry once more)
__device__ __forceinline__ u64 add_lazy(u64 a, u64 b) {
    u64 s, t;
    const bool c1 = __builtin_uaddll_overflow(a, b, &s);
    const bool c2 = __builtin_uaddll_overflow(s, c1 ? gl::EPS : 0, &t);
    return t + (c2 ? gl::EPS : 0);
}

template <int N>
__device__ __forceinline__ u64 row_ror(u64 x) {  // lane i of each 16-lane row receives lane (i - N) mod 16
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u32)x, 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u32)(x >> 32), 0x120 + N, 0xF, 0xF, false);
    return (u64)(u32)lo | ((u64)(u32)hi << 32);
}
// sum of the row's sixteen values, in every lane (as field elements; lanes may hold different representatives)
__device__ __forceinline__ u64 row_sum(u64 x) {
    x = add_lazy(x, row_ror<8>(x));
    x = add_lazy(x, row_ror<4>(x));
    x = add_lazy(x, row_ror<2>(x));
    return add_lazy(x, row_ror<1>(x));
}

// One permutation per row.  `x`: this lane's word (lanes 12..15: anything; they are kept at zero).  `sh` = this row's 16 u64 of
// LDS.  Returns the lane's output word as a lazy residue (call gl::canon before storing); lanes >= 12 return 0.
__device__ __forceinline__ u64 permute(u64 x, u32 l, u64* __restrict__ sh) {
    const bool live = l < 12;
    const u32 lc = live ? l : 0;       // a valid table index for the idle lanes
    x = poseidon_gl::to_mont(x);       // the shared s-box and additive constants work on Montgomery-form residues (poseidon_gl.hpp)
    auto mds = [&](u64 v) {            // res[l] = sum_i v[(l + i) % 12] CIRC[i] + (l == 0) v[0] DIAG[0]   (:547-557)
        __syncthreads();               // the previous reads of sh are done
        sh[l] = v;
        __syncthreads();
        u64 sl = 0, shh = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u32 j = lc + i;
            j = j >= 12 ? j - 12 : j;
            const u64 w = sh[j];
            sl += (u64)(u32)w * poseidon_gl::mds_circ(i);
            shh += (u64)(u32)(w >> 32) * poseidon_gl::mds_circ(i);
        }
        if (l == 0) {
            sl += (u64)(u32)v * poseidon_gl::MDS_DIAG0;
            shh += (u64)(u32)(v >> 32) * poseidon_gl::MDS_DIAG0;
        }
        // sl + 2^32 shh with shh < 2^42: the bits of shh above 32 are worth EPS each (2^64 = EPS)
        const u64 t = sl + (shh >> 32) * gl::EPS;
        u64 r;
        const bool cy = __builtin_uaddll_overflow(t, shh << 32, &r);
        r += cy ? gl::EPS : 0;
        return live ? r : 0;
    };
    // ---- first half: four full rounds (:889-897)
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        x = add_lazy(x, live ? GB_RC[12 * r + lc] : 0);
        x = mds(sbox(x));
    }
    // ---- partial_first_constant_layer + mds_partial_layer_init (:632-683): s_0 stays, s_c = sum_{r >= 1} s_r INIT[r-1][c-1]
    x = add_lazy(x, live ? GB_FP_FIRST[lc] : 0);
    __syncthreads();
    sh[l] = x;
    __syncthreads();
    u64 s0 = sh[0];
    const bool upper = live && l >= 1;  // the lanes holding s_1..s_11
    const u32 t = upper ? l - 1 : 0;
    {
        u64 acc = 0;
#pragma unroll 1
        for (int r = 1; r < 12; r++) acc = mul_add_lazy(sh[r], INIT[(r - 1) * 11 + t], acc);
        x = upper ? acc : 0;
    }
    // ---- 22 partial rounds (:718-744): s_0 <- sbox(s_0) + c_k; d = 25 s_0 + sum_i s_i w_hat_i; s_i += s_0 v_i; s_0 <- d.
    // Uniform control flow: lane 0 multiplies s_0 by 25 where the others multiply s_i by w_hat_i, the idle lanes contribute 0.
    const u64 c00 = poseidon_gl::mds_circ(0) + poseidon_gl::MDS_DIAG0;
    u64 wh = upper ? WHATS[t] : c00, vs = FP_VS[t];
#pragma unroll 1
    for (int k = 0; k < 22; k++) {
        const u64 wh_k = wh, vs_k = vs;
        if (k + 1 < 22) {  // the next round's per-lane constants, in flight while this round computes
            wh = upper ? WHATS[11 * (k + 1) + t] : c00;
            vs = FP_VS[11 * (k + 1) + t];
        }
        s0 = add_lazy(sbox(s0), GB_FP_RC[k]);
        const u64 term = mul_lazy(upper ? x : s0, wh_k);
        const u64 upd = mul_add_lazy(s0, vs_k, x);
        x = upper ? upd : 0;
        s0 = row_sum(live ? term : 0);
    }
    if (l == 0) x = s0;
    // ---- second half: four full rounds
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        x = add_lazy(x, live ? GB_RC[12 * (4 + 22 + r) + lc] : 0);
        x = mds(sbox(x));
    }
    return poseidon_gl::mont_fold((u32)x, (u32)(x >> 32), 0u, 0u);   // out of Montgomery form; still a lazy residue
}

}  // namespace poseidon_gl_coop
