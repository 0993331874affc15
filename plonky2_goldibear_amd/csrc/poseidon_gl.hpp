// Poseidon width-12 permutation over Goldilocks, one permutation per lane (gfx950).
//
// Same function as the reference's PoseidonGoldilocks::poseidon (hash/poseidon_goldilocks.rs:912-922):
// 4 full rounds, 22 partial rounds in the "fast" (v, w_hat, M_init) form (:899-909, :718-744),
// 4 full rounds; x^7 s-box (:840-846); MDS = circulant(MDS_CIRC) + diag(MDS_DIAG) (:301-302,:547-557).
// Round index loops stay rolled (uniform index -> scalar loads of the constants); the 12-lane
// state loops are unrolled so the state lives in VGPRs.
#pragma once
#include "gl_field.hpp"
#include "poseidon_constants.h"

namespace poseidon_gl {

using gl::u32;
using gl::u64;

static constexpr int WIDTH = 12, RATE = 8, HOUT = 4, N_PARTIAL = 22, HALF_FULL = 4;

__device__ static const u64 RC[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN] = {GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST};
__device__ static const u64 FP_FIRST[12] = {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST};
__device__ static const u64 FP_RC[22] = {GL_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS_LIST};
__device__ static const u64 FP_VS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST};
__device__ static const u64 FP_WHATS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST};
__device__ static const u64 FP_INIT[11 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST};

// MDS_MATRIX_CIRC / MDS_MATRIX_DIAG (hash/poseidon_goldilocks.rs:301-302) as immediates
__device__ __forceinline__ constexpr u32 mds_circ(int i) {
    constexpr u32 c[12] = {GL_POSEIDON_MDS_CIRC_LIST};
    return c[i];
}
static constexpr u32 MDS_DIAG0 = 8;

__device__ __forceinline__ u64 sbox(u64 x) {
    u64 x2 = gl::sqr(x), x4 = gl::sqr(x2), x3 = gl::mul(x, x2);
    return gl::mul(x3, x4);
}

// res[r] = sum_i s[(i+r)%12] * CIRC[i] + s[r]*DIAG[r].  Entries < 2^6: split each element into
// 32-bit halves, accumulate the two 12-term sums in 64 bits (< 2^42), recombine and reduce once
// (same decomposition as the reference's mds_layer :497-528, without its FFT form).
__device__ __forceinline__ void mds_layer(u64 (&s)[12]) {
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 sl = 0, sh = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            sl += (u64)lo[(i + r) % 12] * mds_circ(i);
            sh += (u64)hi[(i + r) % 12] * mds_circ(i);
        }
        if (r == 0) {
            sl += (u64)lo[0] * MDS_DIAG0;
            sh += (u64)hi[0] * MDS_DIAG0;
        }
        // value = sl + 2^32 * sh  (< 2^75)
        u64 t_lo = sh << 32, t_hi = sh >> 32;
        u64 l = sl + t_lo;
        u64 h = t_hi + (l < sl ? 1 : 0);
        s[r] = gl::reduce128(l, h);
    }
}

__device__ __forceinline__ void full_rounds(u64 (&s)[12], int round0) {
    for (int k = 0; k < HALF_FULL; k++) {
        const u64* rc = RC + 12 * (round0 + k);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox(gl::add(s[i], rc[i]));
        mds_layer(s);
    }
}

__device__ __forceinline__ void partial_rounds(u64 (&s)[12]) {
    // partial_first_constant_layer (:632-638) + mds_partial_layer_init (:657-683)
    u64 t[12];
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = gl::add(s[i], FP_FIRST[i]);
    s[0] = t[0];
#pragma unroll
    for (int c = 1; c < 12; c++) {
        u64 acc = 0;
#pragma unroll
        for (int r = 1; r < 12; r++) acc = gl::add(acc, gl::mul(t[r], FP_INIT[(r - 1) * 11 + (c - 1)]));
        s[c] = acc;
    }
    for (int k = 0; k < N_PARTIAL; k++) {
        const u64* wh = FP_WHATS + 11 * k;
        const u64* vs = FP_VS + 11 * k;
        u64 s0 = gl::add(sbox(s[0]), FP_RC[k]);
        // mds_partial_layer_fast (:718-744): d = s0*(CIRC[0]+DIAG[0]) + sum_i s[i]*w_hat[i-1]
        u64 d = gl::mul(s0, (u64)(mds_circ(0) + MDS_DIAG0));
#pragma unroll
        for (int i = 1; i < 12; i++) d = gl::add(d, gl::mul(s[i], wh[i - 1]));
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = gl::add(s[i], gl::mul(s0, vs[i - 1]));
        s[0] = d;
    }
}

__device__ __forceinline__ void permute(u64 (&s)[12]) {
    full_rounds(s, 0);
    partial_rounds(s);
    full_rounds(s, HALF_FULL + N_PARTIAL);
}

}  // namespace poseidon_gl
