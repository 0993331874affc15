// Host-side Poseidon-12 + Challenger for the Fiat-Shamir transcript (a few hundred permutations per
// proof, microseconds): iop/challenger.rs:18-150, hash/hashing.rs:100-123, plonk/config.rs:58-84.
// The bulk hashing (Merkle trees, PoW grinding) runs on the GPU (poseidon_gl.hpp).
#pragma once
#include <cstring>
#include <vector>

#include "gl_field.hpp"
#include "poseidon_constants.h"

namespace poseidon_gl_host {

using gl::u64;

inline void permute(u64 (&s)[12]) {
    static const u64 RC[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN] = {GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST};
    static const u64 CIRC[12] = {GL_POSEIDON_MDS_CIRC_LIST};
    static const u64 DIAG[12] = {GL_POSEIDON_MDS_DIAG_LIST};
    static const u64 FIRST[12] = {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST};
    static const u64 PRC[22] = {GL_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS_LIST};
    static const u64 VS[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST};
    static const u64 WHAT[22 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST};
    static const u64 INIT[11 * 11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST};
    auto sbox = [](u64 x) {
        u64 x2 = gl::sqr(x), x4 = gl::sqr(x2), x3 = gl::mul(x, x2);
        return gl::mul(x3, x4);
    };
    // MDS entries are below 2^6: a row is a sum of thirteen 64 x 6-bit products (< 2^74) in one 128-bit accumulator and ONE reduction
    auto mds = [&](u64 (&st)[12]) {
        u64 d[24];
        for (int i = 0; i < 12; i++) d[i] = d[i + 12] = st[i];
        for (int r = 0; r < 12; r++) {
            unsigned __int128 acc = (unsigned __int128)d[r] * DIAG[r];
            for (int i = 0; i < 12; i++) acc += (unsigned __int128)d[r + i] * CIRC[i];
            st[r] = gl::reduce128((u64)acc, (u64)(acc >> 64));
        }
    };
    auto full_round = [&](int round) {
        for (int i = 0; i < 12; i++) s[i] = sbox(gl::add(s[i], RC[12 * round + i]));
        mds(s);
    };
    // sum of up to 16 full 64 x 64 products, reduced once: the carries out of bit 128 are worth 2^128 = -2^32 (mod p) each
    struct Acc {
        unsigned __int128 v = 0;
        u64 over = 0;
        void mad(u64 a, u64 b) {
            const unsigned __int128 t = (unsigned __int128)a * b;
            v += t;
            over += v < t;
        }
        u64 value() const { return gl::sub(gl::reduce128((u64)v, (u64)(v >> 64)), over << 32); }
    };
    for (int r = 0; r < 4; r++) full_round(r);
    // the 22 partial rounds in the (v, w_hat, M_init) form of hash/poseidon.rs:718-744, :899-909 - one s-box and 23 products a round
    // instead of a 12 x 12 matrix (round 6: 4.5 -> 1.6 us a permutation; the transcript of a 2^12-row proof is ~120 of them)
    for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], FIRST[i]);
    {
        u64 t[12];
        t[0] = s[0];
        for (int c = 0; c < 11; c++) {
            Acc a;
            for (int r = 1; r < 12; r++) a.mad(s[r], INIT[(r - 1) * 11 + c]);
            t[c + 1] = a.value();
        }
        std::memcpy(s, t, sizeof t);
    }
    for (int k = 0; k < 22; k++) {
        const u64 s0 = gl::add(sbox(s[0]), PRC[k]);
        Acc d;
        d.mad(s0, CIRC[0] + DIAG[0]);
        for (int i = 1; i < 12; i++) {
            d.mad(s[i], WHAT[k * 11 + i - 1]);
            s[i] = gl::add(s[i], gl::mul(s0, VS[k * 11 + i - 1]));
        }
        s[0] = d.value();
    }
    for (int r = 26; r < 30; r++) full_round(r);
}

// hash_n_to_hash_no_pad (hash/hashing.rs:100-133)
inline void hash_no_pad(const u64* in, size_t n, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t k = n - off < 8 ? n - off : 8;
        std::memcpy(st, in + off, k * sizeof(u64));
        permute(st);
    }
    std::memcpy(out, st, 4 * sizeof(u64));
}

// iop/challenger.rs:18-150
struct Challenger {
    u64 state[12] = {0};
    u64 in[8];
    int nin = 0;
    u64 out[8];
    int nout = 0;
    void duplexing() {
        for (int i = 0; i < nin; i++) state[i] = in[i];
        nin = 0;
        permute(state);
        std::memcpy(out, state, sizeof out);
        nout = 8;
    }
    void observe(u64 e) {
        nout = 0;
        in[nin++] = e;
        if (nin == 8) duplexing();
    }
    void observe(const u64* e, size_t n) { for (size_t i = 0; i < n; i++) observe(e[i]); }
    u64 get() {
        if (nin != 0 || nout == 0) duplexing();
        return out[--nout];
    }
    gl::ext2 get_ext() {
        u64 a = get(), b = get();
        return gl::e2(a, b);
    }
};

}  // namespace poseidon_gl_host
