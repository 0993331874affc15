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
    // the defining (naive) form: hash/poseidon_goldilocks.rs:927-948
    auto sbox = [](u64 x) {
        u64 x2 = gl::sqr(x), x4 = gl::sqr(x2), x3 = gl::mul(x, x2);
        return gl::mul(x3, x4);
    };
    // MDS entries are below 2^6: each output is two 64-bit sums over the 32-bit halves of the inputs (< 2^42 each, no carries) and ONE
    // reduction - loops over r with unit stride that the host compiler vectorises - instead of thirteen 64 x 64 -> 128 products
    // (round 6: the transcript's ~80 permutations per proof were 0.4 ms of a 4 ms recursion-shaped proof)
    auto mds = [&](u64 (&st)[12]) {
        u64 lo2[24], hi2[24], sl[12], sh[12];
        for (int i = 0; i < 12; i++) {
            lo2[i] = lo2[i + 12] = (uint32_t)st[i];
            hi2[i] = hi2[i + 12] = st[i] >> 32;
        }
        for (int r = 0; r < 12; r++) {
            sl[r] = lo2[r] * DIAG[r];
            sh[r] = hi2[r] * DIAG[r];
        }
        for (int i = 0; i < 12; i++) {
            const u64 c = CIRC[i];
            for (int r = 0; r < 12; r++) {
                sl[r] += lo2[r + i] * c;
                sh[r] += hi2[r + i] * c;
            }
        }
        for (int r = 0; r < 12; r++) {
            const unsigned __int128 acc = (unsigned __int128)sl[r] + ((unsigned __int128)sh[r] << 32);
            st[r] = gl::reduce128((u64)acc, (u64)(acc >> 64));
        }
    };
    int round = 0;
    for (int phase = 0; phase < 3; phase++) {
        int cnt = phase == 1 ? 22 : 4;
        for (int k = 0; k < cnt; k++, round++) {
            for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], RC[12 * round + i]);
            if (phase == 1) s[0] = sbox(s[0]);
            else for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
            mds(s);
        }
    }
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
