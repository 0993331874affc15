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
    auto mds = [&](u64 (&st)[12]) {
        u64 out[12];
        for (int r = 0; r < 12; r++) {
            unsigned __int128 acc = 0;
            for (int i = 0; i < 12; i++) acc += (unsigned __int128)st[(i + r) % 12] * CIRC[i];
            acc += (unsigned __int128)st[r] * DIAG[r];
            out[r] = gl::reduce128((u64)acc, (u64)(acc >> 64));
        }
        std::memcpy(st, out, sizeof out);
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
