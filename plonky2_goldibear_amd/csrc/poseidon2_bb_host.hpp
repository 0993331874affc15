// Host-side Poseidon2-16 over BabyBear (canonical words) for the Fiat-Shamir transcript of prove():
// Challenger (iop/challenger.rs:18-150), hash_n_to_hash_no_pad (hash/hashing.rs:100-133) and the circuit digest.
// Same permutation as poseidon2_bb.hpp (reference: hash/poseidon2_babybear.rs:150-159, order restated in
// gates/poseidon2_babybear.rs:609-672); a few dozen permutations per proof, so plain % arithmetic.
#pragma once
#include <cstring>

#include "bb_field.hpp"
#include "poseidon_constants.h"

namespace poseidon2_bb_host {

typedef unsigned int u32;
typedef unsigned long long u64;
static constexpr u32 P = bb::P;

inline u32 addm(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
inline u32 subm(u32 a, u32 b) { return a >= b ? a - b : a + P - b; }
inline u32 mulm(u32 a, u32 b) { return (u32)(((u64)a * b) % P); }
inline u32 sbox7(u32 x) { u32 x2 = mulm(x, x), x4 = mulm(x2, x2), x3 = mulm(x, x2); return mulm(x3, x4); }

inline void external_layer(u32 s[16]) {  // gates/poseidon2_babybear.rs:804-832, :903-917
    for (int i = 0; i < 16; i += 4) {
        u32 a = s[i], b = s[i + 1], c = s[i + 2], d = s[i + 3];
        u32 t01 = addm(a, b), t23 = addm(c, d), t0123 = addm(t01, t23);
        u32 t01123 = addm(t0123, b), t01233 = addm(t0123, d);
        s[i + 3] = addm(t01233, addm(a, a));
        s[i + 1] = addm(t01123, addm(c, c));
        s[i] = addm(t01123, t01);
        s[i + 2] = addm(t01233, t23);
    }
    u32 sums[4];
    for (int k = 0; k < 4; k++) sums[k] = addm(addm(s[k], s[4 + k]), addm(s[8 + k], s[12 + k]));
    for (int i = 0; i < 16; i++) s[i] = addm(s[i], sums[i & 3]);
}
inline void internal_layer(u32 s[16]) {  // :787-802
    static const int SH[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    for (int i = 0; i < 16; i++) s[i] = mulm(s[i], 943718400u);  // 2^-32 mod p
    u32 part = 0;
    for (int i = 1; i < 16; i++) part = addm(part, s[i]);
    const u32 full = addm(part, s[0]);
    s[0] = subm(part, s[0]);
    for (int i = 0; i < 15; i++) s[i + 1] = addm(full, mulm(s[i + 1], 1u << SH[i]));
}
inline void permute(u32 s[16]) {
    static const u32 EXT[128] = {BB_POSEIDON2_EXTERNAL_CONSTANTS_LIST};
    static const u32 INT[13] = {BB_POSEIDON2_INTERNAL_CONSTANTS_LIST};
    external_layer(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(addm(s[i], EXT[16 * r + i]));
        external_layer(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7(addm(s[0], INT[r]));
        internal_layer(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(addm(s[i], EXT[16 * r + i]));
        external_layer(s);
    }
}

inline void hash_no_pad(const u32* in, size_t n, u32 out[8]) {
    u32 st[16] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t k = n - off < 8 ? n - off : 8;
        std::memcpy(st, in + off, k * sizeof(u32));
        permute(st);
    }
    std::memcpy(out, st, 8 * sizeof(u32));
}

struct Challenger {
    u32 state[16] = {0};
    u32 in[8];
    int nin = 0;
    u32 out[8];
    int nout = 0;
    void duplexing() {
        for (int i = 0; i < nin; i++) state[i] = in[i];
        nin = 0;
        permute(state);
        std::memcpy(out, state, sizeof out);
        nout = 8;
    }
    void observe(u32 e) {
        nout = 0;
        in[nin++] = e;
        if (nin == 8) duplexing();
    }
    void observe(const u32* e, size_t n) { for (size_t i = 0; i < n; i++) observe(e[i]); }
    u32 get() {
        if (nin != 0 || nout == 0) duplexing();
        return out[--nout];
    }
};

}  // namespace poseidon2_bb_host
