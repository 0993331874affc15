// The transcript's host permutation (csrc/poseidon_gl_host.hpp: fast partial rounds, 128-bit MDS rows - round 6) against the
// DEFINING form of Poseidon-12 over Goldilocks written here from the round constants and the MDS matrix alone
// (hash/poseidon.rs:632-770 `poseidon_naive`: constant layer, s-box layer, MDS layer, 4 + 22 + 4 rounds), on random and extreme
// states.  Test infrastructure only.
#include <cstdio>
#include <cstdlib>

#include "poseidon_gl_host.hpp"

typedef unsigned long long u64;
typedef unsigned __int128 u128;
static const u64 P = 0xFFFFFFFF00000001ULL;
static u64 mulm(u64 a, u64 b) { return (u64)((u128)(a % P) * (b % P) % P); }
static u64 addm(u64 a, u64 b) { return (u64)(((u128)(a % P) + (b % P)) % P); }

static void naive(u64 (&s)[12]) {
    static const u64 RC[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN] = {GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST};
    static const u64 CIRC[12] = {GL_POSEIDON_MDS_CIRC_LIST};
    static const u64 DIAG[12] = {GL_POSEIDON_MDS_DIAG_LIST};
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = addm(s[i], RC[12 * r + i]);
        const bool full = r < 4 || r >= 26;
        for (int i = 0; i < (full ? 12 : 1); i++) {
            const u64 x = s[i], x2 = mulm(x, x), x4 = mulm(x2, x2), x3 = mulm(x, x2);
            s[i] = mulm(x3, x4);
        }
        u64 t[12];
        for (int row = 0; row < 12; row++) {   // res[row] = sum_i s[(row + i) % 12] CIRC[i] + s[row] DIAG[row]   (poseidon.rs:547-557)
            u64 acc = mulm(s[row], DIAG[row]);
            for (int i = 0; i < 12; i++) acc = addm(acc, mulm(s[(row + i) % 12], CIRC[i]));
            t[row] = acc;
        }
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
}

int main(int argc, char** argv) {
    const long count = argc > 1 ? std::atol(argv[1]) : 20000;
    u64 x = 0x9E3779B97F4A7C15ULL;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const u64 extreme[6] = {0, 1, P - 1, P - 2, 0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
    long bad = 0;
    for (long k = 0; k < count; k++) {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) {
            const u64 r = next();
            a[i] = b[i] = (k % 7 == 0 && (r & 3) == 0) ? extreme[(r >> 2) % 6] : r % P;
        }
        poseidon_gl_host::permute(a);
        naive(b);
        for (int i = 0; i < 12; i++)
            if (a[i] != b[i]) { if (bad < 5) std::printf("state %ld word %d: %016llx != %016llx\n", k, i, a[i], b[i]); bad++; }
        // the permutation chained on its own (canonical) output: every later transcript state is such a state
        if (k % 100 == 0) { for (int i = 0; i < 12; i++) if (a[i] >= P) bad++; }
    }
    std::printf("states=%ld mismatches=%ld\n", count, bad);
    return bad ? 1 : 0;
}
