// The lane-per-state Poseidon2-16 BabyBear permutation of the hash kernels (csrc/poseidon2_bb.hpp: signed Montgomery products,
// lazy words with compile-time offsets, scale tracking) run on the CPU against the canonical host mirror
// (csrc/poseidon2_bb_host.hpp, the transcript's permutation, itself pinned by the oracle).  argv[1] = number of random states.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "poseidon2_bb.hpp"
#include "poseidon2_bb_host.hpp"

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 100000;
    std::mt19937_64 rng(7);
    long bad = 0;
    for (long t = 0; t < n; t++) {
        uint32_t in[16], dev[16], ref[16];
        for (int i = 0; i < 16; i++) {
            in[i] = (uint32_t)(rng() % bb::P);
            if (t % 7 == 0) in[i] = (i & 1) ? bb::P - 1 : 0;                        // extremes of the canonical range
            if (t % 11 == 0) in[i] = bb::P - 1 - (uint32_t)(rng() % 3);
            if (t % 13 == 0) in[i] = (uint32_t)(rng() % 3);
        }
        for (int i = 0; i < 16; i++) dev[i] = bb::to_mont(in[i]);
        poseidon2_bb::permute(dev);
        for (int i = 0; i < 16; i++) dev[i] = bb::from_mont(dev[i]);
        for (int i = 0; i < 16; i++) ref[i] = in[i];
        poseidon2_bb_host::permute(ref);
        for (int i = 0; i < 16; i++)
            if (dev[i] != ref[i]) {
                if (++bad < 5) printf("mismatch: state %ld word %d: %u != %u\n", t, i, dev[i], ref[i]);
                break;
            }
    }
    // the leaf kernels' use of it (kernels_bb.hip): several absorptions with the permutation left at its final scale, the words
    // that stay brought back lazily (renorm_lazy), the digest through canonical_out
    for (long t = 0; t < n / 4; t++) {
        uint32_t dev[16] = {0}, ref[16] = {0};
        for (int a = 0; a < 3; a++) {
            if (a) for (int i = 8; i < 16; i++) dev[i] = poseidon2_bb::renorm_lazy(dev[i]);
            const int take = (a == 2) ? 1 + (int)(rng() % 8) : 8;          // ragged last absorption
            if (a && take < 8) for (int i = 0; i < 8; i++) dev[i] = poseidon2_bb::renorm_lazy(dev[i]);
            for (int i = 0; i < take; i++) {
                const uint32_t v = (t % 5 == 0) ? bb::P - 1 : (uint32_t)(rng() % bb::P);
                dev[i] = bb::to_mont(v);
                ref[i] = v;
            }
            poseidon2_bb::permute_scaled(dev);
            poseidon2_bb_host::permute(ref);
        }
        for (int i = 0; i < 8; i++)
            if (poseidon2_bb::canonical_out(dev[i]) != ref[i]) {
                if (++bad < 5) printf("sponge mismatch: case %ld word %d\n", t, i);
                break;
            }
    }
    printf("states=%ld mismatches=%ld\n", n, bad);
    return bad != 0;
}
