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
    printf("states=%ld mismatches=%ld\n", n, bad);
    return bad != 0;
}
