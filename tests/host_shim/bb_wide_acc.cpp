// BbF's unreduced 96-bit sums (csrc/field_traits.hpp: acc_from / acc_mac / acc_mac2 / acc_finish - the quotient kernel's alpha
// fold) against sums of Montgomery products and canonical additions, on random and extreme operands and with enough terms for the
// high word to count.  The device build replaces acc_mac2's body by two multiply-adds and two add-with-carry; what is checked
// here is the arithmetic both share - the accumulation and the final reduction.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "field_traits.hpp"

using gbk::BbF;

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000;
    std::mt19937_64 rng(11);
    long bad = 0;
    for (long t = 0; t < n; t++) {
        const int terms = 1 + (int)(rng() % 1200);
        const uint32_t start = (uint32_t)(rng() % bb::P);
        BbF::Acc a0 = BbF::acc_from(start), a1 = BbF::acc_from(start), a2 = BbF::acc_from(start);
        uint32_t r0 = start, r1 = start;
        for (int i = 0; i < terms; i++) {
            uint32_t x = (uint32_t)(rng() % bb::P), c0 = (uint32_t)(rng() % bb::P), c1 = (uint32_t)(rng() % bb::P);
            if (t % 3 == 0) { x = bb::P - 1; c0 = bb::P - 1; c1 = bb::P - 1 - (uint32_t)(i & 1); }   // the largest products
            BbF::acc_mac2(a0, a1, x, c0, c1);
            BbF::acc_mac(a2, x, c0);
            r0 = bb::add(r0, bb::mul(x, c0));
            r1 = bb::add(r1, bb::mul(x, c1));
        }
        if (BbF::acc_finish(a0) != r0 || BbF::acc_finish(a1) != r1 || BbF::acc_finish(a2) != r0) {
            if (++bad < 5) printf("mismatch: case %ld (%d terms)\n", t, terms);
        }
    }
    printf("cases=%ld mismatches=%ld\n", n, bad);
    return bad != 0;
}
