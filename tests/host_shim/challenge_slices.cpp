// Every challenge count 1..16 of both fields: the slices cover the count exactly, in order, with widths the kernels are compiled for.
#include <cstdio>
#include <initializer_list>
#include "challenge_slices.hpp"

int main() {
    int bad = 0;
    for (uint32_t field = 0; field < 2; field++)
        for (uint32_t wmax : {4u, 2u})
          for (uint32_t bbw : {10u, 8u})
            for (uint32_t count = 0; count <= 17; count++) {
                uint32_t w[32] = {};
                const uint32_t ns = gbk::challenge_slices(field, wmax, count, w, bbw);
                const bool gl = field == 0;
                const bool must = count >= (gl ? 1u : 4u) && count <= 16;
                if ((ns != 0) != must) { printf("field %u wmax %u count %u: ns %u\n", field, wmax, count, ns); bad++; continue; }
                if (!ns) continue;
                uint32_t sum = 0, lo = ~0u, hi = 0;
                for (uint32_t i = 0; i < ns; i++) { sum += w[i]; lo = w[i] < lo ? w[i] : lo; hi = w[i] > hi ? w[i] : hi; }
                bool ok = sum == count && hi - lo <= 1;
                if (ns == 1) ok = ok && (gl ? (w[0] >= 1 && w[0] <= wmax) : (w[0] >= 4 && w[0] <= bbw));     // the plain instances
                else ok = ok && (gl ? (lo >= 1 && hi <= wmax) : (lo >= (bbw == 10 ? 5u : 4u) && hi <= 8));       // the SLICE instances
                if (!ok) { printf("field %u wmax %u count %u: bad widths\n", field, wmax, count); bad++; }
            }
    printf("mismatches=%d\n", bad);
    return bad != 0;
}
