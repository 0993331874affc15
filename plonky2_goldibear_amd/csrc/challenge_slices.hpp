// Host-side launch planning of the quotient / gate kernels; no dependencies, so that tests/test_device_headers_on_host.py can
// compile it with the host compiler.
#pragma once
#include <cstdint>

namespace gbk {

static constexpr uint32_t MAX_CHALLENGE_SLICES = 16;   // = MAX_CHALLENGES (kernels.hpp)

// The quotient and gate kernels keep their per-challenge accumulators in registers, so they are compiled per challenge count:
// Goldilocks 1 .. gl_wmax, BabyBear 4 .. 10 ((31 - degree_bits) c >= 100 needs c >= 4).  Any other count <= MAX_CHALLENGE_SLICES runs
// as balanced slices (every width floor or ceil of count / slices; BabyBear slices are 5 .. 8 wide).  Returns the number of
// launches and their widths, 0 if the count cannot be covered.  bb_wmax = 8: the reduced BabyBear set of the quotient degree
// factors other than 8 (plain and slice instances 4 .. 8 wide).
static inline uint32_t challenge_slices(uint32_t field, uint32_t gl_wmax, uint32_t count, uint32_t* widths, uint32_t bb_wmax = 10) {
    const bool gl = field == 0;
    const uint32_t wmax = gl ? gl_wmax : bb_wmax, wmin = gl ? 1u : 4u;
    if (count == 0 || count > MAX_CHALLENGE_SLICES) return 0;
    if (count >= wmin && count <= wmax) { widths[0] = count; return 1; }
    const uint32_t smax = gl ? gl_wmax : 8u, smin = gl ? 1u : (bb_wmax == 10 ? 5u : 4u);
    const uint32_t ns = (count + smax - 1) / smax;
    if (ns < 2 || count / ns < smin) return 0;
    for (uint32_t i = 0; i < ns; i++) widths[i] = count / ns + (i < count % ns ? 1 : 0);
    return ns;
}

}  // namespace gbk
