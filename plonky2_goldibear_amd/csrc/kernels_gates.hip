// Gate constraints of a general gate set on the LDE domain: evaluate_gate_constraints_base_batch
// (plonk/vanishing_poly.rs:741-774) -> eval_filtered_base_batch (gates/gate.rs:188-215).  One thread per LDE point j (leaf
// order); for every gate of the circuit's gate set: filter(selector) * sum_i alpha^(t0 + i) * constraint_i, summed over the
// gates, per challenge.  The result is the gate part of the alpha-folded vanishing polynomial; k_quotient (kernels_prover.hip,
// ext_gates = 1) adds the permutation-argument terms to it and divides by Z_H.  Folding on the fly keeps the per-thread state
// at C accumulators instead of num_gate_constraints (123 for PoseidonGate) values; the field is exact, so the order of the
// additions does not change the result.  Two launches: the short gates, then the in-circuit hash gate (gates.hpp GateSubset).
#include "gates.hpp"
#include "kernels.hpp"

namespace gbk {

namespace {
__device__ __forceinline__ u32 brev32g(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// sum_i c_i * a_i of one gate's constraints against the powers of alpha, kept UNREDUCED until the gate is done: a term is a
// product plus carry adds instead of a modular multiplication and a modular addition (Goldilocks 15 instructions against 29,
// BabyBear 3 against 9 - and BabyBear folds against 6..10 challenges).  At most a few hundred terms per gate.
template <class F>
struct FoldAcc;
template <>
struct FoldAcc<GlF> {  // 160-bit sum of 128-bit products; fewer than 2^31 terms (gl::fold160)
    u32 l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0;
    __device__ __forceinline__ void acc(u64 c, u64 a) {
        u32 p0, p1, p2, p3, k0, k1, k2, k3;
        gl::mul_limbs(c, a, p0, p1, p2, p3);
        l0 = __builtin_addc(l0, p0, 0u, &k0);
        l1 = __builtin_addc(l1, p1, k0, &k1);
        l2 = __builtin_addc(l2, p2, k1, &k2);
        l3 = __builtin_addc(l3, p3, k2, &k3);
        l4 += k3;
    }
    __device__ __forceinline__ u64 finish() const { return gl::canon(gl::fold160(l0, l1, l2, l3, l4)); }
};
template <>
struct FoldAcc<BbF> {  // Montgomery words: the sum S of (c R)(a R) is brought back with one reduction, S R^-1 = (sum c a) R
    u64 lo = 0;
    u32 hi = 0;
    __device__ __forceinline__ void acc(u32 c, u32 a) {
        const u64 p = (u64)c * a, s = lo + p;
        hi += s < p;
        lo = s;
    }
    // S = w0 + 2^32 w1 + 2^64 hi:  S 2^-32 = w0 2^-32 + w1 + hi 2^32  (mod p)
    __device__ __forceinline__ u32 finish() const {
        u32 w1 = (u32)(lo >> 32);  // < 2^32 < 3 p
        w1 = w1 >= bb::P ? w1 - bb::P : w1;
        w1 = w1 >= bb::P ? w1 - bb::P : w1;
        return bb::add(bb::add(bb::reduce((u64)(u32)lo), w1), bb::mul(hi, bb::R2));
    }
};
}  // namespace

// SUBSET = LIGHT_GATES writes qv, HEAVY_GATES (the in-circuit hash gate, when the set has one) adds to it.
template <class F, u32 C, int SUBSET>
__global__ __launch_bounds__(256) void k_gate_constraints(GateParams<F> p, const typename F::T* __restrict__ cs,
                                                          const typename F::T* __restrict__ wires,
                                                          const typename F::T* __restrict__ apow,
                                                          const typename F::T* __restrict__ pi_hash, typename F::T* __restrict__ qv) {
    typedef typename F::T T;
    typedef gates::BaseAlg<F> A;
    const u32 lgn = p.log_n, r = p.rate_bits;
    const size_t n = (size_t)1 << lgn, N = n << r;
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const u32 cidx = (u32)(j >> lgn), jl = (u32)(j & (n - 1));
    const u32 il = brev32g(jl, lgn);
    T acc[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) acc[k] = SUBSET == gates::HEAVY_GATES ? qv[(((size_t)k << r) + cidx) * n + il] : F::zero();
    auto wire = [&](u32 col) { return wires[(size_t)col * N + j]; };
    auto konst = [&](u32 i) { return cs[(size_t)(p.gs.num_selectors + i) * N + j]; };
    u32 idx0 = p.t0;
    for (u32 g = 0; g < p.gs.num_gates; g++) {
        const gb_gate& gd = p.gs.g[g];
        if (gd.kind == GB_GATE_NOOP || gates::is_heavy(gd) != (SUBSET == gates::HEAVY_GATES)) continue;
        const T f = gates::filter<F, A>(g, gd, cs[(size_t)gd.selector_index * N + j], p.gs.num_selectors > 1);
        FoldAcc<F> sum[C];
        u32 idx = idx0;
        auto emit = [&](T c) {
#pragma unroll
            for (u32 k = 0; k < C; k++) sum[k].acc(c, apow[k * p.nterms + idx]);
            idx++;
        };
        gates::eval_gate<F, A, SUBSET>(p.gs, gd, wire, konst, pi_hash, emit);
#pragma unroll
        for (u32 k = 0; k < C; k++) acc[k] = F::add(acc[k], F::mul(f, sum[k].finish()));
    }
#pragma unroll
    for (u32 k = 0; k < C; k++) qv[(((size_t)k << r) + cidx) * n + il] = acc[k];
}

static bool has_heavy(const gates::GateSet& gs) {
    for (u32 g = 0; g < gs.num_gates; g++)
        if (gates::is_heavy(gs.g[g])) return true;
    return false;
}

#define GB_G(FF, CC)                                                                                                          \
    do {                                                                                                                      \
        hipLaunchKernelGGL((k_gate_constraints<FF, CC, gates::LIGHT_GATES>), grid, block, 0, st, p, cs, wires, apow, pi_hash, qv); \
        if (has_heavy(p.gs))                                                                                                  \
            hipLaunchKernelGGL((k_gate_constraints<FF, CC, gates::HEAVY_GATES>), grid, block, 0, st, p, cs, wires, apow, pi_hash, qv); \
    } while (0)
template <>
bool gate_constraints<GlF>(const GateParams<GlF>& p, const u64* cs, const u64* wires, const u64* apow, const u64* pi_hash, u64* qv,
                           hipStream_t st) {
    const size_t N = (size_t)1 << (p.log_n + p.rate_bits);
    const dim3 grid((u32)((N + 255) / 256)), block(256);
    switch (p.num_challenges) {
        case 1: GB_G(GlF, 1); return true;
        case 2: GB_G(GlF, 2); return true;
        case 3: GB_G(GlF, 3); return true;
        case 4: GB_G(GlF, 4); return true;
        default: return false;
    }
}
template <>
bool gate_constraints<BbF>(const GateParams<BbF>& p, const u32* cs, const u32* wires, const u32* apow, const u32* pi_hash, u32* qv,
                           hipStream_t st) {
    const size_t N = (size_t)1 << (p.log_n + p.rate_bits);
    const dim3 grid((u32)((N + 255) / 256)), block(256);
    switch (p.num_challenges) {
        case 6: GB_G(BbF, 6); return true;
        case 7: GB_G(BbF, 7); return true;
        case 8: GB_G(BbF, 8); return true;
        case 9: GB_G(BbF, 9); return true;
        case 10: GB_G(BbF, 10); return true;
        default: return false;
    }
}
#undef GB_G

}  // namespace gbk
