// Gate constraints of a general gate set on the LDE domain: evaluate_gate_constraints_base_batch
// (plonk/vanishing_poly.rs:741-774) -> eval_filtered_base_batch (gates/gate.rs:188-215).  One thread per LDE point j (leaf
// order); for every gate of the circuit's gate set: filter(selector) * sum_i alpha^(t0 + i) * constraint_i, summed over the
// gates, per challenge.  The result is the gate part of the alpha-folded vanishing polynomial; k_quotient (kernels_prover.hip,
// ext_gates = 1) adds the permutation-argument terms to it and divides by Z_H.  Folding on the fly keeps the per-thread state
// at C accumulators instead of num_gate_constraints (123 for PoseidonGate) values; the field is exact, so the order of the
// additions does not change the result.
#include "gates.hpp"
#include "kernels.hpp"

namespace gbk {

namespace {
__device__ __forceinline__ u32 brev32g(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
}  // namespace

template <class F, u32 C>
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
    for (u32 k = 0; k < C; k++) acc[k] = F::zero();
    auto wire = [&](u32 col) { return wires[(size_t)col * N + j]; };
    auto konst = [&](u32 i) { return cs[(size_t)(p.gs.num_selectors + i) * N + j]; };
    for (u32 g = 0; g < p.gs.num_gates; g++) {
        const gb_gate& gd = p.gs.g[g];
        if (gd.kind == GB_GATE_NOOP) continue;
        const T f = gates::filter<F, A>(g, gd, cs[(size_t)gd.selector_index * N + j], p.gs.num_selectors > 1);
        T sum[C];
#pragma unroll
        for (u32 k = 0; k < C; k++) sum[k] = F::zero();
        u32 idx = p.t0;
        auto emit = [&](T c) {
#pragma unroll
            for (u32 k = 0; k < C; k++) sum[k] = F::add(sum[k], F::mul(c, apow[k * p.nterms + idx]));
            idx++;
        };
        gates::eval_gate<F, A>(p.gs, gd, wire, konst, pi_hash, emit);
#pragma unroll
        for (u32 k = 0; k < C; k++) acc[k] = F::add(acc[k], F::mul(f, sum[k]));
    }
#pragma unroll
    for (u32 k = 0; k < C; k++) qv[(((size_t)k << r) + cidx) * n + il] = acc[k];
}

#define GB_G(FF, CC) hipLaunchKernelGGL((k_gate_constraints<FF, CC>), grid, block, 0, st, p, cs, wires, apow, pi_hash, qv)
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
