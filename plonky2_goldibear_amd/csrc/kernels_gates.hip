// Gate constraints of a general gate set on the LDE domain: evaluate_gate_constraints_base_batch
// (plonk/vanishing_poly.rs:741-774) -> eval_filtered_base_batch (gates/gate.rs:188-215).  One thread per LDE point j (leaf
// order); for every gate of the circuit's gate set: filter(selector) * sum_i alpha^(t0 + i) * constraint_i, summed over the
// gates, per challenge.  The result is the gate part of the alpha-folded vanishing polynomial; k_quotient (kernels_prover.hip,
// ext_gates = 1) adds the permutation-argument terms to it and divides by Z_H.  Folding on the fly keeps the per-thread state
// at C accumulators instead of num_gate_constraints (123 for PoseidonGate) values; the field is exact, so the order of the
// additions does not change the result.  Two launches: the short gates, then the in-circuit hash gate (gates.hpp GateSubset).
#include <algorithm>
#include <utility>
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
    const size_t n = (size_t)1 << lgn, N = (size_t)1 << p.stride_bits;   // N: column stride of cs / wires (the FRI LDE)
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= (n << r)) return;
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

// The short gates, LDS-tiled.  Every gate reads the wires from column 0 up, so the lane-per-point kernel above fetches a wire
// column once per gate that uses it: 11.7 GB (FETCH_SIZE, raw) per launch for the 16-gate recursion set at 2^18 rows against 2.3 GB
// of distinct data - memory-bound.  Here a workgroup stages the wire / selector / constant values of 64 LDE points in LDS once
// (coalesced 512-byte rows) and its eight waves evaluate eight disjoint groups of gates on them (host-balanced, TiledPlan); the
// per-challenge sums meet in LDS (in the staging area, once every wave is done with it) and wave 0 writes them.  Two such
// workgroups fit a CU's 160 KiB: 4 waves per SIMD, as before.  Same arithmetic, same order inside a gate; sums of field elements.
struct TiledPlan {
    u32 nw, ncs;                         // wire columns / selector + constant columns staged (nw: what the gates read, never
                                         // more than the wires block holds)
    u32 area;                            // columns of the wire area in LDS: max(nw, room for the partial sums)
    unsigned char wave_of[gates::MAX_GATES];
};
static constexpr u32 TILED_WAVES = 8;
template <class F, u32 C>
__global__ __launch_bounds__(64 * TILED_WAVES) void k_gate_constraints_tiled(GateParams<F> p, TiledPlan plan, const typename F::T* __restrict__ cs,
                                                                const typename F::T* __restrict__ wires,
                                                                const typename F::T* __restrict__ apow,
                                                                const typename F::T* __restrict__ pi_hash, typename F::T* __restrict__ qv) {
    typedef typename F::T T;
    typedef gates::BaseAlg<F> A;
    extern __shared__ unsigned char smem_raw[];
    T* shw = reinterpret_cast<T*>(smem_raw);          // [nw][64]
    T* shc = shw + (size_t)plan.area * 64;            // [ncs][64]
    T* red = shw;                                     // [TILED_WAVES - 1][C][64] partial sums, over the wires once they are dead
    const u32 lgn = p.log_n, r = p.rate_bits;
    const size_t n = (size_t)1 << lgn, N = (size_t)1 << p.stride_bits;   // N: column stride of cs / wires (the FRI LDE)
    const size_t j0 = (size_t)blockIdx.x * 64;        // the domain n 2^r is a multiple of 64 (the launcher checks)
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 idx = tid; idx < plan.nw * 64; idx += 64 * TILED_WAVES) shw[idx] = wires[(size_t)(idx >> 6) * N + j0 + (idx & 63)];
    for (u32 idx = tid; idx < plan.ncs * 64; idx += 64 * TILED_WAVES) shc[idx] = cs[(size_t)(idx >> 6) * N + j0 + (idx & 63)];
    __syncthreads();
    T acc[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) acc[k] = F::zero();
    auto wire = [&](u32 col) { return shw[col * 64 + lane]; };
    auto konst = [&](u32 i) { return shc[(p.gs.num_selectors + i) * 64 + lane]; };
    for (u32 g = 0; g < p.gs.num_gates; g++) {
        const gb_gate& gd = p.gs.g[g];
        if (gd.kind == GB_GATE_NOOP || gates::is_heavy(gd) || plan.wave_of[g] != wave) continue;
        const T f = gates::filter<F, A>(g, gd, shc[gd.selector_index * 64 + lane], p.gs.num_selectors > 1);
        FoldAcc<F> sum[C];
        u32 idx = p.t0;
        auto emit = [&](T c) {
#pragma unroll
            for (u32 k = 0; k < C; k++) sum[k].acc(c, apow[k * p.nterms + idx]);
            idx++;
        };
        gates::eval_gate<F, A, gates::LIGHT_GATES>(p.gs, gd, wire, konst, pi_hash, emit);
#pragma unroll
        for (u32 k = 0; k < C; k++) acc[k] = F::add(acc[k], F::mul(f, sum[k].finish()));
    }
    __syncthreads();  // every wave is done reading the staged wires
    if (wave) {
#pragma unroll
        for (u32 k = 0; k < C; k++) red[((wave - 1) * C + k) * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave == 0) {
        const size_t j = j0 + lane;
        const u32 cidx = (u32)(j >> lgn), il = brev32g((u32)(j & (n - 1)), lgn);
#pragma unroll
        for (u32 k = 0; k < C; k++) {
            T v = acc[k];
#pragma unroll
            for (u32 w = 0; w + 1 < TILED_WAVES; w++) v = F::add(v, red[(w * C + k) * 64 + lane]);
            qv[(((size_t)k << r) + cidx) * n + il] = v;
        }
    }
}

// rough instruction count of one gate at one point: modular multiplications of the evaluator (29 instructions each), the
// alpha-fold terms (15 per constraint and challenge for Goldilocks) and the LDS reads - what the wave balance needs
template <class F>
static u32 gate_cost(const gb_gate& g, u32 num_challenges) {
    constexpr u32 D = F::D, ALG_MUL = D * D + D - 1;
    u32 muls = 0;
    switch (g.kind) {
        case GB_GATE_ARITHMETIC: muls = 3 * g.param; break;
        case GB_GATE_ARITHMETIC_EXTENSION: muls = (ALG_MUL + 2 * D) * g.param; break;
        case GB_GATE_MUL_EXTENSION: muls = (ALG_MUL + D) * g.param; break;
        case GB_GATE_BASE_SUM: muls = g.param * gates::base_sum_base(g); break;
        case GB_GATE_REDUCING:
        case GB_GATE_REDUCING_EXTENSION: muls = ALG_MUL * g.param; break;
        case GB_GATE_RANDOM_ACCESS: muls = g.param2 * (g.param + (1u << g.param)); break;
        case GB_GATE_POSEIDON_MDS: muls = 12 * 12 * D; break;
        case GB_GATE_COSET_INTERPOLATION: muls = (3 * ALG_MUL + D) << g.param; break;
        case GB_GATE_EXPONENTIATION: muls = 3 * g.param; break;
        case GB_GATE_POSEIDON2_INTERNAL_PERMUTATION: muls = 32 * D; break;
        default: break;
    }
    return 29 * muls + 15 * gates::num_constraints<F>(g) * num_challenges + 4 * gates::num_wires<F>(g);
}

// wires / constants the short gates touch, and a greedy (longest first) balance of the gates over the waves by gate_cost
template <class F>
static bool make_tiled_plan(const GateParams<F>& p, TiledPlan* plan, size_t* smem_bytes) {
    const size_t N = (size_t)1 << (p.log_n + p.rate_bits);
    if (N % 64) return false;
    u32 nw = 0, nconst = 0, load[TILED_WAVES] = {};
    std::pair<u32, u32> cost[gates::MAX_GATES];
    u32 m = 0;
    for (u32 g = 0; g < p.gs.num_gates; g++) {
        const gb_gate& gd = p.gs.g[g];
        plan->wave_of[g] = 0;
        if (gd.kind == GB_GATE_NOOP || gates::is_heavy(gd)) continue;
        nw = std::max(nw, gates::num_wires<F>(gd));
        nconst = std::max(nconst, gates::num_constants<F>(gd));
        cost[m++] = {gate_cost<F>(gd, p.num_challenges), g};
    }
    if (m == 0) return false;
    std::sort(cost, cost + m, [](const std::pair<u32, u32>& a, const std::pair<u32, u32>& b) { return a.first > b.first; });
    for (u32 i = 0; i < m; i++) {
        const u32 w = (u32)(std::min_element(load, load + TILED_WAVES) - load);
        plan->wave_of[cost[i].second] = (unsigned char)w;
        load[w] += cost[i].first;
    }
    plan->nw = nw;                                                     // staged from `wires`: only columns a gate reads
    plan->area = std::max(nw, (TILED_WAVES - 1) * p.num_challenges);   // the partial sums reuse the wire area
    plan->ncs = p.gs.num_selectors + nconst;
    *smem_bytes = ((size_t)plan->area + plan->ncs) * 64 * sizeof(typename F::T);
    return *smem_bytes <= 150 * 1024;   // 160 KiB of LDS per CU
}

static bool has_heavy(const gates::GateSet& gs) {
    for (u32 g = 0; g < gs.num_gates; g++)
        if (gates::is_heavy(gs.g[g])) return true;
    return false;
}

#define GB_G(FF, CC)                                                                                                          \
    do {                                                                                                                      \
        TiledPlan plan;                                                                                                       \
        size_t smem = 0;                                                                                                      \
        if (make_tiled_plan<FF>(p, &plan, &smem) &&                                                                           \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gate_constraints_tiled<FF, CC>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess)                         \
            hipLaunchKernelGGL((k_gate_constraints_tiled<FF, CC>), dim3((u32)(N / 64)), dim3(64 * TILED_WAVES), smem, st, p, plan, cs, wires, apow, pi_hash, qv); \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_gate_constraints<FF, CC, gates::LIGHT_GATES>), grid, block, 0, st, p, cs, wires, apow, pi_hash, qv); \
        if (has_heavy(p.gs))                                                                                                  \
            hipLaunchKernelGGL((k_gate_constraints<FF, CC, gates::HEAVY_GATES>), grid, block, 0, st, p, cs, wires, apow, pi_hash, qv); \
    } while (0)
// one launch per slice of challenges (challenge_slices, kernels.hpp): a slice [k0, k0 + w) is the same kernel on the alpha powers
// and the qv block of challenge k0 - the gate kernels index both relative to their first challenge
template <class F>
static bool gate_slice(GateParams<F> p, u32 k0, u32 width, const typename F::T* cs, const typename F::T* wires, const typename F::T* apow,
                       const typename F::T* pi_hash, typename F::T* qv, hipStream_t st) {
    const size_t N = (size_t)1 << (p.log_n + p.rate_bits);   // points of the quotient domain
    const dim3 grid((u32)((N + 255) / 256)), block(256);
    apow += (size_t)k0 * p.nterms;
    qv += ((size_t)k0 << p.rate_bits) << p.log_n;
    p.num_challenges = width;
    if constexpr (F::TAG == 0) {
        switch (width) {
            case 1: GB_G(F, 1); return true;
            case 2: GB_G(F, 2); return true;
            case 3: GB_G(F, 3); return true;
            case 4: GB_G(F, 4); return true;
            default: return false;
        }
    } else {
        switch (width) {
            case 4: GB_G(F, 4); return true;
            case 5: GB_G(F, 5); return true;
            case 6: GB_G(F, 6); return true;
            case 7: GB_G(F, 7); return true;
            case 8: GB_G(F, 8); return true;
            case 9: GB_G(F, 9); return true;
            case 10: GB_G(F, 10); return true;
            default: return false;
        }
    }
}
template <class F>
bool gate_constraints(const GateParams<F>& p, const typename F::T* cs, const typename F::T* wires, const typename F::T* apow,
                      const typename F::T* pi_hash, typename F::T* qv, hipStream_t st) {
    u32 widths[MAX_CHALLENGES];
    const u32 ns = challenge_slices(F::TAG, 4, p.num_challenges, widths);
    if (!ns) return false;
    for (u32 i = 0, k0 = 0; i < ns; k0 += widths[i], i++)
        if (!gate_slice<F>(p, k0, widths[i], cs, wires, apow, pi_hash, qv, st)) return false;
    return true;
}
template bool gate_constraints<GlF>(const GateParams<GlF>&, const u64*, const u64*, const u64*, const u64*, u64*, hipStream_t);
template bool gate_constraints<BbF>(const GateParams<BbF>&, const u32*, const u32*, const u32*, const u32*, u32*, hipStream_t);
#undef GB_G

}  // namespace gbk
