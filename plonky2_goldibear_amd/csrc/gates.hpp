// Gate constraint evaluators, written once against an algebra so that the same source serves
//   - the quotient kernel (base field, one thread per LDE point: eval_vanishing_poly_base_batch, plonk/vanishing_poly.rs:177-346)
//   - the host verifier   (extension field at zeta:              eval_vanishing_poly,            plonk/vanishing_poly.rs:40-170)
// Each evaluator yields its constraints in the order of the reference gate's eval_unfiltered:
//   NoopGate                 gates/noop.rs                     (none)
//   ConstantGate{n}          gates/constant.rs:64-72           const_i - wire_i
//   PublicInputGate<H>       gates/public_input.rs:52-60       wire_i - public_inputs_hash_i
//   ArithmeticGate{num_ops}  gates/arithmetic_base.rs:83-100   out - (c0 * m0 * m1 + c1 * addend), wires 4i..4i+3
//   PoseidonGate (GL only)   gates/poseidon_goldilocks.rs:124-221  swap bit, 4 deltas, the s-box inputs of every round but the
//                            first (fast partial-round form: hash/poseidon_goldilocks.rs:632-770), 12 outputs = 123 constraints
//   Poseidon2BabyBearGate{num_ops} (BB only)  gates/poseidon2_babybear.rs:203-313  per operation: swap bit, 8 deltas, the s-box
//                            inputs of every round but the first, 16 outputs = 150 constraints
// The caller multiplies by the gate's filter (gates/gate.rs:391-404) and folds with powers of alpha.
#pragma once
#include "field_traits.hpp"
#include "gate_set.hpp"
#include "poseidon_constants.h"

namespace gbk {
namespace gates {

template <class F>
GB_HD u32 num_constraints(const gb_gate& g) {
    switch (g.kind) {
        case GB_GATE_CONSTANT: return g.param;
        case GB_GATE_PUBLIC_INPUT: return F::H;
        case GB_GATE_ARITHMETIC: return g.param;
        case GB_GATE_POSEIDON: return POSEIDON_NUM_CONSTRAINTS;
        case GB_GATE_POSEIDON2_BABYBEAR: return POSEIDON2_BB_CONSTRAINTS_PER_OP * g.param;
        default: return 0;
    }
}
template <class F>
GB_HD u32 num_wires(const gb_gate& g) {
    switch (g.kind) {
        case GB_GATE_CONSTANT: return g.param;
        case GB_GATE_PUBLIC_INPUT: return F::H;
        case GB_GATE_ARITHMETIC: return 4 * g.param;
        case GB_GATE_POSEIDON: return 135;
        case GB_GATE_POSEIDON2_BABYBEAR: return POSEIDON2_BB_WIRES_PER_OP * g.param;
        default: return 0;
    }
}
template <class F>
GB_HD u32 num_constants(const gb_gate& g) {
    return g.kind == GB_GATE_CONSTANT ? g.param : (g.kind == GB_GATE_ARITHMETIC ? 2 : 0);
}

// base-field algebra (device form) and extension-field algebra; constants handed to mulc / addc are base elements in device form
template <class F>
struct BaseAlg {
    typedef typename F::T V;
    typedef typename F::T T;
    static GB_HD V add(V a, V b) { return F::add(a, b); }
    static GB_HD V sub(V a, V b) { return F::sub(a, b); }
    static GB_HD V mul(V a, V b) { return F::mul(a, b); }
    static GB_HD V mulc(V a, T c) { return F::mul(a, c); }
    static GB_HD V addc(V a, T c) { return F::add(a, c); }
    static GB_HD V cst(T c) { return c; }
};
template <class F>
struct ExtAlg {
    typedef typename F::E V;
    typedef typename F::T T;
    static GB_HD V add(V a, V b) { return F::eadd(a, b); }
    static GB_HD V sub(V a, V b) { return F::esub(a, b); }
    static GB_HD V mul(V a, V b) { return F::emul(a, b); }
    static GB_HD V mulc(V a, T c) { return F::escale(a, c); }
    static GB_HD V addc(V a, T c) { return F::eadd(a, F::efrom(c)); }
    static GB_HD V cst(T c) { return F::efrom(c); }
};

// gate.rs:391-404 compute_filter: prod_{i in group, i != row} (i - s) [* (UNUSED - s) when there are several selectors]
template <class F, class A>
GB_HD typename A::V filter(u32 row, const gb_gate& g, typename A::V s, bool many_selectors) {
    typename A::V f = A::cst(F::one());
    for (u32 i = g.group_start; i < g.group_end; i++)
        if (i != row) f = A::mul(f, A::sub(A::cst(F::enc(i)), s));
    if (many_selectors) f = A::mul(f, A::sub(A::cst(F::enc((u64)UNUSED_SELECTOR % (F::ORDER_BITS == 64 ? 0xFFFFFFFF00000001ull : 2013265921ull))), s));
    return f;
}

struct PoseidonTab {
    u64 rc[360], circ[12], diag[12], first[12], fastc[22], vs[242], whats[242], init[121];
};
#define GB_POSEIDON_TAB_INIT                                                                                               \
    {{GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST}, {GL_POSEIDON_MDS_CIRC_LIST}, {GL_POSEIDON_MDS_DIAG_LIST},                      \
     {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST}, {GL_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS_LIST},                 \
     {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST}, {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST},                                \
     {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST}}
static const PoseidonTab POSEIDON_TAB_HOST = GB_POSEIDON_TAB_INIT;
__device__ static const PoseidonTab POSEIDON_TAB_DEV = GB_POSEIDON_TAB_INIT;
GB_HD const PoseidonTab& poseidon_tab() {
#if defined(__HIP_DEVICE_COMPILE__)
    return POSEIDON_TAB_DEV;
#else
    return POSEIDON_TAB_HOST;
#endif
}

template <class A>
GB_HD typename A::V sbox7(typename A::V x) {  // sbox_monomial (hash/poseidon_goldilocks.rs:840-847)
    typename A::V x2 = A::mul(x, x), x4 = A::mul(x2, x2), x3 = A::mul(x, x2);
    return A::mul(x3, x4);
}
template <class A>
GB_HD void mds_layer(typename A::V (&s)[12]) {  // mds_layer_field (:584-592)
    typedef typename A::V V;
    const PoseidonTab& t = poseidon_tab();
    V out[12];
#pragma unroll
    for (u32 r = 0; r < 12; r++) {
        V acc = A::mulc(s[r], t.circ[0] + t.diag[r]);
#pragma unroll
        for (u32 i = 1; i < 12; i++) acc = A::add(acc, A::mulc(s[(i + r) % 12], t.circ[i]));
        out[r] = acc;
    }
#pragma unroll
    for (u32 r = 0; r < 12; r++) s[r] = out[r];
}

// PoseidonGate::eval_unfiltered (Goldilocks: device form == canonical, so the tables are used as they are)
template <class A, class W, class Emit>
GB_HD void eval_poseidon(W&& wire, Emit&& emit) {
    typedef typename A::V V;
    const PoseidonTab& t = poseidon_tab();
    constexpr u32 WIRE_SWAP = 24, START_DELTA = 25, START_FULL_0 = 29, START_PARTIAL = START_FULL_0 + 36, START_FULL_1 = START_PARTIAL + 22;
    const V swap = wire(WIRE_SWAP);
    emit(A::mul(swap, A::sub(swap, A::cst(1))));
    V s[12];
#pragma unroll
    for (u32 i = 0; i < 4; i++) {
        const V lhs = wire(i), rhs = wire(i + 4), delta = wire(START_DELTA + i);
        emit(A::sub(A::mul(swap, A::sub(rhs, lhs)), delta));
        s[i] = A::add(lhs, delta);
        s[i + 4] = A::sub(rhs, delta);
    }
#pragma unroll
    for (u32 i = 8; i < 12; i++) s[i] = wire(i);
    u32 ctr = 0;
#pragma unroll 1
    for (u32 r = 0; r < 4; r++, ctr++) {
#pragma unroll
        for (u32 i = 0; i < 12; i++) s[i] = A::addc(s[i], t.rc[12 * ctr + i]);
        if (r != 0) {
#pragma unroll
            for (u32 i = 0; i < 12; i++) {
                const V in = wire(START_FULL_0 + 12 * (r - 1) + i);
                emit(A::sub(s[i], in));
                s[i] = in;
            }
        }
#pragma unroll
        for (u32 i = 0; i < 12; i++) s[i] = sbox7<A>(s[i]);
        mds_layer<A>(s);
    }
    // partial_first_constant_layer + mds_partial_layer_init (:632-685)
#pragma unroll
    for (u32 i = 0; i < 12; i++) s[i] = A::addc(s[i], t.first[i]);
    {
        V out[12];
        out[0] = s[0];
#pragma unroll
        for (u32 c = 1; c < 12; c++) {
            V acc = A::mulc(s[1], t.init[c - 1]);
#pragma unroll
            for (u32 r = 2; r < 12; r++) acc = A::add(acc, A::mulc(s[r], t.init[(r - 1) * 11 + (c - 1)]));
            out[c] = acc;
        }
#pragma unroll
        for (u32 i = 0; i < 12; i++) s[i] = out[i];
    }
#pragma unroll 1
    for (u32 r = 0; r < 22; r++) {
        const V in = wire(START_PARTIAL + r);
        emit(A::sub(s[0], in));
        s[0] = sbox7<A>(in);
        if (r != 21) s[0] = A::addc(s[0], t.fastc[r]);
        // mds_partial_layer_fast_field (:747-767)
        V d = A::mulc(s[0], t.circ[0] + t.diag[0]);
#pragma unroll
        for (u32 i = 1; i < 12; i++) d = A::add(d, A::mulc(s[i], t.whats[r * 11 + i - 1]));
#pragma unroll
        for (u32 i = 1; i < 12; i++) s[i] = A::add(A::mulc(s[0], t.vs[r * 11 + i - 1]), s[i]);
        s[0] = d;
    }
    ctr += 22;
#pragma unroll 1
    for (u32 r = 0; r < 4; r++, ctr++) {
#pragma unroll
        for (u32 i = 0; i < 12; i++) {
            const V in = wire(START_FULL_1 + 12 * r + i);
            emit(A::sub(A::addc(s[i], t.rc[12 * ctr + i]), in));
            s[i] = sbox7<A>(in);
        }
        mds_layer<A>(s);
    }
#pragma unroll
    for (u32 i = 0; i < 12; i++) emit(A::sub(s[i], wire(12 + i)));
}

// Poseidon2BabyBearGate::eval_unfiltered.  Wire layout (:56-147): per op 33 routed wires (16 in, 16 out, swap) first for all
// ops, then per op 133 non-routed (8 deltas, s-box inputs of full rounds 1..3, of the 13 internal rounds, of full rounds 4..7).
struct Poseidon2Tab {
    u32 ext[128], internal[13];
};
#define GB_POSEIDON2_TAB_INIT {{BB_POSEIDON2_EXTERNAL_CONSTANTS_LIST}, {BB_POSEIDON2_INTERNAL_CONSTANTS_LIST}}
static const Poseidon2Tab POSEIDON2_TAB_HOST = GB_POSEIDON2_TAB_INIT;
__device__ static const Poseidon2Tab POSEIDON2_TAB_DEV = GB_POSEIDON2_TAB_INIT;
GB_HD const Poseidon2Tab& poseidon2_tab() {
#if defined(__HIP_DEVICE_COMPILE__)
    return POSEIDON2_TAB_DEV;
#else
    return POSEIDON2_TAB_HOST;
#endif
}
template <class A>
GB_HD void p2_external(typename A::V (&s)[16]) {  // permute_external_mut (:804-832), apply_mat4 (:903-917)
    typedef typename A::V V;
#pragma unroll
    for (u32 i = 0; i < 16; i += 4) {
        const V t01 = A::add(s[i], s[i + 1]), t23 = A::add(s[i + 2], s[i + 3]), t0123 = A::add(t01, t23);
        const V t01123 = A::add(t0123, s[i + 1]), t01233 = A::add(t0123, s[i + 3]);
        const V n3 = A::add(t01233, A::add(s[i], s[i])), n1 = A::add(t01123, A::add(s[i + 2], s[i + 2]));
        s[i] = A::add(t01123, t01);
        s[i + 2] = A::add(t01233, t23);
        s[i + 1] = n1;
        s[i + 3] = n3;
    }
    V sums[4];
#pragma unroll
    for (u32 k = 0; k < 4; k++) sums[k] = A::add(A::add(s[k], s[4 + k]), A::add(s[8 + k], s[12 + k]));
#pragma unroll
    for (u32 i = 0; i < 16; i++) s[i] = A::add(s[i], sums[i % 4]);
}
template <class F, class A>
GB_HD void p2_internal(typename A::V (&s)[16]) {  // permute_internal_mut (:787-802)
    typedef typename A::V V;
    constexpr u32 SHIFTS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    const typename F::T k = F::enc(943718400u);
#pragma unroll
    for (u32 i = 0; i < 16; i++) s[i] = A::mulc(s[i], k);
    V part = s[1];
#pragma unroll
    for (u32 i = 2; i < 16; i++) part = A::add(part, s[i]);
    const V full = A::add(part, s[0]);
    s[0] = A::sub(part, s[0]);
#pragma unroll
    for (u32 i = 0; i < 15; i++) s[i + 1] = A::add(full, A::mulc(s[i + 1], F::enc((u64)1 << SHIFTS[i])));
}
template <class F, class A, class W, class Emit>
GB_HD void eval_poseidon2_bb(u32 num_ops, W&& wire, Emit&& emit) {
    typedef typename A::V V;
    const Poseidon2Tab& t = poseidon2_tab();
    constexpr u32 ROUTED = 33, NON_ROUTED = 8 + 16 * 7 + 13;
    for (u32 op = 0; op < num_ops; op++) {
        const u32 in0 = ROUTED * op, out0 = in0 + 16, start_delta = num_ops * ROUTED + op * NON_ROUTED;
        const u32 start_full_0 = start_delta + 8, start_partial = start_full_0 + 48, start_full_1 = start_partial + 13;
        const V swap = wire(in0 + 32);
        emit(A::mul(swap, A::sub(swap, A::cst(F::one()))));
        V s[16];
#pragma unroll
        for (u32 i = 0; i < 8; i++) {
            const V lhs = wire(in0 + i), rhs = wire(in0 + i + 8), delta = wire(start_delta + i);
            emit(A::sub(A::mul(swap, A::sub(rhs, lhs)), delta));
            s[i] = A::add(lhs, delta);
            s[i + 8] = A::sub(rhs, delta);
        }
        p2_external<A>(s);
#pragma unroll 1
        for (u32 r = 0; r < 4; r++) {
#pragma unroll
            for (u32 i = 0; i < 16; i++) s[i] = A::addc(s[i], F::enc(t.ext[16 * r + i]));
            if (r > 0) {
#pragma unroll
                for (u32 i = 0; i < 16; i++) {
                    const V in = wire(start_full_0 + 16 * (r - 1) + i);
                    emit(A::sub(s[i], in));
                    s[i] = in;
                }
            }
#pragma unroll
            for (u32 i = 0; i < 16; i++) s[i] = sbox7<A>(s[i]);
            p2_external<A>(s);
        }
#pragma unroll 1
        for (u32 r = 0; r < 13; r++) {
            const V in = wire(start_partial + r);
            emit(A::sub(A::addc(s[0], F::enc(t.internal[r])), in));
            s[0] = sbox7<A>(in);
            p2_internal<F, A>(s);
        }
#pragma unroll 1
        for (u32 r = 4; r < 8; r++) {
#pragma unroll
            for (u32 i = 0; i < 16; i++) {
                const V in = wire(start_full_1 + 16 * (r - 4) + i);
                emit(A::sub(A::addc(s[i], F::enc(t.ext[16 * r + i])), in));
                s[i] = sbox7<A>(in);
            }
            p2_external<A>(s);
        }
#pragma unroll
        for (u32 i = 0; i < 16; i++) emit(A::sub(s[i], wire(out0 + i)));
    }
}

// wire(col) / konst(i) give the opened (or LDE) value of a wire / of the i-th constant after the selectors
template <class F, class A, class W, class K, class Emit>
GB_HD void eval_gate(const gb_gate& g, W&& wire, K&& konst, const typename F::T* pi_hash, Emit&& emit) {
    typedef typename A::V V;
    switch (g.kind) {
        case GB_GATE_CONSTANT:
            for (u32 i = 0; i < g.param; i++) emit(A::sub(konst(i), wire(i)));
            break;
        case GB_GATE_PUBLIC_INPUT:
            for (u32 i = 0; i < F::H; i++) emit(A::sub(wire(i), A::cst(pi_hash[i])));
            break;
        case GB_GATE_ARITHMETIC: {
            const V c0 = konst(0), c1 = konst(1);
            for (u32 i = 0; i < g.param; i++) {
                const V m0 = wire(4 * i), m1 = wire(4 * i + 1), ad = wire(4 * i + 2), out = wire(4 * i + 3);
                emit(A::sub(out, A::add(A::mul(A::mul(m0, m1), c0), A::mul(ad, c1))));
            }
            break;
        }
        case GB_GATE_POSEIDON:
            if constexpr (F::TAG == 0) eval_poseidon<A>(wire, emit);
            break;
        case GB_GATE_POSEIDON2_BABYBEAR:
            if constexpr (F::TAG == 1) eval_poseidon2_bb<F, A>(g.param, wire, emit);
            break;
        default:
            break;
    }
}

}  // namespace gates
}  // namespace gbk
