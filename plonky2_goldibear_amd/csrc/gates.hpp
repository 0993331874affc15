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
//   ArithmeticExtensionGate  gates/arithmetic_extension.rs:82-100      out - (c0 * m0 * m1 + c1 * addend) on D-tuples, D each
//   MulExtensionGate         gates/multiplication_extension.rs:77-94  out - c0 * m0 * m1, D each
//   BaseSumGate<B>           gates/base_sum.rs:77-93                  sum - reduce_with_powers(limbs, B); prod_{i<B} (limb - i) per limb
//   ReducingGate             gates/reducing.rs:89-115                 acc * alpha + coeff_i - acc_i (base-field coefficients), D each
//   ReducingExtensionGate    gates/reducing_extension.rs:95-120       the same with D-tuple coefficients
//   RandomAccessGate         gates/random_access.rs:150-200           per copy: bits boolean, index reconstruction, folded list - claimed;
//                                                                     then the extra constants
//   PoseidonMdsGate (GL)     gates/poseidon_goldilocks_mds.rs:152-180 out_r - MDS row r on D-tuples
//   CosetInterpolationGate   gates/coset_interpolation.rs:216-268     shifted point, the intermediate (eval, prod) pairs, the value
//   ExponentiationGate       gates/exponentiation.rs:99-135           square-and-multiply chain, output
//   AddManyGate              gates/add_many.rs:80-90                  sum of the addends - sum wire, per operation
//   ApplyMat4Gate            gates/apply_mat4.rs:80-108               out_i - (the Poseidon2 4x4 matrix on four D-tuples)_i
//   Poseidon2InternalPermutationGate (BB)  gates/poseidon2_internal_permutation.rs:75-112  out_i - (M_I on sixteen D-tuples)_i
// A D-tuple of consecutive wires is an element of the extension FIELD on the prover's LDE points (base-field wires) and of the
// extension ALGEBRA F_ext[x]/(x^D - W) at the verifier's zeta (vars.get_local_ext / get_local_ext_algebra, plonk/vars.rs); Tup<>
// below is written over the algebra A and is both.
// The caller multiplies by the gate's filter (gates/gate.rs:391-404) and folds with powers of alpha.
#pragma once
#include "field_traits.hpp"
#include "gate_set.hpp"
#include "poseidon_constants.h"

namespace gbk {
namespace gates {

GB_HD u32 base_sum_base(const gb_gate& g) { return g.param2 ? g.param2 : 2; }
GB_HD u32 interpolation_intermediates(const gb_gate& g) {  // coset_interpolation.rs:148-150
    return g.param2 > 1 ? ((1u << g.param) - 2) / (g.param2 - 1) : 0;
}
template <class F>
GB_HD u32 num_constraints(const gb_gate& g) {
    constexpr u32 D = F::D;
    switch (g.kind) {
        case GB_GATE_CONSTANT: return g.param;
        case GB_GATE_PUBLIC_INPUT: return F::H;
        case GB_GATE_ARITHMETIC: return g.param;
        case GB_GATE_POSEIDON: return POSEIDON_NUM_CONSTRAINTS;
        case GB_GATE_POSEIDON2_BABYBEAR: return POSEIDON2_BB_CONSTRAINTS_PER_OP * g.param;
        case GB_GATE_ARITHMETIC_EXTENSION:
        case GB_GATE_MUL_EXTENSION:
        case GB_GATE_REDUCING:
        case GB_GATE_REDUCING_EXTENSION: return D * g.param;
        case GB_GATE_BASE_SUM: return 1 + g.param;
        case GB_GATE_RANDOM_ACCESS: return (g.param + 2) * g.param2 + g.param3;
        case GB_GATE_POSEIDON_MDS: return 12 * D;
        case GB_GATE_COSET_INTERPOLATION: return 2 * D + 2 * D * interpolation_intermediates(g);
        case GB_GATE_EXPONENTIATION: return g.param + 1;
        case GB_GATE_ADD_MANY: return g.param2;
        case GB_GATE_APPLY_MAT4: return 4 * D * g.param;
        case GB_GATE_POSEIDON2_INTERNAL_PERMUTATION: return 16 * D;
        default: return 0;
    }
}
template <class F>
GB_HD u32 num_wires(const gb_gate& g) {
    constexpr u32 D = F::D;
    switch (g.kind) {
        case GB_GATE_CONSTANT: return g.param;
        case GB_GATE_PUBLIC_INPUT: return F::H;
        case GB_GATE_ARITHMETIC: return 4 * g.param;
        case GB_GATE_POSEIDON: return 135;
        case GB_GATE_POSEIDON2_BABYBEAR: return POSEIDON2_BB_WIRES_PER_OP * g.param;
        case GB_GATE_ARITHMETIC_EXTENSION: return 4 * D * g.param;
        case GB_GATE_MUL_EXTENSION: return 3 * D * g.param;
        case GB_GATE_BASE_SUM: return 1 + g.param;
        case GB_GATE_REDUCING: return 2 * D + g.param * (D + 1);
        case GB_GATE_REDUCING_EXTENSION: return 2 * D + 2 * D * g.param;
        case GB_GATE_RANDOM_ACCESS: return (2 + (1u << g.param)) * g.param2 + g.param3 + g.param2 * g.param;
        case GB_GATE_POSEIDON_MDS: return 24 * D;
        case GB_GATE_COSET_INTERPOLATION: return 1 + (D << g.param) + 2 * D + D * (2 * interpolation_intermediates(g) + 1);
        case GB_GATE_EXPONENTIATION: return 2 + 2 * g.param;
        case GB_GATE_ADD_MANY: return (g.param + 1) * g.param2;
        case GB_GATE_APPLY_MAT4: return 8 * D * g.param;
        case GB_GATE_POSEIDON2_INTERNAL_PERMUTATION: return 32 * D;
        default: return 0;
    }
}
template <class F>
GB_HD u32 num_constants(const gb_gate& g) {
    switch (g.kind) {
        case GB_GATE_CONSTANT: return g.param;
        case GB_GATE_ARITHMETIC:
        case GB_GATE_ARITHMETIC_EXTENSION: return 2;
        case GB_GATE_MUL_EXTENSION: return 1;
        case GB_GATE_RANDOM_ACCESS: return g.param3;
        default: return 0;
    }
}

// base-field algebra (device form) and extension-field algebra; constants handed to mulc / addc are base elements in device form
template <class F>
struct BaseAlg {
    typedef typename F::T V;
    typedef typename F::T T;
    static constexpr bool IS_GL_BASE = F::TAG == 0;  // base-field Goldilocks values: the device fast paths apply
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
    static constexpr bool IS_GL_BASE = false;
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
GB_HD const PoseidonTab& poseidon_tab() { return GB_DEV_OR_HOST(POSEIDON_TAB_DEV, POSEIDON_TAB_HOST); }

template <class A>
GB_HD typename A::V sbox7(typename A::V x) {  // sbox_monomial (hash/poseidon_goldilocks.rs:840-847)
    typename A::V x2 = A::mul(x, x), x4 = A::mul(x2, x2), x3 = A::mul(x, x2);
    return A::mul(x3, x4);
}
// The MDS layer on base-field values in the quotient kernel: entries < 2^6, so each output is two 64-bit sums over the 32-bit
// halves of the inputs and ONE reduction (the decomposition of the reference's mds_layer, hash/poseidon_goldilocks.rs:497-528) -
// 24 v_mad_u64_u32 per output where twelve modular multiplications by small constants would be ~350 instructions.  Half of
// PoseidonGate's multiplications are these.
GB_HD void mds_layer_gl_base(u64 (&s)[12]) {
    const PoseidonTab& t = poseidon_tab();
    u32 lo[12], hi[12];
#pragma unroll
    for (u32 i = 0; i < 12; i++) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (u32 r = 0; r < 12; r++) {
        u64 sl = (u64)lo[r] * (u32)t.diag[r], sh = (u64)hi[r] * (u32)t.diag[r];
#pragma unroll
        for (u32 i = 0; i < 12; i++) {
            sl += (u64)lo[(i + r) % 12] * (u32)t.circ[i];
            sh += (u64)hi[(i + r) % 12] * (u32)t.circ[i];
        }
        // sl + 2^32 sh with sh < 2^42: the bits of sh above 32 are worth EPS each (2^64 = EPS); one carry fix, then canonical
        const u64 tt = sl + (sh >> 32) * gl::EPS;
        u64 r2;
        const bool cy = __builtin_uaddll_overflow(tt, sh << 32, &r2);
        r2 += cy ? gl::EPS : 0;
        s[r] = gl::canon(r2);
    }
}

template <class A>
GB_HD void mds_layer(typename A::V (&s)[12]) {  // mds_layer_field (:584-592)
    typedef typename A::V V;
    if constexpr (A::IS_GL_BASE) {
        mds_layer_gl_base(s);
        return;
    }
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
GB_HD const Poseidon2Tab& poseidon2_tab() { return GB_DEV_OR_HOST(POSEIDON2_TAB_DEV, POSEIDON2_TAB_HOST); }
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

// A D-tuple of values of the algebra A: the extension field over base-field wires, the extension algebra over opened wires
// (field/src/extension/algebra.rs: coordinate-wise add, x^D = W for the product, scalar_mul coordinate-wise).
template <class F, class A>
struct Tup {
    typedef typename A::V V;
    typedef typename F::T T;
    static constexpr u32 D = F::D;
    V c[D];
    template <class W>
    static GB_HD Tup load(W&& wire, u32 start) {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = wire(start + k);
        return r;
    }
    static GB_HD Tup zero() {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = A::cst(F::zero());
        return r;
    }
    static GB_HD Tup from_base(V v) {
        Tup r = zero();
        r.c[0] = v;
        return r;
    }
    GB_HD Tup operator+(const Tup& o) const {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = A::add(c[k], o.c[k]);
        return r;
    }
    GB_HD Tup operator-(const Tup& o) const {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = A::sub(c[k], o.c[k]);
        return r;
    }
    GB_HD Tup operator*(const Tup& o) const {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) {
            V lo = A::mul(c[0], o.c[k]);
#pragma unroll
            for (u32 i = 1; i <= k; i++) lo = A::add(lo, A::mul(c[i], o.c[k - i]));
            if (k + 1 < D) {
                V hi = A::mul(c[k + 1], o.c[D - 1]);
#pragma unroll
                for (u32 i = k + 2; i < D; i++) hi = A::add(hi, A::mul(c[i], o.c[k + D - i]));
                lo = A::add(lo, A::mulc(hi, F::ext_w()));
            }
            r.c[k] = lo;
        }
        return r;
    }
    GB_HD Tup scalar(V s) const {  // ExtensionAlgebra::scalar_mul
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = A::mul(c[k], s);
        return r;
    }
    GB_HD Tup scalar_c(T s) const {
        Tup r;
#pragma unroll
        for (u32 k = 0; k < D; k++) r.c[k] = A::mulc(c[k], s);
        return r;
    }
    template <class Emit>
    GB_HD void emit_all(Emit&& emit) const {
#pragma unroll
        for (u32 k = 0; k < D; k++) emit(c[k]);
    }
};

// ArithmeticExtensionGate (with_addend) / MulExtensionGate
template <class F, class A, class W, class K, class Emit>
GB_HD void eval_arithmetic_extension(u32 num_ops, bool with_addend, W&& wire, K&& konst, Emit&& emit) {
    typedef Tup<F, A> X;
    constexpr u32 D = F::D;
    const typename A::V c0 = konst(0), c1 = with_addend ? konst(1) : c0;
    const u32 stride = (with_addend ? 4 : 3) * D;
    for (u32 i = 0; i < num_ops; i++) {
        const X m0 = X::load(wire, stride * i), m1 = X::load(wire, stride * i + D);
        X computed = (m0 * m1).scalar(c0);
        if (with_addend) computed = computed + X::load(wire, stride * i + 2 * D).scalar(c1);
        (X::load(wire, stride * i + stride - D) - computed).emit_all(emit);
    }
}

template <class F, class A, class W, class Emit>
GB_HD void eval_base_sum(u32 num_limbs, u32 base, W&& wire, Emit&& emit) {
    typedef typename A::V V;
    V acc = A::cst(F::zero());
    for (u32 i = num_limbs; i-- > 0;) acc = A::add(A::mulc(acc, F::enc(base)), wire(1 + i));  // reduce_with_powers
    emit(A::sub(acc, wire(0)));
    for (u32 i = 0; i < num_limbs; i++) {
        const V limb = wire(1 + i);
        V prod = limb;
        for (u32 b = 1; b < base; b++) prod = A::mul(prod, A::sub(limb, A::cst(F::enc(b))));
        emit(prod);
    }
}

// ReducingGate (base-field coefficients, one wire each) / ReducingExtensionGate (D-tuple coefficients)
template <class F, class A, class W, class Emit>
GB_HD void eval_reducing(u32 num_coeffs, bool extension_coeffs, W&& wire, Emit&& emit) {
    typedef Tup<F, A> X;
    constexpr u32 D = F::D;
    const X alpha = X::load(wire, D);
    X acc = X::load(wire, 2 * D);
    const u32 start_coeffs = 3 * D, start_accs = start_coeffs + num_coeffs * (extension_coeffs ? D : 1);
    for (u32 i = 0; i < num_coeffs; i++) {
        const X coeff = extension_coeffs ? X::load(wire, start_coeffs + i * D) : X::from_base(wire(start_coeffs + i));
        const X acc_i = X::load(wire, i == num_coeffs - 1 ? 0 : start_accs + D * i);  // the last accumulator is the output
        (acc * alpha + coeff - acc_i).emit_all(emit);
        acc = acc_i;
    }
}

template <class F, class A, class W, class K, class Emit>
GB_HD void eval_random_access(u32 bits, u32 num_copies, u32 num_extra, W&& wire, K&& konst, Emit&& emit) {
    typedef typename A::V V;
    const u32 vec = 1u << bits, routed = (2 + vec) * num_copies + num_extra;
    const V one = A::cst(F::one());
    for (u32 copy = 0; copy < num_copies; copy++) {
        const u32 base = (2 + vec) * copy, bit0 = routed + copy * bits;
        for (u32 i = 0; i < bits; i++) {
            const V b = wire(bit0 + i);
            emit(A::mul(b, A::sub(b, one)));
        }
        V rec = A::cst(F::zero());
        for (u32 i = bits; i-- > 0;) rec = A::add(A::add(rec, rec), wire(bit0 + i));
        emit(A::sub(rec, wire(base)));
        // fold the list pairwise, level k selecting with bit k (x + b (y - x)); the items are visited once, in order, with one
        // pending value per level instead of the reference's shrinking vector
        V pending[MAX_RANDOM_ACCESS_BITS + 1];
        V cur = A::cst(F::zero());
        for (u32 i = 0; i < vec; i++) {
            cur = wire(base + 2 + i);
            u32 lvl = 0;
            for (u32 t = i; t & 1; t >>= 1, lvl++) {
                const V x = pending[lvl];
                cur = A::add(x, A::mul(wire(bit0 + lvl), A::sub(cur, x)));
            }
            pending[lvl] = cur;
        }
        emit(A::sub(cur, wire(base + 1)));
    }
    for (u32 i = 0; i < num_extra; i++) emit(A::sub(konst(i), wire((2 + vec) * num_copies + i)));
}

template <class F, class A, class W, class Emit>
GB_HD void eval_poseidon_mds(W&& wire, Emit&& emit) {
    typedef Tup<F, A> X;
    constexpr u32 D = F::D;
    const PoseidonTab& t = poseidon_tab();
    // the inputs are read where they are used (a degree-1 gate: twelve live D-tuples would only cost registers)
#pragma unroll 1
    for (u32 r = 0; r < 12; r++) {
        X res = X::load(wire, r * D).scalar_c(F::enc(t.circ[0] + t.diag[r]));
#pragma unroll 1
        for (u32 i = 1; i < 12; i++) {
            const u32 j = r + i >= 12 ? r + i - 12 : r + i;
            res = res + X::load(wire, j * D).scalar_c(F::enc(t.circ[i]));
        }
        (X::load(wire, (12 + r) * D) - res).emit_all(emit);
    }
}

template <class F, class A, class W, class Emit>
GB_HD void eval_coset_interpolation(const GateSet& gs, const gb_gate& g, W&& wire, Emit&& emit) {
    typedef Tup<F, A> X;
    typedef typename F::T T;
    constexpr u32 D = F::D;
    const u32 bits = g.param, degree = g.param2, npts = 1u << bits, nint = interpolation_intermediates(g);
    const u32 start_point = 1 + npts * D, start_int = start_point + 2 * D;
    const X shifted = X::load(wire, start_int + 2 * D * nint);
    (X::load(wire, start_point) - shifted.scalar(wire(0))).emit_all(emit);
    const T inv_m = (T)gs.inv_pow2[bits];
    X ev = X::zero(), prod = X::from_base(A::cst(F::one()));
    auto partial = [&](u32 lo, u32 hi) {  // partial_interpolate_ext_algebra (:637-664)
        for (u32 i = lo; i < hi; i++) {
            const T x_i = (T)gs.subgroup16[i << (MAX_INTERPOLATION_BITS - bits)];
            const X val = X::load(wire, 1 + i * D).scalar_c(F::mul(x_i, inv_m));  // barycentric weight x_i / 2^bits
            X term = shifted;
            term.c[0] = A::sub(term.c[0], A::cst(x_i));
            ev = ev * term + val * prod;
            prod = prod * term;
        }
    };
    partial(0, degree);
    for (u32 i = 0; i < nint; i++) {
        const X iev = X::load(wire, start_int + D * i), iprod = X::load(wire, start_int + D * (nint + i));
        (iev - ev).emit_all(emit);
        (iprod - prod).emit_all(emit);
        ev = iev;
        prod = iprod;
        const u32 lo = 1 + (degree - 1) * (i + 1), hi = lo + degree - 1 < npts ? lo + degree - 1 : npts;
        partial(lo, hi);
    }
    (X::load(wire, start_point + D) - ev).emit_all(emit);
}

template <class F, class A, class W, class Emit>
GB_HD void eval_exponentiation(u32 nbits, W&& wire, Emit&& emit) {
    typedef typename A::V V;
    const V base = wire(0), one = A::cst(F::one());
    V prev = one;
    for (u32 i = 0; i < nbits; i++) {
        const V bit = wire(1 + (nbits - i - 1));  // power bits are little-endian, the chain runs big-endian
        const V inter = wire(2 + nbits + i);
        emit(A::sub(A::mul(prev, A::add(A::mul(bit, base), A::sub(one, bit))), inter));
        prev = A::mul(inter, inter);
    }
    emit(A::sub(wire(1 + nbits), wire(2 + nbits + nbits - 1)));
}

template <class F, class A, class W, class Emit>
GB_HD void eval_add_many(u32 num_addends, u32 num_ops, W&& wire, Emit&& emit) {
    typedef typename A::V V;
    for (u32 i = 0; i < num_ops; i++) {
        const u32 base = (num_addends + 1) * i;
        V sum = A::cst(F::zero());
        for (u32 j = 0; j < num_addends; j++) sum = A::add(sum, wire(base + j));
        emit(A::sub(sum, wire(base + num_addends)));
    }
}

// the 4x4 MDS block of Poseidon2's external layer ([[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]) on four D-tuples
template <class F, class A, class W, class Emit>
GB_HD void eval_apply_mat4(u32 num_ops, W&& wire, Emit&& emit) {
    typedef Tup<F, A> X;
    constexpr u32 D = F::D;
    for (u32 op = 0; op < num_ops; op++) {
        const u32 base = op * 8 * D;
        const X x0 = X::load(wire, base), x1 = X::load(wire, base + D), x2 = X::load(wire, base + 2 * D), x3 = X::load(wire, base + 3 * D);
        const X t01 = x0 + x1, t23 = x2 + x3, t0123 = t01 + t23, t01123 = t0123 + x1, t01233 = t0123 + x3;
        (X::load(wire, base + 4 * D) - (t01123 + t01)).emit_all(emit);
        (X::load(wire, base + 5 * D) - (t01123 + (x2 + x2))).emit_all(emit);
        (X::load(wire, base + 6 * D) - (t01233 + t23)).emit_all(emit);
        (X::load(wire, base + 7 * D) - (t01233 + (x0 + x0))).emit_all(emit);
    }
}

// Poseidon2's internal linear layer M_I (hash/poseidon2_babybear.rs: 2^-32, then diag + all-ones) on sixteen D-tuples
template <class F, class A, class W, class Emit>
GB_HD void eval_poseidon2_internal_permutation(W&& wire, Emit&& emit) {
    typedef Tup<F, A> X;
    constexpr u32 D = F::D;
    constexpr u32 SHIFTS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15};
    const typename F::T k = F::enc(943718400u);
    X part = X::zero();
#pragma unroll 1
    for (u32 i = 1; i < 16; i++) part = part + X::load(wire, i * D).scalar_c(k);
    const X s0 = X::load(wire, 0).scalar_c(k), full = part + s0;
    (X::load(wire, 16 * D) - (part - s0)).emit_all(emit);
#pragma unroll 1
    for (u32 i = 0; i < 15; i++)
        (X::load(wire, (17 + i) * D) - (full + X::load(wire, (i + 1) * D).scalar_c(F::mul(k, F::enc((u64)1 << SHIFTS[i]))))).emit_all(emit);
}

// wire(col) / konst(i) give the opened (or LDE) value of a wire / of the i-th constant after the selectors
// The in-circuit hash gates carry the permutation's whole state through ~120 constraints (200+ VGPRs in the quotient kernel); the
// other gates are short.  The kernel evaluates the two families in separate launches (SUBSET) so that the light ones are not
// compiled - and run - at the heavy ones' register budget.
enum GateSubset { ALL_GATES = 0, HEAVY_GATES = 1, LIGHT_GATES = 2 };
GB_HD bool is_heavy(const gb_gate& g) { return g.kind == GB_GATE_POSEIDON || g.kind == GB_GATE_POSEIDON2_BABYBEAR; }

template <class F, class A, int SUBSET = ALL_GATES, class W, class K, class Emit>
GB_HD void eval_gate(const GateSet& gs, const gb_gate& g, W&& wire, K&& konst, const typename F::T* pi_hash, Emit&& emit) {
    typedef typename A::V V;
    if constexpr (SUBSET == HEAVY_GATES) {
        if (g.kind == GB_GATE_POSEIDON) {
            if constexpr (F::TAG == 0) eval_poseidon<A>(wire, emit);
        } else if (g.kind == GB_GATE_POSEIDON2_BABYBEAR) {
            if constexpr (F::TAG == 1) eval_poseidon2_bb<F, A>(g.param, wire, emit);
        }
        return;
    }
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
            if constexpr (F::TAG == 0 && SUBSET != LIGHT_GATES) eval_poseidon<A>(wire, emit);
            break;
        case GB_GATE_POSEIDON2_BABYBEAR:
            if constexpr (F::TAG == 1 && SUBSET != LIGHT_GATES) eval_poseidon2_bb<F, A>(g.param, wire, emit);
            break;
        case GB_GATE_ARITHMETIC_EXTENSION: eval_arithmetic_extension<F, A>(g.param, true, wire, konst, emit); break;
        case GB_GATE_MUL_EXTENSION: eval_arithmetic_extension<F, A>(g.param, false, wire, konst, emit); break;
        case GB_GATE_BASE_SUM: eval_base_sum<F, A>(g.param, base_sum_base(g), wire, emit); break;
        case GB_GATE_REDUCING: eval_reducing<F, A>(g.param, false, wire, emit); break;
        case GB_GATE_REDUCING_EXTENSION: eval_reducing<F, A>(g.param, true, wire, emit); break;
        case GB_GATE_RANDOM_ACCESS: eval_random_access<F, A>(g.param, g.param2, g.param3, wire, konst, emit); break;
        case GB_GATE_POSEIDON_MDS:
            if constexpr (F::TAG == 0) eval_poseidon_mds<F, A>(wire, emit);
            break;
        case GB_GATE_COSET_INTERPOLATION: eval_coset_interpolation<F, A>(gs, g, wire, emit); break;
        case GB_GATE_EXPONENTIATION: eval_exponentiation<F, A>(g.param, wire, emit); break;
        case GB_GATE_ADD_MANY: eval_add_many<F, A>(g.param, g.param2, wire, emit); break;
        case GB_GATE_APPLY_MAT4: eval_apply_mat4<F, A>(g.param, wire, emit); break;
        case GB_GATE_POSEIDON2_INTERNAL_PERMUTATION:
            if constexpr (F::TAG == 1) eval_poseidon2_internal_permutation<F, A>(wire, emit);
            break;
        default:
            break;
    }
}

}  // namespace gates
}  // namespace gbk
