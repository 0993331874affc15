"""TEST ORACLE - the gate constraint evaluators at one extension-field point (the verifier's side).

Test infrastructure only.  Follows eval_unfiltered of (paths relative to /root/reference/plonky2/src/gates):
  noop.rs, constant.rs:64-72, public_input.rs:52-60, arithmetic_base.rs:83-100, poseidon_goldilocks.rs:124-221,
  poseidon2_babybear.rs:203-313, arithmetic_extension.rs:82-100, multiplication_extension.rs:77-94, base_sum.rs:77-93,
  reducing.rs:89-115, reducing_extension.rs:95-120, random_access.rs:150-200, poseidon_goldilocks_mds.rs:152-180,
  coset_interpolation.rs:216-268 (+ partial_interpolate_ext_algebra :637-664), exponentiation.rs:99-135, add_many.rs:80-90,
  apply_mat4.rs:80-108, poseidon2_internal_permutation.rs:75-112
and compute_filter (gate.rs:391-404).  `e` is a Field of oracle/fields.py; extension elements are tuples.
A gate is the tuple the C oracle and the product ABI use: (kind, param, selector_index, group_start, group_end[, param2,
param3]); param2/param3 only for BaseSumGate (base B), RandomAccessGate (num_copies, num_extra_constants) and
CosetInterpolationGate (degree).

In the verifier the wires are extension elements and a D-tuple of consecutive wires is an element of the extension ALGEBRA
F_ext[x]/(x^D - W) (vars.get_local_ext_algebra, plonk/vars.rs); on the prover's LDE points the wires are base elements and the
same tuple is an element of the extension FIELD.  The evaluators below are written over the algebra, which covers both (a base
value is the extension element (v, 0, ..)).
"""
import os
import re

NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON, POSEIDON2_BABYBEAR = 0, 1, 2, 3, 4, 5
ARITHMETIC_EXTENSION, MUL_EXTENSION, BASE_SUM, REDUCING, REDUCING_EXTENSION = 6, 7, 8, 9, 10
RANDOM_ACCESS, POSEIDON_MDS, COSET_INTERPOLATION, EXPONENTIATION = 11, 12, 13, 14
ADD_MANY, APPLY_MAT4, POSEIDON2_INTERNAL_PERMUTATION = 15, 16, 17
UNUSED_SELECTOR = 0xFFFFFFFF  # selectors.rs:13
POSEIDON_NUM_CONSTRAINTS = 12 * 7 + 22 + 12 + 1 + 4
POSEIDON2_BB_CONSTRAINTS_PER_OP = 1 + 8 + 16 * 7 + 13 + 16

_TABLES = None


def poseidon_tables():
    """The Poseidon-12 tables of oracle/poseidon_constants.h (hash/poseidon_goldilocks.rs:114-492)."""
    global _TABLES
    if _TABLES is None:
        text = open(os.path.join(os.path.dirname(__file__), "poseidon_constants.h")).read()
        _TABLES = {}
        for m in re.finditer(r"#define GL_POSEIDON_(\w+)_LIST \\\n((?:[^\n]*\\\n)*[^\n]*)", text):
            _TABLES[m.group(1)] = [int(x.rstrip("uUlL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", m.group(2))]
    return _TABLES


def _p(gate, i, default=0):
    return gate[i] if len(gate) > i else default


def num_intermediates(gate):
    """CosetInterpolationGate::num_intermediates (coset_interpolation.rs:148-150)"""
    return ((1 << gate[1]) - 2) // (_p(gate, 5) - 1)


def num_constraints(gate, hout, D=2):
    kind, param = gate[0], gate[1]
    if kind == RANDOM_ACCESS:
        return (param + 2) * _p(gate, 5) + _p(gate, 6)
    if kind == COSET_INTERPOLATION:
        return 2 * D + 2 * D * num_intermediates(gate)
    if kind == ADD_MANY:
        return _p(gate, 5)
    return {APPLY_MAT4: 4 * D * param, POSEIDON2_INTERNAL_PERMUTATION: 16 * D,
            NOOP: 0, CONSTANT: param, PUBLIC_INPUT: hout, ARITHMETIC: param, POSEIDON: POSEIDON_NUM_CONSTRAINTS,
            POSEIDON2_BABYBEAR: POSEIDON2_BB_CONSTRAINTS_PER_OP * param, ARITHMETIC_EXTENSION: D * param,
            MUL_EXTENSION: D * param, BASE_SUM: 1 + param, REDUCING: D * param, REDUCING_EXTENSION: D * param,
            POSEIDON_MDS: 12 * D, EXPONENTIATION: param + 1}[kind]


def num_gate_constants(gate):
    """Gate::num_constants"""
    kind = gate[0]
    if kind == CONSTANT:
        return gate[1]
    if kind in (ARITHMETIC, ARITHMETIC_EXTENSION):
        return 2
    if kind == MUL_EXTENSION:
        return 1
    if kind == RANDOM_ACCESS:
        return _p(gate, 6)
    return 0


def gate_degree(gate):
    """Gate::degree"""
    kind = gate[0]
    return {NOOP: 0, CONSTANT: 1, PUBLIC_INPUT: 1, ARITHMETIC: 3, POSEIDON: 7, POSEIDON2_BABYBEAR: 7, ARITHMETIC_EXTENSION: 3,
            MUL_EXTENSION: 3, BASE_SUM: _p(gate, 5, 2), REDUCING: 2, REDUCING_EXTENSION: 2, RANDOM_ACCESS: gate[1] + 1,
            POSEIDON_MDS: 1, COSET_INTERPOLATION: _p(gate, 5), EXPONENTIATION: 4, ADD_MANY: 1, APPLY_MAT4: 1,
            POSEIDON2_INTERNAL_PERMUTATION: 1}[kind]


def barycentric_weights(e, subgroup_bits):
    """field/src/interpolation.rs:56-69 on two_adic_subgroup(subgroup_bits) (field/src/types.rs:14-17)"""
    P = e.P
    g = e.two_adic_generator(subgroup_bits)
    xs = [pow(g, i, P) for i in range(1 << subgroup_bits)]
    out = []
    for i, xi in enumerate(xs):
        d = 1
        for j, xj in enumerate(xs):
            if j != i:
                d = d * (xi - xj) % P
        out.append(pow(d, P - 2, P))
    return xs, out


def compute_filter(e, row, gate, s, many_selectors):
    f = e.one
    for i in range(gate[3], gate[4]):
        if i != row:
            f = e.emul(f, e.esub(e.efrom(i), s))
    if many_selectors:
        f = e.emul(f, e.esub(e.efrom(UNUSED_SELECTOR % e.P), s))
    return f


def _poseidon(e, w):
    T = poseidon_tables()
    rc, circ, diag = T["ALL_ROUND_CONSTANTS"], T["MDS_CIRC"], T["MDS_DIAG"]
    WIRE_SWAP, START_DELTA, START_FULL_0 = 24, 25, 29
    START_PARTIAL = START_FULL_0 + 36
    START_FULL_1 = START_PARTIAL + 22
    add, sub, mul, sc, k = e.eadd, e.esub, e.emul, e.escale, e.efrom
    out = []
    swap = w[WIRE_SWAP]
    out.append(mul(swap, sub(swap, e.one)))
    for i in range(4):
        out.append(sub(mul(swap, sub(w[i + 4], w[i])), w[START_DELTA + i]))
    s = [None] * 12
    for i in range(4):
        s[i] = add(w[i], w[START_DELTA + i])
        s[i + 4] = sub(w[i + 4], w[START_DELTA + i])
    for i in range(8, 12):
        s[i] = w[i]

    def sbox(x):
        x2 = mul(x, x)
        return mul(mul(x, x2), mul(x2, x2))

    def mds(v):
        res = []
        for r in range(12):
            acc = e.zero
            for i in range(12):
                acc = add(acc, sc(v[(i + r) % 12], circ[i]))
            res.append(add(acc, sc(v[r], diag[r])))
        return res

    ctr = 0
    for r in range(4):
        s = [add(s[i], k(rc[12 * ctr + i] % e.P)) for i in range(12)]
        if r:
            for i in range(12):
                sin = w[START_FULL_0 + 12 * (r - 1) + i]
                out.append(sub(s[i], sin))
                s[i] = sin
        s = mds([sbox(x) for x in s])
        ctr += 1
    s = [add(s[i], k(T["FAST_PARTIAL_FIRST_ROUND_CONSTANT"][i] % e.P)) for i in range(12)]
    init = T["FAST_PARTIAL_ROUND_INITIAL_MATRIX"]
    res = [s[0]] + [e.zero] * 11
    for r in range(1, 12):
        for c in range(1, 12):
            res[c] = add(res[c], sc(s[r], init[(r - 1) * 11 + (c - 1)] % e.P))
    s = res
    for r in range(22):
        sin = w[START_PARTIAL + r]
        out.append(sub(s[0], sin))
        s[0] = sbox(sin)
        if r != 21:
            s[0] = add(s[0], k(T["FAST_PARTIAL_ROUND_CONSTANTS"][r] % e.P))
        d = sc(s[0], circ[0] + diag[0])
        for i in range(1, 12):
            d = add(d, sc(s[i], T["FAST_PARTIAL_ROUND_W_HATS"][r * 11 + i - 1] % e.P))
        s = [d] + [add(sc(s[0], T["FAST_PARTIAL_ROUND_VS"][r * 11 + i - 1] % e.P), s[i]) for i in range(1, 12)]
    ctr += 22
    for r in range(4):
        s = [add(s[i], k(rc[12 * ctr + i] % e.P)) for i in range(12)]
        for i in range(12):
            sin = w[START_FULL_1 + 12 * r + i]
            out.append(sub(s[i], sin))
            s[i] = sin
        s = mds([sbox(x) for x in s])
        ctr += 1
    for i in range(12):
        out.append(sub(s[i], w[12 + i]))
    return out


def _bb_tables():
    text = open(os.path.join(os.path.dirname(__file__), "poseidon_constants.h")).read()
    out = {}
    for m in re.finditer(r"#define BB_POSEIDON2_(\w+)_LIST \\\n((?:[^\n]*\\\n)*[^\n]*)", text):
        out[m.group(1)] = [int(x.rstrip("uUlL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", m.group(2))]
    return out


def _poseidon2_bb(e, w, num_ops):
    T = _bb_tables()
    ext, internal = T["EXTERNAL_CONSTANTS"], T["INTERNAL_CONSTANTS"]
    shifts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
    add, sub, mul, sc, k = e.eadd, e.esub, e.emul, e.escale, e.efrom

    def sbox(x):
        x2 = mul(x, x)
        return mul(mul(x, x2), mul(x2, x2))

    def external(s):
        s = list(s)
        for i in range(0, 16, 4):
            x = s[i:i + 4]
            t01, t23 = add(x[0], x[1]), add(x[2], x[3])
            t0123 = add(t01, t23)
            t01123, t01233 = add(t0123, x[1]), add(t0123, x[3])
            s[i:i + 4] = [add(t01123, t01), add(t01123, add(x[2], x[2])), add(t01233, t23), add(t01233, add(x[0], x[0]))]
        sums = [e.zero] * 4
        for j in range(16):
            sums[j % 4] = add(sums[j % 4], s[j])
        return [add(s[i], sums[i % 4]) for i in range(16)]

    def internal_layer(s):
        s = [sc(x, 943718400) for x in s]
        part = e.zero
        for x in s[1:]:
            part = add(part, x)
        full = add(part, s[0])
        return [sub(part, s[0])] + [add(full, sc(s[i + 1], 1 << shifts[i])) for i in range(15)]

    out = []
    for op in range(num_ops):
        in0, start_delta = 33 * op, num_ops * 33 + op * 133
        out0, start_full_0 = in0 + 16, start_delta + 8
        start_partial = start_full_0 + 48
        start_full_1 = start_partial + 13
        swap = w[in0 + 32]
        out.append(mul(swap, sub(swap, e.one)))
        for i in range(8):
            out.append(sub(mul(swap, sub(w[in0 + i + 8], w[in0 + i])), w[start_delta + i]))
        s = [add(w[in0 + i], w[start_delta + i]) for i in range(8)] + [sub(w[in0 + i + 8], w[start_delta + i]) for i in range(8)]
        s = external(s)
        for r in range(4):
            s = [add(s[i], k(ext[16 * r + i])) for i in range(16)]
            if r:
                for i in range(16):
                    sin = w[start_full_0 + 16 * (r - 1) + i]
                    out.append(sub(s[i], sin))
                    s[i] = sin
            s = external([sbox(x) for x in s])
        for r in range(13):
            s[0] = add(s[0], k(internal[r]))
            sin = w[start_partial + r]
            out.append(sub(s[0], sin))
            s[0] = sbox(sin)
            s = internal_layer(s)
        for r in range(4, 8):
            s = [add(s[i], k(ext[16 * r + i])) for i in range(16)]
            for i in range(16):
                sin = w[start_full_1 + 16 * (r - 4) + i]
                out.append(sub(s[i], sin))
                s[i] = sin
            s = external([sbox(x) for x in s])
        for i in range(16):
            out.append(sub(s[i], w[out0 + i]))
    return out


# ---- the extension algebra F_ext[x]/(x^D - W): lists of D extension elements (field/src/extension/algebra.rs)
def _alg(w, start, D):
    return list(w[start:start + D])


def _alg_add(e, a, b):
    return [e.eadd(x, y) for x, y in zip(a, b)]


def _alg_sub(e, a, b):
    return [e.esub(x, y) for x, y in zip(a, b)]


def _alg_scalar(e, a, s):
    """ExtensionAlgebra::scalar_mul: every coordinate times the extension element s"""
    return [e.emul(x, s) for x in a]


def _alg_mul(e, a, b):
    D = e.D
    r = [e.zero] * D
    for i in range(D):
        for j in range(D):
            t = e.emul(a[i], b[j])
            if i + j >= D:
                t = e.escale(t, e.W)
            r[(i + j) % D] = e.eadd(r[(i + j) % D], t)
    return r


def _alg_from_base(e, x):
    return [x] + [e.zero] * (e.D - 1)


def _arithmetic_extension(e, w, c, num_ops):
    D, out = e.D, []
    for i in range(num_ops):
        m0, m1, ad, o = (_alg(w, 4 * D * i + k * D, D) for k in range(4))
        comp = _alg_add(e, _alg_scalar(e, _alg_mul(e, m0, m1), c[0]), _alg_scalar(e, ad, c[1]))
        out += _alg_sub(e, o, comp)
    return out


def _mul_extension(e, w, c, num_ops):
    D, out = e.D, []
    for i in range(num_ops):
        m0, m1, o = (_alg(w, 3 * D * i + k * D, D) for k in range(3))
        out += _alg_sub(e, o, _alg_scalar(e, _alg_mul(e, m0, m1), c[0]))
    return out


def _base_sum(e, w, num_limbs, B):
    limbs = w[1:1 + num_limbs]
    acc = e.zero
    for limb in reversed(limbs):   # reduce_with_powers (plonk/plonk_common.rs)
        acc = e.eadd(e.escale(acc, B), limb)
    out = [e.esub(acc, w[0])]
    for limb in limbs:
        prod = e.one
        for i in range(B):
            prod = e.emul(prod, e.esub(limb, e.efrom(i)))
        out.append(prod)
    return out


def _reducing(e, w, num_coeffs, extension_coeffs):
    D = e.D
    alpha, acc = _alg(w, D, D), _alg(w, 2 * D, D)
    start_coeffs = 3 * D
    start_accs = start_coeffs + num_coeffs * (D if extension_coeffs else 1)
    out = []
    for i in range(num_coeffs):
        coeff = _alg(w, start_coeffs + i * D, D) if extension_coeffs else _alg_from_base(e, w[start_coeffs + i])
        acc_i = _alg(w, 0, D) if i == num_coeffs - 1 else _alg(w, start_accs + D * i, D)
        out += _alg_sub(e, _alg_add(e, _alg_mul(e, acc, alpha), coeff), acc_i)
        acc = acc_i
    return out


def _random_access(e, w, c, bits, num_copies, num_extra):
    vec = 1 << bits
    routed = (2 + vec) * num_copies + num_extra
    out = []
    for copy in range(num_copies):
        base = (2 + vec) * copy
        access_index, claimed = w[base], w[base + 1]
        items = [w[base + 2 + i] for i in range(vec)]
        bs = [w[routed + copy * bits + i] for i in range(bits)]
        for b in bs:
            out.append(e.emul(b, e.esub(b, e.one)))
        rec = e.zero
        for b in reversed(bs):
            rec = e.eadd(e.eadd(rec, rec), b)
        out.append(e.esub(rec, access_index))
        for b in bs:
            items = [e.eadd(x, e.emul(b, e.esub(y, x))) for x, y in zip(items[0::2], items[1::2])]
        out.append(e.esub(items[0], claimed))
    for i in range(num_extra):
        out.append(e.esub(c[i], w[(2 + vec) * num_copies + i]))
    return out


def _poseidon_mds(e, w):
    T = poseidon_tables()
    circ, diag = T["MDS_CIRC"], T["MDS_DIAG"]
    D = e.D
    ins = [_alg(w, i * D, D) for i in range(12)]
    out = []
    for r in range(12):
        res = [e.zero] * D
        for i in range(12):
            res = _alg_add(e, res, [e.escale(x, circ[i]) for x in ins[(i + r) % 12]])
        res = _alg_add(e, res, [e.escale(x, diag[r]) for x in ins[r]])
        out += _alg_sub(e, _alg(w, (12 + r) * D, D), res)
    return out


def _coset_interpolation(e, w, subgroup_bits, degree):
    D = e.D
    npts = 1 << subgroup_bits
    nint = (npts - 2) // (degree - 1)
    domain, weights = barycentric_weights(e, subgroup_bits)
    start_eval_point = 1 + npts * D
    start_int = start_eval_point + 2 * D
    shift = w[0]
    point = _alg(w, start_eval_point, D)
    shifted = _alg(w, start_int + 2 * D * nint, D)
    out = _alg_sub(e, point, _alg_scalar(e, shifted, shift))
    values = [_alg(w, 1 + i * D, D) for i in range(npts)]

    def partial(lo, hi, ev, prod):
        for i in range(lo, hi):
            val = [e.escale(x, weights[i]) for x in values[i]]
            term = _alg_sub(e, shifted, _alg_from_base(e, e.efrom(domain[i])))
            ev = _alg_add(e, _alg_mul(e, ev, term), _alg_mul(e, val, prod))
            prod = _alg_mul(e, prod, term)
        return ev, prod

    ev, prod = partial(0, degree, [e.zero] * D, _alg_from_base(e, e.one))
    for i in range(nint):
        iev, iprod = _alg(w, start_int + D * i, D), _alg(w, start_int + D * (nint + i), D)
        out += _alg_sub(e, iev, ev)
        out += _alg_sub(e, iprod, prod)
        lo = 1 + (degree - 1) * (i + 1)
        ev, prod = partial(lo, min(lo + degree - 1, npts), iev, iprod)
    out += _alg_sub(e, _alg(w, start_eval_point + D, D), ev)
    return out


def _exponentiation(e, w, nbits):
    base, output = w[0], w[1 + nbits]
    inter = [w[2 + nbits + i] for i in range(nbits)]
    out = []
    for i in range(nbits):
        prev = e.one if i == 0 else e.emul(inter[i - 1], inter[i - 1])
        bit = w[1 + (nbits - i - 1)]
        out.append(e.esub(e.emul(prev, e.eadd(e.emul(bit, base), e.esub(e.one, bit))), inter[i]))
    out.append(e.esub(output, inter[nbits - 1]))
    return out


def _add_many(e, w, num_addends, num_ops):
    out = []
    for i in range(num_ops):
        base, acc = (num_addends + 1) * i, e.zero
        for j in range(num_addends):
            acc = e.eadd(acc, w[base + j])
        out.append(e.esub(acc, w[base + num_addends]))
    return out


def _apply_mat4(e, w, num_ops):
    D, out = e.D, []
    for op in range(num_ops):
        x = [_alg(w, op * 8 * D + i * D, D) for i in range(4)]
        t01, t23 = _alg_add(e, x[0], x[1]), _alg_add(e, x[2], x[3])
        t0123 = _alg_add(e, t01, t23)
        t01123, t01233 = _alg_add(e, t0123, x[1]), _alg_add(e, t0123, x[3])
        new = [_alg_add(e, t01123, t01), _alg_add(e, t01123, _alg_add(e, x[2], x[2])), _alg_add(e, t01233, t23),
               _alg_add(e, t01233, _alg_add(e, x[0], x[0]))]
        for i in range(4):
            out += _alg_sub(e, _alg(w, op * 8 * D + (4 + i) * D, D), new[i])
    return out


def _poseidon2_internal_permutation(e, w):
    D = e.D
    shifts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
    s = [[e.escale(x, 943718400) for x in _alg(w, i * D, D)] for i in range(16)]
    part = [e.zero] * D
    for x in s[1:]:
        part = _alg_add(e, part, x)
    full = _alg_add(e, part, s[0])
    new = [_alg_sub(e, part, s[0])] + [_alg_add(e, full, [e.escale(x, 1 << shifts[i]) for x in s[i + 1]]) for i in range(15)]
    out = []
    for i in range(16):
        out += _alg_sub(e, _alg(w, (16 + i) * D, D), new[i])
    return out


def eval_unfiltered(e, gate, wires, consts, pi_hash):
    """consts = local_constants after the selectors (vars.remove_prefix, gate.rs:165-186)"""
    kind, param = gate[0], gate[1]
    if kind == NOOP:
        return []
    if kind == CONSTANT:
        return [e.esub(consts[i], wires[i]) for i in range(param)]
    if kind == PUBLIC_INPUT:
        return [e.esub(wires[i], e.efrom(int(pi_hash[i]))) for i in range(e.hout)]
    if kind == ARITHMETIC:
        c0, c1 = consts[0], consts[1]
        return [e.esub(wires[4 * i + 3], e.eadd(e.emul(e.emul(wires[4 * i], wires[4 * i + 1]), c0), e.emul(wires[4 * i + 2], c1)))
                for i in range(param)]
    if kind == POSEIDON:
        assert e.name == "goldilocks", "PoseidonGate is the Goldilocks gate"
        return _poseidon(e, wires)
    if kind == POSEIDON2_BABYBEAR:
        assert e.name == "babybear", "Poseidon2BabyBearGate is the BabyBear gate"
        return _poseidon2_bb(e, wires, param)
    if kind == ARITHMETIC_EXTENSION:
        return _arithmetic_extension(e, wires, consts, param)
    if kind == MUL_EXTENSION:
        return _mul_extension(e, wires, consts, param)
    if kind == BASE_SUM:
        return _base_sum(e, wires, param, _p(gate, 5, 2))
    if kind == REDUCING:
        return _reducing(e, wires, param, False)
    if kind == REDUCING_EXTENSION:
        return _reducing(e, wires, param, True)
    if kind == RANDOM_ACCESS:
        return _random_access(e, wires, consts, param, _p(gate, 5), _p(gate, 6))
    if kind == POSEIDON_MDS:
        assert e.name == "goldilocks", "PoseidonMdsGate is the Goldilocks gate"
        return _poseidon_mds(e, wires)
    if kind == COSET_INTERPOLATION:
        return _coset_interpolation(e, wires, param, _p(gate, 5))
    if kind == EXPONENTIATION:
        return _exponentiation(e, wires, param)
    if kind == ADD_MANY:
        return _add_many(e, wires, param, _p(gate, 5))
    if kind == APPLY_MAT4:
        return _apply_mat4(e, wires, param)
    if kind == POSEIDON2_INTERNAL_PERMUTATION:
        assert e.name == "babybear", "Poseidon2InternalPermutationGate is the BabyBear gate"
        return _poseidon2_internal_permutation(e, wires)
    raise ValueError("gate kind %r" % (kind,))
