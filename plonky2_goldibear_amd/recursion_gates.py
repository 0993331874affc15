"""Host-side mirror of the remaining gates of the reference's recursion circuits - the gate structs with their wire layouts
and witness generators - for CircuitBuilder.add_gate():

    ArithmeticExtensionGate  gates/arithmetic_extension.rs     MulExtensionGate   gates/multiplication_extension.rs
    BaseSumGate<B>           gates/base_sum.rs                 ReducingGate       gates/reducing.rs
    ReducingExtensionGate    gates/reducing_extension.rs       RandomAccessGate   gates/random_access.rs
    PoseidonMdsGate          gates/poseidon_goldilocks_mds.rs  CosetInterpolationGate  gates/coset_interpolation.rs
    ExponentiationGate       gates/exponentiation.rs           AddManyGate        gates/add_many.rs
    ApplyMat4Gate            gates/apply_mat4.rs               Poseidon2InternalPermutationGate  gates/poseidon2_internal_permutation.rs

The constraint evaluators are in csrc/gates.hpp (one source for the quotient kernel and gb_verify); what is here is what a
Rust host already has: the layouts and the SimpleGenerators that fill a row's dependent wires from its inputs.  A row is used
as `row = builder.add_gate(gate, constants)`, inputs set with `pw.set_target(wire(row, column), value)`.
Extension elements are D-tuples of Python integers.  Not the GPU hot path.
"""
from . import native as N
from .circuit_builder import Gate, _poseidon_tables, wire

(GATE_ARITHMETIC_EXTENSION, GATE_MUL_EXTENSION, GATE_BASE_SUM, GATE_REDUCING, GATE_REDUCING_EXTENSION, GATE_RANDOM_ACCESS,
 GATE_POSEIDON_MDS, GATE_COSET_INTERPOLATION, GATE_EXPONENTIATION, GATE_ADD_MANY, GATE_APPLY_MAT4,
 GATE_POSEIDON2_INTERNAL_PERMUTATION) = range(6, 18)  # gb_gate.kind

_GL_NAME = "p3_goldilocks::goldilocks::Goldilocks"
_BB_NAME = "p3_monty_31::monty_31::MontyField31<p3_baby_bear::baby_bear::BabyBearParameters>"


class Ext:
    """BinomialExtensionField<F, D>: F[x]/(x^D - W) (field/src/types.rs:19-29); Goldilocks D = 2, W = 7; BabyBear D = 4, W = 11"""

    def __init__(self, field):
        gl = field == N.GB_GOLDILOCKS
        self.field = field
        self.p, self.D, self.W = (0xFFFFFFFF00000001, 2, 7) if gl else (2013265921, 4, 11)
        self.two_adic = (1753635133440165772, 32) if gl else (0x1a427a41, 27)
        self.name = _GL_NAME if gl else _BB_NAME
        self.zero, self.one = (0,) * self.D, (1,) + (0,) * (self.D - 1)

    def from_base(self, x):
        return (x % self.p,) + (0,) * (self.D - 1)

    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple((x - y) % self.p for x, y in zip(a, b))

    def scale(self, a, s):
        return tuple(x * s % self.p for x in a)

    def mul(self, a, b):
        D, r = self.D, [0] * (2 * self.D - 1)
        for i in range(D):
            for j in range(D):
                r[i + j] += a[i] * b[j]
        for k in range(2 * D - 2, D - 1, -1):
            r[k - D] += self.W * r[k]
        return tuple(x % self.p for x in r[:D])

    def inv(self, a):
        r, b, e = self.one, a, self.p ** self.D - 2
        while e:
            if e & 1:
                r = self.mul(r, b)
            b = self.mul(b, b)
            e >>= 1
        return r

    def subgroup(self, bits):
        g = pow(self.two_adic[0], 1 << (self.two_adic[1] - bits), self.p)
        return [pow(g, i, self.p) for i in range(1 << bits)]


def _ext_wires(row, start, D):
    return [wire(row, start + k) for k in range(D)]


def _get_ext(w, row, start, D):
    return tuple(w.get(wire(row, start + k)) for k in range(D))


def _set_ext(w, row, start, v):
    for k, x in enumerate(v):
        w.set(wire(row, start + k), x)


# --------------------------------------------------------------------------------------------- arithmetic on extension targets
class ArithmeticExtensionGate(Gate):
    """gates/arithmetic_extension.rs:27-58: num_ops x (out = c0 * m0 * m1 + c1 * addend) on D-tuples at 4 D i .."""
    kind, degree, num_constants = GATE_ARITHMETIC_EXTENSION, 3, 2
    OPERANDS = 4
    NAME = "ArithmeticExtensionGate"

    def __init__(self, num_ops, field=N.GB_GOLDILOCKS):
        self.E = Ext(field)
        self.param = self.num_ops = num_ops
        self.num_wires = self.OPERANDS * self.E.D * num_ops
        self.num_constraints = self.E.D * num_ops
        self.id = "%s { num_ops: %d }" % (self.NAME, num_ops)

    @classmethod
    def new_from_config(cls, cfg):
        return cls(cfg.num_routed_wires // (cls.OPERANDS * Ext(cfg.field).D), cfg.field)

    def generators(self, row, constants):
        return [_ArithmeticExtensionGenerator(self, row, constants, i) for i in range(self.num_ops)]


class MulExtensionGate(ArithmeticExtensionGate):
    """gates/multiplication_extension.rs:27-53: num_ops x (out = c0 * m0 * m1) on D-tuples at 3 D i .."""
    kind, num_constants = GATE_MUL_EXTENSION, 1
    OPERANDS = 3
    NAME = "MulExtensionGate"


class _ArithmeticExtensionGenerator:
    """arithmetic_extension.rs:186-233 / multiplication_extension.rs:160-200"""

    def __init__(self, gate, row, constants, i):
        self.g, self.row, self.c, self.i = gate, row, constants, i
        D, k = gate.E.D, gate.OPERANDS
        self.deps = [t for j in range(k - 1) for t in _ext_wires(row, k * D * i + j * D, D)]

    def run(self, w, p):
        E, k, D = self.g.E, self.g.OPERANDS, self.g.E.D
        base = k * D * self.i
        m0, m1 = _get_ext(w, self.row, base, D), _get_ext(w, self.row, base + D, D)
        out = E.scale(E.mul(m0, m1), self.c[0])
        if k == 4:
            out = E.add(out, E.scale(_get_ext(w, self.row, base + 2 * D, D), self.c[1]))
        _set_ext(w, self.row, base + (k - 1) * D, out)


# --------------------------------------------------------------------------------------------- base-B decomposition
class BaseSumGate(Gate):
    """gates/base_sum.rs:27-50: wire 0 = sum, wires 1.. = num_limbs little-endian base-B limbs"""
    kind = GATE_BASE_SUM

    def __init__(self, num_limbs, base=2):
        self.param, self.param2, self.degree = num_limbs, base, base
        self.num_limbs, self.base = num_limbs, base
        self.num_wires, self.num_constraints = 1 + num_limbs, 1 + num_limbs
        self.id = "BaseSumGate { num_limbs: %d } + Base: %d" % (num_limbs, base)

    def generators(self, row, constants):
        return [_BaseSplitGenerator(self, row)]


class _BaseSplitGenerator:
    """base_sum.rs:178-225"""

    def __init__(self, gate, row):
        self.g, self.row, self.deps = gate, row, [wire(row, 0)]

    def run(self, w, p):
        v = w.get(wire(self.row, 0))
        for i in range(self.g.num_limbs):
            w.set(wire(self.row, 1 + i), v % self.g.base)
            v //= self.g.base
        assert v == 0, "Integer too large to fit in given number of limbs"


# --------------------------------------------------------------------------------------------- sum alpha^i c_i
class ReducingGate(Gate):
    """gates/reducing.rs:27-64: output 0..D, alpha D..2D, old_acc 2D..3D, num_coeffs base-field coefficients, then the
    intermediate accumulators (the last one is the output)"""
    kind, degree = GATE_REDUCING, 2
    EXTENSION_COEFFS = False
    NAME = "ReducingGate"

    def __init__(self, num_coeffs, field=N.GB_GOLDILOCKS):
        self.E = Ext(field)
        D = self.E.D
        self.param = self.num_coeffs = num_coeffs
        self.coeff_width = D if self.EXTENSION_COEFFS else 1
        self.start_coeffs = 3 * D
        self.start_accs = self.start_coeffs + num_coeffs * self.coeff_width
        self.num_wires = self.start_accs + D * (num_coeffs - 1)
        self.num_constraints = D * num_coeffs
        self.id = "%s { num_coeffs: %d }" % (self.NAME, num_coeffs)

    def wires_acc(self, i):
        return 0 if i == self.num_coeffs - 1 else self.start_accs + self.E.D * i

    def generators(self, row, constants):
        return [_ReducingGenerator(self, row)]


class ReducingExtensionGate(ReducingGate):
    """gates/reducing_extension.rs:27-66: the same with D-tuple coefficients"""
    kind = GATE_REDUCING_EXTENSION
    EXTENSION_COEFFS = True
    NAME = "ReducingExtensionGate"


class _ReducingGenerator:
    """reducing.rs:205-250 / reducing_extension.rs:203-245"""

    def __init__(self, gate, row):
        self.g, self.row = gate, row
        D = gate.E.D
        self.deps = _ext_wires(row, D, 2 * D) + [wire(row, gate.start_coeffs + k) for k in range(gate.num_coeffs * gate.coeff_width)]

    def run(self, w, p):
        g, E, row = self.g, self.g.E, self.row
        D = E.D
        alpha, acc = _get_ext(w, row, D, D), _get_ext(w, row, 2 * D, D)
        for i in range(g.num_coeffs):
            if g.EXTENSION_COEFFS:
                c = _get_ext(w, row, g.start_coeffs + i * D, D)
            else:
                c = E.from_base(w.get(wire(row, g.start_coeffs + i)))
            acc = E.add(E.mul(acc, alpha), c)
            _set_ext(w, row, g.wires_acc(i), acc)


# --------------------------------------------------------------------------------------------- list[index]
class RandomAccessGate(Gate):
    """gates/random_access.rs:32-117: per copy access_index, claimed_element, 2^bits list items (routed); then the extra
    constants (routed); then per copy the index bits (not routed)"""
    kind = GATE_RANDOM_ACCESS

    def __init__(self, bits, num_copies, num_extra_constants, field=N.GB_GOLDILOCKS):
        E = Ext(field)
        self.bits, self.num_copies, self.num_extra_constants = bits, num_copies, num_extra_constants
        self.param, self.param2, self.param3 = bits, num_copies, num_extra_constants
        self.vec_size = 1 << bits
        self.degree = bits + 1
        self.num_constants = num_extra_constants
        self.num_routed = (2 + self.vec_size) * num_copies + num_extra_constants
        self.num_wires = self.num_routed + bits * num_copies
        self.num_constraints = (bits + 2) * num_copies + num_extra_constants
        self.id = ("RandomAccessGate { bits: %d, num_copies: %d, num_extra_constants: %d, _phantom: PhantomData<%s> }<D=%d>"
                   % (bits, num_copies, num_extra_constants, E.name, E.D))

    @classmethod
    def new_from_config(cls, cfg, bits):
        vec = 1 << bits
        copies = min(cfg.num_routed_wires // (2 + vec), cfg.num_wires // (2 + vec + bits))
        extra = min(cfg.num_routed_wires - (2 + vec) * copies, cfg.num_constants)
        return cls(bits, copies, extra, cfg.field)

    def wire_access_index(self, copy):
        return (2 + self.vec_size) * copy

    def wire_claimed_element(self, copy):
        return (2 + self.vec_size) * copy + 1

    def wire_list_item(self, i, copy):
        return (2 + self.vec_size) * copy + 2 + i

    def wire_extra_constant(self, i):
        return (2 + self.vec_size) * self.num_copies + i

    def wire_bit(self, i, copy):
        return self.num_routed + copy * self.bits + i

    def extra_constant_wires(self):
        return [(i, self.wire_extra_constant(i)) for i in range(self.num_extra_constants)]

    def generators(self, row, constants):
        return [_RandomAccessGenerator(self, row, copy) for copy in range(self.num_copies)]


class _RandomAccessGenerator:
    """random_access.rs:383-440: the bits of the access index"""

    def __init__(self, gate, row, copy):
        self.g, self.row, self.copy = gate, row, copy
        self.deps = [wire(row, gate.wire_access_index(copy))]

    def run(self, w, p):
        idx = w.get(self.deps[0])
        assert idx < self.g.vec_size, "Access index %d is larger than the vector size %d" % (idx, self.g.vec_size)
        for i in range(self.g.bits):
            w.set(wire(self.row, self.g.wire_bit(i, self.copy)), (idx >> i) & 1)


# --------------------------------------------------------------------------------------------- MDS on extension targets
class PoseidonMdsGate(Gate):
    """gates/poseidon_goldilocks_mds.rs:30-45: inputs i D .., outputs (12 + i) D .. (Goldilocks)"""
    kind, degree = GATE_POSEIDON_MDS, 1
    id = "PoseidonMdsGate(PhantomData<%s>)<WIDTH=12>" % _GL_NAME

    def __init__(self):
        self.E = Ext(N.GB_GOLDILOCKS)
        self.num_wires, self.num_constraints = 24 * self.E.D, 12 * self.E.D

    def generators(self, row, constants):
        return [_PoseidonMdsGenerator(self, row)]


class _PoseidonMdsGenerator:
    """poseidon_goldilocks_mds.rs:255-300"""

    def __init__(self, gate, row):
        self.g, self.row = gate, row
        self.deps = _ext_wires(row, 0, 12 * gate.E.D)

    def run(self, w, p):
        E, D = self.g.E, self.g.E.D
        T = _poseidon_tables()
        circ, diag = T["MDS_CIRC"], T["MDS_DIAG"]
        ins = [_get_ext(w, self.row, i * D, D) for i in range(12)]
        for r in range(12):
            acc = E.scale(ins[r], diag[r])
            for i in range(12):
                acc = E.add(acc, E.scale(ins[(i + r) % 12], circ[i]))
            _set_ext(w, self.row, (12 + r) * D, acc)


# --------------------------------------------------------------------------------------------- interpolation on a coset
class CosetInterpolationGate(Gate):
    """gates/coset_interpolation.rs:57-185: wire 0 = coset shift, then the 2^subgroup_bits values (D each), the evaluation
    point, the evaluation value (all routed); then the intermediate evals, the intermediate products and the shifted point"""
    kind = GATE_COSET_INTERPOLATION

    def __init__(self, subgroup_bits, field=N.GB_GOLDILOCKS, max_degree=None):
        E = self.E = Ext(field)
        D, n_points = E.D, 1 << subgroup_bits
        max_degree = max_degree or n_points
        assert max_degree > 1, "need at least quadratic constraints"
        n_intermediates = (n_points - 2) // (max_degree - 1)
        self.degree = (n_points - 2) // (n_intermediates + 1) + 2       # with_max_degree (:70-99)
        self.subgroup_bits, self.num_points = subgroup_bits, n_points
        self.param, self.param2 = subgroup_bits, self.degree
        self.num_intermediates = (n_points - 2) // (self.degree - 1)
        self.domain = E.subgroup(subgroup_bits)
        m_inv = pow(n_points, E.p - 2, E.p)
        self.barycentric_weights = [x * m_inv % E.p for x in self.domain]  # 1 / prod_{j != i} (x_i - x_j) = x_i / 2^bits
        self.start_point = 1 + n_points * D
        self.start_value = self.start_point + D
        self.start_intermediates = self.start_value + D
        self.num_wires = self.start_intermediates + D * (2 * self.num_intermediates + 1)
        self.num_constraints = 2 * D + 2 * D * self.num_intermediates
        self.id = "%d,%d,%s<D=%d>" % (subgroup_bits, self.degree, "[" + ", ".join(map(str, self.barycentric_weights)) + "]", D)

    def wires_value(self, i):
        return 1 + i * self.E.D

    def wires_intermediate_eval(self, i):
        return self.start_intermediates + self.E.D * i

    def wires_intermediate_prod(self, i):
        return self.start_intermediates + self.E.D * (self.num_intermediates + i)

    def wires_shifted_evaluation_point(self):
        return self.start_intermediates + self.E.D * 2 * self.num_intermediates

    def generators(self, row, constants):
        return [_InterpolationGenerator(self, row)]


class _InterpolationGenerator:
    """coset_interpolation.rs:451-560"""

    def __init__(self, gate, row):
        self.g, self.row = gate, row
        D = gate.E.D
        self.deps = [wire(row, 0)] + _ext_wires(row, 1, gate.num_points * D) + _ext_wires(row, gate.start_point, D)

    def run(self, w, p):
        g, E, row = self.g, self.g.E, self.row
        D = E.D
        shift = w.get(wire(row, 0))
        values = [_get_ext(w, row, g.wires_value(i), D) for i in range(g.num_points)]
        point = _get_ext(w, row, g.start_point, D)
        shifted = E.scale(point, pow(shift, E.p - 2, E.p))
        _set_ext(w, row, g.wires_shifted_evaluation_point(), shifted)

        def partial(lo, hi, ev, prod):
            for i in range(lo, hi):
                term = E.sub(shifted, E.from_base(g.domain[i]))
                ev = E.add(E.mul(ev, term), E.mul(E.scale(values[i], g.barycentric_weights[i]), prod))
                prod = E.mul(prod, term)
            return ev, prod

        ev, prod = partial(0, g.degree, E.zero, E.one)
        for i in range(g.num_intermediates):
            _set_ext(w, row, g.wires_intermediate_eval(i), ev)
            _set_ext(w, row, g.wires_intermediate_prod(i), prod)
            lo = 1 + (g.degree - 1) * (i + 1)
            ev, prod = partial(lo, min(lo + g.degree - 1, g.num_points), ev, prod)
        _set_ext(w, row, g.start_value, ev)


# --------------------------------------------------------------------------------------------- base^power
class ExponentiationGate(Gate):
    """gates/exponentiation.rs:30-70: wire 0 = base, 1.. = num_power_bits little-endian bits, then the output, then the
    intermediate values of the square-and-multiply chain"""
    kind, degree = GATE_EXPONENTIATION, 4

    def __init__(self, num_power_bits, field=N.GB_GOLDILOCKS):
        E = Ext(field)
        self.param = self.num_power_bits = num_power_bits
        self.num_wires, self.num_constraints = 2 + 2 * num_power_bits, num_power_bits + 1
        self.id = "ExponentiationGate { num_power_bits: %d, _phantom: PhantomData<%s> }<D=%d>" % (num_power_bits, E.name, E.D)

    @classmethod
    def new_from_config(cls, cfg):
        return cls(min(cfg.num_routed_wires - 2, (cfg.num_wires - 2) // 2), cfg.field)

    def generators(self, row, constants):
        return [_ExponentiationGenerator(self, row)]


class _ExponentiationGenerator:
    """exponentiation.rs:248-300"""

    def __init__(self, gate, row):
        self.g, self.row = gate, row
        self.deps = [wire(row, c) for c in range(1 + gate.num_power_bits)]

    def run(self, w, p):
        n, row = self.g.num_power_bits, self.row
        base = w.get(wire(row, 0))
        bits = [w.get(wire(row, 1 + i)) for i in range(n)]
        cur = 1
        for i in range(n):
            cur = cur * cur % p if i else 1
            if bits[n - 1 - i]:
                cur = cur * base % p
            w.set(wire(row, 2 + n + i), cur)
        w.set(wire(row, 1 + n), cur)


# --------------------------------------------------------------------------------------------- the linear gates of the BabyBear recursion
class AddManyGate(Gate):
    """gates/add_many.rs:23-49: num_ops x (num_addends wires, then their sum)"""
    kind, degree = GATE_ADD_MANY, 1

    def __init__(self, num_addends, num_ops):
        self.num_addends, self.num_ops = num_addends, num_ops
        self.param, self.param2 = num_addends, num_ops
        self.num_wires, self.num_constraints = (num_addends + 1) * num_ops, num_ops
        self.id = "AddManyGate { num_addends: %d, num_ops: %d }" % (num_addends, num_ops)

    @classmethod
    def new_from_config(cls, cfg, num_addends):
        return cls(num_addends, cfg.num_routed_wires // (num_addends + 1))

    def generators(self, row, constants):
        return [_AddManyGenerator(self, row, i) for i in range(self.num_ops)]


class _AddManyGenerator:
    """add_many.rs:167-215"""

    def __init__(self, gate, row, i):
        self.g, self.row, self.i = gate, row, i
        self.deps = [wire(row, (gate.num_addends + 1) * i + j) for j in range(gate.num_addends)]

    def run(self, w, p):
        w.set(wire(self.row, (self.g.num_addends + 1) * self.i + self.g.num_addends), sum(w.get(t) for t in self.deps) % p)


class ApplyMat4Gate(Gate):
    """gates/apply_mat4.rs:26-50: num_ops x (four input D-tuples, four output D-tuples = the Poseidon2 4x4 block applied)"""
    kind, degree = GATE_APPLY_MAT4, 1

    def __init__(self, num_ops, field=N.GB_GOLDILOCKS):
        E = self.E = Ext(field)
        self.param = self.num_ops = num_ops
        self.num_wires, self.num_constraints = 8 * E.D * num_ops, 4 * E.D * num_ops
        self.id = "ApplyMat4Gate { num_ops: %d, _phantom: PhantomData<%s> } number of operations = %d" % (num_ops, E.name, num_ops)

    @classmethod
    def new_from_config(cls, cfg):
        return cls(cfg.num_routed_wires // (8 * Ext(cfg.field).D), cfg.field)

    def generators(self, row, constants):
        return [_ApplyMat4Generator(self, row, op) for op in range(self.num_ops)]


class _ApplyMat4Generator:
    """apply_mat4.rs:216-270"""

    def __init__(self, gate, row, op):
        self.g, self.row, self.op = gate, row, op
        self.deps = _ext_wires(row, op * 8 * gate.E.D, 4 * gate.E.D)

    def run(self, w, p):
        E, D = self.g.E, self.g.E.D
        base = self.op * 8 * D
        x = [_get_ext(w, self.row, base + i * D, D) for i in range(4)]
        t01, t23 = E.add(x[0], x[1]), E.add(x[2], x[3])
        t0123 = E.add(t01, t23)
        t01123, t01233 = E.add(t0123, x[1]), E.add(t0123, x[3])
        new = [E.add(t01123, t01), E.add(t01123, E.add(x[2], x[2])), E.add(t01233, t23), E.add(t01233, E.add(x[0], x[0]))]
        for i in range(4):
            _set_ext(w, self.row, base + (4 + i) * D, new[i])


class Poseidon2InternalPermutationGate(Gate):
    """gates/poseidon2_internal_permutation.rs:30-48: sixteen input D-tuples, sixteen output D-tuples = M_I applied (BabyBear)"""
    kind, degree = GATE_POSEIDON2_INTERNAL_PERMUTATION, 1
    SHIFTS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
    id = "Poseidon2InternalPermutationGate(PhantomData<%s>)<WIDTH=16>" % _BB_NAME

    def __init__(self):
        self.E = Ext(N.GB_BABYBEAR)
        self.num_wires, self.num_constraints = 32 * self.E.D, 16 * self.E.D

    def generators(self, row, constants):
        return [_Poseidon2InternalGenerator(self, row)]


class _Poseidon2InternalGenerator:
    """poseidon2_internal_permutation.rs:225-290"""

    def __init__(self, gate, row):
        self.g, self.row = gate, row
        self.deps = _ext_wires(row, 0, 16 * gate.E.D)

    def run(self, w, p):
        E, D = self.g.E, self.g.E.D
        s = [E.scale(_get_ext(w, self.row, i * D, D), 943718400) for i in range(16)]
        part = E.zero
        for x in s[1:]:
            part = E.add(part, x)
        full = E.add(part, s[0])
        new = [E.sub(part, s[0])] + [E.add(full, E.scale(s[i + 1], 1 << self.g.SHIFTS[i])) for i in range(15)]
        for i in range(16):
            _set_ext(w, self.row, (16 + i) * D, new[i])
