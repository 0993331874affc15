"""Host-side mirror of the reference's CircuitBuilder / PartialWitness for the gate set the GPU prover evaluates.

What a Rust host already has (plonky2/src/plonk/circuit_builder.rs, gadgets/arithmetic.rs, iop/generator.rs), restated in
Python so that circuits other than the dummy one - the reference's `factorial` example is the model
(plonky2/examples/factorial.rs) - can be built, witnessed and handed to `gb_circuit_create_gates` / `gb_prove`:

    builder = CircuitBuilder(CircuitConfig.standard_recursion_config_gl())
    x = builder.add_virtual_target(); y = builder.mul(x, builder.constant(3)); builder.register_public_input(y)
    data = builder.build(ctx)                   # CircuitData: constants||sigmas committed on the GPU
    pw = PartialWitness(); pw.set_target(x, 5)
    proof = data.prove(pw)                      # witness generation on the host, prove() on the GPU
    data.verify(proof)

Gates: NoopGate, ConstantGate, PublicInputGate, ArithmeticGate (gates/arithmetic_base.rs) and the in-circuit hash of the
configuration - PoseidonGate (gates/poseidon_goldilocks.rs) for Goldilocks, Poseidon2BabyBearGate (gates/poseidon2_babybear.rs)
for BabyBear - which `build()` itself needs as soon as a circuit has public inputs, because it hashes them in-circuit
(circuit_builder.rs:1126-1137).  The other gates of the recursion circuits are in recursion_gates.py (add_gate() takes them).
Not the GPU hot path: plain Python integers.
"""
import os
import re

import numpy as np

from . import native as N
from .dummy_circuit import BB_P, P as GL_P, bb_mul, bb_powers, gl_mul, gl_powers

GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON, GATE_POSEIDON2_BABYBEAR = 0, 1, 2, 3, 4, 5  # gb_gate.kind


class CircuitConfig:
    """plonk/circuit_data.rs:63-93"""

    def __init__(self, field=N.GB_GOLDILOCKS, num_wires=135, num_routed_wires=80, num_constants=2, num_challenges=2,
                 max_quotient_degree_factor=8, rate_bits=3, cap_height=4, proof_of_work_bits=16, num_query_rounds=28,
                 arity_bits=4, final_poly_bits=5, security_bits=100, fri_reduction_strategy=None):
        """fri_reduction_strategy: None = FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (every stock
        configuration); ("fixed", [bits..]) or ("min_size", max_arity_bits or None) as in fri/reduction_strategies.rs:11-25"""
        self.__dict__.update(locals())
        del self.__dict__["self"]

    def reduction_arity_bits(self, degree_bits):
        """FriConfig::fri_params' list for a circuit of 2^degree_bits rows (fri/mod.rs; None = let the library derive it)"""
        if self.fri_reduction_strategy is None:
            return None
        from . import fri_params
        return fri_params.reduction_arity_bits(self.fri_reduction_strategy, degree_bits, self.rate_bits, self.cap_height, self.num_query_rounds)

    @classmethod
    def standard_recursion_config_gl(cls, **kw):
        """circuit_data.rs:102-116"""
        return cls(**kw)

    @classmethod
    def recursion_config_bb_narrow(cls, **kw):
        """circuit_data.rs:131-139"""
        d = dict(field=N.GB_BABYBEAR, num_wires=167, num_routed_wires=41, arity_bits=3, num_challenges=6)
        d.update(kw)
        return cls(**d)


class _Field:
    def __init__(self, field):
        gl = field == N.GB_GOLDILOCKS
        self.field, self.p = field, GL_P if gl else BB_P
        self.order_bits, self.ext_degree, self.hout = (64, 2, 4) if gl else (31, 4, 8)
        self.generator = 7 if gl else 31
        self.dtype = np.uint64 if gl else np.uint32
        self._two_adic = (1753635133440165772, 32) if gl else (0x1a427a41, 27)
        self._mul, self._powers = (gl_mul, gl_powers) if gl else (bb_mul, bb_powers)

    def two_adic_generator(self, bits):
        g, a = self._two_adic
        return pow(g, 1 << (a - bits), self.p)


# --------------------------------------------------------------------------------------------- gates
class Gate:
    kind, param, degree, num_constants, num_constraints = None, 0, 0, 0, 0
    num_ops = 1

    def __eq__(self, o):
        return self.id == o.id

    def __hash__(self):
        return hash(self.id)

    def generators(self, row, constants):
        return []

    def extra_constant_wires(self):
        return []


class NoopGate(Gate):
    """gates/noop.rs"""
    kind, id = GATE_NOOP, "NoopGate"
    num_wires = 0


class ConstantGate(Gate):
    """gates/constant.rs:22-140"""
    kind, degree = GATE_CONSTANT, 1

    def __init__(self, num_consts):
        self.param = self.num_constants = self.num_constraints = self.num_wires = num_consts
        self.id = "ConstantGate { num_consts: %d }" % num_consts

    def extra_constant_wires(self):
        return [(i, i) for i in range(self.param)]


class PublicInputGate(Gate):
    """gates/public_input.rs:24-110: wires 0..H are the public-inputs hash"""
    kind, degree = GATE_PUBLIC_INPUT, 1

    def __init__(self, hout):
        self.param = self.num_constraints = self.num_wires = hout
        self.id = "PublicInputGate<%d>" % hout


class ArithmeticGate(Gate):
    """gates/arithmetic_base.rs:27-190: num_ops x (out = c0 * m0 * m1 + c1 * addend) on wires 4i..4i+3"""
    kind, degree, num_constants = GATE_ARITHMETIC, 3, 2

    def __init__(self, num_ops):
        self.param = self.num_ops = self.num_constraints = num_ops
        self.num_wires = 4 * num_ops
        self.id = "ArithmeticGate { num_ops: %d }" % num_ops

    @classmethod
    def new_from_config(cls, cfg):
        return cls(cfg.num_routed_wires // 4)

    def generators(self, row, constants):
        return [_ArithmeticGenerator(row, constants[0], constants[1], i) for i in range(self.num_ops)]


class PoseidonGate(Gate):
    """gates/poseidon_goldilocks.rs:37-434: one width-12 permutation per row; wires 0..11 in, 12..23 out, 24 swap,
    25..28 delta, then the s-box inputs of full rounds 1..3, the 22 partial rounds and full rounds 4..7."""
    kind, degree = GATE_POSEIDON, 7
    WIRE_SWAP, START_DELTA, START_FULL_0 = 24, 25, 29
    START_PARTIAL = START_FULL_0 + 12 * 3
    START_FULL_1 = START_PARTIAL + 22
    num_wires = START_FULL_1 + 12 * 4            # 135
    num_constraints = 12 * 7 + 22 + 12 + 1 + 4   # 123
    id = "PoseidonGate(PhantomData<p3_goldilocks::goldilocks::Goldilocks>)<WIDTH=12>"

    def generators(self, row, constants):
        return [_PoseidonGenerator(row)]


class Poseidon2BabyBearGate(Gate):
    """gates/poseidon2_babybear.rs:48-147, 473-497: num_ops width-16 Poseidon2 permutations per row; per op 33 routed wires
    (16 in, 16 out, swap) for all ops first, then per op 133 non-routed wires (8 deltas, the s-box inputs of full rounds 1..3,
    of the 13 internal rounds and of full rounds 4..7).  recursion_config_bb_narrow (167 wires, 41 routed) fits one op."""
    kind, degree = GATE_POSEIDON2_BABYBEAR, 7
    ROUTED, NON_ROUTED = 33, 8 + 16 * 7 + 13

    def __init__(self, num_ops):
        self.param = self.num_ops = num_ops
        self.num_wires = (self.ROUTED + self.NON_ROUTED) * num_ops
        self.num_constraints = (1 + 8 + 16 * 7 + 13 + 16) * num_ops
        # Debug of the gate struct; the type name inside PhantomData is p3's (BabyBear is an alias of MontyField31<..>).  Only
        # the relative order of ids within one degree matters (circuit_builder.rs:1195-1196) and this is the only degree-7 gate.
        self.id = ("Poseidon2BabyBearGate { num_ops: %d, _phantom: PhantomData<p3_monty_31::monty_31::MontyField31<"
                   "p3_baby_bear::baby_bear::BabyBearParameters>> }<WIDTH=16>" % num_ops)

    @classmethod
    def new_from_config(cls, cfg):
        return cls(min(cfg.num_wires // (cls.ROUTED + cls.NON_ROUTED), cfg.num_routed_wires // cls.ROUTED))

    def generators(self, row, constants):
        return [_Poseidon2Generator(row, self.num_ops, op) for op in range(self.num_ops)]


# --------------------------------------------------------------------------------------------- targets, generators
def wire(row, column):
    return ("w", row, column)


class _ArithmeticGenerator:
    def __init__(self, row, c0, c1, i):
        self.row, self.c0, self.c1, self.i = row, c0, c1, i
        self.deps = [wire(row, 4 * i), wire(row, 4 * i + 1), wire(row, 4 * i + 2)]

    def run(self, w, p):
        m0, m1, a = (w.get(t) for t in self.deps)
        w.set(wire(self.row, 4 * self.i + 3), (m0 * m1 * self.c0 + a * self.c1) % p)


class _ConstantGenerator:
    """iop/generator.rs ConstantGenerator"""
    deps = []

    def __init__(self, row, constant_index, wire_index):
        self.row, self.constant_index, self.wire_index, self.constant = row, constant_index, wire_index, 0

    def run(self, w, p):
        w.set(wire(self.row, self.wire_index), self.constant)


class _RandomValueGenerator:
    deps = []

    def __init__(self, target):
        self.target = target

    def run(self, w, p):
        w.set(self.target, w.random())


class _PoseidonGenerator:
    """gates/poseidon_goldilocks.rs:436-531"""

    def __init__(self, row):
        self.row = row
        self.deps = [wire(row, c) for c in range(12)] + [wire(row, PoseidonGate.WIRE_SWAP)]

    def run(self, w, p):
        G, row = PoseidonGate, self.row
        state = [w.get(wire(row, c)) for c in range(12)]
        swap = w.get(wire(row, G.WIRE_SWAP))
        assert swap in (0, 1)
        for i in range(4):
            w.set(wire(row, G.START_DELTA + i), swap * (state[i + 4] - state[i]) % p)
        if swap:
            state = state[4:8] + state[0:4] + state[8:]
        out = poseidon_gate_trace(state)
        for col, v in out.items():
            w.set(wire(row, col), v)


class _Poseidon2Generator:
    """gates/poseidon2_babybear.rs:536-676"""

    def __init__(self, row, num_ops, op):
        self.row, self.num_ops, self.op = row, num_ops, op
        self.deps = [wire(row, 33 * op + c) for c in range(16)] + [wire(row, 33 * op + 32)]

    def run(self, w, p):
        row, op = self.row, self.op
        state = [w.get(t) for t in self.deps[:16]]
        swap = w.get(self.deps[16])
        assert swap in (0, 1)
        start_delta = self.num_ops * 33 + op * 133
        for i in range(8):
            w.set(wire(row, start_delta + i), swap * (state[i + 8] - state[i]) % p)
        if swap:
            state = state[8:] + state[:8]
        for col, v in poseidon2_gate_trace(state, start_delta + 8, 33 * op + 16).items():
            w.set(wire(row, col), v)


_POSEIDON_TABLES = None


def _poseidon_tables():
    """The Poseidon-12 tables of csrc/poseidon_constants.h (generated from hash/poseidon_goldilocks.rs by tools/gen_constants.py)."""
    global _POSEIDON_TABLES
    if _POSEIDON_TABLES is None:
        text = open(os.path.join(os.path.dirname(__file__), "csrc", "poseidon_constants.h")).read()
        tabs = {}
        for m in re.finditer(r"#define (?:GL_POSEIDON|BB_POSEIDON2)_(\w+)_LIST \\\n((?:[^\n]*\\\n)*[^\n]*)", text):
            tabs[m.group(1)] = [int(x.rstrip("uUlL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*|\d+[uUlL]*", m.group(2))]
        _POSEIDON_TABLES = tabs
    return _POSEIDON_TABLES


def poseidon_gate_trace(state):
    """The permutation exactly as PoseidonGenerator runs it (fast partial rounds), returning {wire column: value} for the
    s-box-input and output wires of the gate."""
    T, p, G = _poseidon_tables(), GL_P, PoseidonGate
    rc, circ, diag = T["ALL_ROUND_CONSTANTS"], T["MDS_CIRC"], T["MDS_DIAG"]
    out = {}

    def mds(s):
        return [(sum(s[(i + r) % 12] * circ[i] for i in range(12)) + s[r] * diag[r]) % p for r in range(12)]

    s = list(state)
    ctr = 0
    for r in range(4):
        s = [(s[i] + rc[12 * ctr + i]) % p for i in range(12)]
        if r:
            for i in range(12):
                out[G.START_FULL_0 + 12 * (r - 1) + i] = s[i]
        s = mds([pow(x, 7, p) for x in s])
        ctr += 1
    s = [(s[i] + T["FAST_PARTIAL_FIRST_ROUND_CONSTANT"][i]) % p for i in range(12)]
    init = T["FAST_PARTIAL_ROUND_INITIAL_MATRIX"]
    s = [s[0]] + [sum(s[r] * init[(r - 1) * 11 + (c - 1)] for r in range(1, 12)) % p for c in range(1, 12)]
    for r in range(22):
        out[G.START_PARTIAL + r] = s[0]
        s[0] = pow(s[0], 7, p)
        if r < 21:
            s[0] = (s[0] + T["FAST_PARTIAL_ROUND_CONSTANTS"][r]) % p
        d = (s[0] * (circ[0] + diag[0]) + sum(s[i] * T["FAST_PARTIAL_ROUND_W_HATS"][r * 11 + i - 1] for i in range(1, 12))) % p
        s = [d] + [(s[0] * T["FAST_PARTIAL_ROUND_VS"][r * 11 + i - 1] + s[i]) % p for i in range(1, 12)]
    ctr += 22
    for r in range(4):
        s = [(s[i] + rc[12 * ctr + i]) % p for i in range(12)]
        for i in range(12):
            out[G.START_FULL_1 + 12 * r + i] = s[i]
        s = mds([pow(x, 7, p) for x in s])
        ctr += 1
    for i in range(12):
        out[12 + i] = s[i]
    return out


def poseidon2_gate_trace(state, start_full_0, out0):
    """The width-16 Poseidon2 permutation as Poseidon2BabyBearGenerator runs it (gates/poseidon2_babybear.rs:608-676),
    returning {wire column: value} for the s-box-input wires (from start_full_0) and the outputs (from out0)."""
    T, p = _poseidon_tables(), BB_P
    ext, internal = T["EXTERNAL_CONSTANTS"], T["INTERNAL_CONSTANTS"]
    shifts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
    out = {}

    def external(s):
        s = list(s)
        for i in range(0, 16, 4):
            x0, x1, x2, x3 = s[i:i + 4]
            s[i:i + 4] = [2 * x0 + 3 * x1 + x2 + x3, x0 + 2 * x1 + 3 * x2 + x3, x0 + x1 + 2 * x2 + 3 * x3, 3 * x0 + x1 + x2 + 2 * x3]
        sums = [sum(s[k::4]) for k in range(4)]
        return [(s[i] + sums[i % 4]) % p for i in range(16)]

    s = external(state)
    for r in range(4):
        s = [(s[i] + ext[16 * r + i]) % p for i in range(16)]
        if r:
            for i in range(16):
                out[start_full_0 + 16 * (r - 1) + i] = s[i]
        s = external([pow(x, 7, p) for x in s])
    start_partial = start_full_0 + 48
    for r in range(13):
        s[0] = (s[0] + internal[r]) % p
        out[start_partial + r] = s[0]
        s[0] = pow(s[0], 7, p)
        s = [x * 943718400 % p for x in s]
        part = sum(s[1:]) % p
        full = (part + s[0]) % p
        s = [(part - s[0]) % p] + [(full + (s[i + 1] << shifts[i])) % p for i in range(15)]
    start_full_1 = start_partial + 13
    for r in range(4, 8):
        s = [(s[i] + ext[16 * r + i]) % p for i in range(16)]
        for i in range(16):
            out[start_full_1 + 16 * (r - 4) + i] = s[i]
        s = external([pow(x, 7, p) for x in s])
    for i in range(16):
        out[out0 + i] = s[i]
    return out


class PartialWitness:
    """iop/witness.rs PartialWitness: target -> value"""

    def __init__(self):
        self.values = {}

    def set_target(self, target, value):
        self.values[target] = int(value)


class _PartitionWitness:
    """iop/witness.rs:288-357: one value per copy class (representative)"""

    def __init__(self, find, p, rng):
        self.find, self.p, self.rng, self.v = find, p, rng, {}

    def get(self, t):
        return self.v[self.find(t)]

    def has(self, t):
        return self.find(t) in self.v

    def set(self, t, value):
        r = self.find(t)
        value %= self.p
        if r in self.v and self.v[r] != value:
            raise ValueError("Partition containing %r was set twice with different values: %d != %d" % (t, self.v[r], value))
        self.v[r] = value

    def random(self):
        return int(self.rng.integers(0, self.p, dtype=np.uint64))


# --------------------------------------------------------------------------------------------- builder
class CircuitBuilder:
    """plonk/circuit_builder.rs:142-1409 (no lookups, no zero-knowledge blinding, base arithmetic gate only)."""

    def __init__(self, config):
        self.config, self.F = config, _Field(config.field)
        c = config
        # check_fri_security_bits (circuit_builder.rs:266-290)
        ext_bits = self.F.order_bits * self.F.ext_degree
        if min(ext_bits, c.num_query_rounds * c.rate_bits + c.proof_of_work_bits) < c.security_bits:
            raise ValueError("FRI params fall short of target security")
        self.gates, self.gate_instances = set(), []   # instances: [gate, constants]
        self.public_inputs, self.virtual_target_index = [], 0
        self.copy_constraints, self.generators = [], []
        self.constants_to_targets, self.targets_to_constants = {}, {}
        self.base_arithmetic_results, self.current_slots, self.constant_generators = {}, {}, []
        self.random_wire = None

    # ---- targets
    def add_virtual_target(self):
        self.virtual_target_index += 1
        return ("v", self.virtual_target_index - 1)

    def add_virtual_targets(self, n):
        return [self.add_virtual_target() for _ in range(n)]

    def add_virtual_public_input(self):
        t = self.add_virtual_target()
        self.register_public_input(t)
        return t

    def register_public_input(self, t):
        self.public_inputs.append(t)

    def register_public_inputs(self, ts):
        for t in ts:
            self.register_public_input(t)

    def num_gates(self):
        return len(self.gate_instances)

    def is_routable(self, t):
        return t[0] == "v" or t[2] < self.config.num_routed_wires

    def connect(self, x, y):
        if not (self.is_routable(x) and self.is_routable(y)):
            raise ValueError("Tried to route a wire that isn't routable")
        self.copy_constraints.append((x, y))

    def assert_zero(self, x):
        self.connect(x, self.zero())

    def assert_one(self, x):
        self.connect(x, self.one())

    def constant(self, c):
        c = int(c) % self.F.p
        t = self.constants_to_targets.get(c)
        if t is None:
            t = self.add_virtual_target()
            self.constants_to_targets[c] = t
            self.targets_to_constants[t] = c
        return t

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def two(self):
        return self.constant(2)

    def neg_one(self):
        return self.constant(self.F.p - 1)

    def target_as_constant(self, t):
        return self.targets_to_constants.get(t)

    # ---- gates
    def add_gate(self, gate, constants=()):
        """circuit_builder.rs:478-514"""
        cfg = self.config
        if gate.num_wires > cfg.num_wires or gate.num_constants > cfg.num_constants:
            raise ValueError("%s does not fit the CircuitConfig" % gate.id)
        consts = list(constants) + [0] * (gate.num_constants - len(constants))
        row = len(self.gate_instances)
        self.constant_generators += [_ConstantGenerator(row, ci, wi) for ci, wi in gate.extra_constant_wires()]
        self.gates.add(gate)
        self.gate_instances.append([gate, consts])
        return row

    def add_gate_to_gate_set(self, gate):
        self.gates.add(gate)

    def find_slot(self, gate, params, constants):
        """circuit_builder.rs:821-852"""
        slots = self.current_slots.setdefault(gate.id, {})
        key = tuple(params)
        if key in slots:
            gate_idx, slot_idx = slots[key]
        else:
            gate_idx, slot_idx = self.add_gate(gate, constants), 0
        if slot_idx == gate.num_ops - 1:
            slots.pop(key, None)
        else:
            slots[key] = (gate_idx, slot_idx + 1)
        return gate_idx, slot_idx

    # ---- gadgets/arithmetic.rs
    def arithmetic(self, const_0, const_1, m0, m1, addend):
        p = self.F.p
        const_0, const_1 = const_0 % p, const_1 % p
        r = self._arithmetic_special_cases(const_0, const_1, m0, m1, addend)
        if r is not None:
            return r
        op = (const_0, const_1, m0, m1, addend)
        if op in self.base_arithmetic_results:
            return self.base_arithmetic_results[op]
        gate = ArithmeticGate.new_from_config(self.config)
        row, i = self.find_slot(gate, (const_0, const_1), (const_0, const_1))
        self.connect(m0, wire(row, 4 * i))
        self.connect(m1, wire(row, 4 * i + 1))
        self.connect(addend, wire(row, 4 * i + 2))
        out = wire(row, 4 * i + 3)
        self.base_arithmetic_results[op] = out
        return out

    def _arithmetic_special_cases(self, c0, c1, m0, m1, addend):
        """gadgets/arithmetic.rs:112-165"""
        p = self.F.p
        zero = self.zero()
        k0, k1, ka = self.target_as_constant(m0), self.target_as_constant(m1), self.target_as_constant(addend)
        first_zero = c0 == 0 or m0 == zero or m1 == zero
        second_zero = c1 == 0 or addend == zero
        first_const = 0 if first_zero else (k0 * k1 * c0 % p if k0 is not None and k1 is not None else None)
        second_const = 0 if second_zero else (ka * c1 % p if ka is not None else None)
        if first_const is not None and second_const is not None:
            return self.constant((first_const + second_const) % p)
        if first_zero and c1 == 1:
            return addend
        if second_zero:
            if k0 is not None and k0 * c0 % p == 1:
                return m1
            if k1 is not None and k1 * c0 % p == 1:
                return m0
        return None

    def mul(self, x, y):
        return self.arithmetic(1, 0, x, y, x)

    def add(self, x, y):
        return self.arithmetic(1, 1, x, self.one(), y)

    def sub(self, x, y):
        return self.arithmetic(1, self.F.p - 1, x, self.one(), y)

    def mul_add(self, x, y, z):
        return self.arithmetic(1, 1, x, y, z)

    def mul_sub(self, x, y, z):
        return self.arithmetic(1, self.F.p - 1, x, y, z)

    def neg(self, x):
        return self.mul(x, self.neg_one())

    def square(self, x):
        return self.mul(x, x)

    def mul_const(self, c, x):
        return self.mul(self.constant(c), x)

    def add_const(self, x, c):
        return self.add(x, self.constant(c))

    def mul_many(self, terms):
        acc = self.one()
        for t in terms:
            acc = self.mul(acc, t)
        return acc

    # ---- hashing (plonk/config.rs:135-166, hash/poseidon_goldilocks.rs:1116-1143)
    def permute(self, state):
        if self.config.field != N.GB_GOLDILOCKS:
            # hash/poseidon2_babybear.rs:183-213: a slot of a Poseidon2BabyBearGate
            gate = Poseidon2BabyBearGate.new_from_config(self.config)
            if gate.num_ops != 1:
                raise NotImplementedError("Poseidon2BabyBearGate with several operations per row (complete_wires) is not restated")
            row, op = self.find_slot(gate, (), ())
            self.connect(self.zero(), wire(row, 33 * op + 32))
            for i in range(16):
                self.connect(state[i], wire(row, 33 * op + i))
            return [wire(row, 33 * op + 16 + i) for i in range(16)]
        row = self.add_gate(PoseidonGate())
        self.connect(self.zero(), wire(row, PoseidonGate.WIRE_SWAP))   # swap = _false()
        for i in range(12):
            self.connect(state[i], wire(row, i))
        return [wire(row, 12 + i) for i in range(12)]

    def hash_n_to_hash_no_pad(self, inputs):
        width = 12 if self.config.field == N.GB_GOLDILOCKS else 16
        state = [self.zero()] * width
        for c in range(0, len(inputs), 8):
            chunk = inputs[c:c + 8]
            state[:len(chunk)] = chunk
            state = self.permute(state)
        return state[:self.F.hout]

    # ---- build (circuit_builder.rs:1110-1378)
    def build(self, ctx=None):
        cfg, F = self.config, self.F
        p = F.p
        pi_hash = self.hash_n_to_hash_no_pad(list(self.public_inputs))
        pi_gate = self.add_gate(PublicInputGate(F.hout))
        for i, h in enumerate(pi_hash):
            self.connect(h, wire(pi_gate, i))
        # randomize_unused_pi_wires (:1064-1080)
        for w in range(F.hout, cfg.num_wires):
            if w == cfg.num_wires - 1:
                self.random_wire = (pi_gate, w)
            self.generators.append(_RandomValueGenerator(wire(pi_gate, w)))
        while len(self.constants_to_targets) > len(self.constant_generators):
            self.add_gate(ConstantGate(cfg.num_constants))
        for (c, t), gen in zip(sorted(self.constants_to_targets.items()), self.constant_generators):
            self.gate_instances[gen.row][1][gen.constant_index] = c
            self.connect(wire(gen.row, gen.wire_index), t)
            gen.constant = c
            self.generators.append(gen)
        while len(self.gate_instances) & (len(self.gate_instances) - 1):   # blind_and_pad, zero_knowledge = false
            self.add_gate(NoopGate())
        degree = len(self.gate_instances)
        degree_bits = degree.bit_length() - 1
        if F.order_bits * F.ext_degree - degree_bits < cfg.security_bits:
            raise ValueError("The degree of the extension is not enough for the soundness of the evaluation point")
        if (F.order_bits - degree_bits) * cfg.num_challenges < cfg.security_bits:
            raise ValueError("The number of challenges is not sufficient for the soundness of permutation argument and "
                             "combining constraints")
        gates = sorted(self.gates, key=lambda g: (g.degree, g.id))
        sel_cols, selector_indices, groups = selector_polynomials(gates, self.gate_instances, cfg.max_quotient_degree_factor + 1, p)
        max_constants = max(g.num_constants for g in gates)
        consts = np.zeros((max_constants, degree), dtype=F.dtype)
        for row, (_, kc) in enumerate(self.gate_instances):
            for i, c in enumerate(kc):
                consts[i, row] = c
        k_is = np.array([pow(F.generator, i, p) for i in range(cfg.num_routed_wires)], dtype=F.dtype)
        subgroup = F._powers(F.two_adic_generator(degree_bits), degree)
        find = self._forest()
        sigma = self._sigma(find, degree, k_is, subgroup)
        constants_sigmas = np.concatenate([np.array(sel_cols, dtype=F.dtype).reshape(len(groups), degree), consts, sigma])
        # gate generators, dropping the unused operations of partly filled gates (:1252-1270)
        incomplete = {}
        for slots in self.current_slots.values():
            for gate_idx, slot_idx in slots.values():
                incomplete[gate_idx] = slot_idx
        gens = list(self.generators)
        for row, (g, kc) in enumerate(self.gate_instances):
            gg = g.generators(row, kc)
            if row in incomplete:
                gg = gg[:incomplete[row]]
            gens += gg
        gate_table = [(g.kind, g.param, selector_indices[i], groups[selector_indices[i]][0], groups[selector_indices[i]][1],
                       getattr(g, "param2", 0), getattr(g, "param3", 0)) for i, g in enumerate(gates)]
        return BuiltCircuit(ctx, cfg, F, degree_bits, constants_sigmas, k_is, gate_table, len(groups), max_constants, gens, find,
                            list(self.public_inputs), self.random_wire, [g.id for g in gates],
                            sorted({t for ab in self.copy_constraints for t in ab if t[0] == "w"}))

    def _forest(self):
        """Disjoint sets over the targets that appear in copy constraints (plonk/permutation_argument.rs:13-101)."""
        parent = {}

        def find(t):
            root = t
            while parent.get(root, root) != root:
                root = parent[root]
            while parent.get(t, t) != t:
                parent[t], t = root, parent[t]
            return root

        for a, b in self.copy_constraints:
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[rb] = ra
        return find

    def _sigma(self, find, degree, k_is, subgroup):
        """sigma_vecs (:1005-1040): identity, then every copy class's routed wires, in (row, column) order, point to the next."""
        F, nr = self.F, self.config.num_routed_wires
        sig = np.empty((nr, degree), dtype=F.dtype)
        for j in range(nr):
            sig[j] = F._mul(subgroup, k_is[j])
        classes = {}
        touched = set()
        for a, b in self.copy_constraints:
            touched.add(a)
            touched.add(b)
        for t in touched:
            if t[0] == "w" and t[2] < nr:
                classes.setdefault(find(t), []).append((t[1], t[2]))
        for members in classes.values():
            members.sort()
            for i, (row, col) in enumerate(members):
                nrow, ncol = members[(i + 1) % len(members)]
                sig[col, row] = int(k_is[ncol]) * int(subgroup[nrow]) % F.p
        return sig


def selector_polynomials(gates, instances, max_degree, p):
    """gates/selectors.rs:125-209 -> (selector columns, selector_indices, groups as (start, end))"""
    n, num_gates = len(instances), len(gates)
    index = {g.id: i for i, g in enumerate(gates)}
    max_gate_degree = gates[-1].degree
    if max_gate_degree + num_gates - 1 <= max_degree:
        return [[index[g.id] for g, _ in instances]], [0] * num_gates, [(0, num_gates)]
    if max_gate_degree >= max_degree:
        raise ValueError("%s has too high degree. Consider increasing `quotient_degree_factor`." % gates[-1].id)
    groups, start = [], 0
    while start < num_gates:
        size = 0
        while start + size < num_gates and size + gates[start + size].degree < max_degree:
            size += 1
        groups.append((start, start + size))
        start += size
    group = [next(k for k, (a, b) in enumerate(groups) if a <= i < b) for i in range(num_gates)]
    unused = 0xFFFFFFFF % p
    cols = [[unused] * n for _ in groups]
    for j, (g, _) in enumerate(instances):
        i = index[g.id]
        cols[group[i]][j] = i
    return cols, group, groups


class BuiltCircuit:
    """CircuitData (plonk/circuit_data.rs:153-300) for a built circuit: prove(PartialWitness) / verify(proof)."""

    def __init__(self, ctx, cfg, F, degree_bits, constants_sigmas, k_is, gate_table, num_selectors, max_constants, generators, find,
                 public_inputs, random_wire, gate_ids, copy_wires):
        self.config, self.F, self.degree_bits = cfg, F, degree_bits
        self.constants_sigmas, self.k_is, self.gate_table, self.gate_ids = constants_sigmas, k_is, gate_table, gate_ids
        self.num_selectors, self.max_constants = num_selectors, max_constants
        self.generators, self.find, self.public_inputs, self.random_wire = generators, find, public_inputs, random_wire
        self.copy_wires = copy_wires
        self.data = None
        if ctx is not None:
            from .prover import CircuitData
            self.data = CircuitData(ctx, degree_bits, constants_sigmas, k_is, num_wires=cfg.num_wires,
                                    num_routed_wires=cfg.num_routed_wires, num_constants=max_constants,
                                    num_challenges=cfg.num_challenges, max_quotient_degree_factor=cfg.max_quotient_degree_factor,
                                    rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, proof_of_work_bits=cfg.proof_of_work_bits,
                                    num_query_rounds=cfg.num_query_rounds, arity_bits=cfg.arity_bits,
                                    final_poly_bits=cfg.final_poly_bits, num_selectors=num_selectors, field=cfg.field,
                                    gates=gate_table, num_public_inputs=len(public_inputs),
                                    reduction_arity_bits=cfg.reduction_arity_bits(degree_bits))

    def generate_witness(self, pw, rng=None):
        """generate_partial_witness + full_witness (iop/generator.rs:25-117, iop/witness.rs:359-371)
        -> (wire_values [num_wires][n], public input values)"""
        p = self.F.p
        w = _PartitionWitness(self.find, p, rng or np.random.default_rng(0))
        for t, v in pw.values.items():
            w.set(t, v)
        pending = list(self.generators)
        while pending:
            rest = []
            for g in pending:
                if all(w.has(t) for t in g.deps):
                    g.run(w, p)
                else:
                    rest.append(g)
            if len(rest) == len(pending):
                break   # the remaining generators wait on targets nobody sets; their wires stay zero like unset wires
            pending = rest
        # full_witness: every wire carries the value of its copy class; unset wires are zero
        wires = np.zeros((self.config.num_wires, 1 << self.degree_bits), dtype=self.F.dtype)
        for rep, v in w.v.items():
            if rep[0] == "w":
                wires[rep[2], rep[1]] = v
        for t in self.copy_wires:
            r = self.find(t)
            if r in w.v:
                wires[t[2], t[1]] = w.v[r]
        return wires, [w.get(t) for t in self.public_inputs]

    def prove(self, pw, rng=None):
        wires, pis = self.generate_witness(pw, rng)
        col_row = (self.random_wire[1], self.random_wire[0]) if self.random_wire else None
        return self.data.prove(wires, pis, random_wire=col_row, rng=rng)

    def verify(self, proof):
        return self.data.verify(proof)
