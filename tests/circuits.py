"""Shared circuit constructions for the builder / gate tests (product-side CircuitBuilder -> oracle BuiltCircuit)."""
import numpy as np

from oracle import plonk_dummy as PD
from oracle.fields import BB, GL
from plonky2_goldibear_amd import native as N
from plonky2_goldibear_amd.circuit_builder import CircuitBuilder, CircuitConfig, PartialWitness


def factorial_circuit(start=1, count=99, **cfg_kw):
    """plonky2/examples/factorial.rs: cur = initial * 2 * 3 * ... ; public inputs (initial, result)"""
    b = CircuitBuilder(CircuitConfig.standard_recursion_config_gl(**cfg_kw))
    initial = b.add_virtual_target()
    cur = initial
    for i in range(2, 2 + count):
        cur = b.mul(cur, b.constant(i))
    b.register_public_input(initial)
    b.register_public_input(cur)
    pw = PartialWitness()
    pw.set_target(initial, start)
    return b, pw


def fibonacci_circuit(terms=99, **cfg_kw):
    """plonky2/examples/fibonacci.rs: adds only; the initial values and the result are public inputs"""
    b = CircuitBuilder(CircuitConfig.standard_recursion_config_gl(**cfg_kw))
    a0, a1 = b.add_virtual_target(), b.add_virtual_target()
    prev, cur = a0, a1
    for _ in range(terms):
        prev, cur = cur, b.add(prev, cur)
    b.register_public_input(a0)
    b.register_public_input(a1)
    b.register_public_input(cur)
    pw = PartialWitness()
    pw.set_target(a0, 0)
    pw.set_target(a1, 1)
    return b, pw


def poly_chain_circuit(config, steps, x0=3):
    """no public inputs (so no in-circuit hash): y <- y * y + x, `steps` times; the result is tied to its constant value with a
    copy constraint.  Works for either field."""
    b = CircuitBuilder(config)
    p = b.F.p
    x = b.add_virtual_target()
    y, val = x, x0
    for _ in range(steps):
        y = b.mul_add(y, y, x)
        val = (val * val + x0) % p
    b.connect(y, b.constant(val))
    pw = PartialWitness()
    pw.set_target(x, x0)
    return b, pw


def babybear_public_input_circuit(steps=58, x0=5, **cfg_kw):
    """y <- y * i + x for i = 2.., with x and the result as public inputs: build() hashes them through a Poseidon2BabyBearGate"""
    b = CircuitBuilder(CircuitConfig.recursion_config_bb_narrow(**cfg_kw))
    x = b.add_virtual_target()
    cur = x
    for i in range(2, 2 + steps):
        cur = b.mul_add(cur, b.constant(i), x)
    b.register_public_input(x)
    b.register_public_input(cur)
    pw = PartialWitness()
    pw.set_target(x, x0)
    return b, pw


def oracle_circuit(built, num_public_inputs):
    cfg = built.config
    F = GL if cfg.field == N.GB_GOLDILOCKS else BB
    ocfg = PD.CircuitConfig(num_challenges=cfg.num_challenges, num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires,
                            num_constants=cfg.num_constants, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height,
                            proof_of_work_bits=cfg.proof_of_work_bits, num_query_rounds=cfg.num_query_rounds,
                            arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits,
                            max_quotient_degree_factor=cfg.max_quotient_degree_factor)
    oc = PD.BuiltCircuit(ocfg, F, built.degree_bits, built.constants_sigmas, built.k_is, built.gate_table, built.num_selectors,
                         num_public_inputs)
    bits = cfg.reduction_arity_bits(built.degree_bits) if hasattr(cfg, "reduction_arity_bits") else None
    if bits is not None:   # FriReductionStrategy::Fixed / MinSize: FriParams carries the list
        oc.reduction_arity_bits = list(bits)
    return oc


def recursion_gates_circuit(field=N.GB_GOLDILOCKS, seed=1, public_inputs=True, **cfg_kw):
    """One row of each gate of plonky2_goldibear_amd/recursion_gates.py with random inputs (the generators fill the rest), next to
    a short arithmetic chain.  Returns (builder, partial witness, {gate name: row})."""
    from plonky2_goldibear_amd import recursion_gates as R
    from plonky2_goldibear_amd.circuit_builder import wire
    gl = field == N.GB_GOLDILOCKS
    cfg = CircuitConfig.standard_recursion_config_gl(**cfg_kw) if gl else CircuitConfig.recursion_config_bb_narrow(**cfg_kw)
    b = CircuitBuilder(cfg)
    p, D = b.F.p, b.F.ext_degree
    rng = np.random.default_rng(seed)
    rnd = lambda: int(rng.integers(0, p, dtype=np.uint64))
    pw = PartialWitness()
    rows = {}

    def fill(row, cols):
        for c in cols:
            pw.set_target(wire(row, c), rnd())

    x = b.add_virtual_target()
    y = b.mul_add(x, x, b.constant(5))
    if public_inputs:
        b.register_public_input(x)
        b.register_public_input(y)
    pw.set_target(x, 3)

    g = R.ArithmeticExtensionGate.new_from_config(cfg)
    rows["arithmetic_extension"] = r = b.add_gate(g, [rnd(), rnd()])
    fill(r, [4 * D * i + k for i in range(g.num_ops) for k in range(3 * D)])
    g = R.MulExtensionGate.new_from_config(cfg)
    rows["mul_extension"] = r = b.add_gate(g, [rnd()])
    fill(r, [3 * D * i + k for i in range(g.num_ops) for k in range(2 * D)])
    g = R.BaseSumGate(min(63 if gl else 30, cfg.num_routed_wires - 1), 2)
    rows["base_sum"] = r = b.add_gate(g)
    pw.set_target(wire(r, 0), int(rng.integers(0, 1 << min(g.num_limbs, 62))))
    g = R.BaseSumGate(10, 4)
    rows["base_sum_4"] = r = b.add_gate(g)
    pw.set_target(wire(r, 0), int(rng.integers(0, 4 ** 10)))
    nc = min(cfg.num_routed_wires - 3 * D, (cfg.num_wires - 2 * D) // (D + 1))
    g = R.ReducingGate(nc, field)
    rows["reducing"] = r = b.add_gate(g)
    fill(r, list(range(D, 3 * D + nc)))
    nc = min((cfg.num_routed_wires - 3 * D) // D, (cfg.num_wires - 2 * D) // (2 * D))
    g = R.ReducingExtensionGate(nc, field)
    rows["reducing_extension"] = r = b.add_gate(g)
    fill(r, list(range(D, 3 * D + nc * D)))
    g = R.RandomAccessGate.new_from_config(cfg, 4 if gl else 3)
    rows["random_access"] = r = b.add_gate(g)
    for copy in range(g.num_copies):
        items = [rnd() for _ in range(g.vec_size)]
        idx = int(rng.integers(0, g.vec_size))
        pw.set_target(wire(r, g.wire_access_index(copy)), idx)
        pw.set_target(wire(r, g.wire_claimed_element(copy)), items[idx])
        for i, v in enumerate(items):
            pw.set_target(wire(r, g.wire_list_item(i, copy)), v)
    if gl:
        g = R.PoseidonMdsGate()
        rows["poseidon_mds"] = r = b.add_gate(g)
        fill(r, list(range(12 * D)))
    g = R.CosetInterpolationGate(4 if gl else 3, field, max_degree=6 if gl else 4)
    rows["coset_interpolation"] = r = b.add_gate(g)
    fill(r, [0] + list(range(1, 1 + g.num_points * D)) + list(range(g.start_point, g.start_point + D)))
    g = R.AddManyGate.new_from_config(cfg, 7)
    rows["add_many"] = r = b.add_gate(g)
    fill(r, [8 * i + j for i in range(g.num_ops) for j in range(7)])
    g = R.ApplyMat4Gate.new_from_config(cfg)
    rows["apply_mat4"] = r = b.add_gate(g)
    fill(r, [8 * D * op + k for op in range(g.num_ops) for k in range(4 * D)])
    if not gl:
        g = R.Poseidon2InternalPermutationGate()
        rows["poseidon2_internal_permutation"] = r = b.add_gate(g)
        fill(r, list(range(16 * D)))
    g = R.ExponentiationGate.new_from_config(cfg)
    rows["exponentiation"] = r = b.add_gate(g)
    fill(r, [0])
    for i in range(g.num_power_bits):
        pw.set_target(wire(r, 1 + i), int(rng.integers(0, 2)))
    return b, pw, rows
