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
    return PD.BuiltCircuit(ocfg, F, built.degree_bits, built.constants_sigmas, built.k_is, built.gate_table, built.num_selectors,
                           num_public_inputs)
