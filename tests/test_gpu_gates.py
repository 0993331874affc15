"""GPU prove() for general gate sets (gb_circuit_create_gates: ArithmeticGate, PoseidonGate next to the dummy trio), through
the C ABI: proof BYTES identical to the CPU oracle prover's on the same circuit and witness, accepted by gb_verify and by
the oracle verifier; the reference's `factorial` and `fibonacci` examples are the circuits.  -m gpu only."""
import math

import numpy as np
import pytest

from oracle import plonk_dummy as PD
from oracle.fields import GL
from plonky2_goldibear_amd import CircuitData, GpuContext, ShapeError, VerifyError
from plonky2_goldibear_amd import native as N
from plonky2_goldibear_amd.circuit_builder import CircuitConfig

from circuits import babybear_public_input_circuit, factorial_circuit, fibonacci_circuit, oracle_circuit, poly_chain_circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("make,kw", [(factorial_circuit, {}), (fibonacci_circuit, {}), (factorial_circuit, dict(count=1500)),
                                     (factorial_circuit, dict(num_challenges=3))])
def test_example_circuits_bytes_match_oracle(ctx, make, kw):
    b, pw = make(**kw)
    c = b.build(ctx)
    w, pis = c.generate_witness(pw)
    oc = oracle_circuit(c, len(pis))
    assert (c.data.circuit_digest == oc.circuit_digest).all()
    assert (c.data.constants_sigmas_cap == oc.constants_sigmas_cap).all()
    want, _ = PD.prove_cpu(oc, w, pis)
    got = c.data.prove(w, pis)
    assert got == want
    assert c.data.verify(got)
    assert PD.verify(oc, got)
    if make is factorial_circuit and not kw:
        assert pis == [1, math.factorial(100) % GL.P]
    # BuiltCircuit.prove = witness generation + prove
    assert c.prove(pw) == want


def test_bad_witness_and_public_inputs_rejected(ctx):
    b, pw = factorial_circuit()
    c = b.build(ctx)
    w, pis = c.generate_witness(pw)
    good = c.data.prove(w, pis)
    assert c.data.verify(good)
    # a proof for other public inputs: the PublicInputGate constraint fails
    with pytest.raises(VerifyError, match="vanishing"):
        c.data.verify(c.data.prove(w, [pis[0], (pis[1] + 1) % GL.P]))
    # a wrong product in an ArithmeticGate row / a wrong s-box input in the PoseidonGate row
    arith = next(r for r, (g, _) in enumerate(b.gate_instances) if g.kind == 3)
    pos = next(r for r, (g, _) in enumerate(b.gate_instances) if g.kind == 4)
    for col, row in ((3, arith), (60, pos), (24, pos)):
        w2 = w.copy()
        w2[col, row] = (int(w2[col, row]) + 1) % GL.P
        with pytest.raises(VerifyError, match="vanishing"):
            c.data.verify(c.data.prove(w2, pis))


def test_babybear_arithmetic_bytes_match_oracle(ctx):
    b, pw = poly_chain_circuit(CircuitConfig.recursion_config_bb_narrow(), steps=150)
    c = b.build(ctx)
    w, pis = c.generate_witness(pw)
    oc = oracle_circuit(c, 0)
    want, _ = PD.prove_cpu(oc, w, pis)
    got = c.data.prove(w, pis)
    assert got == want
    assert c.data.verify(got) and PD.verify(oc, got)


@pytest.mark.parametrize("kw", [{}, dict(steps=2000, num_challenges=7)])
def test_babybear_public_inputs_bytes_match_oracle(ctx, kw):
    """a BabyBear circuit with public inputs: Poseidon2BabyBearGate row + two selector groups"""
    b, pw = babybear_public_input_circuit(**kw)
    c = b.build(ctx)
    w, pis = c.generate_witness(pw)
    oc = oracle_circuit(c, 2)
    assert (c.data.circuit_digest == oc.circuit_digest).all()
    want, _ = PD.prove_cpu(oc, w, pis)
    got = c.data.prove(w, pis)
    assert got == want
    assert c.data.verify(got) and PD.verify(oc, got)
    row = next(r for r, (g, _) in enumerate(b.gate_instances) if g.kind == 5)
    w[70, row] ^= 1
    with pytest.raises(VerifyError, match="vanishing"):
        c.data.verify(c.data.prove(w, pis))


def test_goldilocks_arithmetic_only_single_selector(ctx):
    b, pw = poly_chain_circuit(CircuitConfig.standard_recursion_config_gl(), steps=3000)
    c = b.build(ctx)
    assert c.num_selectors == 1 and c.degree_bits == 8
    w, pis = c.generate_witness(pw)
    oc = oracle_circuit(c, 0)
    want, _ = PD.prove_cpu(oc, w, pis)
    assert c.data.prove(w, pis) == want


def test_general_path_equals_dummy_path(ctx):
    """the dummy circuit through gb_circuit_create_gates (gate kernel + k_quotient) gives the same bytes as through
    gb_circuit_create (gates evaluated inside k_quotient)"""
    circ = PD.DummyCircuit(9, PD.CircuitConfig(num_challenges=2))
    cfg = circ.cfg
    kw = dict(num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants,
              num_challenges=cfg.num_challenges)
    a = CircuitData(ctx, 9, circ.constants_sigmas, circ.k_is, gate_constant=circ.GATE_CONSTANT, gate_pi=circ.GATE_PI, **kw)
    g = CircuitData(ctx, 9, circ.constants_sigmas, circ.k_is, gates=circ.gate_table, **kw)
    w = circ.witness(seed=5)
    pa, pg = a.prove(w), g.prove(w)
    assert pa == pg and g.verify(pa) and a.verify(pg)


def test_gate_table_validation(ctx):
    b, pw = factorial_circuit()
    c = b.build()
    kw = dict(num_constants=c.max_constants, num_selectors=c.num_selectors)
    with pytest.raises(N.GoldibearError, match="no constraint evaluator"):   # e.g. LookupGate
        CircuitData(ctx, c.degree_bits, c.constants_sigmas, c.k_is, gates=[(99, 0, 0, 0, 1)] + c.gate_table[1:], **kw)
    with pytest.raises(ShapeError, match="param 0"):
        CircuitData(ctx, c.degree_bits, c.constants_sigmas, c.k_is, gates=[(9, 0, 0, 0, 1)] + c.gate_table[1:], **kw)
    with pytest.raises(ShapeError, match="selector group"):
        CircuitData(ctx, c.degree_bits, c.constants_sigmas, c.k_is, gates=[(0, 0, 0, 1, 4)] + c.gate_table[1:], **kw)
    with pytest.raises(ShapeError, match="needs"):
        CircuitData(ctx, c.degree_bits, c.constants_sigmas, c.k_is, gates=c.gate_table[:3] + [(3, 40, 0, 0, 4)] + c.gate_table[4:], **kw)
