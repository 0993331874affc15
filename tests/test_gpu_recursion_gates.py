"""GPU prove() of circuits that use the recursion gate set (csrc/gates.hpp: ArithmeticExtension, MulExtension, BaseSum,
Reducing, ReducingExtension, RandomAccess, PoseidonMds, CosetInterpolation, Exponentiation, AddMany, ApplyMat4,
Poseidon2InternalPermutation), both fields, through the C ABI.

Parity bar (round 3): an ORACLE COMPARISON on the prover's coset.  oracle/plonk_dummy.gate_constraint_terms evaluates every gate
of the set at every LDE point with the evaluators of oracle/gates.py - the ones the reference's own regression proof pins one by
one (tests/test_oracle_fixture.py) - and the CPU oracle prover builds its quotient from them (plonk/vanishing_poly.rs:741-774):
the GPU's quotient chunk coefficients (gb_quotient_polys) equal the oracle's element for element, and the proof bytes are
identical.  On top of that the proof is accepted by the oracle verifier and by gb_verify, and a perturbed gate row is rejected
by both.  The 2^12-row circuit below stays acceptance-only (the pure-Python evaluators take ~30 ms per LDE point).  -m gpu only."""
import numpy as np
import pytest

from oracle import plonk_dummy as PD
from plonky2_goldibear_amd import GpuContext, ShapeError, VerifyError
from plonky2_goldibear_amd import native as N
from plonky2_goldibear_amd import recursion_gates as R
from plonky2_goldibear_amd.circuit_builder import CircuitBuilder, CircuitConfig, PartialWitness, wire

from circuits import oracle_circuit, recursion_gates_circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("field,public_inputs", [(N.GB_GOLDILOCKS, True), (N.GB_GOLDILOCKS, False), (N.GB_BABYBEAR, True),
                                                 (N.GB_BABYBEAR, False)])
def test_recursion_gate_rows_equal_the_oracle_prover(ctx, field, public_inputs):
    b, pw, rows = recursion_gates_circuit(field, seed=7, public_inputs=public_inputs)
    c = b.build(ctx)
    w, pis = c.generate_witness(pw)
    proof = c.data.prove(w, pis, random_wire=(c.random_wire[1], c.random_wire[0]))
    assert c.data.verify(proof)
    oc = oracle_circuit(c, len(pis))
    assert (c.data.circuit_digest == oc.circuit_digest).all()
    # the oracle prover with every gate evaluated on the whole coset by oracle/gates.py: same quotient, same bytes
    oc.set_cap(c.data.constants_sigmas_cap)
    dump, mid = {}, {}
    want, _ = PD.prove_cpu(oc, w, pis, dump=dump)
    assert proof == want
    from test_gpu_stage_abi import prove_by_stages
    assert prove_by_stages(c.data, oc, w, pis, field, mid) == want
    assert (mid["zs_partial_products"] == dump["zs_partial_products"]).all()
    assert (mid["quotient_chunks"] == dump["quotient_chunks"]).all()
    stats = {}
    assert PD.verify(oc, proof, stats)
    assert c.data.prove(w, pis) == proof  # deterministic
    # every gate row is live: a perturbed dependent wire makes the proof fail the identity under both verifiers
    for name, row in rows.items():
        gate = b.gate_instances[row][0]
        bad = w.copy()
        p = 0xFFFFFFFF00000001 if field == N.GB_GOLDILOCKS else 2013265921
        bad[gate.num_wires - 1, row] = (int(bad[gate.num_wires - 1, row]) + 1) % p
        bad_proof = c.data.prove(bad, pis)
        with pytest.raises(VerifyError, match="vanishing"):
            c.data.verify(bad_proof)
        with pytest.raises(AssertionError, match="vanishing"):
            PD.verify(oc, bad_proof)
    c.data.free()


def test_larger_circuit_with_many_interpolation_and_random_access_rows(ctx):
    """2^12 rows: 1500 rows each of the two highest-degree recursion gates next to an arithmetic chain."""
    cfg = CircuitConfig.standard_recursion_config_gl()
    b = CircuitBuilder(cfg)
    rng = np.random.default_rng(3)
    p = b.F.p
    rnd = lambda: int(rng.integers(0, p, dtype=np.uint64))
    pw = PartialWitness()
    ci = R.CosetInterpolationGate(4, max_degree=6)
    ra = R.RandomAccessGate.new_from_config(cfg, 4)
    for _ in range(1500):
        r = b.add_gate(ci)
        for col in [0] + list(range(1, 1 + 32)) + [ci.start_point, ci.start_point + 1]:
            pw.set_target(wire(r, col), rnd())
        r = b.add_gate(ra)
        for copy in range(ra.num_copies):
            items = [rnd() for _ in range(16)]
            idx = int(rng.integers(0, 16))
            pw.set_target(wire(r, ra.wire_access_index(copy)), idx)
            pw.set_target(wire(r, ra.wire_claimed_element(copy)), items[idx])
            for i, v in enumerate(items):
                pw.set_target(wire(r, ra.wire_list_item(i, copy)), v)
    x = b.add_virtual_target()
    b.register_public_input(b.mul(x, x))
    pw.set_target(x, 9)
    c = b.build(ctx)
    assert c.degree_bits == 12
    w, pis = c.generate_witness(pw)
    proof = c.data.prove(w, pis)
    assert c.data.verify(proof)
    assert PD.verify(oracle_circuit(c, len(pis)), proof)
    c.data.free()


def test_unsupported_and_malformed_gates(ctx):
    from plonky2_goldibear_amd.prover import CircuitData
    cs = np.zeros((1 + 2 + 80, 8), dtype=np.uint64)
    k = np.ones(80, dtype=np.uint64)
    mk = lambda gates, **kw: CircuitData(ctx, 3, cs, k, gates=gates, num_selectors=1, **kw)
    with pytest.raises(N.GoldibearError):   # LookupGate and friends: no evaluator
        mk([(0, 0, 0, 0, 2, 0, 0), (99, 1, 0, 0, 2, 0, 0)])
    with pytest.raises(ShapeError):         # degree that with_max_degree() never yields
        mk([(0, 0, 0, 0, 2, 0, 0), (13, 4, 0, 0, 2, 7, 0)])
    with pytest.raises(ShapeError):         # ReducingGate with more coefficients than wires
        mk([(0, 0, 0, 0, 2, 0, 0), (9, 60, 0, 0, 2, 0, 0)])
    with pytest.raises(N.GoldibearError):   # PoseidonMdsGate is Goldilocks only
        CircuitData.babybear(ctx, 3, np.zeros((1 + 2 + 41, 8), dtype=np.uint32), np.ones(41, dtype=np.uint32),
                             gates=[(0, 0, 0, 0, 2, 0, 0), (12, 0, 0, 0, 2, 0, 0)], num_selectors=1)
