"""Host mirror of the recursion gate set (plonky2_goldibear_amd/recursion_gates.py) against the oracle's evaluators, which the
reference's regression proof pins (tests/test_oracle_fixture.py): on a generated witness every constraint of every row
evaluates to zero (the reference's generator <-> eval_unfiltered contract), a perturbed wire breaks its gate, and the gate
ids / sort order / selector groups of a recursion-shaped gate set come out as in the reference's fixture.  No GPU."""
import numpy as np
import pytest

from oracle import gates as G
from oracle import verifier as V
from oracle.fields import BB, GL
from plonky2_goldibear_amd import native as N

from circuits import recursion_gates_circuit


def _row_constraints(c, F, wires, row):
    sel = c.num_selectors
    consts = [F.efrom(int(c.constants_sigmas[sel + i, row])) for i in range(c.max_constants)]
    w = [F.efrom(int(v)) for v in wires[:, row]]
    sel_vals = [int(c.constants_sigmas[i, row]) for i in range(sel)]
    out = {}
    for gi, g in enumerate(c.gate_table):
        if g[0] in (G.NOOP, G.PUBLIC_INPUT) or sel_vals[g[2]] != gi:
            continue
        out[gi] = G.eval_unfiltered(F, g, w, consts, None)
    return out


@pytest.mark.parametrize("field", [N.GB_GOLDILOCKS, N.GB_BABYBEAR])
def test_generated_witness_satisfies_every_gate(field):
    F = GL if field == N.GB_GOLDILOCKS else BB
    b, pw, rows = recursion_gates_circuit(field)
    c = b.build()
    wires, pis = c.generate_witness(pw)
    kinds = {g[0] for g in c.gate_table}
    want = {G.ARITHMETIC_EXTENSION, G.MUL_EXTENSION, G.BASE_SUM, G.REDUCING, G.REDUCING_EXTENSION, G.RANDOM_ACCESS,
            G.COSET_INTERPOLATION, G.EXPONENTIATION, G.ADD_MANY, G.APPLY_MAT4} | (
        {G.POSEIDON_MDS} if F is GL else {G.POSEIDON2_INTERNAL_PERMUTATION})
    assert want <= kinds
    seen = set()
    for row in range(1 << c.degree_bits):
        for gi, cons in _row_constraints(c, F, wires, row).items():
            g = c.gate_table[gi]
            assert len(cons) == G.num_constraints(g, F.hout, F.D), c.gate_ids[gi]
            assert all(x == F.zero for x in cons), "%s row %d" % (c.gate_ids[gi], row)
            seen.add(g[0])
    assert want <= seen
    # a perturbed dependent wire breaks exactly its own row's gate
    for name, row in rows.items():
        gate = b.gate_instances[row][0]
        col = gate.num_wires - 1
        bad = wires.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % F.P
        cons = _row_constraints(c, F, bad, row)
        assert any(x != F.zero for v in cons.values() for x in v), name


def test_gate_order_and_selector_groups_match_the_reference_fixture(golden_dir):
    """The fixture's CommonCircuitData lists its gates sorted by (degree, id) with selectors_info; a builder holding the same
    gate structs must produce the same order and groups (circuit_builder.rs:1194-1196, gates/selectors.rs:125-209)."""
    import os
    from plonky2_goldibear_amd import recursion_gates as R
    from plonky2_goldibear_amd.circuit_builder import (ArithmeticGate, CircuitConfig, NoopGate, PoseidonGate, PublicInputGate,
                                                       selector_polynomials)
    common = open(os.path.join(golden_dir, "recursive_verifier_gl_common_data.bin"), "rb").read()
    cd = V.read_common_data(common)
    ref = V.read_gates(common, cd)
    cfg = CircuitConfig.standard_recursion_config_gl()
    mine = [NoopGate(), PoseidonGate(), PublicInputGate(4), ArithmeticGate.new_from_config(cfg), R.PoseidonMdsGate(),
            R.BaseSumGate(63, 2), R.ReducingExtensionGate(32), R.ReducingGate(43), R.ArithmeticExtensionGate.new_from_config(cfg),
            R.MulExtensionGate.new_from_config(cfg), R.RandomAccessGate.new_from_config(cfg, 4),
            R.CosetInterpolationGate(4, max_degree=6)]
    gates = sorted(mine, key=lambda g: (g.degree, g.id))
    _, sel, groups = selector_polynomials(gates, [(gates[0], [])], cfg.max_quotient_degree_factor + 1, GL.P)
    got = [(g.kind, g.param, sel[i], groups[sel[i]][0], groups[sel[i]][1], getattr(g, "param2", 0), getattr(g, "param3", 0))
           for i, g in enumerate(gates)]
    assert got == ref
    assert cd["num_gate_constraints"] == max(g.num_constraints for g in gates)
    # the interpolation gate's weights are the serialized ones
    ci = next(g for g in gates if g.kind == G.COSET_INTERPOLATION)
    assert ci.barycentric_weights == G.barycentric_weights(GL, 4)[1]
    assert (ci.degree, ci.num_intermediates) == (6, 2)


@pytest.mark.parametrize("field", [N.GB_GOLDILOCKS, N.GB_BABYBEAR])
def test_oracle_prover_evaluates_the_recursion_gates_on_the_coset(field):
    """The CPU oracle prover with the gate terms of oracle/plonk_dummy.gate_constraint_terms (every gate of the recursion set at
    every LDE point, evaluators of oracle/gates.py): its proof satisfies the vanishing identity under the pinned verifier, and a
    witness that breaks one gate row does not - the oracle side of tests/test_gpu_recursion_gates.py, runnable without a GPU."""
    from oracle import plonk_dummy as PD
    from circuits import oracle_circuit, recursion_gates_circuit
    b, pw, rows = recursion_gates_circuit(field, seed=11)
    c = b.build(None)
    w, pis = c.generate_witness(pw)
    oc = oracle_circuit(c, len(pis))
    terms = PD.gate_constraint_terms(oc, w, pis)
    n, r = 1 << c.degree_bits, oc.cfg.rate_bits
    assert terms.shape[0] == n << r and terms.any()       # off the subgroup the constraint polynomials do not vanish
    proof, _ = PD.prove_cpu(oc, w, pis)
    assert PD.verify(oc, proof)
    row = rows["coset_interpolation"]
    gate = b.gate_instances[row][0]
    bad = w.copy()
    bad[gate.num_wires - 1, row] = (int(bad[gate.num_wires - 1, row]) + 1) % oc.F.P
    with pytest.raises(AssertionError, match="vanishing"):
        PD.verify(oc, PD.prove_cpu(oc, bad, pis)[0])
