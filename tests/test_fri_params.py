"""FriParams.reduction_arity_bits beyond the stock ConstantArityBits strategy (fri/reduction_strategies.rs:11-160): the host
mirror's restatement of the three strategies, the library's setter / getter (gb_circuit_set_fri_reduction_arity_bits) on a
verify-only circuit object - no device - with the reference's own serialized proof, and the CPU oracle prover with a Fixed list.
No GPU."""
import itertools
import os

import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle import verifier as V
from oracle.fields import BB, GL
from plonky2_goldibear_amd import VerifierCircuitData, VerifyError, fri_params as FP, native as N

from test_abi_verify_fixture import _fixture_circuit


def test_constant_arity_bits_is_the_oracles():
    for lg, ab, fpb, r, cap in itertools.product((3, 8, 12, 14, 20), (1, 3, 4), (0, 5), (1, 3), (0, 4)):
        cfg = D.CircuitConfig(arity_bits=ab, final_poly_bits=fpb, rate_bits=r, cap_height=cap)
        try:
            mine = FP.constant_arity_bits(ab, fpb, lg, r, cap)
        except AssertionError:     # the reference's own assert!(degree_bits >= arity_bits): a configuration it panics on
            continue
        assert mine == D.reduction_arity_bits(cfg, lg)
        assert FP.reduction_arity_bits(("constant", ab, fpb), lg, r, cap, 28) == mine
    assert FP.reduction_arity_bits(("fixed", [3, 2, 1]), 12, 3, 4, 28) == [3, 2, 1]


@pytest.mark.parametrize("degree_bits,rate_bits,num_queries,max_bits", [(6, 3, 28, None), (9, 1, 84, 3), (10, 8, 10, None), (12, 3, 28, 2)])
def test_min_size_is_the_first_smallest_non_increasing_sequence(degree_bits, rate_bits, num_queries, max_bits):
    """MinSize (:58-127) against a brute force over every monotonically non-increasing arity sequence, visited in the reference's
    depth-first order with its strict `<` (the first of equally small sequences wins)."""
    got = FP.min_size_arity_bits(degree_bits, rate_bits, num_queries, max_bits)
    top = 4 if max_bits is None else max_bits

    def sequences(prefix, cap):
        yield prefix
        for nxt in range(1, min(cap, degree_bits - sum(prefix)) + 1):
            yield from sequences(prefix + [nxt], nxt)
    best, best_size = None, None
    for seq in sequences([], top):     # pre-order = the helper's own visiting order
        size = FP.relative_proof_size(degree_bits, rate_bits, num_queries, seq)
        if best_size is None or size < best_size:
            best, best_size = seq, size
    assert FP.relative_proof_size(degree_bits, rate_bits, num_queries, got) == best_size
    assert got == sorted(got, reverse=True) and sum(got) <= degree_bits
    assert FP.reduction_arity_bits(("min_size", max_bits), degree_bits, rate_bits, 4, num_queries) == got


def test_setter_on_the_reference_proofs_circuit(golden_dir):
    """The regression fixture's CommonCircuitData carries fri_params.reduction_arity_bits = [4, 4, 4] (its strategy is
    ConstantArityBits(4, 5), degree_bits 14, cap_height 4): handed over explicitly the reference's proof still verifies; any
    other list changes the proof's shape; lists the reference would panic on are rejected."""
    circ, cd, raw = _fixture_circuit(golden_dir)
    want = cd["fri_params"]["reduction_arity_bits"]
    assert circ.reduction_arity_bits == want == [4, 4, 4]
    circ.set_reduction_arity_bits(want)
    assert circ.verify(raw)
    for other in ([4, 4], [4, 4, 3], [3, 3, 3, 3]):
        circ.set_reduction_arity_bits(other)
        assert circ.reduction_arity_bits == other
        with pytest.raises((N.ShapeError, VerifyError)):
            circ.verify(raw)
    circ.set_reduction_arity_bits(want)
    assert circ.verify(raw)
    for bad in ([0], [9], [4] * 33, [8, 7], [4, 4, 4, 4]):     # zero / too wide / too many / past degree_bits / tree below the cap
        with pytest.raises(N.GoldibearError) as e:
            circ.set_reduction_arity_bits(bad)
        assert e.value.status == N.GB_ERR_INVALID
        assert circ.reduction_arity_bits == want              # untouched by a rejected list
    circ.set_reduction_arity_bits([])                          # no reduction at all is a valid FriParams
    assert circ.reduction_arity_bits == []
    circ.free()


@pytest.mark.parametrize("F,bits", [(GL, [2, 1, 1]), (GL, []), (BB, [3, 1])])
def test_oracle_prover_with_a_fixed_list(F, bits):
    """the CPU oracle prover and verifier with FriReductionStrategy::Fixed: accepted with the same list, rejected with another"""
    cfg = D.CircuitConfig(num_challenges=2) if F is GL else D.CircuitConfig.babybear(6)
    circ = D.DummyCircuit(6, cfg, F=F)
    circ.reduction_arity_bits = list(bits)
    w = circ.witness(seed=3)
    proof, _ = D.prove_cpu(circ, w)
    assert D.verify(circ, proof)
    stock = D.DummyCircuit(6, cfg, F=F)
    assert stock.reduction_arity_bits != bits
    with pytest.raises(Exception):
        D.verify(stock, proof)


def test_library_mirrors_the_assert_of_constant_arity_bits(golden_dir):
    """fri/reduction_strategies.rs:45 `assert!(degree_bits >= arity_bits)`: the fixture's circuit has cap_height 4; with
    ConstantArityBits(5, 0) at rate_bits 8 a reduction comes up that needs 5 bits where fewer are left while the tree would still
    be high enough - the reference panics in build().  The library never derives a list that wraps around: since round 6
    gb_verifier_create (no device) leaves the list OPEN for such a pair - a circuit with a Fixed / MinSize strategy has no meaningful
    ConstantArityBits parameters and hands its list over after the create call - and everything that needs the list answers
    GB_ERR_INVALID, naming the reference's assert, until gb_circuit_set_fri_reduction_arity_bits has been called."""
    _, cd, _ = _fixture_circuit(golden_dir)
    lg = cd["fri_params"]["degree_bits"]
    with pytest.raises(AssertionError):
        FP.constant_arity_bits(5, 0, lg, 8, 4)
    circ, _, raw = _fixture_circuit(golden_dir, arity_bits=5, final_poly_bits=0, rate_bits=8)
    assert circ.reduction_arity_bits == []
    with pytest.raises(N.GoldibearError) as e:
        circ.verify(raw)
    assert e.value.status == N.GB_ERR_INVALID and "fri/reduction_strategies.rs:45" in str(e.value) and "ConstantArityBits" in str(e.value)
    circ.free()
    for ab, fpb in ((4, 0), (3, 1), (7, 2), (8, 5)):    # pairs the reference accepts: the library derives the same list
        want = FP.constant_arity_bits(ab, fpb, lg, 3, 4)
        circ, _, _ = _fixture_circuit(golden_dir, arity_bits=ab, final_poly_bits=fpb)
        assert circ.reduction_arity_bits == want
