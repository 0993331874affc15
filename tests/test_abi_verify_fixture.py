"""gb_verify (the product's host-side verifier, csrc/verifier_host.inc + the gate evaluators of csrc/gates.hpp) on the
reference's OWN serialized recursion proof (recursion/regression_test_data.rs:5,62,93; verified by the reference at
recursion/recursive_verifier.rs:280-314).  gb_verifier_create touches no device, so this runs without a GPU.

The proof is zero-knowledge (salted leaves) and its circuit uses twelve gates - Noop, PoseidonMds, PublicInput, BaseSum<2>,
ReducingExtension, Reducing, ArithmeticExtension, Arithmetic, MulExtension, RandomAccess, CosetInterpolation, Poseidon - so the
vanishing identity at zeta pins the extension-field instantiation of every one of those evaluators with the reference's numbers;
the quotient kernel runs the same source over the base field (tests/test_gpu_gates.py compares the two on the GPU)."""
import os

import numpy as np
import pytest

from oracle import verifier as V
from plonky2_goldibear_amd import VerifierCircuitData, VerifyError, native as N


def _fixture_circuit(golden_dir, **override):
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    common = rd("recursive_verifier_gl_common_data.bin")
    cd = V.read_common_data(common)
    vd = V.read_verifier_data(rd("recursive_verifier_gl_verifier_data.bin"))
    gates = V.read_gates(common, cd)
    cfg, fc = cd["config"], cd["config"]["fri_config"]
    assert fc["reduction_strategy"] == ("ConstantArityBits", 4, 5)
    nsel = len(cd["selectors_info"]["groups"])
    kw = dict(num_wires=cfg["num_wires"], num_routed_wires=cfg["num_routed_wires"], num_constants=cd["num_constants"] - nsel,
              num_challenges=cfg["num_challenges"], max_quotient_degree_factor=cd["quotient_degree_factor"],
              rate_bits=fc["rate_bits"], cap_height=fc["cap_height"], proof_of_work_bits=fc["proof_of_work_bits"],
              num_query_rounds=fc["num_query_rounds"], arity_bits=4, final_poly_bits=5, num_selectors=nsel,
              zero_knowledge=cd["fri_params"]["hiding"], num_public_inputs=cd["num_public_inputs"])
    kw.update(override)
    circ = VerifierCircuitData(cd["fri_params"]["degree_bits"], kw.pop("gates", gates), np.array(cd["k_is"], dtype=np.uint64),
                               np.array(vd["constants_sigmas_cap"], dtype=np.uint64),
                               np.array(vd["circuit_digest"], dtype=np.uint64), **kw)
    return circ, cd, rd("recursive_verifier_gl_proof.bin")


def test_reference_regression_proof_verifies_through_the_c_abi(golden_dir):
    circ, cd, raw = _fixture_circuit(golden_dir)
    assert circ.verify(raw)


def test_tampered_opening_fails_the_vanishing_identity(golden_dir):
    circ, cd, raw = _fixture_circuit(golden_dir)
    proof, pis = V.read_proof_with_pis(raw, cd)
    w0 = proof["openings"]["wires"][17]
    proof["openings"]["wires"][17] = ((w0[0] + 1) % V.P, w0[1])
    with pytest.raises(VerifyError, match="vanishing"):
        circ.verify(V.write_proof_with_pis(proof, pis))
    with pytest.raises(N.ShapeError):
        circ.verify(raw[:-9])


@pytest.mark.parametrize("victim", range(1, 12))
def test_every_gate_evaluator_is_pinned(golden_dir, victim):
    """Replace one gate of the set by a NoopGate (its constraints drop out of the sum): the identity must fail, i.e. the
    reference's proof pins each evaluator separately."""
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    common = rd("recursive_verifier_gl_common_data.bin")
    gates = V.read_gates(common, V.read_common_data(common))
    g = gates[victim]
    gates[victim] = (0, 0) + tuple(g[2:5]) + (0, 0)
    circ, cd, raw = _fixture_circuit(golden_dir, gates=gates)
    with pytest.raises(VerifyError, match="vanishing"):
        circ.verify(raw)


def test_wrong_salt_setting_is_malformed(golden_dir):
    circ, cd, raw = _fixture_circuit(golden_dir, zero_knowledge=False)
    with pytest.raises((N.ShapeError, VerifyError)):
        circ.verify(raw)


def test_malformed_path_lengths_are_rejected_not_read(golden_dir):
    """A Merkle path shorter than its tree's depth would leave an index beyond the cap (hash/merkle_proofs.rs:62-75 indexes the
    cap with the remaining bits): the library refuses the bytes instead of reading past the cap."""
    circ, cd, raw = _fixture_circuit(golden_dir)
    proof, pis = V.read_proof_with_pis(raw, cd)
    vals, path = proof["opening_proof"]["query_round_proofs"][3]["initial_trees_proof"][2]
    proof["opening_proof"]["query_round_proofs"][3]["initial_trees_proof"][2] = (vals, path[:-2])
    bad = V.write_proof_with_pis(proof, pis)
    with pytest.raises(N.ShapeError, match="wrong length"):
        circ.verify(bad)
    with pytest.raises(N.ShapeError, match="wrong length"):
        circ.compress(bad)


def test_public_input_count_is_checked(golden_dir):
    """plonk/validate_shape.rs:22-25: hash_no_pad does not pad, so the reference's proof re-serialised with a zero public
    input appended (or, for a circuit whose last public input is zero, stripped) keeps its transcript and its PublicInputGate
    constraint - only the count check rejects it.  Same for the compressed form, which carries no count at all."""
    circ, cd, raw = _fixture_circuit(golden_dir)
    proof, pis = V.read_proof_with_pis(raw, cd)
    assert len(pis) == cd["num_public_inputs"]
    padded = V.write_proof_with_pis(proof, list(pis) + [0])
    with pytest.raises(N.ShapeError, match="public inputs"):
        circ.verify(padded)
    two, _, _ = _fixture_circuit(golden_dir, num_public_inputs=2)   # the stripped direction: [0, 0] -> [0] -> []
    with pytest.raises(N.ShapeError, match="public inputs"):
        two.verify(V.write_proof_with_pis(proof, [0]))
    small = circ.compress(raw)
    assert circ.verify_compressed(small)
    with pytest.raises(N.ShapeError, match="public inputs"):
        circ.verify_compressed(small + bytes(8))
    with pytest.raises(N.ShapeError, match="public inputs"):
        circ.decompress(small + bytes(8))
    # a verifier told the wrong count refuses the honest proof
    other, _, _ = _fixture_circuit(golden_dir, num_public_inputs=cd["num_public_inputs"] + 1)
    with pytest.raises(N.ShapeError, match="public inputs"):
        other.verify(raw)


def test_a_circuit_without_constant_arity_bits_waits_for_its_list(golden_dir):
    """a circuit whose FriReductionStrategy is Fixed(..) / MinSize(..) has no meaningful (arity_bits, final_poly_bits) -
    it passes arity_bits = 0 (or a pair ConstantArityBits would panic on, fri/reduction_strategies.rs:45), the create call succeeds,
    everything that needs the list answers GB_ERR_INVALID until gb_circuit_set_fri_reduction_arity_bits has handed it over, and the
    reference's own proof - whose CommonCircuitData carries [4, 4, 4] - verifies afterwards."""
    circ, cd, raw = _fixture_circuit(golden_dir, arity_bits=0, final_poly_bits=0)
    with pytest.raises(N.ShapeError, match="gb_circuit_set_fri_reduction_arity_bits"):
        circ.verify(raw)
    with pytest.raises(N.ShapeError, match="gb_circuit_set_fri_reduction_arity_bits"):
        circ.compress(raw)
    assert circ.reduction_arity_bits == []
    circ.set_reduction_arity_bits([4, 4, 4])
    assert circ.verify(raw)
    circ.free()
