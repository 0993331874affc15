"""The CPU oracle prover for the dummy circuit, checked by the restated verifier (no GPU).

The verifier restatement (oracle/verifier.py) is pinned by the reference's serialized regression
proof (tests/test_oracle_fixture.py); here it accepts the oracle prover's proofs, including the
PLONK identity vanishing(zeta) = Z_H(zeta) * quotient(zeta) for the dummy circuit's gate set."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle import verifier as V


@pytest.mark.parametrize("degree_bits,num_challenges", [(3, 2), (5, 2), (8, 2), (10, 3)])
def test_oracle_proof_verifies(degree_bits, num_challenges):
    circ = D.DummyCircuit(degree_bits, D.CircuitConfig(num_challenges=num_challenges))
    proof, dbg = D.prove_cpu(circ, circ.witness(seed=degree_bits))
    stats = {}
    assert D.verify(circ, proof, stats)
    assert stats["merkle_paths"] == 28 * (4 + len(circ.reduction_arity_bits))
    # byte round trip through the restated (de)serialiser
    pr, pis = V.read_proof_with_pis(proof, circ.common_data())
    assert V.write_proof_with_pis(pr, pis) == proof
    # PoW response has >= 16 leading zeros and the nonce is minimal by construction
    assert int(dbg[3 * num_challenges + 4]).bit_length() <= 48


def test_tampered_proofs_are_rejected():
    circ = D.DummyCircuit(5)
    proof, _ = D.prove_cpu(circ, circ.witness())
    cd = circ.common_data()
    pr, pis = V.read_proof_with_pis(proof, cd)
    pr["openings"]["wires"][7] = (pr["openings"]["wires"][7][0] ^ 1, pr["openings"]["wires"][7][1])
    with pytest.raises(AssertionError):
        D.verify(circ, V.write_proof_with_pis(pr, pis))
    pr, pis = V.read_proof_with_pis(proof, cd)
    pr["opening_proof"]["pow_witness"] += 1
    with pytest.raises(AssertionError):
        D.verify(circ, V.write_proof_with_pis(pr, pis))


def test_bad_witness_fails_the_identity():
    # a witness violating the PublicInputGate constraint (wire 0 != pi_hash[0] = 0) or a copy
    # constraint must not verify (the quotient is then not a polynomial of the right degree)
    circ = D.DummyCircuit(5)
    w = circ.witness()
    w[0, circ.pi_row] = 5
    proof, _ = D.prove_cpu(circ, w)
    with pytest.raises(AssertionError):
        D.verify(circ, proof)


def test_witness_seed_changes_proof_but_not_digest():
    circ = D.DummyCircuit(4)
    p0, _ = D.prove_cpu(circ, circ.witness(0))
    p1, _ = D.prove_cpu(circ, circ.witness(1))
    assert p0 != p1
    assert D.verify(circ, p0) and D.verify(circ, p1)
    assert np.array_equal(circ.circuit_digest, D.DummyCircuit(4).circuit_digest)


@pytest.mark.parametrize("field_name,degree_bits,rate_bits,qdf", [("goldilocks", 5, 4, 8), ("goldilocks", 6, 7, 8), ("goldilocks", 7, 8, 8),
                                                                  ("goldilocks", 6, 5, 16), ("babybear", 5, 5, 8), ("babybear", 7, 8, 8)])
def test_rate_above_the_quotient_degree(field_name, degree_bits, rate_bits, qdf):
    """step = 2^(rate_bits - log2 quotient_degree_factor) > 1 (plonk/prover.rs:735-749; the reference's size-optimised recursion
    proofs run at rate_bits 7 and 8): the quotient computed on every step-th LDE point satisfies the verifier's identity at zeta,
    with the verifier that the reference's own regression proof pins."""
    from oracle.fields import BB, GL
    F = GL if field_name == "goldilocks" else BB
    mk = D.CircuitConfig if F is GL else D.CircuitConfig.babybear
    circ = D.DummyCircuit(degree_bits, mk(rate_bits=rate_bits, max_quotient_degree_factor=qdf), F=F)
    proof, _ = D.prove_cpu(circ, circ.witness(seed=rate_bits))
    stats = {}
    assert D.verify(circ, proof, stats)
    pr, pis = V.read_proof_with_pis(proof, circ.common_data(), F)
    assert len(pr["openings"]["quotient_polys"]) == circ.cfg.num_challenges * qdf
    # a constraint violation is still caught on the subsampled domain
    w = circ.witness(seed=rate_bits)
    w[0, circ.pi_row] = 5
    bad, _ = D.prove_cpu(circ, w)
    with pytest.raises(AssertionError):
        D.verify(circ, bad)
