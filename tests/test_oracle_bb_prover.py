"""BabyBear (D = 4, Poseidon2-16, H = 8) instantiation of the CPU oracle prover, checked by the restated
verifier (no GPU).  PARITY UNPINNED for the BabyBear field constants (oracle/oracle_bb.c header): what is
checked here is self-consistency of the transcript, the PLONK identity at zeta and FRI, not reference bytes."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle import verifier as V
from oracle.fields import BB


def test_ext4_is_a_field():
    # x^4 - 11 irreducible over BabyBear: 11 is not a square and p = 1 mod 4 (binomial criterion), spot-check inverses
    assert pow(11, (BB.P - 1) // 2, BB.P) == BB.P - 1
    assert BB.P % 4 == 1
    rng = np.random.default_rng(1)
    for _ in range(8):
        a = tuple(int(x) for x in rng.integers(1, BB.P, 4))
        assert BB.emul(a, BB.einv(a)) == BB.one


@pytest.mark.parametrize("degree_bits,num_challenges", [(3, 6), (5, 6), (8, 6), (10, 6)])
def test_oracle_bb_proof_verifies(degree_bits, num_challenges):
    circ = D.DummyCircuit(degree_bits, D.CircuitConfig.babybear(num_challenges), F=BB)
    proof, dbg = D.prove_cpu(circ, circ.witness(seed=degree_bits))
    stats = {}
    assert D.verify(circ, proof, stats)
    assert circ.reduction_arity_bits == [3] * len(circ.reduction_arity_bits)
    assert stats["merkle_paths"] == 28 * (4 + len(circ.reduction_arity_bits))
    pr, pis = V.read_proof_with_pis(proof, circ.common_data(), BB)
    assert V.write_proof_with_pis(pr, pis, BB) == proof
    # fri/prover.rs:147: leading_zeros(u64) >= 16 + (64 - 31)
    assert int(dbg[3 * num_challenges + 8]).bit_length() <= 15


def test_bb_security_check_needs_ten_challenges_at_2p20():
    # circuit_builder.rs:1190-1192 with F::bits() = 31
    with pytest.raises(AssertionError):
        D.DummyCircuit(20, D.CircuitConfig.babybear(6), F=BB)
    assert (31 - 20) * 10 >= 100


def test_bb_tampered_proofs_are_rejected():
    circ = D.DummyCircuit(5, F=BB)
    proof, _ = D.prove_cpu(circ, circ.witness())
    cd = circ.common_data()
    pr, pis = V.read_proof_with_pis(proof, cd, BB)
    w = list(pr["openings"]["wires"][9])
    w[2] ^= 1
    pr["openings"]["wires"][9] = tuple(w)
    with pytest.raises(AssertionError):
        D.verify(circ, V.write_proof_with_pis(pr, pis, BB))
    pr, pis = V.read_proof_with_pis(proof, cd, BB)
    pr["opening_proof"]["pow_witness"] += 1
    with pytest.raises(AssertionError):
        D.verify(circ, V.write_proof_with_pis(pr, pis, BB))


def test_bb_bad_witness_fails_the_identity():
    circ = D.DummyCircuit(5, F=BB)
    w = circ.witness()
    w[7, circ.pi_row] = 5  # PublicInputGate<8>: wire 7 must equal pi_hash[7] = 0
    proof, _ = D.prove_cpu(circ, w)
    with pytest.raises(AssertionError):
        D.verify(circ, proof)


def test_verifier_view_accepts_what_the_full_circuit_accepts():
    """The GPU tests at 2^20 rows verify through DummyCircuit.verifier_view (no CPU commit of the constants/sigmas
    columns); here the view is checked against the full object on a size the oracle prover finishes."""
    circ = D.DummyCircuit(6, D.CircuitConfig.babybear(6), F=BB)
    proof, _ = D.prove_cpu(circ, circ.witness(seed=2))
    view = D.DummyCircuit.verifier_view(6, circ.cfg, BB, circ.k_is)
    view.set_cap(circ.constants_sigmas_cap)
    assert (view.circuit_digest == circ.circuit_digest).all()
    assert view.common_data() == circ.common_data()
    assert D.verify(view, proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(AssertionError):
        D.verify(view, bytes(bad))
