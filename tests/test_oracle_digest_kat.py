"""NTT -> LDE -> Merkle cap -> circuit_digest pinned end to end against the reference's known answer
(recursion/recursive_verifier.rs:427-436: 16 000 NoopGates, stock Goldilocks config, degree_bits 14)."""
import numpy as np

from oracle import oracle as O
from oracle import plonk_dummy as D


def test_circuit_digest_kat_cpu(kats):
    want = kats["circuit_digest_gl"][0]
    cs, degree_bits = D.test_form_constants_sigmas(16000)
    assert degree_bits == want["degree_bits"] == 14 and cs.shape == (84, 1 << 14)
    batch = O.PolynomialBatch.from_values(cs, 3, 4)
    digest = D.circuit_digest_from_cap(batch.cap, degree_bits)
    assert digest.tolist() == want["digest"]
