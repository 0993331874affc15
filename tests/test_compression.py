"""Compressed proofs (hash/path_compression.rs, fri/proof.rs:137-384, plonk/proof.rs:96-260).  The reference holds no serialized
compressed proof, so the pin is the round trip on ITS regression proof: compress -> bytes -> decompress reproduces the reference's
bytes exactly (every Merkle path, every inferred FRI element).  No GPU: the oracle (python) and the product (C ABI,
gb_proof_compress / gb_proof_decompress / gb_verify_compressed run on the host) are compared byte for byte."""
import os
import struct

import numpy as np
import pytest

from oracle import compression as Z
from oracle import plonk_dummy as D
from oracle import verifier as V
from oracle.fields import BB, GL


@pytest.fixture(scope="module")
def fixture(golden_dir):
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    cd = V.read_common_data(rd("recursive_verifier_gl_common_data.bin"))
    vd = V.read_verifier_data(rd("recursive_verifier_gl_verifier_data.bin"))
    return cd, vd, rd("recursive_verifier_gl_proof.bin")


def test_path_compression_round_trip():
    """hash/path_compression.rs:127-166 test_path_compression: random leaves, random (repeating) indices"""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    h, cap_height, n = 10, 3, 1 << 10
    leaves = O.splitmix64_fill(1, n * 7).reshape(n, 7)
    tree = O.MerkleTree(leaves, cap_height)
    idx = [int(x) for x in rng.integers(0, n, 40)]
    proofs = [[[int(x) for x in s] for s in tree.prove(i)] for i in idx]
    comp = Z.compress_merkle_proofs(cap_height, idx, proofs)
    assert sum(map(len, comp)) < sum(map(len, proofs))
    assert Z.decompress_merkle_proofs(GL, [leaves[i] for i in idx], idx, comp, h, cap_height) == proofs


def test_reference_regression_proof_round_trips_through_the_compressed_form(fixture):
    cd, vd, raw = fixture
    small = Z.compress_bytes(raw, vd["circuit_digest"], cd)
    assert len(small) == 137620 < len(raw)   # the Merkle paths shrink; the opened leaves dominate
    cpr, pis = Z.read_compressed_proof_with_pis(small, cd)
    assert Z.write_compressed_proof_with_pis(cpr, pis) == small
    assert len(cpr["opening_proof"]["initial_trees_proofs"]) == len(set(cpr["opening_proof"]["indices"]))
    assert Z.decompress_bytes(small, vd["circuit_digest"], cd) == raw


@pytest.mark.parametrize("F", [GL, BB], ids=["goldilocks", "babybear"])
def test_oracle_proofs_round_trip(F):
    circ = D.DummyCircuit(5, F=F) if F is GL else D.DummyCircuit(5, D.CircuitConfig.babybear(6), F=BB)
    proof, _ = D.prove_cpu(circ, circ.witness(seed=3))   # 2^8 LDE points, 28 queries: repeated cosets are certain
    cd = circ.common_data()
    small = Z.compress_bytes(proof, circ.circuit_digest, cd, F)
    assert len(small) < len(proof)
    assert Z.decompress_bytes(small, circ.circuit_digest, cd, F) == proof
    # a flipped sibling / eval in the compressed form cannot decompress to a verifying proof
    bad = bytearray(small)
    bad[len(bad) // 2] ^= 1
    try:
        out = Z.decompress_bytes(bytes(bad), circ.circuit_digest, cd, F)
    except (AssertionError, KeyError, StopIteration, ValueError, struct.error):
        return
    with pytest.raises(AssertionError):
        D.verify(circ, out)


# ------------------------------------------------------------------------------------------------ the product (C ABI, host)
def _abi_verifier(golden_dir):
    from test_abi_verify_fixture import _fixture_circuit
    return _fixture_circuit(golden_dir)


def test_c_abi_compression_matches_oracle_on_the_reference_proof(golden_dir, fixture):
    from plonky2_goldibear_amd import VerifyError
    from plonky2_goldibear_amd import native as N
    cd, vd, raw = fixture
    circ, _, _ = _abi_verifier(golden_dir)
    small = circ.compress(raw)
    assert small == Z.compress_bytes(raw, vd["circuit_digest"], cd)
    assert circ.decompress(small) == raw
    assert circ.verify_compressed(small)
    bad = bytearray(small)
    bad[9000] ^= 1   # inside the openings: the transcript changes
    with pytest.raises((VerifyError, N.ShapeError)):
        circ.verify_compressed(bytes(bad))
    with pytest.raises(N.ShapeError):
        circ.decompress(small[:-5])


@pytest.mark.parametrize("F", [GL, BB], ids=["goldilocks", "babybear"])
def test_c_abi_compression_matches_oracle_on_small_proofs(F):
    """2^8 LDE points and 28 queries: repeated indices and shared cosets in every layer"""
    from plonky2_goldibear_amd import VerifierCircuitData
    from plonky2_goldibear_amd import native as N
    from test_zero_knowledge import _verifier
    for zk in (False, True):
        circ = D.DummyCircuit(5, F=F) if F is GL else D.DummyCircuit(5, D.CircuitConfig.babybear(6), F=BB)
        salts = None
        if zk:
            circ.zero_knowledge = True
            nl = circ.n << circ.cfg.rate_bits
            salts = F.fill(77, 12 * nl).reshape(3, 4, nl)
        proof, _ = D.prove_cpu(circ, circ.witness(seed=3), salts=salts)
        v = _verifier(circ, zk)
        small = v.compress(proof)
        assert small == Z.compress_bytes(proof, circ.circuit_digest, circ.common_data(), F)
        assert v.decompress(small) == proof and v.verify_compressed(small)
