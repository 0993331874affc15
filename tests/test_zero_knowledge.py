"""Zero-knowledge (salted) proofs: CircuitConfig.zero_knowledge = FriParams.hiding (plonk/prover.rs:267,334,382,
fri/oracle.rs:133-148, fri/verifier.rs:155-162).  The salted layout is pinned by the reference's regression proof, which is
zero-knowledge (tests/test_oracle_fixture.py, tests/test_abi_verify_fixture.py).  Here: the oracle prover with salts against the
oracle verifier and against gb_verify (host-only, no GPU); the GPU prover against the oracle prover byte for byte (-m gpu)."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import VerifierCircuitData, VerifyError
from plonky2_goldibear_amd import native as N


def _zk_circuit(F, degree_bits):
    circ = D.DummyCircuit(degree_bits, F=F) if F is GL else D.DummyCircuit(degree_bits, D.CircuitConfig.babybear(6), F=BB)
    circ.zero_knowledge = True
    n_lde = circ.n << circ.cfg.rate_bits
    salts = F.fill(0x5A17, 3 * 4 * n_lde).reshape(3, 4, n_lde)
    return circ, salts


def _verifier(circ, zero_knowledge):
    cfg = circ.cfg
    return VerifierCircuitData(circ.degree_bits, circ.gate_table, circ.k_is, circ.constants_sigmas_cap, circ.circuit_digest,
                               num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants,
                               num_challenges=cfg.num_challenges, arity_bits=cfg.arity_bits, num_selectors=circ.num_selectors,
                               zero_knowledge=zero_knowledge, field=N.GB_GOLDILOCKS if circ.F is GL else N.GB_BABYBEAR)


@pytest.mark.parametrize("F", [GL, BB], ids=["goldilocks", "babybear"])
def test_salted_oracle_proof_verifies_everywhere(F):
    circ, salts = _zk_circuit(F, 6)
    w = circ.witness(seed=4)
    proof, _ = D.prove_cpu(circ, w, salts=salts)
    assert D.verify(circ, proof)
    assert _verifier(circ, True).verify(proof)
    with pytest.raises((N.ShapeError, VerifyError)):
        _verifier(circ, False).verify(proof)  # read as unsalted the bytes do not even parse
    # different salts: different commitments, same statement
    proof2, _ = D.prove_cpu(circ, w, salts=(salts + 1) % F.P)
    assert proof2 != proof and D.verify(circ, proof2)
    plain = D.DummyCircuit(6, F=F) if F is GL else D.DummyCircuit(6, D.CircuitConfig.babybear(6), F=BB)
    unsalted, _ = D.prove_cpu(plain, w)
    assert len(proof) == len(unsalted) + 28 * 3 * 4 * F.elem_bytes
    # a salt in an opened leaf is covered by the Merkle hash: flip one and the path check fails
    from oracle import verifier as V
    pr, pis = V.read_proof_with_pis(proof, circ.common_data(), F)
    vals, path = pr["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"][1]
    vals[-1] = (vals[-1] + 1) % F.P
    bad = V.write_proof_with_pis(pr, pis, F)
    with pytest.raises(AssertionError):
        D.verify(circ, bad)
    with pytest.raises(VerifyError, match="Merkle"):
        _verifier(circ, True).verify(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("F,degree_bits", [(GL, 5), (GL, 10), (BB, 6), (BB, 11)], ids=["gl5", "gl10", "bb6", "bb11"])
def test_gpu_salted_proof_bytes_match_oracle(F, degree_bits):
    from plonky2_goldibear_amd import CircuitData, GpuContext
    ctx = GpuContext(0)
    circ, salts = _zk_circuit(F, degree_bits)
    cfg = circ.cfg
    gpu = CircuitData(ctx, degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                      num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants, num_challenges=cfg.num_challenges,
                      arity_bits=cfg.arity_bits, field=N.GB_GOLDILOCKS if F is GL else N.GB_BABYBEAR, zero_knowledge=True)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    w = circ.witness(seed=9)
    want, _ = D.prove_cpu(circ, w, salts=salts)
    got = gpu.prove(w, salts=salts)
    assert got == want
    assert gpu.verify(got) and D.verify(circ, got)
    # device-resident witness and salts
    import torch
    view = np.int64 if F is GL else np.int32
    wd = torch.from_numpy(w.view(view)).to("cuda:0")
    sd = torch.from_numpy(np.ascontiguousarray(salts).view(view)).to("cuda:0")
    assert gpu.prove(wd, salts=sd) == want
    with pytest.raises(N.ShapeError):
        gpu.prove(w)             # a zero-knowledge circuit needs its salts
    # the witness as separately allocated columns and, like the salts (F::rand_vec columns), in the field type's in-memory words
    # (gb_prove_salted_cols + GB_INPUT_P3_REPR): the same bytes
    def p3(a):
        if F is GL:
            out = a.copy()
            small = a < np.uint64(0xFFFFFFFF)
            out[small] = a[small] + np.uint64(GL.P)      # a non-canonical representative wherever one fits in 64 bits
            return out
        return ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(BB.P)).astype(np.uint32)
    assert gpu.prove([np.array(c, copy=True) for c in w], salts=salts) == want
    assert gpu.prove([np.array(c, copy=True) for c in p3(w)], salts=p3(np.ascontiguousarray(salts)), p3_repr=True) == want
    assert gpu.prove(p3(w), salts=p3(np.ascontiguousarray(salts)), p3_repr=True) == want
    gpu.free()
    ctx.close()
