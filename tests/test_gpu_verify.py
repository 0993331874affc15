"""gb_verify - the product's host-side restatement of the reference's verifier (plonk/verifier.rs:17-128,
fri/verifier.rs:67-250) - against the oracle's independently written Python verifier: both must accept the GPU's
and the oracle prover's proofs and reject the same tampered proofs.  -m gpu only (a circuit needs a context)."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle import verifier as V
from oracle.fields import BB, GL
from plonky2_goldibear_amd import CircuitData, GpuContext, ShapeError, VerifyError
from plonky2_goldibear_amd import native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _setup(ctx, F, degree_bits, nch):
    cfg = D.CircuitConfig(num_challenges=nch) if F is GL else D.CircuitConfig.babybear(nch)
    circ = D.DummyCircuit(degree_bits, cfg, F=F)
    gpu = CircuitData(ctx, degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                      num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants, num_challenges=nch,
                      arity_bits=cfg.arity_bits, field=N.GB_GOLDILOCKS if F is GL else N.GB_BABYBEAR)
    circ.set_cap(gpu.constants_sigmas_cap)
    return circ, gpu


@pytest.mark.parametrize("field,degree_bits,nch", [("gl", 3, 2), ("gl", 8, 2), ("gl", 12, 3), ("gl", 16, 3),
                                                   ("bb", 3, 6), ("bb", 9, 7), ("bb", 13, 6), ("bb", 16, 7)])
def test_accepts_gpu_and_oracle_proofs(ctx, field, degree_bits, nch):
    F = GL if field == "gl" else BB
    circ, gpu = _setup(ctx, F, degree_bits, nch)
    w = circ.witness(seed=degree_bits)
    proof = gpu.prove(w, random_wire=(circ.cfg.num_wires - 1, circ.pi_row), rng=np.random.default_rng(1))
    assert gpu.verify(proof)
    assert D.verify(circ, proof)
    if degree_bits <= 12:
        cpu_proof, _ = D.prove_cpu(circ, w)
        assert gpu.verify(cpu_proof)
    gpu.free()


@pytest.mark.parametrize("field", ["gl", "bb"])
def test_rejects_what_the_oracle_verifier_rejects(ctx, field):
    F = GL if field == "gl" else BB
    circ, gpu = _setup(ctx, F, 6, 2 if F is GL else 6)
    cd = circ.common_data()
    proof = gpu.prove(circ.witness(seed=2))
    assert gpu.verify(proof)

    def both_reject(mutate, match):
        pr, pis = V.read_proof_with_pis(proof, cd, F)
        mutate(pr)
        bad = V.write_proof_with_pis(pr, pis, F)
        with pytest.raises(AssertionError):
            D.verify(circ, bad)
        with pytest.raises(VerifyError, match=match):
            gpu.verify(bad)

    def bump(e, k=0):
        e = list(e)
        e[k] = (e[k] + 1) % F.P
        return tuple(e)

    def m_opening(pr):
        pr["openings"]["wires"][5] = bump(pr["openings"]["wires"][5], F.D - 1)
    both_reject(m_opening, "vanishing polynomial identity")

    def m_quotient(pr):
        pr["openings"]["quotient_polys"][1] = bump(pr["openings"]["quotient_polys"][1])
    both_reject(m_quotient, "vanishing polynomial identity")

    def m_pow(pr):
        pr["opening_proof"]["pow_witness"] = (pr["opening_proof"]["pow_witness"] + 1) % F.P
    both_reject(m_pow, "proof of work")

    def m_row(pr):
        vals, path = pr["opening_proof"]["query_round_proofs"][3]["initial_trees_proof"][1]
        vals[7] = (vals[7] + 1) % F.P
    both_reject(m_row, "Merkle path")

    def m_sibling(pr):
        vals, path = pr["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"][2]
        path[0][0] = (path[0][0] + 1) % F.P
    both_reject(m_sibling, "Merkle path")

    def m_step(pr):
        evals, path = pr["opening_proof"]["query_round_proofs"][5]["steps"][0]
        evals[1] = bump(evals[1])
    both_reject(m_step, "FRI")

    def m_final(pr):
        pr["opening_proof"]["final_poly"][0] = bump(pr["opening_proof"]["final_poly"][0])
    both_reject(m_final, "Final polynomial|FRI consistency|proof of work")  # the final polynomial is in the transcript: the PoW response moves first

    # malformed bytes
    with pytest.raises(ShapeError):
        gpu.verify(proof[:-3])
    with pytest.raises(ShapeError):
        gpu.verify(proof + b"\x00")
    # a proof of another circuit (other degree -> other digest and shapes)
    circ2, gpu2 = _setup(ctx, F, 5, 2 if F is GL else 6)
    other = gpu2.prove(circ2.witness())
    with pytest.raises((VerifyError, ShapeError)):
        gpu.verify(other)
    gpu.free()
    gpu2.free()


def test_invalid_witness_proof_is_rejected(ctx):
    circ, gpu = _setup(ctx, GL, 6, 2)
    w = circ.witness()
    w[0, circ.pi_row] = 5
    with pytest.raises(VerifyError, match="vanishing polynomial identity"):
        gpu.verify(gpu.prove(w))
    gpu.free()
