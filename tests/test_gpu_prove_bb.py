"""GPU prove() over BabyBear (BASELINE config 4) through the C ABI: proof BYTES identical to the CPU oracle
prover's and accepted by the restated verifier.  BabyBear field constants are UNPINNED, so
"identical" means identical to the restated algorithm, not to reference bytes.  -m gpu only."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle.fields import BB
from plonky2_goldibear_amd import CircuitData, GpuContext, PermArgZeroError, ShapeError, TooManyPermArgFailuresError
from plonky2_goldibear_amd import dummy_circuit as DC
from plonky2_goldibear_amd import native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _gpu_circuit(ctx, circ):
    cfg = circ.cfg
    return CircuitData(ctx, circ.degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                       num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants,
                       num_challenges=cfg.num_challenges, max_quotient_degree_factor=cfg.max_quotient_degree_factor,
                       rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, proof_of_work_bits=cfg.proof_of_work_bits,
                       num_query_rounds=cfg.num_query_rounds, arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits,
                       gate_constant=circ.GATE_CONSTANT, gate_pi=circ.GATE_PI, field=N.GB_BABYBEAR)


@pytest.mark.parametrize("degree_bits,num_challenges", [(3, 6), (4, 7), (6, 6), (9, 8), (12, 6), (13, 10)])
def test_bb_proof_bytes_match_oracle(ctx, degree_bits, num_challenges):
    circ = D.DummyCircuit(degree_bits, D.CircuitConfig.babybear(num_challenges), F=BB)
    gpu = _gpu_circuit(ctx, circ)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    assert (gpu.constants_sigmas_cap == circ.constants_sigmas_cap).all()
    w = circ.witness(seed=degree_bits)
    want, _ = D.prove_cpu(circ, w)
    got = gpu.prove(w)
    assert len(got) == len(want)
    assert got == want
    assert D.verify(circ, got)


def test_bb_product_side_circuit_builder_matches_oracle():
    circ = D.DummyCircuit(7, F=BB)
    cs, k_is, pi_row, const_row = DC.build_dummy_circuit_bb(7)
    assert (cs == circ.constants_sigmas).all() and (k_is == circ.k_is).all()
    assert (pi_row, const_row) == (circ.pi_row, circ.const_row)


@pytest.mark.parametrize("degree_bits,num_challenges", [(14, 6), (16, 7)])
def test_bb_larger_proofs_verify(ctx, degree_bits, num_challenges):
    circ = D.DummyCircuit(degree_bits, D.CircuitConfig.babybear(num_challenges), F=BB)
    gpu = _gpu_circuit(ctx, circ)
    circ.set_cap(gpu.constants_sigmas_cap)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    proof = gpu.prove(circ.witness(seed=1))
    stats = {}
    assert D.verify(circ, proof, stats)
    assert stats["merkle_paths"] == 28 * (4 + len(circ.reduction_arity_bits))
    import torch
    w2 = circ.witness(seed=2)
    t = torch.from_numpy(w2.view(np.int32)).to("cuda:0")
    proof2 = gpu.prove(t)
    assert proof2 != proof and D.verify(circ, proof2)
    assert proof2 == gpu.prove(w2)


def test_bb_config4_2p20_rows(ctx):
    """BASELINE config 4: 2^20-row dummy circuit, BabyBear + Poseidon2, num_challenges = 10 (31-bit field)."""
    k = 20
    cs, k_is, pi_row, _ = DC.build_dummy_circuit_bb(k)
    gpu = CircuitData.babybear(ctx, k, cs, k_is, num_challenges=10)
    # verifier-side view without the CPU commit of 44 x 2^20 columns
    circ = D.DummyCircuit.verifier_view(k, D.CircuitConfig.babybear(10), BB, k_is)
    circ.set_cap(gpu.constants_sigmas_cap)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    proof = gpu.prove(DC.dummy_witness_bb(k, pi_row, seed=3))
    stats = {}
    assert D.verify(circ, proof, stats)
    assert stats["merkle_paths"] == 28 * (4 + 5)
    assert gpu.verify(proof)  # the product's own host-side verifier (gb_verify)
    from oracle import compression as Z
    small = gpu.compress(proof)
    assert small == Z.compress_bytes(proof, circ.circuit_digest, circ.common_data(), BB)
    assert gpu.decompress(small) == proof and gpu.verify_compressed(small)


def test_bb_error_behaviour(ctx):
    circ = D.DummyCircuit(5, F=BB)
    with pytest.raises(ShapeError):  # circuit_builder.rs:1191-1192 with F::bits() = 31: 2^20 rows need 10 challenges
        CircuitData.babybear(ctx, 20, np.zeros((44, 1 << 20), np.uint32), circ.k_is, num_challenges=6)


def test_bb_inv_zero_perm_arg_and_retry(ctx):
    """ProverError::InvZeroPermArg (plonk/prover.rs:512-514) is reachable in a 31-bit field: at 2^16 rows, 7 challenges,
    41 routed wires about 1 witness in 115 has a zero denominator.  Find one, check the CPU oracle prover reports the
    same condition for it, and that the retry loop of prove_with_partition_witness (:183-226) re-randomises the random
    wire and produces a verifying proof."""
    k = 16
    circ = D.DummyCircuit(k, D.CircuitConfig.babybear(7), F=BB)
    gpu = _gpu_circuit(ctx, circ)
    circ.set_cap(gpu.constants_sigmas_cap)
    bad = None
    for seed in range(100, 1400):
        w = circ.witness(seed=seed)
        try:
            gpu.prove_once(w)
        except PermArgZeroError:
            bad = w
            break
    if bad is None:
        pytest.skip("no zero denominator in 1300 witnesses (probability ~1e-5)")
    with pytest.raises(RuntimeError, match="rc=1"):  # oracle status 1 = InvZeroPermArg
        D.prove_cpu(circ, bad)
    with pytest.raises(TooManyPermArgFailuresError):  # no random wire given: the reference bails the same way
        gpu.prove(bad.copy())
    w = bad.copy()
    rw = (circ.cfg.num_wires - 1, circ.pi_row)
    proof = gpu.prove(w, random_wire=rw, rng=np.random.default_rng(5))
    assert gpu.perm_arg_retries >= 1
    assert (w != bad).sum() == 1 and w[rw] != bad[rw]
    assert D.verify(circ, proof)
    # The attempts after the first went through gb_prove_retry, which rebuilds only the random wire's column of the failed
    # attempt's wires commitment (2^19 leaves, 167 wires, host witness: the incremental path): same bytes as proving the final
    # witness from scratch, on the GPU and on the CPU oracle.
    assert proof == gpu.prove_once(w)
    assert proof == D.prove_cpu(circ, w)[0]
    # gb_prove_retry is gb_prove whenever there is nothing to build on: no failed attempt before it ...
    assert gpu.prove_once(w, retry_wire=rw) == proof
    # ... a wire outside the last leaf-sponge segment (columns 160..166 here) ...
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(bad)
    w2 = bad.copy()
    w2[5, 7] = (int(w2[5, 7]) + 1) % 2013265921
    try:
        p2 = gpu.prove_once(w2, retry_wire=(5, 7))
        assert p2 == gpu.prove_once(w2)
    except PermArgZeroError:
        pass                                   # the other wire's change did not lift the zero denominator: also gb_prove's answer
    # ... or another proof in between (the kept state belongs to the last failed attempt only)
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(bad)
    other = circ.witness(seed=99)
    gpu.prove_once(other)
    assert gpu.prove_once(w, retry_wire=rw) == proof
    # two failures in a row: the second attempt's state is kept for the third
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(bad)
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(bad, retry_wire=rw)     # "re-drawn" to the same value: fails again, incrementally
    assert gpu.prove_once(w, retry_wire=rw) == proof
    # a device-resident witness: the column is read from the caller's matrix, the kept leaf-sponge state comes from a two-
    # segment hash of the device input
    import torch
    dbad = torch.from_numpy(bad.view(np.int32)).cuda()
    dw = torch.from_numpy(w.view(np.int32)).cuda()
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(dbad)
    assert gpu.prove_once(dw, retry_wire=rw) == proof
    assert gpu.prove_once(dw) == proof
    d = dbad.clone()
    assert gpu.prove(d, random_wire=rw, rng=np.random.default_rng(5)) == proof   # the same draw as the host run above
    assert gpu.perm_arg_retries >= 1
    # a failed host attempt followed by a device retry, and the other way round
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(bad)
    assert gpu.prove_once(dw, retry_wire=rw) == proof
    with pytest.raises(PermArgZeroError):
        gpu.prove_once(dbad)
    assert gpu.prove_once(w, retry_wire=rw) == proof


@pytest.mark.parametrize("degree_bits,num_challenges", [(15, 7), (17, 8), (18, 8), (19, 9)])
def test_bb_odd_sizes_verify(ctx, degree_bits, num_challenges):
    circ = D.DummyCircuit(degree_bits, D.CircuitConfig.babybear(num_challenges), F=BB)
    gpu = _gpu_circuit(ctx, circ)
    circ.set_cap(gpu.constants_sigmas_cap)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    proof = gpu.prove(circ.witness(seed=degree_bits), random_wire=(circ.cfg.num_wires - 1, circ.pi_row),
                      rng=np.random.default_rng(degree_bits))
    stats = {}
    assert D.verify(circ, proof, stats)
    assert stats["merkle_paths"] == 28 * (4 + len(circ.reduction_arity_bits))
    gpu.free()
    ctx.trim()


@pytest.mark.parametrize("field,kw", [("bb", dict(num_wires=334, num_routed_wires=160)),      # recursion_config_bb_wide
                                      ("gl", dict(num_wires=234)), ("gl", dict(num_wires=136))])   # wide_ecc / standard_ecc
def test_other_reference_configs_bytes_match_oracle(ctx, field, kw):
    """The reference's remaining CircuitConfigs (plonk/circuit_data.rs:122-173): same prover, other wire counts (the wide BabyBear
    config has 20 partial-product chunks per challenge)."""
    from oracle.fields import GL
    if field == "bb":
        cfg = D.CircuitConfig.babybear(6, **kw)
        circ = D.DummyCircuit(7, cfg, F=BB)
        gpu = CircuitData.babybear(ctx, 7, circ.constants_sigmas, circ.k_is, num_challenges=6, **kw)
    else:
        cfg = D.CircuitConfig(**kw)
        circ = D.DummyCircuit(7, cfg, F=GL)
        gpu = CircuitData(ctx, 7, circ.constants_sigmas, circ.k_is, **kw)
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    w = circ.witness(seed=5)
    want, _ = D.prove_cpu(circ, w)
    got = gpu.prove(w)
    assert got == want and gpu.verify(got) and D.verify(circ, got)
    gpu.free()
