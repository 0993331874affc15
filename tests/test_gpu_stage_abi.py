"""The stage-level entry points of the C ABI (gb_zs_partial_products, gb_quotient_polys, gb_prove_openings, with
gb_commit_values / gb_commit_coeffs / gb_batch_eval_ext around them) driven the way a Rust host that keeps the reference's prover
loop would drive them (plonk/prover.rs:228-447): the loop and the Challenger run HERE (the oracle's Challenger stands in for the
host's), every heavy step is one ABI call.  Checked against the CPU oracle PROVER:
  * its own intermediates - the Z / partial-product values it hands to from_values, the quotient chunk coefficients it hands to
    from_coeffs - equal the GPU stages' outputs element for element;
  * the bytes assembled from the stages equal its proof bytes, and gb_prove's.
-m gpu only."""
import struct

import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import CircuitData, GpuContext, PermArgZeroError, PolynomialBatch, ShapeError
from plonky2_goldibear_amd import native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _gpu_circuit(ctx, circ, tag):
    cfg = circ.cfg
    return CircuitData(ctx, circ.degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                       num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants, num_challenges=cfg.num_challenges,
                       arity_bits=cfg.arity_bits, gate_constant=circ.GATE_CONSTANT, gate_pi=circ.GATE_PI, field=tag)


def challenger_tuple(ch, F):
    """(sponge_state, input_buffer, output_buffer) of the oracle's Challenger objects (iop/challenger.rs:18-31)"""
    if F is BB:
        return [int(x) for x in ch.state], list(ch.inp), list(ch.out)
    raw = ch.state()   # C struct {state[12], in[8], int nin, out[8], int nout} as u64 words
    nin, nout = int(raw[20]) & 0xFFFFFFFF, int(raw[29]) & 0xFFFFFFFF
    return [int(x) for x in raw[:12]], [int(x) for x in raw[12:12 + nin]], [int(x) for x in raw[21:21 + nout]]


def prove_by_stages(gpu, circ, w, pis, tag, intermediates=None):
    """internal_prove_with_partition_witness (plonk/prover.rs:228-447), every heavy step through one ABI entry point"""
    F, cfg, ctx = circ.F, circ.cfg, gpu.ctx
    c, r, cap_h = cfg.num_challenges, cfg.rate_bits, cfg.cap_height
    pi_hash = F.hash_no_pad(np.asarray(pis, dtype=F.dtype))                                   # :244
    wires = PolynomialBatch.from_values(ctx, w, r, cap_h, field=tag)                          # :261-272
    ch = F.Challenger()
    ch.observe_hash(circ.circuit_digest)                                                       # :277-281
    ch.observe_hash(pi_hash)
    ch.observe_cap(wires.merkle_tree.cap)
    betas, gammas = ch.get_n_challenges(c), ch.get_n_challenges(c)                            # :296-297
    zs_values = gpu.zs_partial_products(w, betas, gammas)                                     # :305-329
    zs = PolynomialBatch.from_values(ctx, zs_values, r, cap_h, field=tag)                     # :331-339
    ch.observe_cap(zs.merkle_tree.cap)
    alphas = ch.get_n_challenges(c)                                                            # :343
    chunks = gpu.quotient_polys(wires, zs, pi_hash, betas, gammas, alphas)                    # :345-376
    quot = PolynomialBatch.from_coeffs(ctx, chunks, r, cap_h, field=tag)                      # :379-387
    ch.observe_cap(quot.merkle_tree.cap)
    zeta = ch.get_extension_challenge(F.D)                                                     # :391
    g = F.two_adic_generator(circ.degree_bits)
    zeta_next = F.escale(zeta, g)
    cs = gpu.constants_sigmas_commitment
    ev = lambda b, z: b.eval_ext(np.array(z, dtype=F.dtype))                                   # OpeningSet::new, plonk/proof.rs:346-387
    o_cs, o_w, o_z, o_zn, o_q = ev(cs, zeta), ev(wires, zeta), ev(zs, zeta), ev(zs, zeta_next), ev(quot, zeta)
    ncst = circ.num_constants
    openings = [o_cs[:ncst], o_cs[ncst:], o_w, o_z[:c], o_zn[:c], o_z[c:], o_q]               # serialization/mod.rs:1514-1529
    for part in (o_cs[:ncst], o_cs[ncst:], o_w, o_z[:c], o_z[c:], o_q, o_zn[:c]):            # observe_openings, proof.rs:388-440
        ch.observe_elements(part)
    fri_bytes, after = gpu.prove_openings(wires, zs, quot, zeta, challenger_tuple(ch, F))     # :422-437
    out = b"".join(np.ascontiguousarray(b.merkle_tree.cap).tobytes() for b in (wires, zs, quot))
    out += b"".join(np.ascontiguousarray(o).tobytes() for o in openings)
    out += fri_bytes
    out += struct.pack("<Q", len(pis)) + np.asarray(pis, dtype=F.dtype).tobytes()
    if intermediates is not None:
        intermediates.update(zs_partial_products=zs_values, quotient_chunks=chunks, challenger_after=after, betas=betas,
                             gammas=gammas, alphas=alphas, zeta=zeta)
    for b in (wires, zs, quot):
        b.free()
    return out


@pytest.mark.parametrize("field_name,degree_bits,num_challenges", [
    ("goldilocks", 4, 2), ("goldilocks", 9, 3), ("goldilocks", 13, 2), ("goldilocks", 16, 3),
    ("babybear", 5, 6), ("babybear", 12, 7), ("babybear", 14, 10),
])
def test_stage_by_stage_equals_oracle_prover(ctx, field_name, degree_bits, num_challenges):
    if field_name == "goldilocks":
        F, tag, cfg = GL, N.GB_GOLDILOCKS, D.CircuitConfig(num_challenges=num_challenges)
    else:
        F, tag, cfg = BB, N.GB_BABYBEAR, D.CircuitConfig.babybear(num_challenges)
    circ = D.DummyCircuit(degree_bits, cfg, F=F)
    gpu = _gpu_circuit(ctx, circ, tag)
    circ.set_cap(gpu.constants_sigmas_cap)
    w = circ.witness(seed=40 + degree_bits)
    dump, mid = {}, {}
    want, dbg = D.prove_cpu(circ, w, dump=dump)
    got = prove_by_stages(gpu, circ, w, [], tag, mid)
    # the oracle prover's own intermediates
    assert (mid["zs_partial_products"] == dump["zs_partial_products"]).all()
    assert (mid["quotient_chunks"] == dump["quotient_chunks"]).all()
    c = cfg.num_challenges
    assert mid["betas"] == [int(x) for x in dbg[:c]] and mid["alphas"] == [int(x) for x in dbg[2 * c:3 * c]]
    assert got == want
    assert got == gpu.prove(w)
    # the transcript left behind is the one the reference leaves: replaying the proof in the verifier's order ends in the same state
    from oracle import verifier as V
    proof, pis = V.read_proof_with_pis(want, circ.common_data(), F)
    ch = F.Challenger()
    ch.observe_hash(circ.circuit_digest)
    ch.observe_hash(F.hash_no_pad(np.asarray(pis, dtype=F.dtype)))
    ch.observe_cap(proof["wires_cap"]); ch.get_n_challenges(2 * c)
    ch.observe_cap(proof["zs_cap"]); ch.get_n_challenges(c)
    ch.observe_cap(proof["quotient_cap"]); ch.get_extension_challenge(F.D)
    for batch in V.fri_openings(proof["openings"]):
        ch.observe_elements([x for e in batch for x in e])
    ch.get_extension_challenge(F.D)
    for cap in proof["opening_proof"]["commit_phase_merkle_caps"]:
        ch.observe_cap(cap)
        ch.get_extension_challenge(F.D)
    ch.observe_elements([x for e in proof["opening_proof"]["final_poly"] for x in e])
    ch.observe_element(proof["opening_proof"]["pow_witness"])
    ch.get_challenge()
    for _ in range(cfg.num_query_rounds):
        ch.get_challenge()
    assert mid["challenger_after"] == challenger_tuple(ch, F)
    gpu.free()


def test_prove_openings_size_query_leaves_the_challenger_alone(ctx):
    """a call with too small a buffer reports the size needed and must not advance the caller's Challenger - the
    retry with that size then returns exactly the bytes (and the transcript) of a call that had room from the start."""
    circ = D.DummyCircuit(8, D.CircuitConfig(num_challenges=2), F=GL)
    gpu = _gpu_circuit(ctx, circ, N.GB_GOLDILOCKS)
    circ.set_cap(gpu.constants_sigmas_cap)
    w = circ.witness(seed=7)
    calls = []
    real = gpu.prove_openings

    def probing(wires, zs, quot, zeta, chal):
        for cap in (0, 100):                      # a NULL buffer, then a short one
            with pytest.raises(N.GoldibearError) as e:
                real(wires, zs, quot, zeta, chal, out_cap=cap)
            assert e.value.status == N.GB_ERR_BUFFER_TOO_SMALL
            assert gpu.last_challenger == (list(chal[0]), list(chal[1]), list(chal[2]))   # untouched
            calls.append(gpu.last_fri_proof_len)
        out = real(wires, zs, quot, zeta, chal, out_cap=calls[-1])                          # exactly the size asked for
        assert len(out[0]) == calls[0] == calls[1]
        return out

    gpu.prove_openings = probing
    got = prove_by_stages(gpu, circ, w, [], N.GB_GOLDILOCKS)
    gpu.prove_openings = real
    assert len(calls) == 2 and got == gpu.prove(w) == D.prove_cpu(circ, w)[0]
    gpu.free()


def test_stages_on_a_general_gate_set(ctx):
    """the factorial example's circuit (ArithmeticGate + PoseidonGate + public inputs, two selector groups): stage by stage =
    oracle prover = gb_prove"""
    from circuits import factorial_circuit, oracle_circuit
    b, pw = factorial_circuit()
    built = b.build(ctx)
    w, pis = built.generate_witness(pw)
    oc = oracle_circuit(built, len(pis))
    dump, mid = {}, {}
    want, _ = D.prove_cpu(oc, w, pis, dump=dump)
    got = prove_by_stages(built.data, oc, w, pis, N.GB_GOLDILOCKS, mid)
    assert (mid["zs_partial_products"] == dump["zs_partial_products"]).all()
    assert (mid["quotient_chunks"] == dump["quotient_chunks"]).all()
    assert got == want == built.data.prove(w, pis)


def test_device_resident_stage_inputs_and_errors(ctx):
    import torch
    circ = D.DummyCircuit(10, D.CircuitConfig(num_challenges=2))
    gpu = _gpu_circuit(ctx, circ, N.GB_GOLDILOCKS)
    w = circ.witness(seed=5)
    betas, gammas = [3, 5], [7, 11]
    host = gpu.zs_partial_products(w, betas, gammas)
    dev = gpu.zs_partial_products(torch.from_numpy(w.view(np.int64)).to("cuda:0"), betas, gammas)
    assert (dev.cpu().numpy().view(np.uint64) == host).all()
    assert (host[:2, 0] == 1).all()   # Z(1) = 1 (prover.rs:531)
    with pytest.raises(ShapeError):
        gpu.zs_partial_products(w[:, :512], betas, gammas)
    with pytest.raises(ShapeError):
        gpu.zs_partial_products(w, [GL.P, 1], gammas)          # non-canonical challenge
    wires = PolynomialBatch.from_values(ctx, w, 3, 4)
    zs = PolynomialBatch.from_values(ctx, host, 3, 4)
    with pytest.raises(ShapeError):                             # batches of the wrong shape are refused, not read
        gpu.quotient_polys(zs, wires, [0, 0, 0, 0], betas, gammas, [1, 2])
    other = PolynomialBatch.from_values(ctx, w[:, :512].copy(), 3, 4)
    with pytest.raises(ShapeError):
        gpu.quotient_polys(other, zs, [0, 0, 0, 0], betas, gammas, [1, 2])
    chunks = gpu.quotient_polys(wires, zs, GL.hash_no_pad(np.zeros(0, np.uint64)), betas, gammas, [1, 2])
    quot = PolynomialBatch.from_coeffs(ctx, chunks, 3, 4)
    with pytest.raises(ShapeError):
        gpu.prove_openings(wires, zs, quot, (1, 0), ([0] * 12, [0] * 9, []))
    from plonky2_goldibear_amd import GoldibearError
    with pytest.raises(GoldibearError, match="subgroup"):       # prover.rs:399-404
        gpu.prove_openings(wires, zs, quot, (1, 0), ([0] * 12, [], []))
    # a witness with a zero permutation denominator: beta = 0, gamma = -w makes w + beta*sigma + gamma vanish
    wv = int(w[5, circ.pi_row])
    with pytest.raises(PermArgZeroError):
        gpu.zs_partial_products(w, [0, 1], [(GL.P - wv) % GL.P, 1])
    for b in (wires, zs, quot, other):
        b.free()
    gpu.free()
