"""Pin the oracle with the reference's serialized regression proof (no GPU).

recursion/regression_test_data.rs:5,62,93 (RECURSIVE_VERIFIER_GL_*), verified by the reference in
recursion/recursive_verifier.rs:280-314.  Replaying it through the restated transcript and FRI
verifier exercises: Poseidon sponge / hash_or_noop / two_to_one, Merkle path indexing, Challenger
ordering, the PoW rule, LDE point order (7 * w_N^bitrev(i)), the two-adic generator, the
extension non-residue and FRI folding - every constant the Goldilocks hot path depends on.
"""
import os

import pytest

from oracle import verifier as V


@pytest.fixture(scope="module")
def fixture(golden_dir):
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    cd = V.read_common_data(rd("recursive_verifier_gl_common_data.bin"))
    vd = V.read_verifier_data(rd("recursive_verifier_gl_verifier_data.bin"))
    raw = rd("recursive_verifier_gl_proof.bin")
    proof, pis = V.read_proof_with_pis(raw, cd)
    return cd, vd, proof, pis, raw


def test_common_data_shape(fixture):
    cd = fixture[0]
    cfg = cd["config"]
    assert (cfg["num_wires"], cfg["num_routed_wires"], cfg["num_constants"]) == (135, 80, 2)
    assert cfg["fri_config"]["rate_bits"] == 3 and cfg["fri_config"]["cap_height"] == 4
    assert cfg["fri_config"]["num_query_rounds"] == 28 and cfg["fri_config"]["proof_of_work_bits"] == 16
    assert cd["k_is"][0] == 1 and cd["k_is"][1] == 7  # F::generator() powers (field/src/cosets.rs:8-21)
    assert len(cd["k_is"]) == 80
    assert cd["num_partial_products"] == 9 and cd["quotient_degree_factor"] == 8


def test_proof_round_trips_bytes(fixture):
    cd, vd, proof, pis, raw = fixture
    assert len(raw) == 149044
    assert V.write_proof_with_pis(proof, pis) == raw


def test_regression_proof_fri_and_merkle_paths_verify(fixture):
    cd, vd, proof, pis, raw = fixture
    ch = V.get_challenges(proof, pis, vd["circuit_digest"], cd)
    lz = 64 - ch["fri_pow_response"].bit_length()
    assert lz >= 16
    caps = [vd["constants_sigmas_cap"], proof["wires_cap"], proof["zs_cap"], proof["quotient_cap"]]
    stats = {}
    assert V.verify_fri(proof, ch, caps, cd, stats)
    n_layers = len(cd["fri_params"]["reduction_arity_bits"])
    assert stats["merkle_paths"] == 28 * (4 + n_layers)


def test_tampered_proof_is_rejected(fixture):
    cd, vd, proof, pis, raw = fixture
    import copy
    bad = copy.deepcopy(proof)
    v, p = bad["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"][1]
    v[0] = (v[0] + 1) % V.P
    ch = V.get_challenges(bad, pis, vd["circuit_digest"], cd)
    caps = [vd["constants_sigmas_cap"], bad["wires_cap"], bad["zs_cap"], bad["quotient_cap"]]
    with pytest.raises(AssertionError):
        V.verify_fri(bad, ch, caps, cd)


def test_regression_proof_full_verify_with_the_recursion_gate_set(golden_dir):
    """plonk/verifier.rs:17-128 on the reference's own recursion proof, INCLUDING vanishing(zeta) == Z_H(zeta) * quotient(zeta)
    with the circuit's twelve gates (Noop, PoseidonMds, PublicInput, BaseSum<2>, ReducingExtension, Reducing,
    ArithmeticExtension, Arithmetic, MulExtension, RandomAccess, CosetInterpolation, Poseidon) evaluated by oracle/gates.py:
    the reference's numbers pin every one of those evaluators, the selector filters and the zero-knowledge (salted) openings."""
    from oracle import gates as G
    from oracle import plonk_dummy as D
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    circ = D.CommonDataCircuit(rd("recursive_verifier_gl_common_data.bin"), rd("recursive_verifier_gl_verifier_data.bin"))
    kinds = [g[0] for g in circ.gate_table]
    assert kinds == [G.NOOP, G.POSEIDON_MDS, G.PUBLIC_INPUT, G.BASE_SUM, G.REDUCING_EXTENSION, G.REDUCING, G.ARITHMETIC_EXTENSION,
                     G.ARITHMETIC, G.MUL_EXTENSION, G.RANDOM_ACCESS, G.COSET_INTERPOLATION, G.POSEIDON]
    assert circ.gate_table[3][1] == 63 and circ.gate_table[9][1:2] + circ.gate_table[9][5:] == (4, 4, 2)
    assert circ.gate_table[10][1] == 4 and circ.gate_table[10][5] == 6
    raw = rd("recursive_verifier_gl_proof.bin")
    stats = {}
    assert D.verify(circ, raw, stats)
    assert stats["merkle_paths"] == 28 * (4 + 3)
    # one opened wire value off by one: the identity must fail (the transcript changes too, so every later check would as well)
    proof, pis = V.read_proof_with_pis(raw, circ.common_data())
    w0 = proof["openings"]["wires"][17]
    proof["openings"]["wires"][17] = ((w0[0] + 1) % V.P, w0[1])
    with pytest.raises(AssertionError):
        D.verify(circ, V.write_proof_with_pis(proof, pis))


def test_every_gate_of_the_fixture_is_pinned_by_its_identity(golden_dir, monkeypatch):
    """Perturbing the last constraint of any one gate kind of the regression circuit breaks the identity: the reference's proof
    pins each evaluator separately (a filter is non-zero at a random zeta, so every gate contributes)."""
    from oracle import gates as G
    from oracle import plonk_dummy as D
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    circ = D.CommonDataCircuit(rd("recursive_verifier_gl_common_data.bin"), rd("recursive_verifier_gl_verifier_data.bin"))
    raw = rd("recursive_verifier_gl_proof.bin")
    orig = G.eval_unfiltered
    for kind in sorted({g[0] for g in circ.gate_table} - {G.NOOP}):
        def perturbed(e, gate, wires, consts, pi_hash, kind=kind):
            out = list(orig(e, gate, wires, consts, pi_hash))
            if gate[0] == kind:
                out[-1] = e.eadd(out[-1], e.one)
            return out
        monkeypatch.setattr(G, "eval_unfiltered", perturbed)
        with pytest.raises(AssertionError, match="vanishing"):
            D.verify(circ, raw)
    monkeypatch.setattr(G, "eval_unfiltered", orig)
    assert D.verify(circ, raw)
