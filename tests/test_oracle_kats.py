"""The CPU oracle against the reference's own known-answer vectors (no GPU)."""
import numpy as np
import pytest

from oracle import oracle as O

P = O.GL_P


def test_poseidon12_reference_kats(kats):
    # hash/poseidon_goldilocks.rs:1158-1193 test_vectors
    for v in kats["poseidon12"]:
        want = np.array(v["output"], dtype=np.uint64)
        assert (O.poseidon(v["input"]) == want).all()
        assert (O.poseidon(v["input"], naive=True) == want).all()


def test_poseidon12_fast_equals_naive_random():
    # hash/poseidon_goldilocks.rs:1196-1198 consistency, widened to random + edge states
    rng = np.random.default_rng(1)
    states = [O.splitmix64_fill(s, 12) for s in range(50)]
    states.append(np.full(12, P - 1, dtype=np.uint64))
    states.append(np.array([P - 1, 0] * 6, dtype=np.uint64))
    for s in states:
        assert (O.poseidon(s) == O.poseidon(s, naive=True)).all()
    del rng


def test_reverse_index_bits_table(kats):
    # plonky2/src/util/mod.rs:59-83
    assert O.reverse_index_bits(np.arange(256)).tolist() == kats["reverse_index_bits_256"]
    assert O.reverse_index_bits([10, 20, 30, 40]).tolist() == [10, 30, 20, 40]
    for lg in (16, 17):  # :101-147 round trips
        a = O.splitmix64_fill(lg, 1 << lg)
        assert (O.reverse_index_bits(O.reverse_index_bits(a)) == a).all()


def _eval_naive(coeffs, lg, shift=1):
    w = pow(1753635133440165772, 1 << (32 - lg), P)
    n = 1 << lg
    out = []
    for j in range(n):
        x = shift * pow(w, j, P) % P
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + int(c)) % P
        out.append(acc)
    return out


def test_fft_and_ifft_reference_case():
    # field/src/fft.rs:219-253 fft_and_ifft: degree 200, coeffs i*1337 % 100, zero-factor shortcut
    degree, n = 200, 256
    coeffs = [(i * 1337) % 100 for i in range(degree)] + [0] * (n - degree)
    points = O.fft(coeffs)
    assert points.tolist() == _eval_naive(coeffs, 8)
    assert O.ifft(points).tolist() == coeffs
    for r in range(4):
        tail = coeffs + [0] * (n * ((1 << r) - 1))
        assert (O.fft(tail) == O.fft(tail, zero_factor=r)).all()


def test_coset_fft_ifft_vs_naive():
    # field/src/polynomial/mod.rs:479-516
    lg = 6
    coeffs = O.splitmix64_fill(99, 1 << lg)
    shift = 7
    vals = O.coset_fft(coeffs, shift)
    assert vals.tolist() == _eval_naive(coeffs.tolist(), lg, shift)
    assert (O.coset_ifft(vals, shift) == coeffs).all()


def test_hash_or_noop_and_two_to_one():
    # plonk/config.rs:70-84, hash/hashing.rs:76-96
    x = np.array([5, 6, 7], dtype=np.uint64)
    assert O.hash_or_noop(x).tolist() == [5, 6, 7, 0]
    l, r = O.splitmix64_fill(1, 4), O.splitmix64_fill(2, 4)
    st = np.concatenate([l, r, np.zeros(4, np.uint64)])
    assert (O.two_to_one(l, r) == O.poseidon(st)[:4]).all()
    y = O.splitmix64_fill(3, 19)
    st = np.zeros(12, np.uint64)
    for off in range(0, 19, 8):
        chunk = y[off:off + 8]
        st[:chunk.size] = chunk
        st = O.poseidon(st)
    assert (O.hash_no_pad(y) == st[:4]).all()


@pytest.mark.parametrize("cap_height", [0, 1, 3, 8])
def test_merkle_all_leaves_round_trip(cap_height):
    # hash/merkle_tree.rs:239-304: n=256, leaf width 7, every leaf proves to the cap
    leaves = O.splitmix64_fill(7, 256 * 7).reshape(256, 7)
    t = O.MerkleTree(leaves, cap_height)
    assert t.digests.shape[0] == 2 * (256 - (1 << cap_height))
    for i in range(256):
        sib = t.prove(i)
        assert sib.shape[0] == 8 - cap_height
        assert O.merkle_verify(leaves[i], i, t.cap, sib)
    bad = leaves[3].copy()
    bad[0] ^= 1
    assert not O.merkle_verify(bad, 3, t.cap, t.prove(3))


def test_merkle_cap_height_too_big():
    # hash/merkle_tree.rs:257-272 should_panic
    with pytest.raises(ValueError):
        O.MerkleTree(np.zeros((256, 7), np.uint64), 9)


def test_challenger_no_duplicate_challenges():
    # iop/challenger.rs tests: no_duplicate_challenges
    ch = O.Challenger()
    seen = set()
    for i in range(10):
        ch.observe_element(i + 1)
        for _ in range(5):
            seen.add(ch.get_challenge())
    assert len(seen) == 50


def test_polynomial_batch_matches_definition():
    # fri/oracle.rs:68-158: leaf i holds P_c(7 * w_N^bitrev(i)); get_lde_values(i, step)
    lg_n, r, ncols = 4, 3, 5
    vals = O.splitmix64_fill(11, ncols << lg_n).reshape(ncols, 1 << lg_n)
    b = O.PolynomialBatch.from_values(vals, r, 2)
    for c in range(ncols):
        assert (O.fft(b.polynomials[c]) == vals[c]).all()
        lde = _eval_naive(b.polynomials[c].tolist(), lg_n + r, 7)
        for i in range(1 << (lg_n + r)):
            assert b.get_lde_values(i, 1)[c] == lde[i]
    t = O.MerkleTree(b.leaves, 2)
    assert (t.cap == b.cap).all() and (t.digests == b.digests).all()
    b2 = O.PolynomialBatch.from_coeffs(b.polynomials, r, 2)
    assert (b2.cap == b.cap).all()
