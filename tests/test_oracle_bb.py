"""BabyBear oracle: self-consistency only - the reference holds no KAT or serialized proof for this field
(BabyBear parity is UNPINNED).  What can be checked without the reference: the recalled
field constants have the required algebraic properties, the NTT evaluates polynomials, Merkle round trips."""
import numpy as np
import pytest

from oracle import oracle_bb as B

P = B.BB_P


def test_field_constants_are_consistent():
    assert P == 2 ** 31 - 2 ** 27 + 1 and P - 1 == (1 << 27) * 15
    for q in (2, 3, 5):  # 31 generates F_p^*
        assert pow(31, (P - 1) // q, P) != 1
    g = 0x1a427a41
    assert pow(g, 1 << 27, P) == 1 and pow(g, 1 << 26, P) == P - 1  # order exactly 2^27
    assert int(B.lib().gbo_bb_two_adic_generator(27)) == g
    assert int(B.lib().gbo_bb_two_adic_generator(1)) == P - 1
    assert 943718400 * (1 << 32) % P == 1  # the internal-layer factor is 2^-32 (gates/poseidon2_babybear.rs:789)


def test_poseidon2_structure():
    # deterministic, input-sensitive, and equal to a from-scratch python transcription of
    # gates/poseidon2_babybear.rs:609-672,787-832,903-917
    import json, os, re
    hdr = open(os.path.join(os.path.dirname(B.__file__), "poseidon_constants.h")).read()
    def grab(name):
        m = re.search(r"#define %s_LIST \\\n(.*?)\n(?:#define|$)" % name, hdr, re.S)
        return [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", m.group(1))]
    ext, internal = grab("BB_POSEIDON2_EXTERNAL_CONSTANTS"), grab("BB_POSEIDON2_INTERNAL_CONSTANTS")
    assert len(ext) == 128 and len(internal) == 13 and max(ext + internal) < P

    def mat4(x):
        t01, t23 = x[0] + x[1], x[2] + x[3]
        t0123 = t01 + t23
        t01123, t01233 = t0123 + x[1], t0123 + x[3]
        return [(t01123 + t01) % P, (t01123 + 2 * x[2]) % P, (t01233 + t23) % P, (t01233 + 2 * x[0]) % P]

    def ext_layer(s):
        s = sum((mat4(s[i:i + 4]) for i in range(0, 16, 4)), [])
        sums = [sum(s[j + k] for j in range(0, 16, 4)) % P for k in range(4)]
        return [(s[i] + sums[i % 4]) % P for i in range(16)]

    def int_layer(s):
        s = [x * 943718400 % P for x in s]
        part = sum(s[1:]) % P
        full = (part + s[0]) % P
        shifts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
        return [(part - s[0]) % P] + [(full + s[i + 1] * (1 << shifts[i])) % P for i in range(15)]

    def perm(s):
        s = ext_layer(list(s))
        for r in range(4):
            s = ext_layer([pow((s[i] + ext[16 * r + i]) % P, 7, P) for i in range(16)])
        for r in range(13):
            s[0] = pow((s[0] + internal[r]) % P, 7, P)
            s = int_layer(s)
        for r in range(4, 8):
            s = ext_layer([pow((s[i] + ext[16 * r + i]) % P, 7, P) for i in range(16)])
        return s

    for st in ([0] * 16, list(range(16)), [P - 1] * 16, B.fill(3, 16).tolist()):
        assert B.poseidon2(st).tolist() == perm(st)


def _eval_naive(coeffs, lg, shift=1):
    w = int(B.lib().gbo_bb_two_adic_generator(lg))
    out = []
    for j in range(1 << lg):
        x = shift * pow(w, j, P) % P
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + int(c)) % P
        out.append(acc)
    return out


def test_fft_ifft_coset():
    # field/src/fft.rs:219-253 shape, BabyBear instance
    coeffs = [(i * 1337) % 100 for i in range(200)] + [0] * 56
    pts = B.fft(coeffs)
    assert pts.tolist() == _eval_naive(coeffs, 8)
    assert B.ifft(pts).tolist() == coeffs
    for r in range(4):
        tail = coeffs + [0] * (256 * ((1 << r) - 1))
        assert (B.fft(tail) == B.fft(tail, zero_factor=r)).all()
    c = B.fill(9, 64)
    assert B.coset_fft(c, 31).tolist() == _eval_naive(c.tolist(), 6, 31)


@pytest.mark.parametrize("cap_height", [0, 1, 8])
def test_merkle_round_trip(cap_height):
    # hash/merkle_tree.rs:239-304 with H = 8 (hash_or_noop pads up to 8 elements)
    vals = B.fill(7, 256 * 9).reshape(256, 9)
    b = B.PolynomialBatch.from_coeffs(vals.T.copy()[:, :32].copy(), 3, cap_height) if False else None
    cols = B.fill(5, 9 * 32).reshape(9, 32)
    bt = B.PolynomialBatch.from_values(cols, 3, cap_height)
    for i in range(256):
        assert B.merkle_verify(bt.leaves[i], i, bt.cap, bt.prove(i))
    assert B.hash_or_noop([1, 2, 3]).tolist() == [1, 2, 3, 0, 0, 0, 0, 0]
    del b, vals


def test_batch_matches_definition():
    lg_n, r, ncols = 4, 3, 5
    vals = B.fill(11, ncols << lg_n).reshape(ncols, 1 << lg_n)
    b = B.PolynomialBatch.from_values(vals, r, 2)
    for c in range(ncols):
        assert (B.fft(b.polynomials[c]) == vals[c]).all()
        lde = _eval_naive(b.polynomials[c].tolist(), lg_n + r, 31)
        for i in range(1 << (lg_n + r)):
            assert b.get_lde_values(i, 1)[c] == lde[i]


def test_reference_kat_poseidon2_r0_babybear(kats):
    """The one BabyBear known-answer test the reference holds (hash/poseidon2_risc0_babybear.rs:321-342,
    `test_against_r0_values`): a width-24 Poseidon2 with RISC0's parameters.  It is not the hot path's hash, but it
    runs on the same field code and the same Poseidon2 round order as the width-16 permutation (one generic
    p3_poseidon2::Poseidon2 drives both), so matching it pins - from the reference's own numbers - the BabyBear
    modulus, canonical arithmetic, the x^7 s-box and the round structure.  Still unpinned afterwards: the multiplicative
    generator 31, the two-adic generator and the extension non-residue 11."""
    k = kats["poseidon2_r0_babybear"]
    out = B.poseidon2_r0(np.array(k["input"], dtype=np.uint32))
    assert [int(x) for x in out] == k["output"]


def test_in_repo_monty_inverse_constant_matches_the_modulus():
    # gates/poseidon2_babybear.rs:776,790 multiply by 943718400 "= 2^-32 mod p" (the Montgomery R^-1): a second in-repo
    # number that only makes sense for p = 2^31 - 2^27 + 1
    assert 943718400 * (1 << 32) % B.BB_P == 1
