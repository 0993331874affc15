"""TEST ORACLE - field descriptors (base field + binomial extension) for the python checker.

Goldilocks: p = 2^64 - 2^32 + 1, D = 2, x^2 - 7, H = 4   (pinned by the reference's fixtures)
BabyBear:   p = 2^31 - 2^27 + 1, D = 4, x^4 - 11, H = 8  (UNPINNED: recalled from upstream Plonky3)
"""
import ctypes as C

import numpy as np

from . import oracle as O
from . import oracle_bb as B


class Field:
    def __init__(self, name, P, D, W, hout, dtype, generator, two_adic_gen, two_adicity, order_bits, hash_no_pad,
                 merkle_verify, challenger_cls, mod):
        self.name, self.P, self.D, self.W, self.hout, self.dtype = name, P, D, W, hout, dtype
        self.generator, self._tag, self.two_adicity, self.order_bits = generator, two_adic_gen, two_adicity, order_bits
        self.hash_no_pad, self.merkle_verify, self.Challenger = hash_no_pad, merkle_verify, challenger_cls
        self.elem_bytes = np.dtype(dtype).itemsize
        self.zero, self.one = (0,) * D, (1,) + (0,) * (D - 1)
        self.mod = mod  # python module with powers / scale_vec / PolynomialBatch over this field
        self.prove_symbol = "gbo_gl_prove_dummy" if name == "goldilocks" else "gbo_bb_prove_dummy"

    def fill(self, seed, count):
        """synthetic canonical elements: SplitMix64 stream reduced mod p"""
        if self.name == "goldilocks":
            return O.splitmix64_fill(seed, count)
        return B.fill(seed, count)

    # base field
    def two_adic_generator(self, bits):
        return pow(self._tag, 1 << (self.two_adicity - bits), self.P)

    def finv(self, a):
        return pow(a, self.P - 2, self.P)

    # binomial extension F[x]/(x^D - W), elements are D-tuples
    def efrom(self, x):
        return (x % self.P,) + (0,) * (self.D - 1)

    def eadd(self, a, b):
        return tuple((x + y) % self.P for x, y in zip(a, b))

    def esub(self, a, b):
        return tuple((x - y) % self.P for x, y in zip(a, b))

    def emul(self, a, b):
        D, P, W = self.D, self.P, self.W
        r = [0] * (2 * D - 1)
        for i in range(D):
            if a[i]:
                for j in range(D):
                    r[i + j] += a[i] * b[j]
        for k in range(2 * D - 2, D - 1, -1):
            r[k - D] += W * r[k]
        return tuple(x % P for x in r[:D])

    def epow(self, a, e):
        r = self.one
        while e:
            if e & 1:
                r = self.emul(r, a)
            a = self.emul(a, a)
            e >>= 1
        return r

    def einv(self, a):
        return self.epow(a, self.P ** self.D - 2)

    def ediv(self, a, b):
        return self.emul(a, self.einv(b))

    def escale(self, a, s):
        return tuple(x * s % self.P for x in a)


class _BbChallenger:
    """iop/challenger.rs:18-150 over BabyBear / Poseidon2-16 (rate 8)"""

    def __init__(self):
        self.state = np.zeros(16, dtype=np.uint32)
        self.inp, self.out = [], []

    def _duplex(self):
        for i, v in enumerate(self.inp):
            self.state[i] = v
        self.inp = []
        self.state = B.poseidon2(self.state)
        self.out = [int(x) for x in self.state[:8]]

    def observe_element(self, x):
        self.out = []
        self.inp.append(int(x))
        if len(self.inp) == 8:
            self._duplex()

    def observe_elements(self, xs):
        for x in np.asarray(xs).ravel().tolist():
            self.observe_element(x)

    observe_cap = observe_elements
    observe_hash = observe_elements

    def get_challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self, d=4):
        return tuple(self.get_n_challenges(d))


GL = Field("goldilocks", O.GL_P, 2, 7, 4, np.uint64, 7, 1753635133440165772, 32, 64, O.hash_no_pad, O.merkle_verify,
           O.Challenger, O)
BB = Field("babybear", B.BB_P, 4, 11, 8, np.uint32, 31, 0x1a427a41, 27, 31, B.hash_no_pad, B.merkle_verify, _BbChallenger, B)
