"""TEST ORACLE loader for the BabyBear restatement (oracle_bb.c).  Test infrastructure only.
PARITY UNPINNED for the BabyBear field constants (see oracle_bb.c header / DESIGN.md)."""
import ctypes as C

import numpy as np

from . import oracle as O

BB_P = 2013265921
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_done = False


def lib():
    global _done
    L = O.lib()
    if not _done:
        L.gbo_bb_poseidon2.argtypes = [_u32p, _u32p]
        L.gbo_bb_hash_no_pad.argtypes = [_u32p, C.c_size_t, _u32p]
        L.gbo_bb_hash_or_noop.argtypes = [_u32p, C.c_size_t, _u32p]
        L.gbo_bb_two_to_one.argtypes = [_u32p, _u32p, _u32p]
        L.gbo_bb_fft.argtypes = [_u32p, C.c_uint, C.c_uint]
        L.gbo_bb_ifft.argtypes = [_u32p, C.c_uint]
        L.gbo_bb_coset_fft.argtypes = [_u32p, C.c_uint, C.c_uint32, C.c_uint]
        L.gbo_bb_merkle_tree.argtypes = [_u32p, C.c_size_t, C.c_size_t, C.c_uint, _u32p, _u32p]
        L.gbo_bb_merkle_tree.restype = C.c_int
        L.gbo_bb_merkle_prove.argtypes = [_u32p, C.c_size_t, C.c_uint, C.c_size_t, _u32p]
        L.gbo_bb_merkle_prove.restype = C.c_int
        L.gbo_bb_merkle_verify.argtypes = [_u32p, C.c_size_t, C.c_size_t, _u32p, _u32p, C.c_uint]
        L.gbo_bb_merkle_verify.restype = C.c_int
        L.gbo_bb_commit.argtypes = [_u32p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_void_p, _u32p, _u32p, _u32p, _u32p]
        L.gbo_bb_commit.restype = C.c_int
        L.gbo_bb_two_adic_generator.argtypes = [C.c_uint]
        L.gbo_bb_two_adic_generator.restype = C.c_uint32
        L.gbo_bb_poseidon2_r0.argtypes = [_u32p, _u32p]
        L.gbo_bb_powers.argtypes = [C.c_uint32, C.c_size_t, _u32p]
        L.gbo_bb_scale_vec.argtypes = [_u32p, C.c_uint32, C.c_size_t, _u32p]
        _done = True
    return L


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint32)


def _lg(n):
    lg = int(n).bit_length() - 1
    assert 1 << lg == n
    return lg


def poseidon2(state):
    out = np.empty(16, dtype=np.uint32)
    lib().gbo_bb_poseidon2(_a(state), out)
    return out


def poseidon2_r0(state):
    """Poseidon2-24 with the RISC0 parameters (hash/poseidon2_risc0_babybear.rs) - KAT pin only, not on the hot path"""
    out = np.empty(24, dtype=np.uint32)
    lib().gbo_bb_poseidon2_r0(_a(state), out)
    return out


def hash_no_pad(x):
    x = _a(x)
    out = np.empty(8, dtype=np.uint32)
    lib().gbo_bb_hash_no_pad(x, x.size, out)
    return out


def hash_or_noop(x):
    x = _a(x)
    out = np.empty(8, dtype=np.uint32)
    lib().gbo_bb_hash_or_noop(x, x.size, out)
    return out


def two_to_one(l, r):
    out = np.empty(8, dtype=np.uint32)
    lib().gbo_bb_two_to_one(_a(l), _a(r), out)
    return out


def fft(c, zero_factor=0):
    v = _a(c).copy()
    lib().gbo_bb_fft(v, _lg(v.size), zero_factor)
    return v


def ifft(vals):
    v = _a(vals).copy()
    lib().gbo_bb_ifft(v, _lg(v.size))
    return v


def coset_fft(c, shift=31, zero_factor=0):
    v = _a(c).copy()
    lib().gbo_bb_coset_fft(v, _lg(v.size), shift, zero_factor)
    return v


def merkle_verify(leaf, index, cap, siblings):
    leaf, cap, siblings = _a(leaf), _a(cap), _a(siblings).reshape(-1, 8)
    sib = siblings if siblings.size else np.zeros((1, 8), np.uint32)
    return bool(lib().gbo_bb_merkle_verify(leaf, leaf.size, index, cap, sib, siblings.shape[0]))


def fill(seed, count):
    """synthetic canonical BabyBear elements (SplitMix64 reduced mod p)"""
    return (O.splitmix64_fill(seed, count, modulus=(1 << 64) - 1) % np.uint64(BB_P)).astype(np.uint32)


def powers(base, n):
    out = np.empty(n, dtype=np.uint32)
    lib().gbo_bb_powers(int(base), n, out)
    return out


def scale_vec(a, k):
    a = _a(a)
    out = np.empty_like(a)
    lib().gbo_bb_scale_vec(a, int(k), a.size, out)
    return out


class PolynomialBatch:
    """fri/oracle.rs:29-158 over BabyBear on the CPU"""

    def __init__(self, cols, rate_bits, cap_height, is_coeffs=False, salts=None):
        cols = _a(cols)
        ncols, n = cols.shape
        self.degree_log, self.rate_bits, self.cap_height = _lg(n), rate_bits, cap_height
        self.blinding = salts is not None
        N = n << rate_bits
        width = ncols + (4 if self.blinding else 0)
        if cap_height > self.degree_log + rate_bits:
            raise ValueError("cap_height too large")
        self.polynomials = np.empty((ncols, n), dtype=np.uint32)
        self.leaves = np.empty((N, width), dtype=np.uint32)
        self.digests = np.zeros((2 * (N - (1 << cap_height)), 8), dtype=np.uint32)
        self.cap = np.zeros((1 << cap_height, 8), dtype=np.uint32)
        sp = None
        if self.blinding:
            self._salts = _a(salts)
            sp = self._salts.ctypes.data
        rc = lib().gbo_bb_commit(cols, ncols, self.degree_log, rate_bits, cap_height, int(is_coeffs), sp, self.polynomials,
                                 self.leaves, self.digests, self.cap)
        assert rc == 0, rc

    @classmethod
    def from_values(cls, values, rate_bits, cap_height, salts=None):
        return cls(values, rate_bits, cap_height, False, salts)

    @classmethod
    def from_coeffs(cls, coeffs, rate_bits, cap_height, salts=None):
        return cls(coeffs, rate_bits, cap_height, True, salts)

    def get_lde_values(self, index, step):
        bits = self.degree_log + self.rate_bits
        i = int(format(index * step, "0%db" % bits)[::-1], 2) if bits else 0
        row = self.leaves[i]
        return row[: row.size - (4 if self.blinding else 0)]

    def prove(self, i):
        n = self.degree_log + self.rate_bits - self.cap_height
        sib = np.zeros((max(n, 1), 8), dtype=np.uint32)
        k = lib().gbo_bb_merkle_prove(self.digests if self.digests.size else np.zeros((1, 8), np.uint32),
                                      self.degree_log + self.rate_bits, self.cap_height, i, sib)
        return sib[:k]
