"""TEST ORACLE loader: ctypes view of oracle/_build/liboracle.so (the CPU restatement).

Test infrastructure only - imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

GL_P = 0xFFFFFFFF00000001
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build with make when sources are newer than the .so (gcc is on both boxes)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.gbo_gl_poseidon.argtypes = [_u64p, _u64p]
        L.gbo_gl_poseidon_naive.argtypes = [_u64p, _u64p]
        L.gbo_gl_hash_no_pad.argtypes = [_u64p, C.c_size_t, _u64p]
        L.gbo_gl_hash_n_to_m_no_pad.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t]
        L.gbo_gl_hash_or_noop.argtypes = [_u64p, C.c_size_t, _u64p]
        L.gbo_gl_two_to_one.argtypes = [_u64p, _u64p, _u64p]
        L.gbo_reverse_index_bits_u64.argtypes = [_u64p, C.c_uint]
        L.gbo_gl_fft.argtypes = [_u64p, C.c_uint, C.c_uint]
        L.gbo_gl_ifft.argtypes = [_u64p, C.c_uint]
        L.gbo_gl_coset_fft.argtypes = [_u64p, C.c_uint, C.c_uint64, C.c_uint]
        L.gbo_gl_coset_ifft.argtypes = [_u64p, C.c_uint, C.c_uint64]
        L.gbo_gl_merkle_tree.argtypes = [_u64p, C.c_size_t, C.c_size_t, C.c_uint, _u64p, _u64p]
        L.gbo_gl_merkle_tree.restype = C.c_int
        L.gbo_gl_merkle_prove.argtypes = [_u64p, C.c_size_t, C.c_uint, C.c_size_t, _u64p]
        L.gbo_gl_merkle_prove.restype = C.c_int
        L.gbo_gl_merkle_verify.argtypes = [_u64p, C.c_size_t, C.c_size_t, _u64p, _u64p, C.c_uint]
        L.gbo_gl_merkle_verify.restype = C.c_int
        L.gbo_gl_commit.argtypes = [_u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_void_p,
                                    _u64p, _u64p, _u64p, _u64p]
        L.gbo_gl_commit.restype = C.c_int
        L.gbo_gl_challenger_sizeof.restype = C.c_size_t
        L.gbo_gl_challenger_init.argtypes = [C.c_void_p]
        L.gbo_gl_challenger_observe.argtypes = [C.c_void_p, _u64p, C.c_size_t]
        L.gbo_gl_challenger_get.argtypes = [C.c_void_p]
        L.gbo_gl_challenger_get.restype = C.c_uint64
        L.gbo_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
        L.gbo_gl_mul.restype = C.c_uint64
        L.gbo_gl_powu.argtypes = [C.c_uint64, C.c_uint64]
        L.gbo_gl_powu.restype = C.c_uint64
        L.gbo_gl_inv.argtypes = [C.c_uint64]
        L.gbo_gl_inv.restype = C.c_uint64
        L.gbo_gl_two_adic_generator.argtypes = [C.c_uint]
        L.gbo_gl_two_adic_generator.restype = C.c_uint64
        L.gbo_gl_powers.argtypes = [C.c_uint64, C.c_size_t, _u64p]
        L.gbo_gl_scale_vec.argtypes = [_u64p, C.c_uint64, C.c_size_t, _u64p]
        L.gbo_num_threads.restype = C.c_int
        L.gbo_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def poseidon(state, naive=False):
    out = np.empty(12, dtype=np.uint64)
    (lib().gbo_gl_poseidon_naive if naive else lib().gbo_gl_poseidon)(_a(state), out)
    return out


def hash_no_pad(x):
    x = _a(x)
    out = np.empty(4, dtype=np.uint64)
    lib().gbo_gl_hash_no_pad(x, x.size, out)
    return out


def hash_n_to_m_no_pad(x, m):
    x = _a(x)
    out = np.empty(m, dtype=np.uint64)
    lib().gbo_gl_hash_n_to_m_no_pad(x, x.size, out, m)
    return out


def hash_or_noop(x):
    x = _a(x)
    out = np.empty(4, dtype=np.uint64)
    lib().gbo_gl_hash_or_noop(x, x.size, out)
    return out


def two_to_one(l, r):
    out = np.empty(4, dtype=np.uint64)
    lib().gbo_gl_two_to_one(_a(l), _a(r), out)
    return out


def reverse_index_bits(a):
    a = _a(a).copy()
    lg = int(a.size).bit_length() - 1
    assert 1 << lg == a.size
    lib().gbo_reverse_index_bits_u64(a, lg)
    return a


def _lg(n):
    lg = int(n).bit_length() - 1
    assert 1 << lg == n, "length must be a power of two"
    return lg


def fft(coeffs, zero_factor=0):
    v = _a(coeffs).copy()
    lib().gbo_gl_fft(v, _lg(v.size), zero_factor)
    return v


def ifft(values):
    v = _a(values).copy()
    lib().gbo_gl_ifft(v, _lg(v.size))
    return v


def coset_fft(coeffs, shift=7, zero_factor=0):
    v = _a(coeffs).copy()
    lib().gbo_gl_coset_fft(v, _lg(v.size), shift, zero_factor)
    return v


def coset_ifft(values, shift=7):
    v = _a(values).copy()
    lib().gbo_gl_coset_ifft(v, _lg(v.size), shift)
    return v


class MerkleTree:
    """hash/merkle_tree.rs: leaves [L][width], digests in the reference's interleaved layout, cap."""

    def __init__(self, leaves, cap_height):
        leaves = _a(leaves)
        L, width = leaves.shape
        self.log_l = _lg(L)
        if cap_height > self.log_l:
            raise ValueError("cap_height=%d should be at most log2(leaves.len())=%d" % (cap_height, self.log_l))
        self.leaves, self.cap_height = leaves, cap_height
        self.digests = np.zeros((2 * (L - (1 << cap_height)), 4), dtype=np.uint64)
        self.cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        rc = lib().gbo_gl_merkle_tree(leaves, self.log_l, width, cap_height, self.digests, self.cap)
        assert rc == 0

    def prove(self, i):
        n = self.log_l - self.cap_height
        sib = np.zeros((max(n, 1), 4), dtype=np.uint64)
        k = lib().gbo_gl_merkle_prove(self.digests if self.digests.size else np.zeros((1, 4), np.uint64),
                                      self.log_l, self.cap_height, i, sib)
        return sib[:k]


def merkle_verify(leaf, index, cap, siblings):
    leaf, cap, siblings = _a(leaf), _a(cap), _a(siblings).reshape(-1, 4)
    sib = siblings if siblings.size else np.zeros((1, 4), np.uint64)
    return bool(lib().gbo_gl_merkle_verify(leaf, leaf.size, index, cap, sib, siblings.shape[0]))


class PolynomialBatch:
    """fri/oracle.rs:29-158 on the CPU: polynomials (coeffs), merkle tree, get_lde_values."""

    def __init__(self, cols, rate_bits, cap_height, is_coeffs=False, salts=None):
        cols = _a(cols)
        ncols, n = cols.shape
        self.degree_log, self.rate_bits, self.cap_height = _lg(n), rate_bits, cap_height
        self.blinding = salts is not None
        N = n << rate_bits
        width = ncols + (4 if self.blinding else 0)
        if cap_height > self.degree_log + rate_bits:
            raise ValueError("cap_height too large")
        self.polynomials = np.empty((ncols, n), dtype=np.uint64)
        self.leaves = np.empty((N, width), dtype=np.uint64)
        self.digests = np.zeros((2 * (N - (1 << cap_height)), 4), dtype=np.uint64)
        self.cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        sp = None
        if self.blinding:
            self._salts = _a(salts)
            assert self._salts.shape == (4, N)
            sp = self._salts.ctypes.data
        rc = lib().gbo_gl_commit(cols, ncols, self.degree_log, rate_bits, cap_height, int(is_coeffs), sp,
                                 self.polynomials, self.leaves, self.digests, self.cap)
        assert rc == 0, rc

    @classmethod
    def from_values(cls, values, rate_bits, cap_height, salts=None):
        return cls(values, rate_bits, cap_height, False, salts)

    @classmethod
    def from_coeffs(cls, coeffs, rate_bits, cap_height, salts=None):
        return cls(coeffs, rate_bits, cap_height, True, salts)

    def get_lde_values(self, index, step):
        """fri/oracle.rs:153-158"""
        bits = self.degree_log + self.rate_bits
        i = int(format(index * step, "0%db" % bits)[::-1], 2) if bits else 0
        row = self.leaves[i]
        return row[: row.size - (4 if self.blinding else 0)]

    def prove(self, i):
        n = self.degree_log + self.rate_bits - self.cap_height
        sib = np.zeros((max(n, 1), 4), dtype=np.uint64)
        k = lib().gbo_gl_merkle_prove(self.digests if self.digests.size else np.zeros((1, 4), np.uint64),
                                      self.degree_log + self.rate_bits, self.cap_height, i, sib)
        return sib[:k]


class Challenger:
    """iop/challenger.rs:18-150"""

    def __init__(self):
        self._buf = C.create_string_buffer(lib().gbo_gl_challenger_sizeof())
        lib().gbo_gl_challenger_init(self._buf)

    def observe_elements(self, xs):
        xs = _a(xs).ravel()
        if xs.size:
            lib().gbo_gl_challenger_observe(self._buf, xs, xs.size)

    observe_cap = observe_elements
    observe_hash = observe_elements

    def observe_element(self, x):
        self.observe_elements([x])

    def get_challenge(self):
        return int(lib().gbo_gl_challenger_get(self._buf))

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self, d=2):
        return tuple(self.get_n_challenges(d))

    def clone(self):
        c = Challenger.__new__(Challenger)
        c._buf = C.create_string_buffer(self._buf.raw, len(self._buf))
        return c

    def state(self):
        return np.frombuffer(self._buf.raw, dtype=np.uint64).copy()


def powers(base, n):
    out = np.empty(n, dtype=np.uint64)
    lib().gbo_gl_powers(int(base), n, out)
    return out


def scale_vec(a, k):
    a = _a(a)
    out = np.empty_like(a)
    lib().gbo_gl_scale_vec(a, int(k), a.size, out)
    return out


def splitmix64_fill(seed, count, modulus=GL_P):
    """SplitMix64 stream reduced mod p (synthetic canonical field elements)."""
    out = np.empty(count, dtype=np.uint64)
    x = np.uint64(seed)
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out[:] = z % np.uint64(modulus)
    return out


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota_us)."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def use_host_cpu_share():
    """Size the OpenMP pool to host_cpu_share(); returns the thread count in use."""
    n = host_cpu_share()
    lib().gbo_set_num_threads(n)
    return int(lib().gbo_num_threads())
