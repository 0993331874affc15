"""Host-side restatement of the reference's dummy circuit (bench form), product side, numpy only.

`dummy_proof` of plonky2/examples/bench_recursion.rs:87-122 builds 2^(k-1)+1 NoopGates; `build()`
(plonky2/src/plonk/circuit_builder.rs:1110-1312) appends the PublicInputGate row (wires 0..3 tied to
the `zero` constant, the rest randomised), one ConstantGate row for that constant, pads to 2^k rows,
and emits one selector column, `num_constants` constant columns and the sigma columns.  This module
produces exactly those inputs for `CircuitData` (constants_sigmas values, k_is) and a `MatrixWitness`
(iop/witness.rs:359-371: unset wires are zero).  It is the synthetic-input generator of bench.py and
the piece a Rust host already has; it is not on the GPU hot path.
"""
import numpy as np

P = 0xFFFFFFFF00000001
_M32 = np.uint64(0xFFFFFFFF)
_EPS = np.uint64(0xFFFFFFFF)
_P = np.uint64(P)

GATE_NOOP, GATE_CONSTANT, GATE_PI = 0, 1, 2  # sorted by (degree, id) (circuit_builder.rs:1195-1196)


def gl_mul(a, b):
    """Vectorised Goldilocks multiplication on uint64 numpy arrays (canonical in, canonical out)."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & _M32, a >> np.uint64(32), b & _M32, b >> np.uint64(32)
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> np.uint64(32)) + (p01 & _M32) + (p10 & _M32)
        lo = (p00 & _M32) | ((mid & _M32) << np.uint64(32))
        hi = p11 + (p01 >> np.uint64(32)) + (p10 >> np.uint64(32)) + (mid >> np.uint64(32))
        # (lo + 2^64 hi) mod p: 2^64 = 2^32 - 1, 2^96 = -1
        hh, hl = hi >> np.uint64(32), hi & _M32
        t0 = lo - hh
        t0 = np.where(lo < hh, t0 - _EPS, t0)
        t1 = (hl << np.uint64(32)) - hl
        t2 = t0 + t1
        t2 = np.where(t2 < t0, t2 + _EPS, t2)
        return np.where(t2 >= _P, t2 - _P, t2)


def gl_powers(base, n):
    """[base^0 .. base^(n-1)] by doubling (n a power of two)."""
    out = np.ones(n, dtype=np.uint64)
    m, bm = 1, int(base) % P
    while m < n:
        out[m:2 * m] = gl_mul(out[:m], np.uint64(bm))
        bm = bm * bm % P
        m *= 2
    return out


def gl_mul7(x):
    """7 x mod p for a canonical uint64 array: 8 x = (x << 3) + (x >> 61) 2^64 with 2^64 = 2^32 - 1, minus x - about ten array
    operations where gl_mul takes twenty-five"""
    with np.errstate(over="ignore"):
        lo = x << np.uint64(3)
        s = lo + (x >> np.uint64(61)) * _EPS
        s = np.where(s < lo, s + _EPS, s)
        r = s - x
        r = np.where(s < x, r - _EPS, r)
        return np.where(r >= _P, r - _P, r)


def _sigma_columns(sig, sub, k_is, step, workers=None):
    """sig[j] = k_is[j] * sub with k_is[j] = k_is[j-1] * g (cosets.rs:8-21): one multiplication by the small generator per column,
    row blocks in parallel (numpy releases the GIL), each block staying in cache across the columns"""
    import concurrent.futures
    import os
    n = sub.shape[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    workers = workers or max(1, min(cores, 16))
    block = max(1 << 12, min(1 << 16, n // workers or n))

    def run(lo):
        hi = min(n, lo + block)
        cur = sub[lo:hi].copy()      # k_is[0] = 1
        for j in range(sig.shape[0]):
            if j:
                cur = step(cur)
            sig[j, lo:hi] = cur
    if workers == 1 or n <= block:
        for lo in range(0, n, block):
            run(lo)
    else:
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, range(0, n, block)))


def splitmix64(seed, count):
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % _P


BB_P = 2013265921  # 2^31 - 2^27 + 1


def bb_mul(a, b):
    return (np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64) % np.uint64(BB_P)).astype(np.uint32)


def bb_powers(base, n):
    out = np.ones(n, dtype=np.uint32)
    m, bm = 1, int(base) % BB_P
    while m < n:
        out[m:2 * m] = bb_mul(out[:m], bm)
        bm = bm * bm % BB_P
        m *= 2
    return out


def build_dummy_circuit_bb(degree_bits, num_routed_wires=41, num_constants=2):
    """BabyBear form of build_dummy_circuit (recursion_config_bb_narrow): PublicInputGate<8>, k_is = 31^i,
    subgroup generator 0x1a427a41^(2^(27 - degree_bits)) (field constants recalled from upstream Plonky3, unpinned)."""
    assert degree_bits >= 3
    n = 1 << degree_bits
    pi_row = (1 << (degree_bits - 1)) + 1
    const_row = pi_row + 1
    cs = np.zeros((1 + num_constants + num_routed_wires, n), dtype=np.uint32)
    cs[0, pi_row] = GATE_PI
    cs[0, const_row] = GATE_CONSTANT
    k_is = np.array([pow(31, i, BB_P) for i in range(num_routed_wires)], dtype=np.uint32)
    sub = bb_powers(pow(0x1a427a41, 1 << (27 - degree_bits), BB_P), n)
    sig = cs[1 + num_constants:]
    _sigma_columns(sig, sub, k_is, lambda x: bb_mul(x, 31))
    cls = [(pi_row, j) for j in range(8)] + [(const_row, 0)]
    for t, (row, col) in enumerate(cls):
        nrow, ncol = cls[(t + 1) % len(cls)]
        sig[col, row] = int(k_is[ncol]) * int(sub[nrow]) % BB_P
    return cs, k_is, pi_row, const_row


def dummy_witness_bb(degree_bits, pi_row, num_wires=167, seed=0):
    """wires 8.. of the PublicInputGate<8> row from SplitMix64 reduced mod p."""
    w = np.zeros((num_wires, 1 << degree_bits), dtype=np.uint32)
    idx = np.arange(1, num_wires - 8 + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(0x9E3779B97F4A7C15 + seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w[8:, pi_row] = (z % np.uint64(BB_P)).astype(np.uint32)
    return w


def build_dummy_circuit(degree_bits, num_routed_wires=80, num_constants=2):
    """-> (constants_sigmas [1 + num_constants + routed][n] uint64, k_is [routed], pi_row, const_row)"""
    assert degree_bits >= 3
    n = 1 << degree_bits
    pi_row = (1 << (degree_bits - 1)) + 1
    const_row = pi_row + 1
    cs = np.zeros((1 + num_constants + num_routed_wires, n), dtype=np.uint64)
    cs[0, pi_row] = GATE_PI        # selector polynomial: gate index per row (gates/selectors.rs:142-159)
    cs[0, const_row] = GATE_CONSTANT
    k_is = np.array([pow(7, i, P) for i in range(num_routed_wires)], dtype=np.uint64)  # field/src/cosets.rs:8-21
    sub = gl_powers(pow(1753635133440165772, 1 << (32 - degree_bits), P), n)
    sig = cs[1 + num_constants:]
    _sigma_columns(sig, sub, k_is, gl_mul7)   # k_is[j] = 7^j: 12 s of single-threaded gl_mul at 2^20 rows before round 4
    # the one copy class {(pi,0..3), (const,0)} in (row, column) order (permutation_argument.rs:108-157)
    cls = [(pi_row, 0), (pi_row, 1), (pi_row, 2), (pi_row, 3), (const_row, 0)]
    for t, (row, col) in enumerate(cls):
        nrow, ncol = cls[(t + 1) % len(cls)]
        sig[col, row] = int(k_is[ncol]) * int(sub[nrow]) % P
    return cs, k_is, pi_row, const_row


def dummy_witness(degree_bits, pi_row, num_wires=135, seed=0):
    """wire_values[column][row]: zeros except the PublicInputGate row's wires 4.. (RandomValueGenerator
    stand-in: SplitMix64(seed))."""
    w = np.zeros((num_wires, 1 << degree_bits), dtype=np.uint64)
    w[4:, pi_row] = splitmix64(0x9E3779B97F4A7C15 + seed, num_wires - 4)
    return w
