"""Seeded random shapes through PolynomialBatch::from_values / from_coeffs (fri/oracle.rs:68-150): field, rows 2^0 .. 2^14, 1 .. 60
polynomials, rate_bits 0 .. 8, every cap_height the tree allows, salted or not, and each way of handing the matrix over (one host
block, separately allocated host columns, the field types' in-memory words, a device tensor).  Every coefficient, every leaf,
every digest and the cap against the CPU oracle; sampled rows and Merkle paths through the accessors.  -m gpu."""
import numpy as np
import pytest

from oracle.fields import BB, GL
from plonky2_goldibear_amd import GpuContext, PolynomialBatch, native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _p3_words(F, a, rng):
    """the in-memory words p3's field types would hold for the canonical values `a`"""
    if F is GL:   # p3-goldilocks: any u64 representative; add p where it fits
        a = a.copy()
        fits = a < (np.uint64(0xFFFFFFFFFFFFFFFF) - np.uint64(F.P) + np.uint64(1))
        pick = fits & (rng.random(a.shape) < 0.5)
        a[pick] += np.uint64(F.P)
        return a
    return ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(F.P)).astype(np.uint32)   # p3-baby-bear: x 2^32 mod p


@pytest.mark.parametrize("seed", range(60))
def test_random_commit_shape(ctx, seed):
    rng = np.random.default_rng(10_000 + seed)
    F, tag = (GL, N.GB_GOLDILOCKS) if seed % 2 == 0 else (BB, N.GB_BABYBEAR)
    log_n = int(rng.choice([0, 1, 2, 3, 5, 7, 9, 11, 12, 13, 14]))
    rate = int(rng.integers(0, 9))
    while log_n + rate > 17:
        rate -= 1
    ncols = int(rng.integers(1, 61 if log_n <= 12 else 25))
    cap = int(rng.integers(0, log_n + rate + 1))
    is_coeffs = bool(rng.random() < 0.3)
    salted = bool(rng.random() < 0.25)
    how = str(rng.choice(["block", "cols", "p3_cols", "p3_block", "device"]))
    vals = F.fill(777 + seed, ncols << log_n).reshape(ncols, -1)
    edge = rng.random(vals.shape) < 0.02          # sprinkle 0 and p - 1
    vals[edge] = np.where(rng.random(int(edge.sum())) < 0.5, 0, F.P - 1).astype(F.dtype)
    salts = F.fill(99 + seed, 4 << (log_n + rate)).reshape(4, -1) if salted else None
    what = "seed %d: %s 2^%d x %d rate %d cap %d %s salted=%r %s" % (seed, F.name, log_n, ncols, rate, cap,
                                                                        "coeffs" if is_coeffs else "values", salted, how)
    make = PolynomialBatch.from_coeffs if is_coeffs else PolynomialBatch.from_values
    cpu = (F.mod.PolynomialBatch.from_coeffs if is_coeffs else F.mod.PolynomialBatch.from_values)(vals, rate, cap, salts=salts)
    kw = dict(field=tag)
    given, gsalts = vals, salts
    if how.startswith("p3"):
        given, kw["p3_repr"] = _p3_words(F, vals, rng), True
        gsalts = _p3_words(F, salts, rng) if salted else None
    if how.endswith("cols"):
        given = [np.array(given[c], copy=True) for c in range(ncols)]   # separately allocated, pageable
    if how == "device":
        import torch
        view = np.int64 if F is GL else np.int32
        given = torch.from_numpy(np.ascontiguousarray(vals).view(view)).to("cuda:0")
        gsalts = torch.from_numpy(np.ascontiguousarray(salts).view(view)).to("cuda:0") if salted else None
    gpu = make(ctx, given, rate, cap, salts=gsalts, **kw)
    assert (gpu.merkle_tree.cap == cpu.cap).all(), what
    assert (gpu.polynomials == cpu.polynomials).all(), what
    assert (gpu.merkle_tree.leaves == cpu.leaves).all(), what
    if cpu.digests.size:
        assert (gpu.merkle_tree.digests == cpu.digests).all(), what
    Nl = cpu.leaves.shape[0]
    for i in {0, Nl - 1, int(rng.integers(0, Nl))}:
        row, sib = gpu._leaf(i)
        assert (row == cpu.leaves[i]).all() and (sib == cpu.prove(i)).all(), what
    gpu.free()
