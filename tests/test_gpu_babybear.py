"""GPU parity for BabyBear + Poseidon2-16 (BASELINE configs[3] commitment path): HIP through the C ABI
against the BabyBear CPU oracle, bit for bit.  The oracle itself is self-consistent but UNPINNED against the
reference for this field (no KAT in the reference).  -m gpu only."""
import numpy as np
import pytest

from oracle import oracle_bb as B
from plonky2_goldibear_amd import GB_BABYBEAR, GpuContext, PolynomialBatch, ShapeError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _cols(ncols, log_n, seed=None):
    seed = (0xC0FFEE ^ (ncols << 32) ^ log_n) if seed is None else seed
    return B.fill(seed, ncols << log_n).reshape(ncols, 1 << log_n)


def test_poseidon2_matches_oracle(ctx):
    st = B.fill(5, 16 * 500).reshape(500, 16)
    st[0, :] = 0
    st[1, :] = B.BB_P - 1
    st[2, :] = np.arange(16)
    got = ctx.permute(st, field=GB_BABYBEAR)
    for i in list(range(6)) + list(range(6, 500, 41)):
        assert (got[i] == B.poseidon2(st[i])).all()


def _check(gpu, cpu):
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.polynomials == cpu.polynomials).all()
    assert (gpu.merkle_tree.leaves == cpu.leaves).all()
    if cpu.digests.size:
        assert (gpu.merkle_tree.digests == cpu.digests).all()
    N = cpu.leaves.shape[0]
    for i in {i for i in (0, 1, N // 2, N - 1, (N * 5) // 7) if i < N}:
        row, sib = gpu._leaf(i)
        assert (row == cpu.leaves[i]).all() and (sib == cpu.prove(i)).all()
        assert B.merkle_verify(row, i, cpu.cap, sib)


@pytest.mark.parametrize("log_n,ncols,rate_bits,cap_height", [
    (0, 1, 0, 0), (1, 3, 1, 0), (3, 8, 3, 2), (4, 9, 3, 4), (6, 5, 3, 4), (8, 17, 3, 4), (10, 167, 3, 4),
    (12, 3, 3, 4), (13, 4, 3, 4), (14, 2, 1, 4), (16, 3, 3, 4), (17, 2, 3, 4), (18, 1, 2, 4), (18, 3, 3, 4), (19, 2, 3, 4),
])
def test_from_values_matches_oracle(ctx, log_n, ncols, rate_bits, cap_height):
    vals = _cols(ncols, log_n)
    gpu = PolynomialBatch.from_values(ctx, vals, rate_bits, cap_height, field=GB_BABYBEAR)
    cpu = B.PolynomialBatch.from_values(vals, rate_bits, cap_height)
    _check(gpu, cpu)
    if log_n >= 2:
        assert (gpu.get_lde_values(3, 1 << rate_bits) == cpu.get_lde_values(3, 1 << rate_bits)).all()
    gpu.free()


@pytest.mark.parametrize("ncols", [9, 10, 15, 16, 17, 23, 24, 25, 31])
def test_lane_per_leaf_sponge_widths_on_a_grid_of_many_workgroups(ctx, ncols):
    """The lane-per-leaf sponge kernels at every absorption shape (whole, ragged last) on 2^17 leaves = 512 workgroups, more than
    one per CU (tests/test_gpu_parity.py has the Goldilocks twin and the reason)."""
    vals = _cols(ncols, 14)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4, field=GB_BABYBEAR)
    cpu = B.PolynomialBatch.from_values(vals, 3, 4)
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.merkle_tree.digests == cpu.digests).all()
    gpu.free()


def test_from_coeffs_salts_and_edges(ctx):
    coeffs = _cols(6, 10, seed=9)
    coeffs[0, :] = 0
    coeffs[1, :] = B.BB_P - 1
    salts = B.fill(77, 4 << 13).reshape(4, -1)
    gpu = PolynomialBatch.from_coeffs(ctx, coeffs, 3, 4, salts=salts, field=GB_BABYBEAR)
    cpu = B.PolynomialBatch.from_coeffs(coeffs, 3, 4, salts=salts)
    assert gpu.blinding and gpu.width == 10
    _check(gpu, cpu)
    with pytest.raises(ShapeError):  # two-adicity 27
        PolynomialBatch.from_values(ctx, _cols(1, 20), 8, 4, field=GB_BABYBEAR)


def test_full_size_2pow20(ctx):
    """BASELINE configs[3] shape: n = 2^20 BabyBear columns, rate 3, cap 4: whole columns against the oracle's
    ifft / coset fft plus Merkle-path verification on the GPU cap."""
    vals = _cols(3, 20)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4, field=GB_BABYBEAR)
    c0 = gpu.polynomial(0)
    assert (c0 == B.ifft(vals[0])).all()
    lde0 = B.coset_fft(np.concatenate([c0, np.zeros((1 << 23) - (1 << 20), np.uint32)]), 31, 3)
    cap = gpu.merkle_tree.cap
    rng = np.random.default_rng(3)
    for i in [0, (1 << 23) - 1] + rng.integers(0, 1 << 23, 10).tolist():
        row, sib = gpu._leaf(int(i))
        assert row[0] == lde0[int(format(int(i), "023b")[::-1], 2)]
        assert sib.shape == (19, 8) and B.merkle_verify(row, int(i), cap, sib)
