"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.  -m gpu only."""
import numpy as np
import pytest

from oracle import oracle as O
from plonky2_goldibear_amd import GpuContext, PolynomialBatch, ShapeError

pytestmark = pytest.mark.gpu
P = O.GL_P


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _cols(ncols, log_n, seed=None):
    seed = (0xC0FFEE ^ (ncols << 32) ^ log_n) if seed is None else seed
    return O.splitmix64_fill(seed, ncols << log_n).reshape(ncols, 1 << log_n)


def test_poseidon12_kats_and_random(ctx, kats):
    # hash/poseidon_goldilocks.rs:1158-1193
    ins = np.array([v["input"] for v in kats["poseidon12"]], dtype=np.uint64)
    outs = np.array([v["output"] for v in kats["poseidon12"]], dtype=np.uint64)
    assert (ctx.permute(ins) == outs).all()
    rnd = O.splitmix64_fill(5, 12 * 1000).reshape(1000, 12)
    rnd[0, :] = P - 1
    got = ctx.permute(rnd)
    for i in range(0, 1000, 37):
        assert (got[i] == O.poseidon(rnd[i])).all()


def test_poseidon12_matrix_pipe_edges(ctx):
    """The permutation's MDS layers run as i8 MFMAs on byte planes of the state (csrc/poseidon_gl.hpp, mds_layer_mfma): states
    made of the bytes where a signed-byte interpretation bites (0x00 / 0x7f / 0x80 / 0xff in every position), counts that leave
    lanes of the last wave without data, and enough random states (2^17 x 30 layers) for the fold's rare carry path (a lane whose
    high accumulator's low word lies within 2^12 of 2^32: ~4e-4 of the wave-layers) to be taken hundreds of times - every word
    against the oracle's scalar permutation."""
    pat = [0, 0x7f7f7f7f7f7f7f7f, 0x8080808080808080, 0xFFFFFFFF00000000, 0x00FF00FF00FF00FF, 0x80007fff0100ff80, P - 1, 1]
    edge = np.array([[pat[(i + j) % len(pat)] for j in range(12)] for i in range(67)], dtype=np.uint64)   # 67: a ragged last wave
    got = ctx.permute(edge)
    for i in range(len(edge)):
        assert (got[i] == O.poseidon(edge[i])).all(), i
    for count in (1, 63, 65):
        st = O.splitmix64_fill(90 + count, 12 * count).reshape(count, 12)
        got = ctx.permute(st)
        for i in range(count):
            assert (got[i] == O.poseidon(st[i])).all()
    big = O.splitmix64_fill(77, 12 << 17).reshape(1 << 17, 12)
    got = ctx.permute(big)
    want = np.empty_like(big)
    for i in range(big.shape[0]):
        want[i] = O.poseidon(big[i])
    assert (got == want).all()


def _check_batch(gpu, cpu, full=True):
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.polynomials == cpu.polynomials).all()
    if full:
        assert (gpu.merkle_tree.leaves == cpu.leaves).all()
        if cpu.digests.size:
            assert (gpu.merkle_tree.digests == cpu.digests).all()
    N = cpu.leaves.shape[0]
    for i in {i for i in (0, 1, N // 2, N - 1, (N * 5) // 7) if i < N}:
        row, sib = gpu._leaf(i)
        assert (row == cpu.leaves[i]).all()
        assert (sib == cpu.prove(i)).all()
        assert O.merkle_verify(row, i, cpu.cap, sib)


@pytest.mark.parametrize("log_n,ncols,rate_bits,cap_height", [
    (0, 1, 0, 0), (0, 3, 3, 2), (1, 2, 1, 0), (2, 5, 3, 4), (4, 1, 3, 4), (5, 4, 3, 0), (6, 9, 3, 4),
    (8, 17, 3, 4), (10, 8, 3, 4), (10, 135, 3, 4), (11, 3, 2, 13), (12, 5, 3, 4), (12, 2, 0, 1),
    (13, 3, 3, 4), (14, 2, 1, 4), (15, 2, 3, 4), (16, 3, 3, 4), (17, 2, 3, 4), (18, 1, 2, 4), (18, 3, 3, 4), (19, 2, 3, 4),
])
def test_from_values_matches_oracle(ctx, log_n, ncols, rate_bits, cap_height):
    vals = _cols(ncols, log_n)
    gpu = PolynomialBatch.from_values(ctx, vals, rate_bits, cap_height)
    cpu = O.PolynomialBatch.from_values(vals, rate_bits, cap_height)
    _check_batch(gpu, cpu)
    assert (gpu.get_lde_values(0, 1) == cpu.get_lde_values(0, 1)).all()
    if log_n >= 2:
        step = 1 << rate_bits  # the prover's next_step (plonk/prover.rs:819-831)
        assert (gpu.get_lde_values(3, step) == cpu.get_lde_values(3, step)).all()
    gpu.free()


@pytest.mark.parametrize("ncols", [5, 6, 7, 8, 9, 12, 15, 16, 17, 23, 24, 25])
def test_lane_per_leaf_sponge_widths_on_a_grid_of_many_workgroups(ctx, ncols):
    """The lane-per-leaf sponge kernels (trees above 2^14 leaves) at every absorption shape - ragged first, whole, ragged last -
    on 2^17 leaves = 512 workgroups, two per CU: a kernel variant of round 3 was right in the first workgroup of every CU and
    wrong in the second (an undefined MFMA operand), and only three shapes of this file happened to see it."""
    vals = _cols(ncols, 14)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4)
    cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.merkle_tree.digests == cpu.digests).all()
    gpu.free()


@pytest.mark.parametrize("log_n,ncols", [(3, 2), (10, 30), (13, 4), (16, 2)])
def test_from_coeffs_and_salts(ctx, log_n, ncols):
    coeffs = _cols(ncols, log_n, seed=77 + log_n)
    salts = O.splitmix64_fill(1234, 4 << (log_n + 3)).reshape(4, -1)
    gpu = PolynomialBatch.from_coeffs(ctx, coeffs, 3, 4 if log_n > 3 else 2, salts=salts)
    cpu = O.PolynomialBatch.from_coeffs(coeffs, 3, 4 if log_n > 3 else 2, salts=salts)
    assert gpu.blinding and gpu.width == ncols + 4
    _check_batch(gpu, cpu)
    assert gpu.get_lde_values(1, 1).size == ncols  # salt columns dropped (oracle.rs:157)
    gpu.free()


def test_device_resident_input(ctx):
    import torch
    vals = _cols(6, 14)
    t = torch.from_numpy(vals.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    gpu = PolynomialBatch.from_values(ctx, t, 3, 4)
    cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    _check_batch(gpu, cpu, full=False)
    assert (t.cpu().numpy().view(np.uint64) == vals).all()  # input not clobbered
    gpu.free()


def test_error_behaviour(ctx):
    with pytest.raises(ShapeError):  # merkle_tree.rs:154-157 / :257-272 should_panic
        PolynomialBatch.from_values(ctx, _cols(2, 5), 3, 9)
    with pytest.raises(ShapeError):
        PolynomialBatch.from_values(ctx, np.zeros((2, 12), np.uint64), 3, 1)  # not a power of two
    b = PolynomialBatch.from_values(ctx, _cols(2, 5), 3, 8)  # cap_height == log2(leaves): allowed (:274-288)
    assert b.merkle_tree.digests.shape[0] == 0
    cpu = O.PolynomialBatch.from_values(_cols(2, 5), 3, 8)
    assert (b.merkle_tree.cap == cpu.cap).all()
    with pytest.raises(ShapeError):
        b._leaf(1 << 8)
    with pytest.raises(ShapeError):
        b.polynomial(2)


def test_edge_values(ctx):
    # all-zero, all p-1 and a single spike: exercises carries / canonical reduction paths
    n = 1 << 13
    vals = np.zeros((4, n), dtype=np.uint64)
    vals[1, :] = P - 1
    vals[2, 7] = P - 1
    vals[3, :] = np.arange(n, dtype=np.uint64) * np.uint64(0xFFFFFFFF) % np.uint64(P)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4)
    cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    _check_batch(gpu, cpu)


def test_full_size_2pow20_properties(ctx):
    """BASELINE size (n = 2^20, N = 2^23): two columns checked against the oracle's own
    ifft / coset fft, plus size-independent properties (Merkle paths verify against the cap,
    from_coeffs(from_values(v).polynomials) reproduces the cap)."""
    log_n, ncols = 20, 3
    vals = _cols(ncols, log_n)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4)
    c0 = gpu.polynomial(0)
    assert (c0 == O.ifft(vals[0])).all()
    assert (O.fft(gpu.polynomial(2)) == vals[2]).all()
    lde0 = O.coset_fft(np.concatenate([c0, np.zeros((1 << 23) - (1 << 20), np.uint64)]), 7, 3)
    cap = gpu.merkle_tree.cap
    rng = np.random.default_rng(7)
    for i in [0, 1, (1 << 23) - 1] + rng.integers(0, 1 << 23, 12).tolist():
        row, sib = gpu._leaf(int(i))
        src = int(format(int(i), "023b")[::-1], 2)
        assert row[0] == lde0[src]
        assert sib.shape == (19, 4)
        assert O.merkle_verify(row, int(i), cap, sib)
    again = PolynomialBatch.from_coeffs(ctx, gpu.polynomials, 3, 4)
    assert (again.merkle_tree.cap == cap).all()


def test_circuit_digest_kat_gpu(ctx, kats):
    """The reference's own golden value (recursion/recursive_verifier.rs:427-436): the constants||sigmas
    commitment of the 16 000-NoopGate test-form circuit, computed on the GPU, hashes to the pinned digest."""
    from oracle import plonk_dummy as D
    want = kats["circuit_digest_gl"][0]
    cs, degree_bits = D.test_form_constants_sigmas(16000)
    gpu = PolynomialBatch.from_values(ctx, cs, 3, 4)
    assert D.circuit_digest_from_cap(gpu.merkle_tree.cap, degree_bits).tolist() == want["digest"]


@pytest.mark.parametrize("field_name,log_n,ncols", [("goldilocks", 0, 2), ("goldilocks", 5, 3), ("goldilocks", 13, 7),
                                                     ("babybear", 4, 3), ("babybear", 13, 5)])
def test_eval_ext_matches_horner(ctx, field_name, log_n, ncols):
    """gb_batch_eval_ext = p.to_extension().eval(z) per polynomial (plonk/proof.rs:359-363), checked against a
    Horner evaluation in the oracle's extension-field restatement (oracle/fields.py)."""
    from oracle.fields import BB, GL
    from plonky2_goldibear_amd import native as N
    F, tag = (GL, N.GB_GOLDILOCKS) if field_name == "goldilocks" else (BB, N.GB_BABYBEAR)
    n = 1 << log_n
    coeffs = F.fill(0xABCDEF + log_n, ncols * n).reshape(ncols, n)
    b = PolynomialBatch.from_coeffs(ctx, coeffs, 1, 0, field=tag)
    z = tuple(int(x) for x in F.fill(77 + log_n, F.D))
    got = b.eval_ext(np.array(z, dtype=F.dtype))
    for c in range(ncols):
        acc = F.zero
        for t in range(n - 1, -1, -1):
            acc = F.eadd(F.emul(acc, z), F.efrom(int(coeffs[c, t])))
        assert tuple(int(x) for x in got[c]) == acc
    with pytest.raises(ShapeError):
        b.eval_ext(np.zeros(F.D + 1, dtype=F.dtype))
    b.free()


def test_pow_grind_minimum_nonce(ctx):
    """gb_pow_grind = fri_proof_of_work (fri/prover.rs:136-188) with the minimum-nonce rule, both fields, checked by
    brute force over the oracle's permutations."""
    from oracle import oracle_bb as B
    from plonky2_goldibear_amd import native as N
    st = O.splitmix64_fill(4242, 12)
    bits = 10
    nonce = ctx.pow_grind(st, 3, bits)
    for cand in range(nonce + 1):
        s2 = st.copy()
        s2[3] = cand
        ok = int(O.poseidon(s2)[7]) >> (64 - bits) == 0
        assert ok == (cand == nonce)
    stb = B.fill(99, 16)
    min_lz = 8 + 33  # proof_of_work_bits + (64 - 31)
    nonce = ctx.pow_grind(stb, 5, min_lz, field=N.GB_BABYBEAR)
    for cand in range(nonce + 1):
        s2 = stb.copy()
        s2[5] = cand
        ok = int(B.poseidon2(s2)[7]) < (1 << (64 - min_lz))
        assert ok == (cand == nonce)
    with pytest.raises(ShapeError):
        ctx.pow_grind(st, 8, bits)
