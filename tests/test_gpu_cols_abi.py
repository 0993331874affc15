"""The column-pointer entry points (gb_commit_values_cols, gb_prove_cols, ...): the reference's own memory layout - ncols
separately allocated, PAGEABLE columns (MatrixWitness.wire_values: Vec<Vec<F>>, iop/witness.rs:277-279; Vec<PolynomialValues<F>>,
fri/oracle.rs:68-75) - staged by the library's page-locked ring, and GB_INPUT_P3_REPR (the field types' in-memory words).
Everything against the CPU oracle, through the C ABI.  -m gpu only.  (2^20 rows: tests/test_gpu_parity_at_size.py.)"""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_bb as B
from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import GB_BABYBEAR, CircuitData, GpuContext, PolynomialBatch
from plonky2_goldibear_amd import native as N
from plonky2_goldibear_amd.native import PermArgZeroError, ShapeError

pytestmark = pytest.mark.gpu
BB_P = 2013265921


@pytest.fixture(scope="module")
def ctx():
    O.use_host_cpu_share()
    c = GpuContext(0)
    yield c
    c.close()


def columns_of(matrix):
    """Vec<Vec<F>>: every column its own (malloc'ed, pageable) allocation"""
    return [np.array(col, copy=True) for col in matrix]


def to_p3_words(a, field):
    """the reference's field types as they lie in memory: p3-goldilocks keeps ANY u64 representative (here x + p wherever that
    fits in 64 bits), p3-baby-bear the Montgomery word x * 2^32 mod p"""
    if field == GB_BABYBEAR:
        return ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(BB_P)).astype(np.uint32)
    out = a.copy()
    small = a < np.uint64(0xFFFFFFFF)          # x + p < 2^64  <=>  x < 2^32 - 1
    out[small] = a[small] + np.uint64(GL.P)
    return out


def _gpu_circuit(ctx, circ, tag, cs=None, k_is=None, p3_repr=False):
    cfg = circ.cfg
    return CircuitData(ctx, circ.degree_bits, circ.constants_sigmas if cs is None else cs, circ.k_is if k_is is None else k_is,
                       num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants,
                       num_challenges=cfg.num_challenges, arity_bits=cfg.arity_bits, gate_constant=circ.GATE_CONSTANT,
                       gate_pi=circ.GATE_PI, field=tag, p3_repr=p3_repr)


@pytest.mark.parametrize("field_name,log_n,ncols", [
    ("goldilocks", 6, 5), ("goldilocks", 10, 9), ("goldilocks", 13, 40), ("goldilocks", 16, 135), ("goldilocks", 18, 21),
    ("babybear", 6, 5), ("babybear", 11, 7), ("babybear", 13, 44), ("babybear", 17, 167), ("babybear", 18, 37),
])
def test_from_values_cols_equals_oracle(ctx, field_name, log_n, ncols):
    """small batches (plain copies), 2^12 rows and up (chunked upload; from ~2 MiB per chunk through the staging ring), more than
    32 columns at >= 2^19 leaves (leaf sponges in segments): cap, every coefficient, sampled rows and paths"""
    seed = 0x5EED ^ (ncols << 8) ^ log_n
    if field_name == "goldilocks":
        vals, tag = O.splitmix64_fill(seed, ncols << log_n).reshape(ncols, 1 << log_n), N.GB_GOLDILOCKS
        cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    else:
        vals, tag = B.fill(seed, ncols << log_n).reshape(ncols, 1 << log_n), GB_BABYBEAR
        cpu = B.PolynomialBatch.from_values(vals, 3, 4)
    gpu = PolynomialBatch.from_values(ctx, columns_of(vals), 3, 4, field=tag)
    assert gpu.num_polys == ncols and gpu.degree_log == log_n
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.polynomials == cpu.polynomials).all()
    for i in (0, 1, (8 << log_n) - 1, (5 << log_n) // 3):
        row, sib = gpu._leaf(i)
        assert (row == cpu.leaves[i]).all() and (sib == cpu.prove(i)).all()
    gpu.free()
    # the same columns as the reference's field types hold them in memory
    p3 = PolynomialBatch.from_values(ctx, columns_of(to_p3_words(vals, tag)), 3, 4, field=tag, p3_repr=True)
    assert (p3.merkle_tree.cap == cpu.cap).all()
    assert (p3.polynomial(ncols - 1) == cpu.polynomials[ncols - 1]).all()
    p3.free()
    # ... as one block, and as coefficients
    p3b = PolynomialBatch.from_values(ctx, to_p3_words(vals, tag), 3, 4, field=tag, p3_repr=True)
    assert (p3b.merkle_tree.cap == cpu.cap).all()
    p3b.free()
    co = PolynomialBatch.from_coeffs(ctx, columns_of(to_p3_words(cpu.polynomials, tag)), 3, 4, field=tag, p3_repr=True)
    assert (co.merkle_tree.cap == cpu.cap).all()
    co.free()
    ctx.trim()


@pytest.mark.parametrize("threads", [-1, 0, 1, 7])
def test_staging_ring_thread_counts(ctx, threads):
    """"copy_threads": no ring (the runtime's pageable path), the calling thread, one and seven copy threads - same batch"""
    log_n, ncols = 16, 70
    vals = O.splitmix64_fill(0x7EAD + 5, ncols << log_n).reshape(ncols, 1 << log_n)
    cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    ctx.set_option("copy_threads", threads)
    try:
        for _ in range(3):   # the ring's slots come round
            gpu = PolynomialBatch.from_values(ctx, columns_of(vals), 3, 4)
            assert (gpu.merkle_tree.cap == cpu.cap).all()
            assert (gpu.merkle_tree.digests == cpu.digests).all()
            gpu.free()
    finally:
        ctx.set_option("copy_threads", 4)
    ctx.trim()


def test_page_locked_columns_and_device_columns(ctx):
    """columns a host placed in page-locked memory (gb_host_alloc, gb_host_register) go to the copy engine directly; separately
    allocated DEVICE columns are gathered"""
    import torch
    log_n, ncols = 16, 40
    vals = O.splitmix64_fill(0xA110C, ncols << log_n).reshape(ncols, 1 << log_n)
    cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    pinned = [ctx.host_alloc((1 << log_n,), np.uint64) for _ in range(ncols)]
    for dst, src in zip(pinned, vals):
        dst[:] = src
    a = PolynomialBatch.from_values(ctx, pinned, 3, 4)
    assert (a.merkle_tree.cap == cpu.cap).all()
    a.free()
    for p in pinned:
        ctx.host_free(p)
    regd = columns_of(vals)
    for c in regd:
        ctx.host_register(c)
    b = PolynomialBatch.from_values(ctx, regd, 3, 4)
    assert (b.merkle_tree.cap == cpu.cap).all()
    b.free()
    for c in regd:
        ctx.host_unregister(c)
    dev = [torch.from_numpy(c.view(np.int64)).cuda() for c in vals]
    d = PolynomialBatch.from_values(ctx, dev, 3, 4)
    assert (d.merkle_tree.cap == cpu.cap).all()
    assert (d.polynomials == cpu.polynomials).all()
    d.free()
    ctx.trim()


def test_cols_errors(ctx):
    vals = O.splitmix64_fill(1, 4 << 6).reshape(4, 1 << 6)
    cols = columns_of(vals)
    lib = ctx._lib
    import ctypes as C
    ptrs = (C.c_void_p * 4)(*[c.ctypes.data for c in cols])
    h = C.c_void_p()
    ptrs[2] = None
    assert lib.gb_commit_values_cols(ctx.handle, 0, ptrs, 4, 6, 3, 4, None, 0, C.byref(h)) == N.GB_ERR_INVALID   # null column
    ptrs[2] = cols[2].ctypes.data
    assert lib.gb_commit_values_cols(ctx.handle, 0, ptrs, 4, 6, 3, 4, None, 0x200, C.byref(h)) == N.GB_ERR_INVALID  # unknown flag bit
    assert lib.gb_commit_values_cols(ctx.handle, 0, ptrs, 4, 6, 3, 4, None, N.GB_INPUT_DEVICE | N.GB_INPUT_P3_REPR, C.byref(h)) == N.GB_ERR_INVALID
    assert lib.gb_commit_values(ctx.handle, 0, vals.ctypes.data, 4, 6, 3, 4, None, 0x100, C.byref(h)) == N.GB_ERR_INVALID   # the internal bit
    # page-locked memory: a zero-sized allocation and the unregistration of a range that was never registered are errors, not crashes
    assert lib.gb_host_alloc(ctx.handle, 0, C.byref(h)) == N.GB_ERR_INVALID
    assert lib.gb_host_unregister(ctx.handle, cols[0].ctypes.data) == N.GB_ERR_INVALID
    assert lib.gb_host_free(ctx.handle, None) == N.GB_OK
    with pytest.raises(ShapeError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(ShapeError):
        ctx.set_option("copy_threads", 1000)


@pytest.mark.parametrize("field_name,degree_bits,num_challenges", [
    ("goldilocks", 5, 2), ("goldilocks", 13, 2), ("goldilocks", 16, 3), ("babybear", 5, 6), ("babybear", 13, 6), ("babybear", 16, 7)])
def test_prove_cols_bytes_match_oracle(ctx, field_name, degree_bits, num_challenges):
    """gb_prove_cols from num_wires separately allocated pageable columns: proof bytes == the oracle prover's == gb_prove's from
    the flat matrix; the same from the p3 in-memory words; gb_zs_partial_products_cols == the flat entry point"""
    if field_name == "goldilocks":
        F, tag, cfg = GL, N.GB_GOLDILOCKS, D.CircuitConfig(num_challenges=num_challenges)
    else:
        F, tag, cfg = BB, N.GB_BABYBEAR, D.CircuitConfig.babybear(num_challenges)
    circ = D.DummyCircuit(degree_bits, cfg, check_security=False, F=F)
    gpu = _gpu_circuit(ctx, circ, tag)
    circ.set_cap(gpu.constants_sigmas_cap)   # prove_cpu asserts that this IS the cap of the oracle's own commitment
    w = circ.witness(seed=degree_bits + 100)
    want, _ = D.prove_cpu(circ, w)
    assert gpu.prove_once(columns_of(w)) == want
    assert gpu.prove_once(w) == want
    assert gpu.prove_once(columns_of(to_p3_words(w, tag)), p3_repr=True) == want
    assert gpu.prove_once(to_p3_words(w, tag), p3_repr=True) == want
    rng = np.random.default_rng(degree_bits)
    while True:   # challenges with no zero denominator (a 31-bit field meets one now and then: InvZeroPermArg, prover.rs:512-514)
        betas = [int(x) for x in rng.integers(1, F.P, num_challenges, dtype=np.uint64)]
        gammas = [int(x) for x in rng.integers(1, F.P, num_challenges, dtype=np.uint64)]
        try:
            flat = gpu.zs_partial_products(w, betas, gammas)
            break
        except PermArgZeroError:
            continue
    assert (gpu.zs_partial_products(columns_of(w), betas, gammas) == flat).all()
    assert (gpu.zs_partial_products(columns_of(to_p3_words(w, tag)), betas, gammas, p3_repr=True) == flat).all()
    # build(): constants_sigmas_vecs as separately allocated columns (gb_circuit_create_cols) - the same circuit
    gpu2 = _gpu_circuit(ctx, circ, tag, cs=columns_of(circ.constants_sigmas))
    assert (gpu2.constants_sigmas_cap == gpu.constants_sigmas_cap).all() and (gpu2.circuit_digest == gpu.circuit_digest).all()
    assert gpu2.prove_once(columns_of(w)) == want
    gpu2.free()
    # ... and as the field types' in-memory words (circuit data and witness alike: nothing canonicalised on the host)
    gpu3 = _gpu_circuit(ctx, circ, tag, cs=columns_of(to_p3_words(circ.constants_sigmas, tag)), k_is=to_p3_words(circ.k_is, tag), p3_repr=True)
    assert (gpu3.constants_sigmas_cap == gpu.constants_sigmas_cap).all() and (gpu3.circuit_digest == gpu.circuit_digest).all()
    assert gpu3.prove_once(columns_of(to_p3_words(w, tag)), p3_repr=True) == want
    gpu3.free()
    gpu.free()
    ctx.trim()


def test_retry_cols_is_incremental_and_byte_identical(ctx):
    """gb_prove_retry_cols: the failed attempt (armed through the test hook) keeps its wires commitment; the retry from the same
    columns with the random wire re-drawn gives the bytes of a proof from scratch and of the oracle - canonical and p3 words"""
    circ = D.DummyCircuit(16, D.CircuitConfig(num_challenges=3))
    gpu = _gpu_circuit(ctx, circ, N.GB_GOLDILOCKS)
    circ.set_cap(gpu.constants_sigmas_cap)
    w0 = circ.witness(seed=11)
    rw = (circ.cfg.num_wires - 1, circ.pi_row)
    w = w0.copy()
    w[rw] = np.uint64(0x0123456789ABCDEF % D.P)
    want = D.prove_cpu(circ, w)[0]
    for p3 in (False, True):
        conv = (lambda a: to_p3_words(a, N.GB_GOLDILOCKS)) if p3 else (lambda a: a)
        gpu.arm_perm_arg_failure()
        with pytest.raises(PermArgZeroError):
            gpu.prove_once(columns_of(conv(w0)), p3_repr=p3)
        assert gpu.prove_once(columns_of(conv(w)), retry_wire=rw, p3_repr=p3) == want
    # "retry_verify": the library compares the caller's columns with the copy the failed attempt kept, column by column
    ctx.set_option("retry_verify", 1)
    try:
        gpu.arm_perm_arg_failure()
        with pytest.raises(PermArgZeroError):
            gpu.prove_once(columns_of(w0))
        assert gpu.prove_once(columns_of(w), retry_wire=rw) == want
        other = columns_of(w)
        other[7][123] = np.uint64(5)
        gpu.arm_perm_arg_failure()
        with pytest.raises(PermArgZeroError):
            gpu.prove_once(columns_of(w0))
        with pytest.raises(ShapeError):
            gpu.prove_once(other, retry_wire=rw)        # differs in more than witness[wire][row]
    finally:
        ctx.set_option("retry_verify", 0)
    # the retry loop of the host mirror over a column list (prover.rs:183-226)
    cols = columns_of(w0)
    gpu.arm_perm_arg_failure()
    proof = gpu.prove(cols, random_wire=rw, rng=np.random.default_rng(5))
    assert gpu.perm_arg_retries == 1 and gpu.verify(proof)
    w1 = np.stack(cols)
    assert proof == D.prove_cpu(circ, w1)[0]
    gpu.free()
    # BabyBear, p3 words: the Montgomery word written into the kept device copy
    bcirc = D.DummyCircuit(16, D.CircuitConfig.babybear(7), F=BB)
    bgpu = _gpu_circuit(ctx, bcirc, N.GB_BABYBEAR)
    bcirc.set_cap(bgpu.constants_sigmas_cap)
    b0 = bcirc.witness(seed=3)
    brw = (bcirc.cfg.num_wires - 1, bcirc.pi_row)
    b1 = b0.copy()
    b1[brw] = np.uint32(0x12345678 % BB_P)
    bwant = D.prove_cpu(bcirc, b1)[0]
    for p3 in (False, True):
        conv = (lambda a: to_p3_words(a, GB_BABYBEAR)) if p3 else (lambda a: a)
        bgpu.arm_perm_arg_failure()
        with pytest.raises(PermArgZeroError):
            bgpu.prove_once(columns_of(conv(b0)), p3_repr=p3)
        assert bgpu.prove_once(columns_of(conv(b1)), retry_wire=brw, p3_repr=p3) == bwant
    bgpu.free()
    ctx.trim()
