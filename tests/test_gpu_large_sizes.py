"""More than 2^20 rows: the reference's transforms have no size cap below the field's two-adicity (field/src/fft.rs:168-205) and
prove() none either (plonk/prover.rs:228-447).  2^21 and 2^22 rows run the library's own passes (round 6: a radix-32 / radix-64 middle
pass of the inverse transform, a 512- / 1024-row strided pass of the LDE); from 2^23 rows one outer radix step per four bits around
them (csrc/ntt_outer.hpp).  -m gpu only.
* PolynomialBatch::from_values at 2^21 rows: EVERY coefficient, EVERY leaf, EVERY digest and the cap against the CPU oracle's batch
  (Goldilocks 5 columns = one sponge permutation per leaf; BabyBear 9 columns), host and device input; 2^22 rows: coefficients + cap.
* prove() of the 2^21-row dummy circuit, both fields: proof BYTES == the CPU oracle prover's, and the oracle prover's own Z /
  partial-product values and quotient chunk coefficients == the stage entry points' outputs.
* prove() at 2^22 and 2^23 rows: accepted by gb_verify and by the independently written oracle verifier (oracle/verifier.py).
* from_values at 2^23 ... 2^26 rows (Goldilocks; BabyBear's two-adicity ends at 2^24 rows with rate_bits 3): size-independent properties."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_bb as B
from oracle import plonk_dummy as D
from oracle.fields import GL
from plonky2_goldibear_amd import GB_BABYBEAR, CircuitData, GpuContext, PolynomialBatch
from plonky2_goldibear_amd import dummy_circuit as DC
from plonky2_goldibear_amd import native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    O.use_host_cpu_share()
    c = GpuContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("field_name,log_n,ncols", [("goldilocks", 21, 5), ("babybear", 21, 9), ("goldilocks", 22, 2), ("babybear", 22, 3)])
def test_from_values_above_2pow20_rows(ctx, field_name, log_n, ncols):
    import torch
    seed = 0xB16 ^ (ncols << 8) ^ log_n
    if field_name == "goldilocks":
        vals, tag, idt = O.splitmix64_fill(seed, ncols << log_n).reshape(ncols, 1 << log_n), N.GB_GOLDILOCKS, np.int64
        cpu = O.PolynomialBatch.from_values(vals, 3, 4)
    else:
        vals, tag, idt = B.fill(seed, ncols << log_n).reshape(ncols, 1 << log_n), GB_BABYBEAR, np.int32
        cpu = B.PolynomialBatch.from_values(vals, 3, 4)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4, field=tag)
    assert (gpu.merkle_tree.cap == cpu.cap).all()
    assert (gpu.polynomials == cpu.polynomials).all()
    if log_n == 21:
        leaves = gpu.merkle_tree.leaves
        bad = np.flatnonzero((leaves != cpu.leaves).any(axis=1))
        assert bad.size == 0, "%d leaves differ, first %d" % (bad.size, bad[0])
        del leaves
        assert (gpu.merkle_tree.digests == cpu.digests).all()
    for i in (0, 1, (1 << log_n) - 1, 1234567):
        assert (gpu.get_lde_values(i, 8) == cpu.get_lde_values(i, 8)).all()
    gpu.free()
    dev = PolynomialBatch.from_values(ctx, torch.from_numpy(vals.view(idt)).cuda(), 3, 4, field=tag)
    assert (dev.merkle_tree.cap == cpu.cap).all()
    dev.free()
    co = PolynomialBatch.from_coeffs(ctx, cpu.polynomials, 3, 4, field=tag)
    assert (co.merkle_tree.cap == cpu.cap).all()
    co.free()
    ctx.trim()


def test_prove_2pow21_rows_bytes_equal_oracle(ctx):
    """prove() of the 2^21-row Goldilocks dummy circuit: the proof BYTES equal the CPU oracle prover's (round 5 only verified
    proofs of this size: 28 queried leaves per tree say nothing about the other 2^24), and - through the stage entry points, driven
    like the reference's prover loop (tests/test_gpu_stage_abi.py) - the oracle prover's own intermediates equal the GPU's element for
    element: the Z / partial-product values (a running product over 2048 blocks) and the quotient chunk coefficients.  (This
    comparison found the overflow of the suffix-total buffer of divide_by_linear - 1024 entries, 2048 blocks at this size.)"""
    from test_gpu_stage_abi import _gpu_circuit, prove_by_stages
    lg, ch = 21, 3
    F, tag, cfg = GL, N.GB_GOLDILOCKS, D.CircuitConfig(num_challenges=ch)
    circ = D.DummyCircuit(lg, cfg, F=F)
    gpu = _gpu_circuit(ctx, circ, tag)
    circ.set_cap(gpu.constants_sigmas_cap)     # prove_cpu() asserts that it IS the cap of the oracle's own commitment
    assert (gpu.circuit_digest == circ.circuit_digest).all()
    w = circ.witness(seed=lg)
    got = gpu.prove(w, random_wire=(cfg.num_wires - 1, circ.pi_row), rng=np.random.default_rng(lg))   # (re-draws in place: InvZeroPermArg)
    dump, mid = {}, {}
    want, _ = D.prove_cpu(circ, w, dump=dump)
    assert len(got) == len(want)
    assert got == want, "first differing byte at %d" % next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
    assert (gpu.constants_sigmas_cap == D.prove_cpu.last_cs_cap).all()
    assert gpu.verify(got)
    assert prove_by_stages(gpu, circ, w, [], tag, mid) == want
    assert (mid["zs_partial_products"] == dump["zs_partial_products"]).all()
    assert (mid["quotient_chunks"] == dump["quotient_chunks"]).all()
    gpu.free()
    ctx.trim()


def test_babybear_prove_2pow21_rows_bytes_equal_golden(ctx):
    """The same for BabyBear (num_challenges 10) against the oracle prover's proof as a committed golden vector
    (tests/golden/bench_proof_sha256.json "babybear_2p21": tests/golden/make_bench_proof_golden.py 21 babybear - the oracle prover,
    2.3 minutes of the GPU box's host cores that the test step does not have twice), plus the stage entry points assembling the same
    bytes."""
    import hashlib
    import json
    import os
    from oracle.fields import BB
    from test_gpu_stage_abi import prove_by_stages
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_proof_sha256.json")))["babybear_2p21"]
    lg, ch = g["log_n"], g["num_challenges"]
    cs, k_is, pi_row, _ = DC.build_dummy_circuit_bb(lg)
    gpu = CircuitData(ctx, lg, cs, k_is, num_wires=167, num_routed_wires=41, num_challenges=ch, arity_bits=3, field=GB_BABYBEAR)
    assert [int(x) for x in gpu.circuit_digest] == g["circuit_digest"]
    assert hashlib.sha256(np.ascontiguousarray(gpu.constants_sigmas_cap).tobytes()).hexdigest() == g["constants_sigmas_cap_sha256"]
    w = DC.dummy_witness_bb(lg, pi_row, seed=g["witness_seed"])
    proof = gpu.prove_once(w)
    assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["sha256"]
    assert gpu.verify(proof)
    view = D.DummyCircuit.verifier_view(lg, D.CircuitConfig.babybear(ch), BB, k_is)   # (what prove_by_stages reads of a circuit)
    view.set_cap(gpu.constants_sigmas_cap)
    assert prove_by_stages(gpu, view, w, [], GB_BABYBEAR) == proof
    gpu.free()
    ctx.trim()


@pytest.mark.parametrize("lg", [22, 23])
def test_prove_above_2pow21_rows_verifies(ctx, lg):
    """prove() of the 2^22- and 2^23-row Goldilocks dummy circuits (num_challenges 3; 2^23 rows: one outer radix-2 step around the
    2^22-row passes, ~170 GB of commitments): gb_verify and the oracle's verifier accept the proof, a flipped byte in the opening set
    is rejected by both"""
    ch = 3
    cs, k_is, pi_row, _ = DC.build_dummy_circuit(lg)
    gpu = CircuitData(ctx, lg, cs, k_is, num_challenges=ch)
    del cs
    w = DC.dummy_witness(lg, pi_row, seed=5)
    proof = gpu.prove(w)
    del w
    assert gpu.verify(proof)
    view = D.DummyCircuit.verifier_view(lg, D.CircuitConfig(num_challenges=ch), GL, k_is)
    view.set_cap(gpu.constants_sigmas_cap)
    assert (view.circuit_digest == gpu.circuit_digest).all()
    assert D.verify(view, proof)
    bad = bytearray(proof)
    bad[3 * 16 * 32 + 100] ^= 1
    with pytest.raises(N.VerifyError):
        gpu.verify(bytes(bad))
    with pytest.raises(AssertionError):
        D.verify(view, bytes(bad))
    gpu.free()
    ctx.trim()


@pytest.mark.parametrize("lg,ch", [(22, 12), (23, 13)])
def test_babybear_prove_above_2pow21_rows_verifies(ctx, lg, ch):
    """BabyBear at 2^22 / 2^23 rows: (31 - degree_bits) c >= 100 (circuit_builder.rs:1190-1192) asks for 12 / 13 challenges -
    run as two slices in the quotient kernel (csrc/challenge_slices.hpp).  The proof may need the reference's retry (a zero
    denominator meets a third to a half of the attempts at these sizes); gb_verify and the oracle verifier accept it."""
    from oracle.fields import BB
    cs, k_is, pi_row, _ = DC.build_dummy_circuit_bb(lg)
    gpu = CircuitData(ctx, lg, cs, k_is, num_wires=167, num_routed_wires=41, num_challenges=ch, arity_bits=3, field=GB_BABYBEAR)
    del cs
    proof = None
    for seed in range(48):   # 2^23 rows, 13 challenges: ~2^23 * 41 * 13 / p = 2.1 zero denominators expected per attempt, 12 % of the attempts get through
        try:
            proof = gpu.prove_once(DC.dummy_witness_bb(lg, pi_row, seed=seed))
            break
        except N.PermArgZeroError:
            continue
    assert proof is not None
    assert gpu.verify(proof)
    view = D.DummyCircuit.verifier_view(lg, D.CircuitConfig.babybear(ch), BB, k_is)
    view.set_cap(gpu.constants_sigmas_cap)
    assert (view.circuit_digest == gpu.circuit_digest).all()
    assert D.verify(view, proof)
    gpu.free()
    ctx.trim()


def test_zs_running_product_above_1024_blocks(ctx):
    """wires_permutation_partial_products_and_zs at 2^21 rows (plonk/prover.rs:480-546): the running product is a scan over 2048
    blocks of 1024 rows - more than one thread per block total.  The dummy circuit's sigma is the identity almost everywhere (every
    quotient 1: a broken carry between blocks would go unnoticed), so the sigma columns are rotated by one and the witness is random:
    Z(0) = 1 and Z(row + 1) prod_j den_j(row) == Z(row) prod_j num_j(row) on every block boundary and on random rows."""
    lg, ch, nr = 21, 3, 80
    n, P = 1 << lg, GL.P
    cs, k_is, pi_row, _ = DC.build_dummy_circuit(lg)
    cs[3:3 + nr] = np.roll(cs[3:3 + nr], 1, axis=0)      # sigma_j <- sigma_(j-1): any field values will do for the formula
    gpu = CircuitData(ctx, lg, cs, k_is, num_challenges=ch)
    rng = np.random.default_rng(21)
    w = np.zeros((135, n), dtype=np.uint64)
    w[:nr] = rng.integers(0, P, (nr, n), dtype=np.uint64)
    betas = [int(x) for x in rng.integers(1, P, ch, dtype=np.uint64)]
    gammas = [int(x) for x in rng.integers(1, P, ch, dtype=np.uint64)]
    out = gpu.zs_partial_products(w, betas, gammas)
    assert out.shape == (ch * 10, n)
    sub = DC.gl_powers(pow(1753635133440165772, 1 << (32 - lg), P), n)
    rows = sorted(set([0, 1, n - 2] + [1024 * b - 1 for b in range(1, 2048)] + [int(r) for r in rng.integers(0, n - 1, 512)]))
    ks = [int(k) for k in k_is]
    for c in range(ch):
        Z = out[c]
        assert int(Z[0]) == 1
        for r in rows:
            x = int(sub[r])
            num = den = 1
            for j in range(nr):
                wv = int(w[j, r])
                num = num * ((wv + betas[c] * ks[j] % P * x + gammas[c]) % P) % P
                den = den * ((wv + betas[c] * int(cs[3 + j, r]) + gammas[c]) % P) % P
            assert int(Z[r + 1]) * den % P == int(Z[r]) * num % P, (c, r)
    gpu.free()
    ctx.trim()


@pytest.mark.parametrize("field_name,log_n", [("goldilocks", 23), ("babybear", 23), ("goldilocks", 24), ("babybear", 24), ("goldilocks", 26)])
def test_from_values_2pow23_and_up_sparse_polynomials(ctx, field_name, log_n):
    """2^23 ... 2^26 rows (one outer radix-2 ... radix-16 step around the 2^22-row passes; BabyBear's two-adicity ends at 2^24 rows
    with rate_bits 3; round 5 stopped at 2^24) through size-independent
    properties, no oracle run of that size: the values of SPARSE polynomials a x^k1 + b x^k2 + c with exponents all over [0, n) are
    built on the host with vectorised powers; from_values must return exactly those coefficients (EVERY one of the n compared), the LDE
    rows must equal the polynomials evaluated directly at 7 w_N^i, and sampled Merkle paths must verify against the cap."""
    n = 1 << log_n
    if field_name == "goldilocks":
        P, tag, dt, powers, hmod = GL.P, N.GB_GOLDILOCKS, np.uint64, DC.gl_powers, O
        gen_N = pow(1753635133440165772, 1 << (32 - log_n - 3), P)
        mul = DC.gl_mul
        shift = 7
    else:
        P, tag, dt, powers, hmod = DC.BB_P, GB_BABYBEAR, np.uint32, DC.bb_powers, B
        gen_N = pow(0x1a427a41, 1 << (27 - log_n - 3), P)
        mul = DC.bb_mul
        shift = 31
    w_n = pow(gen_N, 8, P)
    rng = np.random.default_rng(log_n)
    polys = []
    for _ in range(2 if log_n <= 24 else 1):     # (the host side of this test is numpy: one polynomial is enough work at 2^26 rows)
        ks = sorted({0, int(rng.integers(1, n)), n - 1 - int(rng.integers(0, 1000)), int(rng.integers(1, 1 << 20))})
        polys.append({k: int(rng.integers(1, P, dtype=np.uint64)) for k in ks})
    vals = np.zeros((len(polys), n), dtype=dt)
    for c, poly in enumerate(polys):
        acc = np.zeros(n, dtype=np.uint64)
        for k, a in poly.items():
            term = mul(powers(pow(w_n, k, P), n), dt(a)).astype(np.uint64)      # a (w_n^k)^i
            with np.errstate(over="ignore"):
                t = acc + term                                                   # BabyBear: no wrap; Goldilocks: 2^64 = 2^32 - 1 (mod p)
                if P > (1 << 32):
                    t = np.where(t < acc, t + np.uint64(0xFFFFFFFF), t)
                acc = np.where(t >= np.uint64(P), t - np.uint64(P), t)
        vals[c] = acc.astype(dt)
    gpu = PolynomialBatch.from_values(ctx, vals, 3, 4, field=tag)
    for c, poly in enumerate(polys):
        want = np.zeros(n, dtype=dt)
        for k, a in poly.items():
            want[k] = a
        assert (gpu.polynomial(c) == want).all(), "coefficients of column %d" % c
    Nn = n << 3
    for i in (0, 1, n - 1, 5 * n + 12345, Nn - 1):
        x = shift * pow(gen_N, i, P) % P
        want = [sum(a * pow(x, k, P) for k, a in poly.items()) % P for poly in polys]
        got = gpu.get_lde_values(i, 1)
        assert [int(v) for v in got] == want, i
    cap = gpu.merkle_tree.cap
    for leaf in (0, 77, Nn - 1, int(rng.integers(0, Nn))):
        row, sib = gpu._leaf(leaf)
        assert hmod.merkle_verify(row, leaf, cap, sib), leaf
    gpu.free()
    ctx.trim()
