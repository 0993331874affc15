"""Seeded random circuits through the host mirror of CircuitBuilder (arithmetic, constants, copy constraints, in-circuit hashing,
public inputs - the gate kinds the CPU oracle prover evaluates natively) under random configurations (field, num_challenges,
rate_bits 3 .. 8 against quotient_degree_factor 8, query rounds and proof-of-work bits that meet the builder's security check):
prove() through the C ABI == the oracle prover on the same constants / sigmas / witness, byte for byte; gb_verify and the oracle
verifier accept; a wrong public input is refused.  -m gpu."""
import numpy as np
import pytest

from oracle import plonk_dummy as PD
from plonky2_goldibear_amd import GpuContext, VerifyError, native as N
from plonky2_goldibear_amd.circuit_builder import CircuitBuilder, CircuitConfig, PartialWitness

from circuits import oracle_circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def random_circuit(seed):
    rng = np.random.default_rng(20_000 + seed)
    gl = seed % 2 == 0
    rate = int(rng.choice([3, 3, 4, 5, 7, 8]))
    pow_bits = int(rng.integers(8, 19))
    queries = -(-(100 - pow_bits) // rate)                       # conjectured security >= 100 bits (circuit_builder.rs:1176-1186)
    nops = int(rng.choice([3, 20, 90, 300, 900]))
    # rows are not known before build(): take the challenge count that is enough for 2^12 rows (circuit_builder.rs:1190-1192)
    nch = (2 if gl else 6) + int(rng.choice([0, 0, 1, 3]))
    kw = dict(rate_bits=rate, proof_of_work_bits=pow_bits, num_query_rounds=queries, num_challenges=nch,
              cap_height=int(rng.integers(0, 5)))
    cfg = CircuitConfig.standard_recursion_config_gl(**kw) if gl else CircuitConfig.recursion_config_bb_narrow(**kw)
    b = CircuitBuilder(cfg)
    p = b.F.p
    pw = PartialWitness()
    targets, values = [], []
    for _ in range(int(rng.integers(1, 5))):
        t, v = b.add_virtual_target(), int(rng.integers(0, p, dtype=np.uint64))
        pw.set_target(t, v)
        targets.append(t), values.append(v)
    pick = lambda: int(rng.integers(0, len(targets)))
    for _ in range(nops):
        op = int(rng.integers(0, 7))
        i, j, k = pick(), pick(), pick()
        if op == 0:
            t, v = b.add(targets[i], targets[j]), (values[i] + values[j]) % p
        elif op == 1:
            t, v = b.mul(targets[i], targets[j]), values[i] * values[j] % p
        elif op == 2:
            t, v = b.mul_add(targets[i], targets[j], targets[k]), (values[i] * values[j] + values[k]) % p
        elif op == 3:
            t, v = b.sub(targets[i], targets[j]), (values[i] - values[j]) % p
        elif op == 4:
            c = int(rng.integers(0, p, dtype=np.uint64))
            t, v = b.mul_const(c, targets[i]), c * values[i] % p
        elif op == 5:
            c = int(rng.integers(0, 1000))
            t, v = b.constant(c), c
        else:
            t, v = b.square(targets[i]), values[i] * values[i] % p
        targets.append(t), values.append(v)
        if rng.random() < 0.05:       # a copy constraint that holds: tie the value to its constant
            b.connect(t, b.constant(v))
    pis = []
    for _ in range(int(rng.choice([0, 1, 2, 5]))):
        i = pick()
        b.register_public_input(targets[i])
        pis.append(values[i])
    return b, pw, pis


@pytest.mark.parametrize("seed", range(24))
def test_random_circuit(ctx, seed):
    b, pw, want_pis = random_circuit(seed)
    c = b.build(ctx)
    cfg = c.config
    what = "seed %d: field %d, 2^%d rows, rate %d, %d challenges, %d gates in the set, %d public inputs" % (
        seed, cfg.field, c.degree_bits, cfg.rate_bits, cfg.num_challenges, len(c.gate_table), len(want_pis))
    w, pis = c.generate_witness(pw)
    assert [int(x) for x in pis] == want_pis, what
    oc = oracle_circuit(c, len(pis))
    assert (c.data.circuit_digest == oc.circuit_digest).all(), what      # the oracle's own constants/sigmas commitment
    proof = None
    for attempt in range(6):   # BabyBear: a zero denominator is a natural event; the reference re-draws a wire, here: next blinding
        try:
            want, _ = PD.prove_cpu(oc, w, pis)
        except RuntimeError as e:
            assert "rc=1" in str(e), what
            w, pis = c.generate_witness(pw, rng=np.random.default_rng(seed * 100 + attempt))
            continue
        proof = c.data.prove_once(w, pis)
        assert proof == want, what
        break
    assert proof is not None, what
    assert c.data.verify(proof) and PD.verify(oc, proof), what
    if want_pis:   # the same proof under another public input: refused
        bad = bytearray(proof)
        bad[-(8 if cfg.field == N.GB_GOLDILOCKS else 4)] ^= 1     # the low byte of the last public input
        with pytest.raises((VerifyError, N.ShapeError)):
            c.data.verify(bytes(bad))
    c.data.free()
