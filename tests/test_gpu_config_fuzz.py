"""Seeded random walk over the configuration space of prove() (CircuitConfig / FriConfig, plonk/circuit_data.rs:63-93,
fri/mod.rs:25-45): field, rows, wire counts, num_challenges, rate_bits against quotient_degree_factor 8, cap_height,
proof-of-work bits, query rounds, the three FriReductionStrategy variants with arities up to 2^8, zero-knowledge salting.
Every drawn configuration is one the reference accepts (its build() asserts are mirrored when drawing); for each the proof
BYTES of the GPU prover equal the CPU oracle prover's, and gb_verify and the oracle verifier accept them.  The fixed-parameter
tests next to this file pin the shapes the kernels special-case; this one is for the combinations nobody thought of.  -m gpu."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import GpuContext, fri_params as FP, native as N
from plonky2_goldibear_amd.prover import CircuitData

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def draw(seed):
    """one configuration the reference's build() would accept, as (F, tag, degree_bits, oracle config, arity list or None, zk)"""
    rng = np.random.default_rng(seed)
    F, tag = (GL, N.GB_GOLDILOCKS) if seed % 2 == 0 else (BB, N.GB_BABYBEAR)
    lg = int(rng.integers(3, 11))
    rate = int(rng.choice([3, 3, 4, 5, 6, 7, 8]))
    min_c = -(-100 // (F.order_bits - lg))                       # circuit_builder.rs:1190-1192
    nch = int(min(16, min_c + rng.choice([0, 0, 1, 2, 3, 5, 9])))
    routed = int(rng.choice([16, 25, 37, 41, 64, 80]))
    wires = routed + int(rng.choice([0, 1, 20, 55, 94]))
    wires = max(wires, F.hout + 2)                                # PublicInputGate wires + something to fill
    cap = int(rng.integers(0, 5))
    queries = int(rng.integers(1, 31))
    pow_bits = int(rng.integers(0, 17))
    kind = int(rng.integers(0, 3))
    cap = min(cap, lg + rate - 1)
    ab, fpb, bits = 4, 5, None          # the stock pair where the list comes from another strategy
    while kind == 0:                    # ConstantArityBits(ab, fpb): redraw what the reference's own assert refuses (:45)
        ab, fpb = int(rng.integers(1, 9)), int(rng.integers(0, 6))
        try:
            FP.constant_arity_bits(ab, fpb, lg, rate, cap)
            break
        except AssertionError:
            pass
    cfg_kw = dict(num_challenges=nch, num_wires=wires, num_routed_wires=routed, rate_bits=rate, cap_height=cap,
                  proof_of_work_bits=pow_bits, num_query_rounds=queries, arity_bits=ab, final_poly_bits=fpb)
    cfg = D.CircuitConfig(**cfg_kw)
    if kind == 1:     # FriReductionStrategy::Fixed: a random list the layers' trees allow (merkle_tree.rs:154-157)
        bits, db = [], lg
        while db > 0 and len(bits) < 6:
            a = int(rng.integers(1, min(8, db) + 1))
            if db + rate - a < cap:
                break
            bits.append(a)
            db -= a
            if rng.random() < 0.25:
                break
    elif kind == 2:   # MinSize(None | Some(max))
        mx = None if rng.random() < 0.5 else int(rng.integers(1, 7))
        bits = FP.reduction_arity_bits(("min_size", mx), lg, rate, cap, queries)
        if any(lg + rate - sum(bits[:i + 1]) < cap for i in range(len(bits))):   # the reference would panic in MerkleTree::new
            bits = None
    zk = bool(rng.random() < 0.25)
    return F, tag, lg, cfg, bits, zk


@pytest.mark.parametrize("seed", range(36))   # (48 until round 6; tools/fuzz_configs.py walks hundreds more offline: profiles/r05_config_fuzz.txt)
def test_random_configuration(ctx, seed):
    F, tag, lg, cfg, bits, zk = draw(seed)
    circ = D.DummyCircuit(lg, cfg, F=F)
    if bits is not None:
        circ.reduction_arity_bits = list(bits)
    salts = None
    if zk:
        circ.zero_knowledge = True
        n_lde = circ.n << cfg.rate_bits
        salts = F.fill(0x5A17 + seed, 3 * 4 * n_lde).reshape(3, 4, n_lde)
    gpu = CircuitData(ctx, lg, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires,
                      num_constants=cfg.num_constants, num_challenges=cfg.num_challenges, rate_bits=cfg.rate_bits,
                      cap_height=cfg.cap_height, proof_of_work_bits=cfg.proof_of_work_bits, num_query_rounds=cfg.num_query_rounds,
                      arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits, gate_constant=circ.GATE_CONSTANT,
                      gate_pi=circ.GATE_PI, field=tag, zero_knowledge=zk, reduction_arity_bits=bits)
    assert gpu.reduction_arity_bits == list(circ.reduction_arity_bits)
    assert (gpu.circuit_digest == circ.circuit_digest).all()      # the oracle's own constants/sigmas commitment
    what = "seed %d: %s 2^%d rows, %r, arities %r, zk %r" % (seed, F.name, lg, cfg.__dict__, circ.reduction_arity_bits, zk)
    for attempt in range(6):   # a zero denominator of the permutation argument is a natural event for BabyBear: next witness
        w = circ.witness(seed=seed + 1000 * attempt)
        try:
            want, _ = D.prove_cpu(circ, w, salts=salts)
        except RuntimeError as e:
            assert "rc=1" in str(e), what
            continue
        got = gpu.prove_once(w, salts=salts)
        assert got == want, what
        assert gpu.verify(got) and D.verify(circ, got), what
        comp = gpu.compress(got)
        assert gpu.decompress(comp) == got and gpu.verify_compressed(comp), what
        break
    else:
        raise AssertionError("six witnesses in a row met a zero denominator: " + what)
    gpu.free()


def test_constant_arity_that_does_not_fit_the_degree_is_rejected(ctx):
    """`assert!(degree_bits >= arity_bits)` inside ConstantArityBits' loop (fri/reduction_strategies.rs:45): 2^6 rows, arity 2^4,
    cap_height 0 - the second reduction would need 4 of the 2 bits that are left.  The reference panics in build(); the library
    (which used to derive a list that wrapped around) leaves the list open - such a pair is what a circuit with a Fixed / MinSize
    strategy passes - and prove() answers GB_ERR_INVALID with the reference's assert until a list has been handed over; with the list
    the valid circuit's strategy would have derived, the proof equals the oracle's."""
    cfg = D.CircuitConfig(num_challenges=2, cap_height=0, arity_bits=4, final_poly_bits=0)
    with pytest.raises(AssertionError):
        D.DummyCircuit(6, cfg, F=GL)
    with pytest.raises(AssertionError):
        FP.constant_arity_bits(4, 0, 6, 3, 0)
    ok = D.DummyCircuit(6, D.CircuitConfig(num_challenges=2, cap_height=0), F=GL)    # the columns of a valid circuit of that size
    gpu = CircuitData(ctx, 6, ok.constants_sigmas, ok.k_is, num_challenges=2, cap_height=0, arity_bits=4, final_poly_bits=0,
                      gate_constant=ok.GATE_CONSTANT, gate_pi=ok.GATE_PI)
    assert gpu.reduction_arity_bits == []
    w = ok.witness(seed=2)
    with pytest.raises(N.GoldibearError) as e:
        gpu.prove(w)
    assert e.value.status == N.GB_ERR_INVALID and "fri/reduction_strategies.rs:45" in str(e.value)
    gpu.set_reduction_arity_bits(ok.reduction_arity_bits)
    assert gpu.prove(w) == D.prove_cpu(ok, w)[0]
    gpu.free()
