"""prove() with FriReductionStrategy::Fixed / MinSize lists (fri/reduction_strategies.rs:11-56) handed over through
gb_circuit_set_fri_reduction_arity_bits: proof bytes identical to the CPU oracle prover with the same list, accepted by gb_verify
and the oracle verifier, compress / decompress round trip, rejected under a different list.  -m gpu."""
import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import GpuContext, VerifyError, fri_params as FP, native as N
from plonky2_goldibear_amd.prover import CircuitData

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = GpuContext(0)
    yield c
    c.close()


def _gpu(ctx, circ, tag, bits):
    cfg = circ.cfg
    return CircuitData(ctx, circ.degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                       num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants, num_challenges=cfg.num_challenges,
                       arity_bits=cfg.arity_bits, gate_constant=circ.GATE_CONSTANT, gate_pi=circ.GATE_PI, field=tag,
                       reduction_arity_bits=bits)


@pytest.mark.parametrize("field_name,degree_bits,strategy", [
    ("goldilocks", 10, ("fixed", [3, 2, 2])), ("goldilocks", 10, ("fixed", [1, 1, 1, 1])), ("goldilocks", 12, ("min_size", None)),
    ("goldilocks", 9, ("fixed", [])), ("goldilocks", 13, ("min_size", 3)),
    ("babybear", 10, ("fixed", [2, 2, 1])), ("babybear", 12, ("min_size", None)),
])
def test_proofs_with_other_reduction_strategies(ctx, field_name, degree_bits, strategy):
    if field_name == "goldilocks":
        F, tag, cfg = GL, N.GB_GOLDILOCKS, D.CircuitConfig(num_challenges=2)
    else:
        F, tag, cfg = BB, N.GB_BABYBEAR, D.CircuitConfig.babybear(7)
    bits = FP.reduction_arity_bits(strategy, degree_bits, cfg.rate_bits, cfg.cap_height, cfg.num_query_rounds)
    circ = D.DummyCircuit(degree_bits, cfg, F=F)
    circ.reduction_arity_bits = list(bits)
    gpu = _gpu(ctx, circ, tag, bits)
    assert gpu.reduction_arity_bits == list(bits)
    circ.set_cap(gpu.constants_sigmas_cap)
    w = circ.witness(seed=degree_bits)
    got = gpu.prove(w)
    want, _ = D.prove_cpu(circ, w)
    assert got == want
    assert gpu.verify(got) and D.verify(circ, got)
    comp = gpu.compress(got)
    assert gpu.decompress(comp) == got and gpu.verify_compressed(comp)
    # the stock list of the same configuration is a different FriParams: the proof does not parse / verify under it
    stock = D.reduction_arity_bits(cfg, degree_bits)
    if stock != list(bits):
        gpu.set_reduction_arity_bits(stock)
        with pytest.raises((N.ShapeError, VerifyError)):
            gpu.verify(got)
        stock_circ = D.DummyCircuit(degree_bits, cfg, F=F)
        stock_circ.set_cap(gpu.constants_sigmas_cap)
        assert gpu.prove(w) == D.prove_cpu(stock_circ, w)[0]      # and back: the setter is the whole state
    gpu.free()


def test_invalid_lists_are_rejected(ctx):
    circ = D.DummyCircuit(8, D.CircuitConfig(num_challenges=2), F=GL)
    gpu = _gpu(ctx, circ, N.GB_GOLDILOCKS, None)
    stock = gpu.reduction_arity_bits
    for bad in ([9], [0, 1], [4, 4, 4], [4] * 33):     # wider than 2^8 / zero / past degree_bits (and below the cap) / too many
        with pytest.raises(N.GoldibearError) as e:
            gpu.set_reduction_arity_bits(bad)
        assert e.value.status == N.GB_ERR_INVALID
    assert gpu.reduction_arity_bits == stock
    gpu.free()
