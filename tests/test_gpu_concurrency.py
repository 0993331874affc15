"""Two contexts on one device driven from two host threads: the in-process form of "independent circuits, one per context"
(contexts share nothing; the C ABI only asks the caller to serialise calls on ONE context).  Every proof and
every commitment made concurrently must equal, byte for byte, the one made alone.  -m gpu only."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from oracle import plonk_dummy as D
from oracle.fields import BB, GL
from plonky2_goldibear_amd import CircuitData, GpuContext, PolynomialBatch
from plonky2_goldibear_amd import native as N

pytestmark = pytest.mark.gpu


def _circuit(ctx, circ, tag):
    cfg = circ.cfg
    return CircuitData(ctx, circ.degree_bits, circ.constants_sigmas, circ.k_is, num_wires=cfg.num_wires,
                       num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants, num_challenges=cfg.num_challenges,
                       arity_bits=cfg.arity_bits, gate_constant=circ.GATE_CONSTANT, gate_pi=circ.GATE_PI, field=tag)


def _run_threads(fns):
    errs = []

    def wrap(f):
        def g():
            try:
                f()
            except BaseException as e:  # noqa: BLE001 - reported to the main thread
                errs.append(e)
        return g
    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]


def test_two_contexts_two_threads_same_bytes_as_sequential():
    ctxs = [GpuContext(0), GpuContext(0)]
    # different circuits and fields on the two contexts, so that a shared table or scratch buffer would show
    circs = [D.DummyCircuit(13, D.CircuitConfig(num_challenges=2), F=GL), D.DummyCircuit(12, D.CircuitConfig.babybear(6), F=BB)]
    tags = [N.GB_GOLDILOCKS, N.GB_BABYBEAR]
    gpus = [_circuit(ctxs[i], circs[i], tags[i]) for i in range(2)]
    rounds = 6
    wits = [[circs[i].witness(seed=100 * i + r) for r in range(rounds)] for i in range(2)]
    alone = [[gpus[i].prove(w) for w in wits[i]] for i in range(2)]            # sequential, one context at a time
    assert alone[0][0] == D.prove_cpu(circs[0], wits[0][0])[0]                  # and equal to the oracle's bytes
    assert alone[1][0] == D.prove_cpu(circs[1], wits[1][0])[0]
    together = [[None] * rounds, [None] * rounds]

    def worker(i):
        def f():
            for r in range(rounds):
                together[i][r] = gpus[i].prove(wits[i][r])
        return f
    for _ in range(2):   # twice: the second pass runs on warm pools
        _run_threads([worker(0), worker(1)])
        assert together == alone
    for g in gpus:
        g.free()
    for c in ctxs:
        c.close()


def test_concurrent_commits_match_oracle():
    ctxs = [GpuContext(0), GpuContext(0), GpuContext(0)]
    shapes = [(14, 5), (12, 9), (15, 2)]
    vals = [O.splitmix64_fill(900 + i, c << lg).reshape(c, 1 << lg) for i, (lg, c) in enumerate(shapes)]
    want = [O.PolynomialBatch.from_values(v, 3, 4) for v in vals]
    got = [[None] * 4 for _ in shapes]

    def worker(i):
        def f():
            for r in range(4):
                b = PolynomialBatch.from_values(ctxs[i], vals[i], 3, 4)
                got[i][r] = (b.merkle_tree.cap.copy(), b.polynomials.copy(), b._leaf(r * 37)[1].copy())
                b.free()
        return f
    _run_threads([worker(i) for i in range(3)])
    for i in range(3):
        for r in range(4):
            cap, pol, sib = got[i][r]
            assert (cap == want[i].cap).all() and (pol == want[i].polynomials).all() and (sib == want[i].prove(r * 37)).all()
    for c in ctxs:
        c.close()


def test_two_contexts_stage_pageable_columns_concurrently():
    """each context has its own page-locked ring and copy threads (round 5): two host threads commit wide batches from separately
    allocated pageable columns on two contexts at once, repeatedly (the rings' slots come round) - every tree equals the one made alone"""
    ctxs = [GpuContext(0), GpuContext(0)]
    ctxs[1].set_option("copy_threads", 2)
    log_n = 16
    mats = [O.splitmix64_fill(0xC0 + i, 72 << log_n).reshape(72, 1 << log_n) for i in range(2)]
    want = [O.PolynomialBatch.from_values(m, 3, 4).cap for m in mats]
    cols = [[np.array(c, copy=True) for c in m] for m in mats]
    got = [[None] * 5, [None] * 5]

    def worker(i):
        def f():
            for r in range(5):
                b = PolynomialBatch.from_values(ctxs[i], cols[i], 3, 4)
                got[i][r] = b.merkle_tree.cap
                b.free()
        return f
    _run_threads([worker(0), worker(1)])
    for i in range(2):
        for r in range(5):
            assert (got[i][r] == want[i]).all(), (i, r)
    for c in ctxs:
        c.close()
