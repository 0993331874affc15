"""TEST ORACLE - the gate constraint evaluators at one extension-field point (the verifier's side).

Test infrastructure only.  Follows eval_unfiltered of (paths relative to /root/reference/plonky2/src/gates):
  noop.rs, constant.rs:64-72, public_input.rs:52-60, arithmetic_base.rs:83-100, poseidon_goldilocks.rs:124-221
and compute_filter (gate.rs:391-404).  `e` is a Field of oracle/fields.py; extension elements are tuples.
A gate is the tuple the C oracle and the product ABI use: (kind, param, selector_index, group_start, group_end).
"""
import os
import re

NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON = 0, 1, 2, 3, 4
UNUSED_SELECTOR = 0xFFFFFFFF  # selectors.rs:13
POSEIDON_NUM_CONSTRAINTS = 12 * 7 + 22 + 12 + 1 + 4

_TABLES = None


def poseidon_tables():
    """The Poseidon-12 tables of oracle/poseidon_constants.h (hash/poseidon_goldilocks.rs:114-492)."""
    global _TABLES
    if _TABLES is None:
        text = open(os.path.join(os.path.dirname(__file__), "poseidon_constants.h")).read()
        _TABLES = {}
        for m in re.finditer(r"#define GL_POSEIDON_(\w+)_LIST \\\n((?:[^\n]*\\\n)*[^\n]*)", text):
            _TABLES[m.group(1)] = [int(x.rstrip("uUlL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", m.group(2))]
    return _TABLES


def num_constraints(gate, hout):
    kind, param = gate[0], gate[1]
    return {NOOP: 0, CONSTANT: param, PUBLIC_INPUT: hout, ARITHMETIC: param, POSEIDON: POSEIDON_NUM_CONSTRAINTS}[kind]


def compute_filter(e, row, gate, s, many_selectors):
    f = e.one
    for i in range(gate[3], gate[4]):
        if i != row:
            f = e.emul(f, e.esub(e.efrom(i), s))
    if many_selectors:
        f = e.emul(f, e.esub(e.efrom(UNUSED_SELECTOR % e.P), s))
    return f


def _poseidon(e, w):
    T = poseidon_tables()
    rc, circ, diag = T["ALL_ROUND_CONSTANTS"], T["MDS_CIRC"], T["MDS_DIAG"]
    WIRE_SWAP, START_DELTA, START_FULL_0 = 24, 25, 29
    START_PARTIAL = START_FULL_0 + 36
    START_FULL_1 = START_PARTIAL + 22
    add, sub, mul, sc, k = e.eadd, e.esub, e.emul, e.escale, e.efrom
    out = []
    swap = w[WIRE_SWAP]
    out.append(mul(swap, sub(swap, e.one)))
    for i in range(4):
        out.append(sub(mul(swap, sub(w[i + 4], w[i])), w[START_DELTA + i]))
    s = [None] * 12
    for i in range(4):
        s[i] = add(w[i], w[START_DELTA + i])
        s[i + 4] = sub(w[i + 4], w[START_DELTA + i])
    for i in range(8, 12):
        s[i] = w[i]

    def sbox(x):
        x2 = mul(x, x)
        return mul(mul(x, x2), mul(x2, x2))

    def mds(v):
        res = []
        for r in range(12):
            acc = e.zero
            for i in range(12):
                acc = add(acc, sc(v[(i + r) % 12], circ[i]))
            res.append(add(acc, sc(v[r], diag[r])))
        return res

    ctr = 0
    for r in range(4):
        s = [add(s[i], k(rc[12 * ctr + i] % e.P)) for i in range(12)]
        if r:
            for i in range(12):
                sin = w[START_FULL_0 + 12 * (r - 1) + i]
                out.append(sub(s[i], sin))
                s[i] = sin
        s = mds([sbox(x) for x in s])
        ctr += 1
    s = [add(s[i], k(T["FAST_PARTIAL_FIRST_ROUND_CONSTANT"][i] % e.P)) for i in range(12)]
    init = T["FAST_PARTIAL_ROUND_INITIAL_MATRIX"]
    res = [s[0]] + [e.zero] * 11
    for r in range(1, 12):
        for c in range(1, 12):
            res[c] = add(res[c], sc(s[r], init[(r - 1) * 11 + (c - 1)] % e.P))
    s = res
    for r in range(22):
        sin = w[START_PARTIAL + r]
        out.append(sub(s[0], sin))
        s[0] = sbox(sin)
        if r != 21:
            s[0] = add(s[0], k(T["FAST_PARTIAL_ROUND_CONSTANTS"][r] % e.P))
        d = sc(s[0], circ[0] + diag[0])
        for i in range(1, 12):
            d = add(d, sc(s[i], T["FAST_PARTIAL_ROUND_W_HATS"][r * 11 + i - 1] % e.P))
        s = [d] + [add(sc(s[0], T["FAST_PARTIAL_ROUND_VS"][r * 11 + i - 1] % e.P), s[i]) for i in range(1, 12)]
    ctr += 22
    for r in range(4):
        s = [add(s[i], k(rc[12 * ctr + i] % e.P)) for i in range(12)]
        for i in range(12):
            sin = w[START_FULL_1 + 12 * r + i]
            out.append(sub(s[i], sin))
            s[i] = sin
        s = mds([sbox(x) for x in s])
        ctr += 1
    for i in range(12):
        out.append(sub(s[i], w[12 + i]))
    return out


def eval_unfiltered(e, gate, wires, consts, pi_hash):
    """consts = local_constants after the selectors (vars.remove_prefix, gate.rs:165-186)"""
    kind, param = gate[0], gate[1]
    if kind == NOOP:
        return []
    if kind == CONSTANT:
        return [e.esub(consts[i], wires[i]) for i in range(param)]
    if kind == PUBLIC_INPUT:
        return [e.esub(wires[i], e.efrom(int(pi_hash[i]))) for i in range(e.hout)]
    if kind == ARITHMETIC:
        c0, c1 = consts[0], consts[1]
        return [e.esub(wires[4 * i + 3], e.eadd(e.emul(e.emul(wires[4 * i], wires[4 * i + 1]), c0), e.emul(wires[4 * i + 2], c1)))
                for i in range(param)]
    if kind == POSEIDON:
        assert e.name == "goldilocks", "PoseidonGate is the Goldilocks gate"
        return _poseidon(e, wires)
    raise ValueError("gate kind %r" % (kind,))
