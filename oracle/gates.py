"""TEST ORACLE - the gate constraint evaluators at one extension-field point (the verifier's side).

Test infrastructure only.  Follows eval_unfiltered of (paths relative to /root/reference/plonky2/src/gates):
  noop.rs, constant.rs:64-72, public_input.rs:52-60, arithmetic_base.rs:83-100, poseidon_goldilocks.rs:124-221,
  poseidon2_babybear.rs:203-313
and compute_filter (gate.rs:391-404).  `e` is a Field of oracle/fields.py; extension elements are tuples.
A gate is the tuple the C oracle and the product ABI use: (kind, param, selector_index, group_start, group_end).
"""
import os
import re

NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON, POSEIDON2_BABYBEAR = 0, 1, 2, 3, 4, 5
UNUSED_SELECTOR = 0xFFFFFFFF  # selectors.rs:13
POSEIDON_NUM_CONSTRAINTS = 12 * 7 + 22 + 12 + 1 + 4
POSEIDON2_BB_CONSTRAINTS_PER_OP = 1 + 8 + 16 * 7 + 13 + 16

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
    return {NOOP: 0, CONSTANT: param, PUBLIC_INPUT: hout, ARITHMETIC: param, POSEIDON: POSEIDON_NUM_CONSTRAINTS,
            POSEIDON2_BABYBEAR: POSEIDON2_BB_CONSTRAINTS_PER_OP * param}[kind]


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


def _bb_tables():
    text = open(os.path.join(os.path.dirname(__file__), "poseidon_constants.h")).read()
    out = {}
    for m in re.finditer(r"#define BB_POSEIDON2_(\w+)_LIST \\\n((?:[^\n]*\\\n)*[^\n]*)", text):
        out[m.group(1)] = [int(x.rstrip("uUlL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", m.group(2))]
    return out


def _poseidon2_bb(e, w, num_ops):
    T = _bb_tables()
    ext, internal = T["EXTERNAL_CONSTANTS"], T["INTERNAL_CONSTANTS"]
    shifts = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
    add, sub, mul, sc, k = e.eadd, e.esub, e.emul, e.escale, e.efrom

    def sbox(x):
        x2 = mul(x, x)
        return mul(mul(x, x2), mul(x2, x2))

    def external(s):
        s = list(s)
        for i in range(0, 16, 4):
            x = s[i:i + 4]
            t01, t23 = add(x[0], x[1]), add(x[2], x[3])
            t0123 = add(t01, t23)
            t01123, t01233 = add(t0123, x[1]), add(t0123, x[3])
            s[i:i + 4] = [add(t01123, t01), add(t01123, add(x[2], x[2])), add(t01233, t23), add(t01233, add(x[0], x[0]))]
        sums = [e.zero] * 4
        for j in range(16):
            sums[j % 4] = add(sums[j % 4], s[j])
        return [add(s[i], sums[i % 4]) for i in range(16)]

    def internal_layer(s):
        s = [sc(x, 943718400) for x in s]
        part = e.zero
        for x in s[1:]:
            part = add(part, x)
        full = add(part, s[0])
        return [sub(part, s[0])] + [add(full, sc(s[i + 1], 1 << shifts[i])) for i in range(15)]

    out = []
    for op in range(num_ops):
        in0, start_delta = 33 * op, num_ops * 33 + op * 133
        out0, start_full_0 = in0 + 16, start_delta + 8
        start_partial = start_full_0 + 48
        start_full_1 = start_partial + 13
        swap = w[in0 + 32]
        out.append(mul(swap, sub(swap, e.one)))
        for i in range(8):
            out.append(sub(mul(swap, sub(w[in0 + i + 8], w[in0 + i])), w[start_delta + i]))
        s = [add(w[in0 + i], w[start_delta + i]) for i in range(8)] + [sub(w[in0 + i + 8], w[start_delta + i]) for i in range(8)]
        s = external(s)
        for r in range(4):
            s = [add(s[i], k(ext[16 * r + i])) for i in range(16)]
            if r:
                for i in range(16):
                    sin = w[start_full_0 + 16 * (r - 1) + i]
                    out.append(sub(s[i], sin))
                    s[i] = sin
            s = external([sbox(x) for x in s])
        for r in range(13):
            s[0] = add(s[0], k(internal[r]))
            sin = w[start_partial + r]
            out.append(sub(s[0], sin))
            s[0] = sbox(sin)
            s = internal_layer(s)
        for r in range(4, 8):
            s = [add(s[i], k(ext[16 * r + i])) for i in range(16)]
            for i in range(16):
                sin = w[start_full_1 + 16 * (r - 4) + i]
                out.append(sub(s[i], sin))
                s[i] = sin
            s = external([sbox(x) for x in s])
        for i in range(16):
            out.append(sub(s[i], w[out0 + i]))
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
    if kind == POSEIDON2_BABYBEAR:
        assert e.name == "babybear", "Poseidon2BabyBearGate is the BabyBear gate"
        return _poseidon2_bb(e, wires, param)
    raise ValueError("gate kind %r" % (kind,))
