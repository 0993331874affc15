"""TEST ORACLE - restated proof (de)serialiser, Fiat-Shamir replay and FRI verifier (Goldilocks, D=2).

Test infrastructure only.  Follows (paths relative to /root/reference/plonky2/src):
  util/serialization/mod.rs:343-379,406-472,490-602,623-836,973-991,1023-1087   byte layout
  plonk/get_challenges.rs:26-101, fri/challenges.rs:15-68                          transcript
  fri/verifier.rs:23-250                                                          FRI checks
  plonk/circuit_data.rs:658-800, plonk/proof.rs:388-440                           FRI instance/openings
  hash/merkle_proofs.rs:54-76                                                     Merkle path rule

Used to pin the C oracle (hashes, Merkle indexing, Challenger order, PoW rule, LDE point
order, two-adic generator, extension non-residue, FRI folding) against the reference's own
serialized regression proof, and later to check proofs produced by the GPU prover.
The PLONK vanishing-polynomial identity (oracle/plonk_dummy.py eval_vanishing_poly) is checked with the gate evaluators of
oracle/gates.py, which cover the recursion circuit's gate set: the regression proof's identity holds under them.
"""
import struct

import numpy as np

from . import oracle as O
from .fields import BB, GL, Field  # noqa: F401

P = O.GL_P
SALT_SIZE = 4

# Goldilocks shorthands (the pinned field); every checker below takes F=GL|BB
eadd, esub, emul, einv, ediv, epow, efrom = GL.eadd, GL.esub, GL.emul, GL.einv, GL.ediv, GL.epow, GL.efrom


def two_adic_generator(bits):
    return GL.two_adic_generator(bits)


def reverse_bits(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


# ----------------------------------------------------------------------------- byte reader
class Reader:
    def __init__(self, data, F=GL):
        self.d, self.o, self.F = memoryview(data), 0, F

    def u8(self):
        v = self.d[self.o]
        self.o += 1
        return v

    def bool(self):
        return bool(self.u8())

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.o)[0]
        self.o += 4
        return v

    def usize(self):
        v = struct.unpack_from("<Q", self.d, self.o)[0]
        self.o += 8
        return v

    def usize_vec(self):
        return [self.usize() for _ in range(self.usize())]

    def field(self):
        v = self.usize() if self.F.elem_bytes == 8 else self.u32()
        assert v < self.F.P, "non-canonical field element"
        return v

    def field_vec(self, n):
        return [self.field() for _ in range(n)]

    def ext(self):
        return tuple(self.field() for _ in range(self.F.D))

    def ext_vec(self, n):
        return [self.ext() for _ in range(n)]

    def hash(self):
        return self.field_vec(self.F.hout)

    def cap(self, h):
        return [self.hash() for _ in range(1 << h)]

    def merkle_proof(self):
        return [self.hash() for _ in range(self.u8())]

    def done(self):
        return self.o == len(self.d)


def read_fri_config(r):
    c = dict(rate_bits=r.usize(), cap_height=r.usize(), num_query_rounds=r.usize(), proof_of_work_bits=r.u32())
    variant = r.u8()
    if variant == 0:
        c["reduction_strategy"] = ("Fixed", r.usize_vec())
    elif variant == 1:
        c["reduction_strategy"] = ("ConstantArityBits", r.usize(), r.usize())
    else:
        c["reduction_strategy"] = ("MinSize", r.usize() if r.u8() else None)
    return c


def read_common_data(data, F=GL):
    """CommonCircuitData up to (not including) the gate list - all the verifier restatement needs."""
    r = Reader(data, F)
    cfg = dict(num_wires=r.usize(), num_routed_wires=r.usize(), num_constants=r.usize(), security_bits=r.usize(),
               num_challenges=r.usize(), max_quotient_degree_factor=r.usize(), use_base_arithmetic_gate=r.bool(),
               zero_knowledge=r.bool())
    cfg["fri_config"] = read_fri_config(r)
    fri_params = dict(config=read_fri_config(r), reduction_arity_bits=r.usize_vec(), degree_bits=r.usize(), hiding=r.bool())
    sel = dict(selector_indices=r.usize_vec())
    sel["groups"] = [(r.usize(), r.usize()) for _ in range(r.usize())]
    cd = dict(config=cfg, fri_params=fri_params, selectors_info=sel, quotient_degree_factor=r.usize(),
              num_gate_constraints=r.usize(), num_constants=r.usize(), num_public_inputs=r.usize())
    cd["k_is"] = r.field_vec(r.usize())
    cd["num_partial_products"] = r.usize()
    cd["num_lookup_polys"] = r.usize()
    cd["num_lookup_selectors"] = r.usize()
    cd["num_luts"] = r.usize()
    cd["_gates_offset"] = r.o
    return cd


def read_gates(data, cd, F=GL):
    """The gate list that closes CommonCircuitData (util/serialization/mod.rs:1910-1913): per gate a u32 tag in the order of
    DefaultGateSerializer (util/serialization/gate_serialization.rs:143-165) and the gate's own serialize() fields.
    Returns the gate tuples of oracle/gates.py with selectors_info merged in:
    (kind, param, selector_index, group_start, group_end, param2, param3)."""
    from . import gates as G
    r = Reader(data, F)
    r.o = cd["_gates_offset"]
    sel = cd["selectors_info"]
    out = []
    n = r.usize()
    for row in range(n):
        tag = r.u32()
        p1 = p2 = p3 = 0
        if tag == 0:
            kind, p1 = G.ARITHMETIC, r.usize()
        elif tag == 1:
            kind, p1 = G.ARITHMETIC_EXTENSION, r.usize()
        elif tag == 2:
            kind, p1, p2 = G.BASE_SUM, r.usize(), 2
        elif tag == 3:
            kind, p1 = G.CONSTANT, r.usize()
        elif tag == 4:
            kind, p1, p2 = G.COSET_INTERPOLATION, r.usize(), r.usize()
            weights = r.field_vec(r.usize())
            assert weights == G.barycentric_weights(F, p1)[1], "barycentric weights are a function of subgroup_bits"
        elif tag == 5:
            kind, p1 = G.EXPONENTIATION, r.usize()
        elif tag == 8:
            kind, p1 = G.MUL_EXTENSION, r.usize()
        elif tag == 9:
            kind = G.NOOP
        elif tag == 10:
            kind = G.POSEIDON_MDS
        elif tag == 11:
            kind = G.POSEIDON
        elif tag == 12:
            kind, p1 = G.PUBLIC_INPUT, F.hout
        elif tag == 13:
            kind, p1, p2, p3 = G.RANDOM_ACCESS, r.usize(), r.usize(), r.usize()
        elif tag == 14:
            kind, p1 = G.REDUCING_EXTENSION, r.usize()
        elif tag == 15:
            kind, p1 = G.REDUCING, r.usize()
        elif tag == 16:
            kind = G.POSEIDON2_BABYBEAR  # num_ops from the config (gates/poseidon2_babybear.rs:65-68)
            p1 = min(cd["config"]["num_wires"] // 166, cd["config"]["num_routed_wires"] // 33)
        elif tag == 17:
            kind, p1, p2 = G.ADD_MANY, r.usize(), r.usize()
        elif tag == 18:
            kind, p1 = G.APPLY_MAT4, r.usize()
        elif tag == 19:
            kind = G.POSEIDON2_INTERNAL_PERMUTATION
        else:
            raise ValueError("gate tag %d (LookupGate / LookupTableGate) is not restated" % tag)
        si = sel["selector_indices"][row]
        gs, ge = sel["groups"][si]
        out.append((kind, p1, si, gs, ge, p2, p3))
    assert r.done()
    return out


def read_verifier_data(data, F=GL):
    r = Reader(data, F)
    h = r.usize()
    vd = dict(constants_sigmas_cap=r.cap(h), circuit_digest=r.hash())
    assert r.done()
    return vd


def read_proof_with_pis(data, cd, F=GL):
    r = Reader(data, F)
    cfg, fp = cd["config"], cd["fri_params"]
    ch = cfg["fri_config"]["cap_height"]
    c = cfg["num_challenges"]
    salt = SALT_SIZE if fp["hiding"] else 0
    nlk = c * cd["num_lookup_polys"]
    pr = dict(wires_cap=r.cap(ch), zs_cap=r.cap(ch), quotient_cap=r.cap(ch))
    pr["openings"] = dict(
        constants=r.ext_vec(cd["num_constants"]), plonk_sigmas=r.ext_vec(cfg["num_routed_wires"]),
        wires=r.ext_vec(cfg["num_wires"]), plonk_zs=r.ext_vec(c), plonk_zs_next=r.ext_vec(c),
        lookup_zs=r.ext_vec(nlk), lookup_zs_next=r.ext_vec(nlk),
        partial_products=r.ext_vec(cd["num_partial_products"] * c),
        quotient_polys=r.ext_vec(cd["quotient_degree_factor"] * c))
    fri = dict(commit_phase_merkle_caps=[r.cap(ch) for _ in fp["reduction_arity_bits"]], query_round_proofs=[])
    widths = [cd["num_constants"] + cfg["num_routed_wires"], cfg["num_wires"] + salt,
              c * (1 + cd["num_partial_products"] + cd["num_lookup_polys"]) + salt,
              c * cd["quotient_degree_factor"] + salt]
    for _ in range(cfg["fri_config"]["num_query_rounds"]):
        initial = [(r.field_vec(w), r.merkle_proof()) for w in widths]
        steps = [(r.ext_vec(1 << ab), r.merkle_proof()) for ab in fp["reduction_arity_bits"]]
        fri["query_round_proofs"].append(dict(initial_trees_proof=initial, steps=steps))
    final_len = 1 << (fp["degree_bits"] - sum(fp["reduction_arity_bits"]))
    fri["final_poly"] = r.ext_vec(final_len)
    fri["pow_witness"] = r.field()
    pr["opening_proof"] = fri
    pis = r.field_vec(r.usize())
    assert r.done(), "trailing bytes in proof"
    return pr, pis


# ----------------------------------------------------------------------------- writer (parity format)
def write_proof_with_pis(pr, pis, F=GL):
    """util/serialization/mod.rs:2103-2151 - the byte layout parity is judged on."""
    out = bytearray()
    fmt = "<Q" if F.elem_bytes == 8 else "<I"

    def f(x):
        out.extend(struct.pack(fmt, int(x)))

    def fv(xs):
        for x in xs:
            f(x)

    def ev(xs):
        for x in xs:
            fv(x)

    def cap(c):
        for h in c:
            fv(h)

    def mp(p):
        out.append(len(p))
        for h in p:
            fv(h)

    cap(pr["wires_cap"]); cap(pr["zs_cap"]); cap(pr["quotient_cap"])
    o = pr["openings"]
    for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "lookup_zs", "lookup_zs_next",
              "partial_products", "quotient_polys"):
        ev(o[k])
    fri = pr["opening_proof"]
    for c in fri["commit_phase_merkle_caps"]:
        cap(c)
    for q in fri["query_round_proofs"]:
        for vals, path in q["initial_trees_proof"]:
            fv(vals); mp(path)
        for evals, path in q["steps"]:
            ev(evals); mp(path)
    ev(fri["final_poly"])
    f(fri["pow_witness"])
    out.extend(struct.pack("<Q", len(pis)))
    fv(pis)
    return bytes(out)


# ----------------------------------------------------------------------------- transcript + FRI
def fri_openings(o):
    """plonk/proof.rs:388-440 to_fri_openings"""
    zeta = o["constants"] + o["plonk_sigmas"] + o["wires"] + o["plonk_zs"] + o["partial_products"] + o["quotient_polys"] + o["lookup_zs"]
    nxt = o["plonk_zs_next"] + o["lookup_zs_next"]
    return [zeta, nxt]


def get_challenges(pr, pis, circuit_digest, cd, F=GL):
    """plonk/get_challenges.rs:26-101 + fri/challenges.rs:24-68"""
    cfg = cd["config"]
    c = cfg["num_challenges"]
    assert cd["num_lookup_polys"] == 0, "lookups are out of scope"
    ch = F.Challenger()
    ch.observe_hash(circuit_digest)
    ch.observe_hash(F.hash_no_pad(np.asarray(pis, dtype=F.dtype)))
    ch.observe_cap(pr["wires_cap"])
    betas = ch.get_n_challenges(c)
    gammas = ch.get_n_challenges(c)
    ch.observe_cap(pr["zs_cap"])
    alphas = ch.get_n_challenges(c)
    ch.observe_cap(pr["quotient_cap"])
    zeta = ch.get_extension_challenge(F.D)
    for batch in fri_openings(pr["openings"]):
        ch.observe_elements([x for e in batch for x in e])
    fri = pr["opening_proof"]
    fri_alpha = ch.get_extension_challenge(F.D)
    fri_betas = []
    for cap in fri["commit_phase_merkle_caps"]:
        ch.observe_cap(cap)
        fri_betas.append(ch.get_extension_challenge(F.D))
    ch.observe_elements([x for e in fri["final_poly"] for x in e])
    ch.observe_element(fri["pow_witness"])
    pow_response = ch.get_challenge()
    lde_size = 1 << (cd["fri_params"]["degree_bits"] + cfg["fri_config"]["rate_bits"])
    idx = [ch.get_challenge() % lde_size for _ in range(cfg["fri_config"]["num_query_rounds"])]
    return dict(plonk_betas=betas, plonk_gammas=gammas, plonk_alphas=alphas, plonk_zeta=zeta, fri_alpha=fri_alpha,
                fri_betas=fri_betas, fri_pow_response=pow_response, fri_query_indices=idx)


def fri_instance(cd, zeta, F=GL):
    """plonk/circuit_data.rs:658-800: (oracle blinding flags, [(point, [(oracle, poly)])])"""
    cfg = cd["config"]
    c = cfg["num_challenges"]
    n_pre = cd["num_constants"] + cfg["num_routed_wires"]
    n_zs_pp = c * (1 + cd["num_partial_products"])
    n_q = c * cd["quotient_degree_factor"]
    all_polys = [(0, i) for i in range(n_pre)] + [(1, i) for i in range(cfg["num_wires"])] + \
                [(2, i) for i in range(n_zs_pp)] + [(3, i) for i in range(n_q)]
    g = F.two_adic_generator(cd["fri_params"]["degree_bits"])
    zeta_next = F.emul(F.efrom(g), zeta)
    next_polys = [(2, i) for i in range(c)]
    blinding = [False, True, True, True]
    return blinding, [(zeta, all_polys), (zeta_next, next_polys)]


def reduce_with_alpha(alpha, xs, F=GL):
    """util/reducing.rs:56-59 ReducingFactor::reduce (Horner from the back); returns (value, count)"""
    acc = F.zero
    for x in reversed(xs):
        acc = F.eadd(F.emul(alpha, acc), x)
    return acc, len(xs)


def interpolate_eval(points, x, F=GL):
    """Lagrange interpolation through `points` evaluated at x (== barycentric form of
    field/src/interpolation.rs used by fri/verifier.rs:23-49; exact arithmetic, same value)."""
    total = F.zero
    for i, (xi, yi) in enumerate(points):
        num, den = F.one, F.one
        for j, (xj, _) in enumerate(points):
            if i != j:
                num = F.emul(num, F.esub(x, xj))
                den = F.emul(den, F.esub(xi, xj))
        total = F.eadd(total, F.emul(yi, F.ediv(num, den)))
    return total


def compute_evaluation(x, x_index_within_coset, arity_bits, evals, beta, F=GL):
    """fri/verifier.rs:23-49"""
    arity = 1 << arity_bits
    g = F.two_adic_generator(arity_bits)
    ev = [evals[reverse_bits(i, arity_bits)] for i in range(arity)]
    rev = reverse_bits(x_index_within_coset, arity_bits)
    coset_start = x * pow(g, arity - rev, F.P) % F.P
    pts = [(F.efrom(coset_start * pow(g, i, F.P) % F.P), ev[i]) for i in range(arity)]
    return interpolate_eval(pts, beta, F)


def pow_ok(resp, bits, F=GL):
    """fri/verifier.rs:51-65: leading_zeros(u64) >= pow_bits + (64 - order.bits())"""
    lz = 64 - int(resp).bit_length()
    return lz >= bits + (64 - F.order_bits)


def verify_fri(pr, challenges, initial_caps, cd, stats=None, F=GL):
    """fri/verifier.rs:67-250.  Raises AssertionError on any failed check."""
    cfg, fp = cd["config"], cd["fri_params"]
    P_ = F.P
    fri = pr["opening_proof"]
    log_n = fp["degree_bits"] + cfg["fri_config"]["rate_bits"]
    assert pow_ok(challenges["fri_pow_response"], cfg["fri_config"]["proof_of_work_bits"], F), "Invalid proof of work witness."
    assert len(fri["query_round_proofs"]) == cfg["fri_config"]["num_query_rounds"]
    blinding, batches = fri_instance(cd, challenges["plonk_zeta"], F)
    alpha = challenges["fri_alpha"]
    openings = fri_openings(pr["openings"])
    reduced_openings = [reduce_with_alpha(alpha, b, F)[0] for b in openings]
    n_paths = 0
    for x_index, rp in zip(challenges["fri_query_indices"], fri["query_round_proofs"]):
        for (vals, path), cap in zip(rp["initial_trees_proof"], initial_caps):
            assert F.merkle_verify(vals, x_index, cap, path), "initial Merkle path"
            n_paths += 1
        subgroup_x = F.generator * pow(F.two_adic_generator(log_n), reverse_bits(x_index, log_n), P_) % P_
        # fri_combine_initial (fri/verifier.rs:121-165)
        total, count = F.zero, 0
        for (point, polys), red_open in zip(batches, reduced_openings):
            evs = []
            for (oi, pi) in polys:
                vals = rp["initial_trees_proof"][oi][0]
                salted = fp["hiding"] and blinding[oi]
                unsalted = vals[: len(vals) - (SALT_SIZE if salted else 0)]
                evs.append(F.efrom(unsalted[pi]))
            red, count = reduce_with_alpha(alpha, evs, F)
            total = F.emul(F.epow(alpha, count), total)  # alpha.shift(sum): count of THIS batch's reduce
            total = F.eadd(total, F.ediv(F.esub(red, red_open), F.esub(F.efrom(subgroup_x), point)))
        old_eval = total
        xi = x_index
        for i, ab in enumerate(fp["reduction_arity_bits"]):
            evals, path = rp["steps"][i]
            coset_index, within = xi >> ab, xi & ((1 << ab) - 1)
            assert evals[within] == old_eval, "FRI consistency (layer %d)" % i
            old_eval = compute_evaluation(subgroup_x, within, ab, evals, challenges["fri_betas"][i], F)
            flat = [x for e in evals for x in e]
            assert F.merkle_verify(flat, coset_index, fri["commit_phase_merkle_caps"][i], path), "FRI layer Merkle path"
            n_paths += 1
            subgroup_x = pow(subgroup_x, 1 << ab, P_)
            xi = coset_index
        acc = F.zero
        for cf in reversed(fri["final_poly"]):
            acc = F.eadd(F.emul(acc, F.efrom(subgroup_x)), cf)
        assert acc == old_eval, "Final polynomial evaluation is invalid."
    if stats is not None:
        stats["merkle_paths"] = n_paths
    return True
