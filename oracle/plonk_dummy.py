"""TEST ORACLE - restatement of the dummy circuit (bench form) and of the PLONK verifier identity.

Test infrastructure only.  Follows (paths relative to /root/reference/plonky2):
  examples/bench_recursion.rs:87-122   dummy_proof: 2^(k-1)+1 NoopGates -> 2^k rows
  src/plonk/circuit_builder.rs:1110-1312  build(): PublicInputGate row, ConstantGate row for the
       `zero` target (src/plonk/config.rs:135-166 with no public inputs), padding, selectors
       (src/gates/selectors.rs:125-159 single-group case), constant_polys (:982-1003),
       k_is (field/src/cosets.rs:8-21), sigma (src/plonk/permutation_argument.rs:108-157),
       circuit_digest (:1300-1312, hash_pad src/plonk/config.rs:58-66)
  src/iop/witness.rs:359-371           full_witness: unset wires are zero
  src/plonk/verifier.rs:17-128, src/plonk/vanishing_poly.rs:40-170   verify()
The gate set is exactly {NoopGate, ConstantGate{2}, PublicInputGate<4>} sorted by (degree, id).
"""
import ctypes as C

import numpy as np

from . import gates as G
from . import oracle as O
from . import verifier as V
from .fields import BB, GL

P = O.GL_P  # test-form digest KAT below is Goldilocks only


class CircuitConfig:
    """standard_recursion_config_gl (src/plonk/circuit_data.rs:102-159) with num_challenges overridable
    (2^16 / 2^20 rows need num_challenges = 3)."""

    def __init__(self, num_challenges=2, num_wires=135, num_routed_wires=80, num_constants=2, rate_bits=3, cap_height=4,
                 proof_of_work_bits=16, num_query_rounds=28, arity_bits=4, final_poly_bits=5, max_quotient_degree_factor=8,
                 security_bits=100):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @classmethod
    def babybear(cls, num_challenges=6, **kw):
        """recursion_config_bb_narrow (circuit_data.rs:130-138): 167 wires, 41 routed, arity 8, 6 challenges
        (31-bit field: (31 - degree_bits) * num_challenges >= 100 needs 10 at 2^20 rows)."""
        d = dict(num_challenges=num_challenges, num_wires=167, num_routed_wires=41, arity_bits=3)
        d.update(kw)
        return cls(**d)


def reduction_arity_bits(cfg, degree_bits):
    """fri/reduction_strategies.rs:39-50 ConstantArityBits, with its assert (the reference panics on an arity that does not fit
    what is left of the degree)"""
    out, db = [], degree_bits
    while db > cfg.final_poly_bits:
        assert db + cfg.rate_bits >= cfg.arity_bits, "usize underflow of degree_bits + rate_bits - arity_bits (:42)"
        if db + cfg.rate_bits - cfg.arity_bits < cfg.cap_height:
            break
        out.append(cfg.arity_bits)
        assert db >= cfg.arity_bits, "ConstantArityBits: degree_bits >= arity_bits (fri/reduction_strategies.rs:45)"
        db -= cfg.arity_bits
    return out


class DummyCircuit:
    """CircuitData of the bench-form dummy circuit with 2^degree_bits rows (degree_bits >= 3)."""

    zero_knowledge = False  # CircuitConfig.zero_knowledge (= FriParams.hiding); set on an instance to get salted commitments
    GATE_NOOP, GATE_CONSTANT, GATE_PI = 0, 1, 2  # sorted by (degree, id): Noop(0), "ConstantGate {..}"(1), "PublicInputGate<4>"(1)

    def __init__(self, degree_bits, cfg=None, check_security=True, F=GL):
        cfg = cfg or (CircuitConfig() if F is GL else CircuitConfig.babybear())
        assert degree_bits >= 3
        if check_security:  # circuit_builder.rs:1187-1192
            assert (F.order_bits - degree_bits) * cfg.num_challenges >= cfg.security_bits, "num_challenges too small for this degree"
        self.cfg, self.degree_bits, self.F = cfg, degree_bits, F
        P, dt, M, H = F.P, F.dtype, F.mod, F.hout
        n = 1 << degree_bits
        self.n = n
        num_noops = (1 << (degree_bits - 1)) + 1
        self.pi_row, self.const_row = num_noops, num_noops + 1
        assert self.const_row < n
        # selector column: gate index per row; constants: all zero (the only constant is 0)
        sel = np.zeros(n, dtype=dt)
        sel[self.pi_row] = self.GATE_PI
        sel[self.const_row] = self.GATE_CONSTANT
        consts = np.zeros((cfg.num_constants, n), dtype=dt)
        self.num_constants = 1 + cfg.num_constants
        self.k_is = np.array([pow(F.generator, i, P) for i in range(cfg.num_routed_wires)], dtype=dt)
        w = F.two_adic_generator(degree_bits)
        self.subgroup = sub = M.powers(w, n)
        # sigma: identity except the single copy class {(pi,0..H-1), (const,0)} in (row, column) order
        sig = np.empty((cfg.num_routed_wires, n), dtype=dt)
        for j in range(cfg.num_routed_wires):
            sig[j] = M.scale_vec(sub, int(self.k_is[j]))
        cls = [(self.pi_row, j) for j in range(H)] + [(self.const_row, 0)]
        for t, (row, col) in enumerate(cls):
            nrow, ncol = cls[(t + 1) % len(cls)]
            sig[col, row] = int(self.k_is[ncol]) * int(sub[nrow]) % P
        self.constants_sigmas = np.concatenate([sel[None, :], consts, sig]).astype(dt)
        self.sigma = sig
        self.num_partial_products = -(-cfg.num_routed_wires // cfg.max_quotient_degree_factor) - 1
        self.reduction_arity_bits = reduction_arity_bits(cfg, degree_bits)
        self._digest = None
        self.constants_sigmas_cap = None
        # CommonCircuitData.gates / selectors_info in table form (oracle/gates.py)
        self.num_selectors, self.num_public_inputs = 1, 0
        self.gate_table = [(G.NOOP, 0, 0, 0, 3), (G.CONSTANT, cfg.num_constants, 0, 0, 3), (G.PUBLIC_INPUT, H, 0, 0, 3)]

    @classmethod
    def verifier_view(cls, degree_bits, cfg, F, k_is):
        """Only what verify() reads (CommonCircuitData of the dummy gate set), without materialising the
        constants/sigmas columns; call set_cap() with the prover side's cap afterwards."""
        self = cls.__new__(cls)
        H = F.hout
        self.cfg, self.degree_bits, self.F, self.n = cfg, degree_bits, F, 1 << degree_bits
        self.k_is = np.asarray(k_is, dtype=F.dtype)
        self.num_constants = 1 + cfg.num_constants
        self.num_partial_products = -(-cfg.num_routed_wires // cfg.max_quotient_degree_factor) - 1
        self.reduction_arity_bits = reduction_arity_bits(cfg, degree_bits)
        self._digest, self.constants_sigmas_cap = None, None
        self.num_selectors, self.num_public_inputs = 1, 0
        self.gate_table = [(G.NOOP, 0, 0, 0, 3), (G.CONSTANT, cfg.num_constants, 0, 0, 3), (G.PUBLIC_INPUT, H, 0, 0, 3)]
        return self

    def set_cap(self, cap):
        """circuit_digest = hash_no_pad(cap.flatten() ++ hash_pad(domain_sep = []) ++ [degree_bits])"""
        F = self.F
        cap = np.asarray(cap, dtype=F.dtype)
        self.constants_sigmas_cap = cap
        dom = F.hash_no_pad(np.array([1, 0, 0, 0, 0, 0, 0, 1], dtype=F.dtype))  # hash_pad([]) (config.rs:58-66), rate 8
        parts = np.concatenate([cap.ravel(), dom, np.array([self.degree_bits], dtype=F.dtype)])
        self._digest = F.hash_no_pad(parts)
        return self._digest

    @property
    def circuit_digest(self):
        if self._digest is None:
            b = self.F.mod.PolynomialBatch.from_values(self.constants_sigmas, self.cfg.rate_bits, self.cfg.cap_height)
            self.set_cap(b.cap)
        return self._digest

    def witness(self, seed=0):
        """MatrixWitness wire_values[column][row]: zeros except the PublicInputGate row's wires H..
        (circuit_builder.rs:1065-1080), filled from SplitMix64 (stands in for RandomValueGenerator's F::rand(),
        SURVEY.md 8(d))."""
        cfg, H = self.cfg, self.F.hout
        w = np.zeros((cfg.num_wires, self.n), dtype=self.F.dtype)
        w[H:, self.pi_row] = self.F.fill(0x9E3779B97F4A7C15 + seed, cfg.num_wires - H)
        return w

    def common_data(self):
        """The dict shape oracle/verifier.py uses."""
        cfg = self.cfg
        fri_cfg = dict(rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, num_query_rounds=cfg.num_query_rounds,
                       proof_of_work_bits=cfg.proof_of_work_bits)
        return dict(
            config=dict(num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires, num_constants=cfg.num_constants,
                        num_challenges=cfg.num_challenges, fri_config=fri_cfg, zero_knowledge=self.zero_knowledge),
            fri_params=dict(config=fri_cfg, reduction_arity_bits=self.reduction_arity_bits, degree_bits=self.degree_bits,
                            hiding=self.zero_knowledge),
            quotient_degree_factor=cfg.max_quotient_degree_factor, num_constants=self.num_constants,
            num_partial_products=self.num_partial_products, num_lookup_polys=0, k_is=[int(k) for k in self.k_is],
            num_public_inputs=self.num_public_inputs)

    def c_cfg(self):
        cfg = self.cfg
        vals = [cfg.num_wires, cfg.num_routed_wires, self.num_constants, cfg.num_challenges, cfg.rate_bits, cfg.cap_height,
                cfg.proof_of_work_bits, cfg.num_query_rounds, cfg.arity_bits, cfg.final_poly_bits,
                cfg.max_quotient_degree_factor, self.degree_bits, self.num_selectors, self.GATE_NOOP, self.GATE_CONSTANT,
                self.GATE_PI, cfg.num_constants, min(16, len(self.gate_table))]
        for g in self.gate_table[:16]:   # beyond 16 kinds the gate terms come from gate_constraint_terms() anyway
            vals += list(g[:5])   # the C oracle prover knows the kinds without a second parameter
        vals += [0] * (5 * (16 - min(16, len(self.gate_table))))
        return (C.c_uint * len(vals))(*vals)


class BuiltCircuit(DummyCircuit):
    """CircuitData of any circuit given by its build() outputs - constants||sigmas values, k_is and the sorted gate set with
    selectors_info as (kind, param, selector_index, group_start, group_end) tuples - for the gate kinds of oracle/gates.py."""

    def __init__(self, cfg, F, degree_bits, constants_sigmas, k_is, gate_table, num_selectors, num_public_inputs):
        self.cfg, self.degree_bits, self.F, self.n = cfg, degree_bits, F, 1 << degree_bits
        self.constants_sigmas = np.ascontiguousarray(constants_sigmas, dtype=F.dtype)
        self.k_is = np.ascontiguousarray(k_is, dtype=F.dtype)
        self.gate_table, self.num_selectors, self.num_public_inputs = [tuple(g) for g in gate_table], num_selectors, num_public_inputs
        self.num_constants = self.constants_sigmas.shape[0] - cfg.num_routed_wires   # selectors + constant columns
        self.sigma = self.constants_sigmas[self.num_constants:]
        self.num_partial_products = -(-cfg.num_routed_wires // cfg.max_quotient_degree_factor) - 1
        self.reduction_arity_bits = reduction_arity_bits(cfg, degree_bits)
        self._digest, self.constants_sigmas_cap = None, None


class CommonDataCircuit(DummyCircuit):
    """The verifier's view of a circuit given as serialized CommonCircuitData + VerifierOnlyCircuitData
    (util/serialization/mod.rs:1862-1916, 1740-1760) - e.g. the reference's recursion regression fixture."""

    def __init__(self, common_bytes, verifier_bytes, F=GL):
        cd = V.read_common_data(common_bytes, F)
        vd = V.read_verifier_data(verifier_bytes, F)
        c, fc = cd["config"], cd["config"]["fri_config"]
        self._cd = cd
        self.F, self.degree_bits = F, cd["fri_params"]["degree_bits"]
        self.n = 1 << self.degree_bits
        self.cfg = CircuitConfig(num_wires=c["num_wires"], num_routed_wires=c["num_routed_wires"], num_constants=c["num_constants"],
                                 num_challenges=c["num_challenges"], max_quotient_degree_factor=c["max_quotient_degree_factor"],
                                 rate_bits=fc["rate_bits"], cap_height=fc["cap_height"], num_query_rounds=fc["num_query_rounds"],
                                 proof_of_work_bits=fc["proof_of_work_bits"])
        self.k_is = np.asarray(cd["k_is"], dtype=F.dtype)
        self.num_constants = cd["num_constants"]
        self.num_partial_products = cd["num_partial_products"]
        self.reduction_arity_bits = cd["fri_params"]["reduction_arity_bits"]
        self.num_public_inputs = cd["num_public_inputs"]
        self.gate_table = V.read_gates(common_bytes, cd, F)
        self.num_selectors = len(cd["selectors_info"]["groups"])
        assert cd["num_gate_constraints"] == max(G.num_constraints(g, F.hout, F.D) for g in self.gate_table)
        assert cd["quotient_degree_factor"] == self.cfg.max_quotient_degree_factor
        self.constants_sigmas_cap = np.asarray(vd["constants_sigmas_cap"], dtype=F.dtype)
        self._digest = np.asarray(vd["circuit_digest"], dtype=F.dtype)

    def common_data(self):
        return self._cd


def gate_constraint_terms(circ, witness, public_inputs=()):
    """The gate part of eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:741-774) on the prover's quotient domain (every
    step-th point of the commitments' LDE, step = 2^(rate_bits - log2 quotient_degree_factor), prover.rs:735-749), for any
    gate set of oracle/gates.py: out[i][j] = sum over the gates of compute_filter(selector)(x_i) * unfiltered constraint j at
    x_i = shift * w_Q^i, Q = n * quotient_degree_factor points, num_gate_constraints columns.  The wires and constants at x_i are base-field values; gates.py's
    evaluators work on the extension algebra, in which a base value v is (v, 0, ..) and every constraint comes out as (c, 0, ..).
    Pure Python: meant for circuits of 2^6..2^8 rows."""
    e, cfg = circ.F, circ.cfg
    r, H, nsel = cfg.rate_bits, e.hout, circ.num_selectors
    qb = cfg.max_quotient_degree_factor.bit_length() - 1
    assert cfg.max_quotient_degree_factor == 1 << qb and qb <= r
    N, step = circ.n << qb, 1 << (r - qb)
    pi_hash = [int(x) for x in e.hash_no_pad(np.asarray(list(public_inputs), dtype=e.dtype))]
    cs = e.mod.PolynomialBatch.from_values(np.ascontiguousarray(circ.constants_sigmas[:circ.num_constants]), r, cfg.cap_height)
    wb = e.mod.PolynomialBatch.from_values(np.ascontiguousarray(witness, dtype=e.dtype), r, cfg.cap_height)
    ngc = max(G.num_constraints(g, H, e.D) for g in circ.gate_table)
    out = np.zeros((N, ngc), dtype=e.dtype)
    pad = (0,) * (e.D - 1)
    for i in range(N):
        consts = [(int(v),) + pad for v in cs.get_lde_values(i, step)[:circ.num_constants]]
        wires = [(int(v),) + pad for v in wb.get_lde_values(i, step)[:cfg.num_wires]]
        acc = [0] * ngc
        for row, g in enumerate(circ.gate_table):
            f = G.compute_filter(e, row, g, consts[g[2]], nsel > 1)
            assert f[1:] == pad
            if f[0] == 0:
                continue
            for j, cj in enumerate(G.eval_unfiltered(e, g, wires, consts[nsel:], pi_hash)):
                assert cj[1:] == pad, "a constraint on base-field wires is a base-field value"
                acc[j] = (acc[j] + f[0] * cj[0]) % e.P
        out[i] = acc
    return out


def prove_cpu(circ, witness, public_inputs=(), salts=None, dump=None):
    """Run the CPU oracle prover; returns (proof_bytes, debug challenges).  `dump`: a dict that receives the prover's own
    intermediates - "zs_partial_products" [c * (1 + num_partial_products)][n] values as handed to from_values (prover.rs:318-329)
    and "quotient_chunks" [c * quotient_degree_factor][n] coefficients as handed to from_coeffs (:361-376).  `salts`: None, or for a zero-knowledge circuit
    (circ.zero_knowledge) the [3][4][N] salt columns of the wires / Zs / quotient commitments in LDE-point order - the values
    the reference draws with F::rand_vec (fri/oracle.rs:144-148), a host input here."""
    L = O.lib()
    F = circ.F
    fn = getattr(L, F.prove_symbol + "_salted")
    fn.restype = C.c_int
    c = circ.cfg.num_challenges
    cs = np.ascontiguousarray(circ.constants_sigmas)
    wit = np.ascontiguousarray(witness, dtype=F.dtype)
    pis = np.ascontiguousarray(list(public_inputs) or [0], dtype=F.dtype)
    dig = np.ascontiguousarray(circ.circuit_digest)
    assert (salts is not None) == bool(getattr(circ, "zero_knowledge", False)), "salts go with zero_knowledge circuits"
    if salts is not None:
        salts = np.ascontiguousarray(salts, dtype=F.dtype)
        assert salts.shape == (3, 4, circ.n << circ.cfg.rate_bits)
    if dump is not None:
        dump["zs_partial_products"] = np.zeros((c * (1 + circ.num_partial_products), circ.n), dtype=F.dtype)
        dump["quotient_chunks"] = np.zeros((c * circ.cfg.max_quotient_degree_factor, circ.n), dtype=F.dtype)
        getattr(L, F.prove_symbol + "_set_dump")(dump["zs_partial_products"].ctypes.data_as(C.c_void_p),
                                                 dump["quotient_chunks"].ctypes.data_as(C.c_void_p))
    cap = 64 << 20
    out = np.zeros(cap, dtype=np.uint8)
    out_len = C.c_size_t()
    dbg = np.zeros(3 * c + 2 * F.D + 1, dtype=F.dtype)  # betas, gammas, alphas, zeta, fri_alpha, pow response
    custom_arities = list(circ.reduction_arity_bits) != reduction_arity_bits(circ.cfg, circ.degree_bits)
    if custom_arities:   # FriReductionStrategy::Fixed / MinSize: the circuit carries its own list (set circ.reduction_arity_bits)
        arr = (C.c_uint * max(1, len(circ.reduction_arity_bits)))(*circ.reduction_arity_bits)
        getattr(L, F.prove_symbol + "_set_reduction_arity_bits")(arr, C.c_uint(len(circ.reduction_arity_bits)))
    gate_terms = None
    if any(g[0] > G.POSEIDON2_BABYBEAR for g in getattr(circ, "gate_table", ())):
        # gates the C prover has no evaluator for (the recursion gate set): their terms come from oracle/gates.py
        gate_terms = np.ascontiguousarray(gate_constraint_terms(circ, wit, public_inputs))
        getattr(L, F.prove_symbol + "_set_gate_terms")(gate_terms.ctypes.data_as(C.c_void_p), C.c_uint(gate_terms.shape[1]))
    try:
        rc = _call_prover(fn, circ, cs, dig, wit, pis, public_inputs, out, cap, out_len, dbg, salts)
    finally:
        if custom_arities:
            getattr(L, F.prove_symbol + "_set_reduction_arity_bits")(None, C.c_uint(0))
        if gate_terms is not None:
            getattr(L, F.prove_symbol + "_set_gate_terms")(None, C.c_uint(0))
        if dump is not None:
            getattr(L, F.prove_symbol + "_set_dump")(None, None)
    if rc != 0:
        raise RuntimeError("oracle prover failed: rc=%d" % rc)
    prove_cpu.last_cs_commit_seconds = C.c_double.in_dll(L, "gbo_last_cs_commit_seconds").value  # build() share of the call
    nb = C.c_size_t.in_dll(L, "gbo_last_cs_cap_bytes").value   # the cap of the constants/sigmas commitment the prover made itself
    raw = (C.c_ubyte * 8192).in_dll(L, "gbo_last_cs_cap")
    prove_cpu.last_cs_cap = np.frombuffer(bytes(raw[:nb]), dtype=F.dtype).reshape(-1, F.hout).copy() if nb else None
    # A cap handed over with set_cap() (the GPU's, in the -m gpu tests) seeds the transcript through circuit_digest: it must be the
    # cap of the commitment this prover made ITSELF from constants_sigmas - otherwise cap and digest would be GPU-vs-GPU.
    if prove_cpu.last_cs_cap is not None and circ.constants_sigmas_cap is not None:
        assert (np.asarray(circ.constants_sigmas_cap) == prove_cpu.last_cs_cap).all(), \
            "constants_sigmas_cap given to set_cap() differs from the oracle prover's own constants/sigmas commitment"
    return out[: out_len.value].tobytes(), dbg


def _call_prover(fn, circ, cs, dig, wit, pis, public_inputs, out, cap, out_len, dbg, salts):
    return fn(circ.c_cfg(), cs.ctypes.data_as(C.c_void_p), dig.ctypes.data_as(C.c_void_p), circ.k_is.ctypes.data_as(C.c_void_p),
            wit.ctypes.data_as(C.c_void_p), pis.ctypes.data_as(C.c_void_p), C.c_size_t(len(public_inputs)),
            out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(out_len), dbg.ctypes.data_as(C.c_void_p),
            salts.ctypes.data_as(C.c_void_p) if salts is not None else None)


# ----------------------------------------------------------------------------- verify()
def eval_vanishing_poly(circ, zeta, openings, pi_hash, betas, gammas, alphas):
    """plonk/vanishing_poly.rs:40-170 for the dummy gate set, at an extension point."""
    cfg = circ.cfg
    e = circ.F
    H = e.hout
    one, zero = e.one, e.zero
    n = circ.n
    consts, wires, sig = openings["constants"], openings["wires"], openings["plonk_sigmas"]
    zs, zs_next, pps = openings["plonk_zs"], openings["plonk_zs_next"], openings["partial_products"]
    qdf, num_prods = cfg.max_quotient_degree_factor, circ.num_partial_products
    # gate constraints: sum over the gate set of filter * unfiltered per constraint index (vanishing_poly.rs:129-170)
    nsel = circ.num_selectors
    cons = [zero] * max(G.num_constraints(g, H, e.D) for g in circ.gate_table)
    for row, g in enumerate(circ.gate_table):
        f = G.compute_filter(e, row, g, consts[g[2]], nsel > 1)
        for j, c in enumerate(G.eval_unfiltered(e, g, wires, consts[nsel:], pi_hash)):
            cons[j] = e.eadd(cons[j], e.emul(f, c))
    # eval_l_0 (plonk_common.rs:56-66)
    zn = e.epow(zeta, n)
    if zeta == one:
        l0 = one
    else:
        l0 = e.ediv(e.esub(zn, one), e.emul(e.efrom(n), e.esub(zeta, one)))
    z1_terms, pp_terms = [], []
    for i in range(cfg.num_challenges):
        z1_terms.append(e.emul(l0, e.esub(zs[i], one)))
        nums = [e.eadd(e.eadd(wires[j], e.emul(e.efrom(betas[i]), e.emul(e.efrom(int(circ.k_is[j])), zeta))), e.efrom(gammas[i]))
                for j in range(cfg.num_routed_wires)]
        dens = [e.eadd(e.eadd(wires[j], e.emul(e.efrom(betas[i]), sig[j])), e.efrom(gammas[i])) for j in range(cfg.num_routed_wires)]
        accs = [zs[i]] + pps[i * num_prods:(i + 1) * num_prods] + [zs_next[i]]
        for m in range(num_prods + 1):
            npd, dpd = one, one
            for j in range(m * qdf, min((m + 1) * qdf, cfg.num_routed_wires)):
                npd, dpd = e.emul(npd, nums[j]), e.emul(dpd, dens[j])
            pp_terms.append(e.esub(e.emul(accs[m], npd), e.emul(accs[m + 1], dpd)))
    terms = z1_terms + pp_terms + cons
    out = []
    for a in alphas:
        cum = zero
        for t in reversed(terms):
            cum = e.eadd(t, e.emul(cum, e.efrom(a)))
        out.append(cum)
    return out


def verify(circ, proof_bytes, stats=None):
    """plonk/verifier.rs:17-128: transcript, vanishing identity at zeta, FRI.  Raises AssertionError."""
    cd = circ.common_data()
    F = circ.F
    proof, pis = V.read_proof_with_pis(proof_bytes, cd, F)
    assert len(pis) == circ.num_public_inputs, "Number of public inputs doesn't match circuit data."
    ch = V.get_challenges(proof, pis, circ.circuit_digest, cd, F)
    pi_hash = F.hash_no_pad(np.asarray(pis, dtype=F.dtype))
    van = eval_vanishing_poly(circ, ch["plonk_zeta"], proof["openings"], pi_hash, ch["plonk_betas"], ch["plonk_gammas"],
                              ch["plonk_alphas"])
    zeta_pow = F.epow(ch["plonk_zeta"], circ.n)
    z_h = F.esub(zeta_pow, F.one)
    q = proof["openings"]["quotient_polys"]
    qdf = circ.cfg.max_quotient_degree_factor
    for i in range(circ.cfg.num_challenges):
        acc = F.zero
        for t in reversed(q[i * qdf:(i + 1) * qdf]):
            acc = F.eadd(F.emul(acc, zeta_pow), t)
        assert van[i] == F.emul(z_h, acc), "vanishing(zeta) != Z_H(zeta) * quotient(zeta) for challenge %d" % i
    caps = [[[int(x) for x in h] for h in circ.constants_sigmas_cap], proof["wires_cap"], proof["zs_cap"], proof["quotient_cap"]]
    V.verify_fri(proof, ch, caps, cd, stats, F)
    return True


# ----------------------------------------------------------------------------- test-form dummy circuit (digest KAT)
def test_form_constants_sigmas(num_noops=16000, cfg=None):
    """constants||sigmas VALUES of the reference's test-form dummy circuit
    (recursion/recursive_verifier.rs:666-697): `num_noops` NoopGates, PoseidonGate added to the gate set, four
    public inputs all equal to the `zero` target.  build() then (circuit_builder.rs:1126-1178):
      row n0   PoseidonGate   in-circuit hash of the 4 public inputs: swap wire (24) and the 12 input wires all
                              tied to the `zero` constant target (hash/poseidon_goldilocks.rs:1116-1143)
      row n0+1 PublicInputGate wires 0..3 tied to the Poseidon outputs (wires 12..15)
      row n0+2 ConstantGate{2} wire 0 tied to `zero`
    gates sorted by (degree, id) = [Noop, Constant, PublicInput, Poseidon]; 7 + 4 - 1 > 9 so two selector
    groups [0..3), [3..4) (gates/selectors.rs:168-206) with UNUSED_SELECTOR = 2^32 - 1 elsewhere.
    Returns (constants_sigmas [4 + routed][n], degree_bits)."""
    cfg = cfg or CircuitConfig()
    rows_used = num_noops + 3
    degree_bits = (rows_used - 1).bit_length()
    n = 1 << degree_bits
    pos_row, pi_row, const_row = num_noops, num_noops + 1, num_noops + 2
    UNUSED = (1 << 32) - 1
    s0 = np.zeros(n, dtype=np.uint64)            # NoopGate = index 0 (padding rows are NoopGates too)
    s0[const_row], s0[pi_row], s0[pos_row] = 1, 2, UNUSED
    s1 = np.full(n, UNUSED, dtype=np.uint64)
    s1[pos_row] = 3
    consts = np.zeros((cfg.num_constants, n), dtype=np.uint64)
    k_is = np.array([pow(7, i, P) for i in range(cfg.num_routed_wires)], dtype=np.uint64)
    sub = O.powers(pow(1753635133440165772, 1 << (32 - degree_bits), P), n)
    sig = np.empty((cfg.num_routed_wires, n), dtype=np.uint64)
    for j in range(cfg.num_routed_wires):
        sig[j] = O.scale_vec(sub, int(k_is[j]))
    # copy classes, members in (row, column) order (permutation_argument.rs:84-101)
    classes = [[(pos_row, c) for c in range(12)] + [(pos_row, 24), (const_row, 0)]]
    classes += [[(pos_row, 12 + i), (pi_row, i)] for i in range(4)]
    for cls in classes:
        for t, (row, col) in enumerate(cls):
            nrow, ncol = cls[(t + 1) % len(cls)]
            sig[col, row] = int(k_is[ncol]) * int(sub[nrow]) % P
    return np.concatenate([s0[None, :], s1[None, :], consts, sig]).astype(np.uint64), degree_bits


def circuit_digest_from_cap(cap, degree_bits):
    """circuit_builder.rs:1300-1312 with the empty domain separator"""
    dom = O.hash_no_pad(np.array([1, 0, 0, 0, 0, 0, 0, 1], dtype=np.uint64))
    parts = np.concatenate([np.asarray(cap, dtype=np.uint64).ravel(), dom, np.array([degree_bits], dtype=np.uint64)])
    return O.hash_no_pad(parts)
