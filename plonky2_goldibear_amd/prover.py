"""Host-side mirror of the reference's circuit/prover entry points, over the C ABI.

`CircuitData` stands for what `CircuitBuilder::build()` returns (plonk/circuit_builder.rs:1373-1378):
`prover_only` (constants_sigmas_commitment resident on the GPU, sigmas, circuit_digest) and
`verifier_only` (constants_sigmas_cap, circuit_digest).  `prove(witness, public_inputs)` is
`prove_with_partition_witness` (plonk/prover.rs:160-447) from an already generated
`MatrixWitness.wire_values` matrix to `ProofWithPublicInputs` bytes.
"""
import ctypes as C
import sys
import weakref

import numpy as np

from . import native as N
from .polynomial_batch import _as_columns, _as_input, _dtype, _live_contexts, is_column_list  # noqa: F401

_live_circuits = weakref.WeakSet()


class gb_circuit_config(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in (
        "field", "degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges",
        "max_quotient_degree_factor", "rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds",
        "arity_bits", "final_poly_bits", "num_selectors", "gate_constant", "gate_pi", "zero_knowledge", "num_public_inputs")]


class gb_challenger_state(C.Structure):
    """Challenger<F, H> by value (iop/challenger.rs:18-31); include/goldibear_gpu.h"""
    _fields_ = [("sponge_state", C.c_uint64 * 16), ("input_buffer", C.c_uint64 * 8), ("output_buffer", C.c_uint64 * 8),
                ("input_len", C.c_uint32), ("output_len", C.c_uint32)]


class gb_gate(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("kind", "param", "selector_index", "group_start", "group_end", "param2", "param3")]


class _ProofBytesOps:
    """compress / decompress / verify_compressed over the C ABI (plonk/proof.rs:96-140, 183-265), shared by CircuitData and
    VerifierCircuitData; all three run on the host."""

    def _bytes_call(self, fn, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        out = np.empty(max(1 << 16, 2 * buf.size), dtype=np.uint8)
        n = C.c_size_t()
        N.check(fn(self.handle, buf.ctypes.data, buf.size, out.ctypes.data, out.size, C.byref(n)), getattr(getattr(self, "ctx", None), "handle", None))
        return out[: n.value].tobytes()

    def compress(self, proof_bytes):
        """ProofWithPublicInputs::compress -> CompressedProofWithPublicInputs bytes"""
        return self._bytes_call(self._lib.gb_proof_compress, proof_bytes)

    def decompress(self, compressed_bytes):
        """CompressedProofWithPublicInputs::decompress -> ProofWithPublicInputs bytes"""
        return self._bytes_call(self._lib.gb_proof_decompress, compressed_bytes)

    def verify_compressed(self, compressed_bytes):
        """CompressedProofWithPublicInputs::verify; True, or raises VerifyError / ShapeError"""
        buf = np.frombuffer(bytes(compressed_bytes), dtype=np.uint8)
        N.check(self._lib.gb_verify_compressed(self.handle, buf.ctypes.data, buf.size), getattr(getattr(self, "ctx", None), "handle", None))
        return True


class _FriParamsOps:
    """FriParams.reduction_arity_bits of a circuit object (gb_circuit_set_fri_reduction_arity_bits / its getter)"""

    def set_reduction_arity_bits(self, bits):
        arr = (C.c_uint32 * max(1, len(bits)))(*[int(b) for b in bits])
        N.check(self._lib.gb_circuit_set_fri_reduction_arity_bits(self.handle, arr, len(bits)), getattr(getattr(self, "ctx", None), "handle", None))

    @property
    def reduction_arity_bits(self):
        arr, n = (C.c_uint32 * 32)(), C.c_uint32()
        N.check(self._lib.gb_circuit_fri_reduction_arity_bits(self.handle, arr, C.byref(n)), getattr(getattr(self, "ctx", None), "handle", None))
        return [int(arr[i]) for i in range(n.value)]


class CircuitData(_ProofBytesOps, _FriParamsOps):
    """Defaults are standard_recursion_config_gl (plonk/circuit_data.rs:102-116); `CircuitData.babybear(...)`
    fills in recursion_config_bb_narrow (:131-139)."""

    @classmethod
    def babybear(cls, ctx, degree_bits, constants_sigmas, k_is, *, num_challenges=6, **kw):
        d = dict(num_wires=167, num_routed_wires=41, arity_bits=3, num_challenges=num_challenges, field=N.GB_BABYBEAR)
        d.update(kw)
        return cls(ctx, degree_bits, constants_sigmas, k_is, **d)

    def __init__(self, ctx, degree_bits, constants_sigmas, k_is, *, num_wires=135, num_routed_wires=80, num_constants=2,
                 num_challenges=2, max_quotient_degree_factor=8, rate_bits=3, cap_height=4, proof_of_work_bits=16,
                 num_query_rounds=28, arity_bits=4, final_poly_bits=5, num_selectors=1, gate_constant=1, gate_pi=2,
                 field=N.GB_GOLDILOCKS, gates=None, zero_knowledge=False, num_public_inputs=0, reduction_arity_bits=None, p3_repr=False):
        """`gates` = None: the dummy circuit's gate set, given by the selector values gate_constant / gate_pi
        (gb_circuit_create).  Otherwise CommonCircuitData.gates with selectors_info, one tuple
        (kind, param, selector_index, group_start, group_end) per gate in sorted order (gb_circuit_create_gates; what
        circuit_builder.CircuitBuilder.build() passes); num_constants then counts the constant columns after the selectors."""
        self.ctx, self._lib = ctx, ctx._lib
        self.field, self._dt = field, _dtype(field)
        hout = 4 if field == N.GB_GOLDILOCKS else 8
        self.cfg = gb_circuit_config(field, degree_bits, num_wires, num_routed_wires, num_constants, num_challenges,
                                     max_quotient_degree_factor, rate_bits, cap_height, proof_of_work_bits,
                                     num_query_rounds, arity_bits, final_poly_bits, num_selectors, gate_constant, gate_pi,
                                     1 if zero_knowledge else 0, num_public_inputs)
        cs_cols = is_column_list(constants_sigmas)   # constants_sigmas_vecs as build() holds them: one allocation per column
        ptr, shape, flags, keep = _as_columns(constants_sigmas, field) if cs_cols else _as_input(constants_sigmas, field)
        want = (num_selectors + num_constants + num_routed_wires, 1 << degree_bits)
        if tuple(shape) != want:
            raise N.ShapeError(N.GB_ERR_INVALID, "constants_sigmas must be %r, got %r" % (want, tuple(shape)))
        k = np.ascontiguousarray(k_is, dtype=self._dt)
        if k.shape != (num_routed_wires,):
            raise N.ShapeError(N.GB_ERR_INVALID, "k_is must have num_routed_wires entries")
        if flags == N.GB_INPUT_DEVICE:
            import torch
            kd = torch.from_numpy(k.view(np.int64 if k.itemsize == 8 else np.int32)).to("cuda:%d" % ctx.device)
            kptr, keep2 = kd.data_ptr(), kd
        else:
            kptr, keep2 = k.ctypes.data, k
        h = C.c_void_p()
        if p3_repr:   # constants_sigmas and k_is are the field types' in-memory words (host input)
            flags |= N.GB_INPUT_P3_REPR
        if gates is None:
            fn = self._lib.gb_circuit_create_cols if cs_cols else self._lib.gb_circuit_create
            N.check(fn(ctx.handle, C.byref(self.cfg), ptr, kptr, flags, C.byref(h)), ctx.handle)
        else:
            arr = (gb_gate * len(gates))(*[gb_gate(*g) for g in gates])
            fn = self._lib.gb_circuit_create_gates_cols if cs_cols else self._lib.gb_circuit_create_gates
            N.check(fn(ctx.handle, C.byref(self.cfg), arr, len(gates), ptr, kptr, flags, C.byref(h)), ctx.handle)
        del keep, keep2
        self.handle = h
        cap = np.empty((1 << cap_height, hout), dtype=self._dt)
        dig = np.empty(hout, dtype=self._dt)
        N.check(self._lib.gb_circuit_verifier_data(h, cap.ctypes.data, dig.ctypes.data), ctx.handle)
        self.constants_sigmas_cap, self.circuit_digest = cap, dig
        self._proof_buf = None
        _live_circuits.add(self)
        if reduction_arity_bits is not None:   # FriReductionStrategy::Fixed / MinSize (fri_params.py); None = ConstantArityBits
            self.set_reduction_arity_bits(reduction_arity_bits)

    MAX_PERM_ARG_RETRIES = 3  # plonk/prover.rs:183

    def prove(self, witness, public_inputs=(), random_wire=None, rng=None, salts=None, p3_repr=False):
        """prove_with_partition_witness (plonk/prover.rs:160-226): the retry loop around the proof proper.  When the
        permutation argument hits a zero denominator (ProverError::InvZeroPermArg - with a 31-bit field and 2^20 rows
        about one proof in five) the reference overwrites `random_wire` (circuit_builder.rs:1073-1075: the last wire of the
        PublicInputGate row, given here as (column, row)) with a fresh F::rand() and tries again, at most 3 attempts.
        `witness` is modified in place in that case, like the reference's `witness.wire_values`.  `witness`: the [num_wires][n]
        matrix, or MatrixWitness.wire_values as the reference holds it - a list of num_wires separately allocated columns
        (gb_prove_cols).  p3_repr: host elements are the reference's in-memory words (GB_INPUT_P3_REPR)."""
        self.perm_arg_retries = 0
        for attempt in range(self.MAX_PERM_ARG_RETRIES):
            if attempt > 0:
                if random_wire is None:
                    raise N.TooManyPermArgFailuresError(N.GB_ERR_PERM_ARG_ZERO, "Permutation argument division by zero but no "
                                                        "random wire was given to randomize the witness")
                rng = rng or np.random.default_rng()
                col, row = random_wire
                p = 0xFFFFFFFF00000001 if self.field == N.GB_GOLDILOCKS else 2013265921
                val = int(rng.integers(0, p, dtype=np.uint64))
                if p3_repr and self.field == N.GB_BABYBEAR:
                    val = (val << 32) % p   # the Montgomery word of p3's BabyBear
                if is_column_list(witness) and isinstance(witness[col], np.ndarray):
                    witness[col][row] = val
                elif isinstance(witness, np.ndarray):
                    witness[col, row] = val
                else:  # torch device tensor(s) holding the same-width integer bit pattern
                    t = witness[col] if is_column_list(witness) else witness
                    sval = val - (1 << (8 * t.element_size())) if val >> (8 * t.element_size() - 1) else val
                    if is_column_list(witness):
                        t[row] = sval
                    else:
                        t[col, row] = sval
                    # that write is on torch's current stream; the library reads the column on its own stream: order them
                    import torch
                    torch.cuda.current_stream(t.device).synchronize()
                self.perm_arg_retries = attempt
            try:
                # the second and third attempts differ from the failed one in the random wire only: gb_prove_retry rebuilds just
                # that column of the wires commitment where the library kept the rest (no salts)
                retry = (col, row) if attempt > 0 and salts is None else None
                return self.prove_once(witness, public_inputs, salts, retry_wire=retry, p3_repr=p3_repr)
            except N.PermArgZeroError:
                if random_wire is None:   # no retry will follow: do not keep the failed attempt's commitment on the device
                    self.drop_retry()
                continue
        self.drop_retry()   # giving up: the failed attempt's wires commitment and witness copy (~12 GB at 2^20 rows) go back
        raise N.TooManyPermArgFailuresError(N.GB_ERR_PERM_ARG_ZERO, "ProverError::TooManyPermArgFailures")

    def drop_retry(self):
        """release what an attempt that ended in PermArgZeroError keeps on the device for the incremental retry"""
        if self.handle:
            N.check(self._lib.gb_circuit_drop_retry(self.handle), self.ctx.handle)

    def arm_perm_arg_failure(self):
        """TEST HOOK (include/goldibear_gpu_test_hooks.h): the next prove_once raises PermArgZeroError once its Z computation is
        done, keeping what gb_prove_retry builds on - the only way to drive the retry path of a 64-bit field"""
        N.check(self._lib.gb_test_arm_perm_arg_failure(self.handle), self.ctx.handle)

    def prove_once(self, witness, public_inputs=(), salts=None, retry_wire=None, p3_repr=False):
        """internal_prove_with_partition_witness (plonk/prover.rs:228-447); raises PermArgZeroError.  A circuit created with
        zero_knowledge=True takes `salts`: [3][4][N] canonical elements (the F::rand_vec columns of the wires / Zs / quotient
        commitments, fri/oracle.rs:144-148), in the same memory space as the witness."""
        cols = is_column_list(witness)
        ptr, shape, flags, keep = _as_columns(witness, self.field) if cols else _as_input(witness, self.field)
        lib, sfx = self._lib, "_cols" if cols else ""
        if p3_repr:
            flags |= N.GB_INPUT_P3_REPR
        want = (self.cfg.num_wires, 1 << self.cfg.degree_bits)
        if tuple(shape) != want:
            raise N.ShapeError(N.GB_ERR_INVALID, "witness must be %r, got %r" % (want, tuple(shape)))
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if self._proof_buf is None:
            self._proof_buf = np.empty(8 << 20, dtype=np.uint8)
        n = C.c_size_t()
        pis_ptr = pis.ctypes.data if pis.size else None
        if salts is None and retry_wire is not None:   # (wire, row): the one element re-drawn since the attempt that failed
            st = getattr(lib, "gb_prove_retry" + sfx)(self.handle, ptr, flags, int(retry_wire[0]), int(retry_wire[1]), pis_ptr, pis.size,
                                                      self._proof_buf.ctypes.data, self._proof_buf.size, C.byref(n))
        elif salts is None:
            st = getattr(lib, "gb_prove" + sfx)(self.handle, ptr, flags, pis_ptr, pis.size, self._proof_buf.ctypes.data,
                                                self._proof_buf.size, C.byref(n))
        else:
            sptr, sshape, sflags, skeep = _as_input(np.reshape(salts, (3 * N.GB_SALT_SIZE, -1)) if isinstance(salts, np.ndarray)
                                                    else salts.reshape(3 * N.GB_SALT_SIZE, -1), self.field)
            if tuple(sshape) != (3 * N.GB_SALT_SIZE, 1 << (self.cfg.degree_bits + self.cfg.rate_bits)) or sflags != (flags & N.GB_INPUT_DEVICE):
                raise N.ShapeError(N.GB_ERR_INVALID, "salts must be [3][4][N] in the same memory space as the witness")
            st = getattr(lib, "gb_prove_salted" + sfx)(self.handle, ptr, flags, pis_ptr, pis.size, sptr, self._proof_buf.ctypes.data,
                                                       self._proof_buf.size, C.byref(n))
            del skeep
        N.check(st, self.ctx.handle)
        del keep
        return self._proof_buf[: n.value].tobytes()

    @property
    def constants_sigmas_commitment(self):
        """ProverOnlyCircuitData.constants_sigmas_commitment (plonk/circuit_data.rs:532-534): a PolynomialBatch view that the
        circuit keeps owning."""
        from .polynomial_batch import PolynomialBatch
        h = C.c_void_p()
        N.check(self._lib.gb_circuit_constants_sigmas_commitment(self.handle, C.byref(h)), self.ctx.handle)
        return PolynomialBatch(self.ctx, h, borrowed=True)

    # ---- the stages of prove() one at a time, for a host that keeps the reference's prover loop and Challenger
    def _nzs(self):
        c = self.cfg
        return c.num_challenges * (-(-c.num_routed_wires // c.max_quotient_degree_factor))

    def _challenges(self, xs):
        a = np.ascontiguousarray(xs, dtype=self._dt)
        if a.shape != (self.cfg.num_challenges,):
            raise N.ShapeError(N.GB_ERR_INVALID, "expected num_challenges field elements")
        return a

    def zs_partial_products(self, witness, betas, gammas, p3_repr=False):
        """wires_permutation_partial_products_and_zs for every challenge (plonk/prover.rs:305-329, 449-546) ->
        [num_challenges * (1 + num_partial_products)][n] values, Zs first; host array for a host witness, device tensor for a
        device one.  Raises PermArgZeroError (InvZeroPermArg)."""
        cols = is_column_list(witness)
        ptr, shape, flags, keep = _as_columns(witness, self.field) if cols else _as_input(witness, self.field)
        n = 1 << self.cfg.degree_bits
        if tuple(shape) != (self.cfg.num_wires, n):
            raise N.ShapeError(N.GB_ERR_INVALID, "witness must be [num_wires][n]")
        b, g = self._challenges(betas), self._challenges(gammas)
        dev = flags == N.GB_INPUT_DEVICE
        if p3_repr:
            flags |= N.GB_INPUT_P3_REPR
        if dev:
            import torch
            w0 = witness[0] if cols else witness
            out = torch.empty((self._nzs(), n), dtype=w0.dtype, device=w0.device)
            optr = out.data_ptr()
        else:
            out = np.empty((self._nzs(), n), dtype=self._dt)
            optr = out.ctypes.data
        fn = self._lib.gb_zs_partial_products_cols if cols else self._lib.gb_zs_partial_products
        N.check(fn(self.handle, ptr, flags, b.ctypes.data, g.ctypes.data, optr), self.ctx.handle)
        del keep
        return out

    def quotient_polys(self, wires, zs_partial_products, public_inputs_hash, betas, gammas, alphas):
        """compute_quotient_polys + the split into chunks (plonk/prover.rs:345-376, 712-926): `wires` / `zs_partial_products` are
        the PolynomialBatch commitments of this proof -> [num_challenges * quotient_degree_factor][n] coefficients (host)."""
        hout = 4 if self.field == N.GB_GOLDILOCKS else 8
        ph = np.ascontiguousarray(public_inputs_hash, dtype=self._dt)
        if ph.shape != (hout,):
            raise N.ShapeError(N.GB_ERR_INVALID, "public_inputs_hash must have NUM_HASH_OUT_ELTS elements")
        b, g, a = self._challenges(betas), self._challenges(gammas), self._challenges(alphas)
        out = np.empty((self.cfg.num_challenges * self.cfg.max_quotient_degree_factor, 1 << self.cfg.degree_bits), dtype=self._dt)
        N.check(self._lib.gb_quotient_polys(self.handle, wires.handle, zs_partial_products.handle, ph.ctypes.data, b.ctypes.data,
                                            g.ctypes.data, a.ctypes.data, N.GB_INPUT_HOST, out.ctypes.data), self.ctx.handle)
        return out

    def prove_openings(self, wires, zs_partial_products, quotient, zeta, challenger, out_cap=None):
        """PolynomialBatch::prove_openings on this circuit's FRI instance (fri/oracle.rs:187-246, plonk/prover.rs:422-437).
        `challenger` = (sponge_state, input_buffer, output_buffer) after observe_openings, canonical ints.
        -> (FriProof bytes, challenger afterwards in the same form).  `out_cap` (tests): capacity handed to the library instead
        of the wrapper's own buffer; when it is too small the call raises (status GB_ERR_BUFFER_TOO_SMALL) after recording the
        size the library asked for in `last_fri_proof_len` and the challenger it handed back in `last_challenger`."""
        d = 2 if self.field == N.GB_GOLDILOCKS else 4
        z = np.ascontiguousarray(zeta, dtype=self._dt)
        if z.shape != (d,):
            raise N.ShapeError(N.GB_ERR_INVALID, "zeta must have %d coordinates" % d)
        st, inp, outb = challenger
        w = 12 if self.field == N.GB_GOLDILOCKS else 16
        if len(st) != w or len(inp) > 8 or len(outb) > 8:
            raise N.ShapeError(N.GB_ERR_INVALID, "challenger state has the wrong shape")
        cs = gb_challenger_state()
        for i, v in enumerate(st):
            cs.sponge_state[i] = int(v)
        for i, v in enumerate(inp):
            cs.input_buffer[i] = int(v)
        for i, v in enumerate(outb):
            cs.output_buffer[i] = int(v)
        cs.input_len, cs.output_len = len(inp), len(outb)
        if self._proof_buf is None:
            self._proof_buf = np.empty(8 << 20, dtype=np.uint8)
        n = C.c_size_t()
        cap = self._proof_buf.size if out_cap is None else min(int(out_cap), self._proof_buf.size)
        st = self._lib.gb_prove_openings(self.handle, wires.handle, zs_partial_products.handle, quotient.handle, z.ctypes.data,
                                         C.byref(cs), self._proof_buf.ctypes.data if cap else None, cap, C.byref(n))
        after = ([int(cs.sponge_state[i]) for i in range(w)], [int(cs.input_buffer[i]) for i in range(cs.input_len)],
                 [int(cs.output_buffer[i]) for i in range(cs.output_len)])
        self.last_fri_proof_len, self.last_challenger = n.value, after
        N.check(st, self.ctx.handle)
        return self._proof_buf[: n.value].tobytes(), after

    def verify(self, proof_bytes):
        """CircuitData::verify (plonk/circuit_data.rs:290-300 -> plonk/verifier.rs:17-128): True, or raises VerifyError naming
        the failed check (ShapeError for malformed bytes).  Runs on the host, like the reference's verifier."""
        buf = np.frombuffer(bytes(proof_bytes), dtype=np.uint8)
        N.check(self._lib.gb_verify(self.handle, buf.ctypes.data, buf.size), self.ctx.handle)
        return True

    def free(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self._lib.gb_circuit_free(self.handle)
        self.handle = None

    def __del__(self):
        if not sys.is_finalizing():
            self.free()


class VerifierCircuitData(_ProofBytesOps, _FriParamsOps):
    """VerifierCircuitData (plonk/circuit_data.rs:358-380): CommonCircuitData + VerifierOnlyCircuitData, enough to verify and
    nothing else.  Built on gb_verifier_create, which touches no device - usable without a GPU context."""

    def __init__(self, degree_bits, gates, k_is, constants_sigmas_cap, circuit_digest, *, num_wires=135, num_routed_wires=80,
                 num_constants=2, num_challenges=2, max_quotient_degree_factor=8, rate_bits=3, cap_height=4,
                 proof_of_work_bits=16, num_query_rounds=28, arity_bits=4, final_poly_bits=5, num_selectors=1,
                 zero_knowledge=False, field=N.GB_GOLDILOCKS, num_public_inputs=0, reduction_arity_bits=None):
        """`gates`: (kind, param, selector_index, group_start, group_end[, param2, param3]) per gate, sorted as in
        CommonCircuitData.gates; num_constants counts the constant columns after the selectors; reduction_arity_bits:
        FriParams.reduction_arity_bits when the strategy is not ConstantArityBits(arity_bits, final_poly_bits)."""
        self._lib = N.load()
        self.field, self._dt = field, _dtype(field)
        hout = 4 if field == N.GB_GOLDILOCKS else 8
        self.cfg = gb_circuit_config(field, degree_bits, num_wires, num_routed_wires, num_constants, num_challenges,
                                     max_quotient_degree_factor, rate_bits, cap_height, proof_of_work_bits, num_query_rounds,
                                     arity_bits, final_poly_bits, num_selectors, 0, 0, 1 if zero_knowledge else 0,
                                     num_public_inputs)
        k = np.ascontiguousarray(k_is, dtype=self._dt)
        cap = np.ascontiguousarray(constants_sigmas_cap, dtype=self._dt)
        dig = np.ascontiguousarray(circuit_digest, dtype=self._dt)
        if k.shape != (num_routed_wires,) or cap.shape != (1 << cap_height, hout) or dig.shape != (hout,):
            raise N.ShapeError(N.GB_ERR_INVALID, "k_is / constants_sigmas_cap / circuit_digest have the wrong shape")
        arr = (gb_gate * len(gates))(*[gb_gate(*g) for g in gates])
        h = C.c_void_p()
        N.check(self._lib.gb_verifier_create(None, C.byref(self.cfg), arr, len(gates), k.ctypes.data, cap.ctypes.data,
                                             dig.ctypes.data, C.byref(h)))
        self.handle = h
        if reduction_arity_bits is not None:
            self.set_reduction_arity_bits(reduction_arity_bits)

    def verify(self, proof_bytes):
        """plonk/verifier.rs:17-128; True, or raises VerifyError naming the failed check."""
        buf = np.frombuffer(bytes(proof_bytes), dtype=np.uint8)
        N.check(self._lib.gb_verify(self.handle, buf.ctypes.data, buf.size))
        return True

    def free(self):
        if getattr(self, "handle", None):
            self._lib.gb_circuit_free(self.handle)
        self.handle = None

    def __del__(self):
        if not sys.is_finalizing():
            self.free()
