#!/usr/bin/env python3
"""bench.py - one JSON line for the driver (see DESIGN.md "Measurement").

  python bench.py --gpus N --steps K --warmup W        (N>1 from a bare shell: starts its own N rank processes; under
                                                        torch.distributed.run it is one of the ranks)

A "step" is one full prove() (plonk/prover.rs:228-447) of the 2^20-row dummy circuit (BASELINE.json configs[2]:
Goldilocks, standard_recursion_config_gl with num_challenges = 3) from a MatrixWitness to ProofWithPublicInputs
bytes, constants/sigmas commitment pre-resident as after build().

`value` is the proofs/s SURVEY.md 8(d) and BASELINE.md define: the witness is a (page-locked) HOST array handed over
every step, as the drop-in boundary hands it over.  `value_hbm_resident` is the same circuit proved from a witness that
is already resident in HBM when the timed region starts (the other boundary the task statement names); both are
timed the same way (W warm-up steps, barrier, exactly K steps, barrier, max over ranks) in the same run.

Objects in the line:
  roofline      NTT pass (IFFT + LDE kernels of every commitment of a step) against the 8 TB/s HBM peak with
                SURVEY.md 8(d)'s algorithmic bytes; durations from HIP events on the context's stream.
  roofline_alu  the dominant kernel, k_gl_merkle_leaves (Poseidon-12 leaf sponges): VALU wave-instructions per second
                (instructions per permutation from SQ_INSTS_VALU, profiles/r*_poseidon_valu_*.json) against the SIMD
                issue peak of /opt/skills/guides/MI355X_MICROARCH.md (one wave64 VALU instruction per 2 cycles per SIMD).
  babybear      BASELINE configs[3] (2^20 rows, BabyBear + Poseidon2-16, num_challenges 10), same measurements, N = 1 only.
  cpu_baseline  the CPU oracle prover (a restatement of the reference algorithm, "port") on a bounded sample, rank 0, N = 1.
Independent circuits shard one per GPU: every rank proves its own circuit, no data-path collective ("scaling": "weak").
`--workload commit` = PolynomialBatch::from_values on the wires matrix only (fri/oracle.rs:68-123).
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SIMDS, CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 2.0  # 256 CUs x 4 SIMD-32; a wave64 VALU op issues over 2 cycles (same guide)
GL_P = 0xFFFFFFFF00000001
SCOPES = ("IFFT", "FFT + blinding", "build Merkle tree", "hash leaves", "compute wires commitment", "compute partial products",
          "compute quotient polys", "construct the opening set", "compute opening proofs", "find proof-of-work witness",
          "fri query rounds", "quotient IFFT", "FRI LDE")


def splitmix64_matrix(seed, rows, cols):
    """SURVEY.md 8(d) synthetic input: SplitMix64 stream reduced mod p, [rows][cols] uint64."""
    count = rows * cols
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(GL_P)).reshape(rows, cols)


def cpu_baseline_commit(ncols, log_n, rate_bits, cap_height, sample_log_n):
    """The CPU oracle (same algorithm as the reference: per-column radix-2 NTTs, transpose,
    recursive Merkle) on all host cores, on a row-reduced sample of the same matrix."""
    from oracle import oracle as O
    cores = O.use_host_cpu_share()  # a 1-GPU box grants 16 of the host's CPUs (cgroup quota): more threads only contend
    vals = splitmix64_matrix(0xC0FFEE ^ (ncols << 32) ^ sample_log_n, ncols, 1 << sample_log_n)
    O.PolynomialBatch.from_values(vals[:, : 1 << 10].copy(), rate_bits, cap_height)  # warm the library
    t0 = time.perf_counter()
    O.PolynomialBatch.from_values(vals, rate_bits, cap_height)
    dt = time.perf_counter() - t0
    scale = float(1 << (log_n - sample_log_n))  # rows ratio; ignores the log factor in the NTT (favours the CPU)
    return {
        "value": 1.0 / (dt * scale), "unit": "commits/s", "cores": cores, "kind": "port",
        "sample": "oracle from_values on %d cols x 2^%d rows (1/%d of the workload's rows) took %.2f s; "
                  "scaled linearly in rows" % (ncols, sample_log_n, int(scale), dt),
        "sample_seconds": dt,
    }


def _oracle_prove_seconds(D, log_n, num_challenges, field):
    """one oracle prove() of the 2^log_n-row dummy circuit -> (seconds without the build() share, proof verified)"""
    if field == "babybear":
        from oracle.fields import BB
        circ = D.DummyCircuit(log_n, D.CircuitConfig.babybear(num_challenges), check_security=False, F=BB)
    else:
        circ = D.DummyCircuit(log_n, D.CircuitConfig(num_challenges=num_challenges), check_security=False)
    _ = circ.circuit_digest  # build(): not part of prove()
    for seed in range(1, 9):   # a witness that meets a zero denominator (BabyBear) would be re-drawn by the retry loop: take the next
        w = circ.witness(seed=seed)
        t0 = time.perf_counter()
        try:
            proof, _dbg = D.prove_cpu(circ, w)
        except RuntimeError:
            continue
        dt = time.perf_counter() - t0 - D.prove_cpu.last_cs_commit_seconds  # the oracle redoes build()'s constants/sigmas commit
        assert D.verify(circ, proof)
        return dt
    raise RuntimeError("no witness without InvZeroPermArg among 8 seeds")


def cpu_baseline_prove(log_n, num_challenges, sample_log_n, field="goldilocks", budget_s=150.0):
    """The CPU oracle prover (restatement of the reference's prove(), OpenMP where the reference uses Rayon) on the host cores of
    the box.  sample_log_n = None: AT THE BENCHMARK'S OWN SIZE when a calibration run (2^15 rows, scaled by rows) says it fits
    `budget_s`, otherwise on the largest smaller circuit that does, scaled linearly in rows ("scaled": true)."""
    from oracle import oracle as O
    from oracle import plonk_dummy as D
    cores = O.use_host_cpu_share()  # a 1-GPU box grants 16 of the host's CPUs (cgroup quota): more threads only contend
    calib = None
    if sample_log_n is None:
        cal_n = min(log_n, 15)
        calib = _oracle_prove_seconds(D, cal_n, num_challenges, field)
        est = calib * (1 << (log_n - cal_n)) * 1.25      # + the log factor of the transforms and colder caches
        sample_log_n = log_n
        while sample_log_n > cal_n and est > budget_s:
            sample_log_n, est = sample_log_n - 1, est / 2
    dt = _oracle_prove_seconds(D, sample_log_n, num_challenges, field)
    scale = float(1 << (log_n - sample_log_n))
    placement = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES", "GOMP_CPU_AFFINITY") if os.environ.get(k)}
    try:
        aff = sorted(os.sched_getaffinity(0))
        placement["affinity_mask"] = "%d cpus [%d..%d]" % (len(aff), aff[0], aff[-1])
    except AttributeError:
        pass
    return {
        "value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": cores, "kind": "port", "scaled": scale != 1.0,
        "sample": ("oracle prove() of the 2^%d-row %s dummy circuit itself (num_challenges %d) took %.2f s and verified" % (
                       sample_log_n, field, num_challenges, dt)) if scale == 1.0 else
                  ("oracle prove() of the 2^%d-row dummy circuit (1/%d of the rows, num_challenges %d) took %.2f s and "
                   "verified; scaled linearly in rows" % (sample_log_n, int(scale), num_challenges, dt)),
        "sample_seconds": dt, "sample_log_n": sample_log_n, "calibration_seconds": calib,
        "threads": "%d OpenMP threads (gbo_set_num_threads = the cgroup's CPU share), no explicit binding beyond the process's "
                   "affinity mask" % cores, "omp_placement": placement,
    }


def _num(x):
    """JSON has no NaN: unavailable figures are null."""
    return None if x is None or x != x or x in (float("inf"), float("-inf")) else x


def _latest_profile(pattern):
    """newest profiles/ summary matching `pattern`, with "stale": True when the library sources have changed since it was measured
    (the summary carries the hash of the csrc tree it was taken on, tools/csrc_hash.py; older files carry none and count as stale)"""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))  # r01e_..., r02_...: newest round last
    if not paths:
        return None
    j = json.load(open(paths[-1]))
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_hash import csrc_sha16
        j["stale"] = j.get("csrc_sha16") != csrc_sha16()
    except Exception:
        j["stale"] = None
    j["profile_file"] = os.path.relpath(paths[-1], ROOT)
    return j


_CIRCUITS = {}   # (field, log_n) -> build_dummy_circuit(): the K = 1 legs and the in-flight legs prove the same circuit


def dummy_circuit_inputs(bb, log_n):
    from plonky2_goldibear_amd import dummy_circuit as DC
    key = (bb, log_n)
    if key not in _CIRCUITS:
        _CIRCUITS[key] = DC.build_dummy_circuit_bb(log_n) if bb else DC.build_dummy_circuit(log_n)
    return _CIRCUITS[key]


class ProveLeg:
    """One field's circuit on this rank's GPU: build() once, then timed prove() steps from a host or an HBM-resident witness."""

    def __init__(self, field, log_n, challenges, local_rank, rank, inflight, ctx):
        import torch
        from plonky2_goldibear_amd import CircuitData, GpuContext
        from plonky2_goldibear_amd import dummy_circuit as DC
        self.torch, self.field, self.log_n, self.rank, self.inflight = torch, field, log_n, rank, inflight
        self.bb = bb = field == "babybear"
        if challenges is None:  # the minimum the reference's security assert allows (circuit_builder.rs:1190-1192)
            challenges = max(6 if bb else 2, -(-100 // ((31 if bb else 64) - log_n)))
        self.challenges = challenges
        # per-field circuit shape: standard_recursion_config_gl / recursion_config_bb_narrow (circuit_data.rs:102-139)
        self.nwires, self.nrouted, self.arity_bits, self.ext_d, self.esz = (167, 41, 3, 4, 4) if bb else (135, 80, 4, 2, 8)
        self.idt = np.int32 if bb else np.int64
        self.dev = "cuda:%d" % local_rank
        t_build = time.perf_counter()
        cs, k_is, pi_row, _ = dummy_circuit_inputs(bb, log_n)
        self.build_s = time.perf_counter() - t_build        # the dummy circuit's constants / sigma columns (numpy, this rank's cores)
        t_build = time.perf_counter()
        cs_dev = torch.from_numpy(cs.view(self.idt)).to(self.dev)
        self.lanes = []  # one (context, circuit, host witness) per proof in flight: independent circuits, as across GPUs
        self.extra_ctx = []
        for li in range(inflight):
            lctx = ctx if li == 0 else GpuContext(local_rank)
            if li:
                self.extra_ctx.append(lctx)
                for k, v in getattr(ProveLeg, "lib_options", ()):
                    lctx.set_option(k, v)
            if bb:  # build(): once per circuit
                circuit = CircuitData.babybear(lctx, log_n, cs_dev, k_is, num_challenges=challenges)
                wit = DC.dummy_witness_bb(log_n, pi_row, seed=rank * inflight + li)
            else:
                circuit = CircuitData(lctx, log_n, cs_dev, k_is, num_challenges=challenges)
                wit = DC.dummy_witness(log_n, pi_row, seed=rank * inflight + li)
            self.lanes.append([lctx, circuit, wit, None])
        # A stream of DIFFERENT witnesses: the dummy witness is zero outside the PublicInputGate row, so witness k of
        # a lane is the lane's buffer with that row re-written from seed k - N_WITNESSES pinned seeds, cycled, so that a field with a
        # natural InvZeroPermArg rate (BabyBear: ~1 proof in 5 at 2^20 rows) shows it in `perm_arg_retries` and pays for it in `value`
        self.pi_row = pi_row
        gen = self._gen = DC.dummy_witness_bb if bb else DC.dummy_witness
        self.rows = [[np.ascontiguousarray(gen(3, 0, seed=(rank * inflight + li) * N_WITNESSES + k)[:, 0]) for k in range(N_WITNESSES)]
                     for li in range(inflight)]
        self.step_no = 0
        # prove_with_partition_witness's retry loop (plonk/prover.rs:183-226): on InvZeroPermArg the random wire - last
        # wire of the PublicInputGate row - is re-drawn and the proof redone; failed attempts stay inside the timed region
        self.random_wire = (self.nwires - 1, pi_row)
        self.ctx = ctx
        self.proof, self.proof_len, self.retries = None, 0, 0
        del cs, cs_dev
        torch.cuda.synchronize()
        self.circuit_create_s = time.perf_counter() - t_build   # gb_circuit_create: the constants/sigmas commitment (build()'s GPU share)

    def to_p3_words(self, a):
        """canonical values -> the words the reference's field types hold in memory (GB_INPUT_P3_REPR): p3-baby-bear's Montgomery
        word x 2^32 mod p; p3-goldilocks keeps a u64 representative, for which the canonical one is as good as any"""
        if not self.bb:
            return a
        return ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(2013265921)).astype(np.uint32)

    def set_witness(self, where):
        """"pinned": one contiguous block of page-locked host memory, as a host that wants the PCIe rate allocates it (hipHostMalloc);
        "hbm": already resident on the device; "vecs": the layout the reference itself has - MatrixWitness.wire_values is Vec<Vec<F>>
        (iop/witness.rs:277-279): num_wires separately malloc'ed, PAGEABLE columns holding the field type's in-memory words,
        handed over as a pointer table (gb_prove_cols + GB_INPUT_P3_REPR) and staged by the library's own page-locked ring."""
        self.where = where
        for lane in self.lanes:
            if where == "vecs":
                lane[3] = [np.array(self.to_p3_words(col), copy=True) for col in lane[2]]
                continue
            t = self.torch.from_numpy(lane[2].view(self.idt))
            lane[3] = t.pin_memory().numpy().view(lane[2].dtype) if where == "pinned" else t.to(self.dev)
        self.torch.cuda.synchronize()

    def next_witness(self):
        """re-write the PublicInputGate row of every lane's witness from the next pinned seed (host: 135 / 167 scattered words;
        HBM-resident: the same through a small copy, ordered before the proof by a device synchronisation)"""
        k = self.step_no % N_WITNESSES
        self.step_no += 1
        for li, lane in enumerate(self.lanes):
            row = self.rows[li][k]
            if isinstance(lane[3], list):
                for col, v in zip(lane[3], self.to_p3_words(row)):
                    col[self.pi_row] = v
            elif isinstance(lane[3], np.ndarray):
                lane[3][:, self.pi_row] = row
            else:
                lane[3][:, self.pi_row] = self.torch.from_numpy(row.view(self.idt)).to(self.dev)
        if self.where == "hbm":
            self.torch.cuda.synchronize()

    def step(self):
        self.next_witness()
        t0 = time.perf_counter()
        if self.inflight == 1:
            lctx, circuit, _, wit = self.lanes[0]
            self.proof = circuit.prove(wit, random_wire=self.random_wire, rng=self.rng[0], p3_repr=self.where == "vecs")
            self.retries += circuit.perm_arg_retries
            self.step_log.append((time.perf_counter() - t0, circuit.perm_arg_retries))
        else:
            out, ret = [None] * self.inflight, [0] * self.inflight

            def run(i):
                out[i] = self.lanes[i][1].prove(self.lanes[i][3], random_wire=self.random_wire, rng=self.rng[i],
                                                p3_repr=self.where == "vecs")
                ret[i] = self.lanes[i][1].perm_arg_retries
            ts = [threading.Thread(target=run, args=(i,)) for i in range(self.inflight)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            self.proof = out[0]
            self.retries += sum(ret)
        self.proof_len = len(self.proof)

    def barrier(self):
        from plonky2_goldibear_amd import sharding
        self.ctx.synchronize()
        for c in self.extra_ctx:
            c.synchronize()
        self.torch.cuda.synchronize()
        sharding.barrier()

    def timed(self, steps, warmup, where):
        """W untimed steps, then exactly K steps between barriers; returns (seconds = max over ranks, scopes, retries)."""
        from plonky2_goldibear_amd import sharding
        self.set_witness(where)
        self.rng = [np.random.default_rng(1234 + self.rank * self.inflight + i) for i in range(self.inflight)]
        self.step_no, self.step_log = 0, []
        for _ in range(warmup):
            self.step()
        self.retries, self.step_log = 0, []
        if self.inflight == 1:
            self.ctx.set_profiling(True)  # per-scope HIP events; with several proofs in flight scopes overlap, so only wall time
        self.ctx.scope_reset()
        self.barrier()
        self.trace_marker()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        dt = sharding.max_over_ranks(time.perf_counter() - t0)  # whole job = the slowest rank
        self.trace_marker()
        scopes = {s: self.ctx.scope_ms(s) for s in SCOPES}
        self.ctx.set_profiling(False)
        return dt, scopes, self.retries

    trace_markers = False

    def trace_marker(self):
        """--trace-markers (tools/make_profiles.sh, under rocprofv3 --kernel-trace): one dispatch of a kernel that a proof never
        launches (the raw permutation of ONE state) in front of and behind every timed region, outside the clock - so that
        tools/roofline_recompute.py can cut exactly the dispatches of the timed steps out of the same process's trace"""
        if self.trace_markers:
            self.ctx.permute(np.zeros((1, 12), dtype=np.uint64))
            self.ctx.synchronize()

    def no_retry_rate(self):
        """proofs/s over the timed steps that needed no InvZeroPermArg retry (this rank, one proof in flight); None if none"""
        clean = [t for t, r in self.step_log if r == 0]
        return len(clean) / sum(clean) if clean else None

    def verify_last(self):
        """outside the timed region: the library's own host-side verifier (gb_verify) on lane 0's last proof"""
        return bool(self.lanes[0][1].verify(self.proof)) if self.proof is not None else None

    def _prove_seed(self, seed):
        """lane 0's witness with the PublicInputGate row of pinned seed `seed` (rank 0's numbering), proved outside the timed region"""
        lctx, circuit, _, wit = self.lanes[0]
        gen = self._gen
        row = np.ascontiguousarray(gen(3, 0, seed=seed)[:, 0])
        if isinstance(wit, list):
            for col, v in zip(wit, self.to_p3_words(row)):
                col[self.pi_row] = v
        elif isinstance(wit, np.ndarray):
            wit[:, self.pi_row] = row
        else:
            wit[:, self.pi_row] = self.torch.from_numpy(row.view(self.idt)).to(self.dev)
            self.torch.cuda.synchronize()
        return circuit.prove(wit, random_wire=self.random_wire, rng=np.random.default_rng(99), p3_repr=self.where == "vecs")

    def verify_all_witnesses(self):
        """every one of the N_WITNESSES distinct witnesses of the run proved once more and checked by gb_verify (host side, outside
        the timed region) -> number verified"""
        ok = 0
        for k in range(N_WITNESSES):
            proof = self._prove_seed(self.rank * self.inflight * N_WITNESSES + k)
            ok += bool(self.lanes[0][1].verify(proof))
        return ok

    def golden_check(self):
        """Byte parity of the line itself: tests/golden/bench_proof_sha256.json holds the SHA-256 of the proof the CPU ORACLE prover
        produces for this very circuit and witness seed (tests/golden/make_bench_proof_golden.py, run in the build container);
        the GPU's proof of the same witness must hash to it.  None when the file has no entry for this shape."""
        import hashlib
        try:
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_proof_sha256.json"))).get("%s_2p%d" % (self.field, self.log_n))
        except OSError:
            g = None
        if not g or g["num_challenges"] != self.challenges:
            return None, None
        proof = self._prove_seed(g["witness_seed"])
        retried = self.lanes[0][1].perm_arg_retries
        got = hashlib.sha256(proof).hexdigest()
        return (got == g["sha256"] and not retried), {"file": "tests/golden/bench_proof_sha256.json", "witness_seed": g["witness_seed"],
                                                      "sha256": g["sha256"], "gpu_sha256": got, "proof_len": len(proof)}

    def counts(self, rate_bits=3, cap_height=4):
        """algorithmic NTT bytes and Poseidon permutations of one proof"""
        n, N, c = 1 << self.log_n, (1 << self.log_n) << rate_bits, self.challenges
        nzs, nq = c * (-(-self.nrouted // 8)), c * 8  # Z + partial products, quotient chunks
        # from_values (2 + 2^r) n s per column (wires, zs/pp), from_coeffs (1 + 2^r) n s (quotient chunks), + the final
        # polynomial's D coordinate columns; the quotient's per-coset inverse NTTs and the small FRI layers are not counted
        alg_bytes = ((2 + 8) * (self.nwires + nzs) + (1 + 8) * (nq + self.ext_d)) * n * self.esz
        nlayers, db = 0, self.log_n
        while db > 5 and db + rate_bits - self.arity_bits >= cap_height:  # fri/reduction_strategies.rs:44-56
            nlayers, db = nlayers + 1, db - self.arity_bits
        leaf_perms = sum(N * (-(-w // 8)) for w in (self.nwires, nzs, nq))   # the three big trees' leaf sponges ("hash leaves")
        perms = leaf_perms + 3 * (N - 16) + sum(
            (N >> (self.arity_bits * (l + 1))) * ((self.ext_d << self.arity_bits) // 8) + ((N >> (self.arity_bits * (l + 1))) - 16)
            for l in range(nlayers))
        return alg_bytes, perms, leaf_perms, nzs, nq

    def report(self, steps, scopes, inflight, retries=0):
        """roofline / roofline_alu / merkle objects of one timed region.  A proof that hit InvZeroPermArg redid (part of) its wires
        commitment (prover.rs:183-226): that work is inside the scopes, so it is inside the per-step byte and permutation counts as
        well.  From a host witness the retry is incremental (gb_prove_retry: the random wire's column, the last absorption of
        every leaf sponge, the tree above); this is the host-witness leg's report."""
        bb, fname = self.bb, self.field
        alg_bytes, perms, leaf_perms, nzs, nq = self.counts()
        redo = retries / float(steps)   # extra (partial) wires commitments per step
        n_, N_ = 1 << self.log_n, (1 << self.log_n) << 3
        incremental = self.log_n + 3 >= 19 and self.nwires > 32      # the library's condition (include/goldibear_gpu.h, gb_prove_retry)
        redo_cols = 1 if incremental else self.nwires
        redo_leaf_perms = N_ * (1 if incremental else -(-self.nwires // 8))
        alg_bytes += int(redo * (2 + 8) * redo_cols * n_ * self.esz)
        leaf_perms += int(redo * redo_leaf_perms)
        perms += int(redo * (redo_leaf_perms + N_ - 16))
        live = inflight == 1
        # every transform the algorithmic bytes count: the commitments' scopes and the first FRI layer's coset_fft of the final
        # polynomial's D coordinate columns (the whole "FRI LDE" scope: the three smaller layers are ~2 % of it); the quotient's
        # per-coset inverse transforms ("quotient IFFT") are neither in the bytes nor in the time
        ntt_ms = (scopes["IFFT"][0] + scopes["FFT + blinding"][0] + scopes["FRI LDE"][0]) / steps if live else None
        merkle_ms = scopes["build Merkle tree"][0] / steps if live else None
        leaves_ms = scopes["hash leaves"][0] / steps if live else None
        achieved = alg_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None
        traffic = None  # physical HBM bytes per column from rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/*.json)
        tj = _latest_profile("r*_ntt_traffic_pmc_%s.json" % fname)
        if self.log_n == 20 and tj:
            traffic = tj["ifft_bytes_per_column"] * (self.nwires + nzs) + tj["lde_bytes_per_column"] * (self.nwires + nzs + nq + self.ext_d)
        rj = _latest_profile("r*_roofline_recompute.json")      # the same object recomputed from the committed rocprofv3 summaries
        rf = (rj or {}).get(fname) if self.log_n == 20 else None
        out = {
            "roofline": {"bound": "hbm", "kernel": "NTT pass = %s (IFFT) + %s (FFT + blinding), all commitments of the step" % (
                             ("k_bb_intt16_*", "k_bb_lde_pa16*+pb16") if bb else ("k_gl_intt16_*", "k_gl_lde_pa16*+pb16")),
                         "achieved": _num(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": _num(achieved / HBM_PEAK_GBS if achieved else None), "traffic": traffic,
                         "traffic_source": tj and {"file": tj["profile_file"], "stale": tj["stale"]},
                         "algorithmic_bytes": alg_bytes, "ms": _num(ntt_ms),
                         "scopes_ms": live and {k: scopes[k][0] / steps for k in ("IFFT", "FFT + blinding", "FRI LDE", "quotient IFFT")},
                         # `frac` is live (HIP-event scopes of this run); frac_from_profile is the same ratio from the kernel trace
                         # committed under profiles/ (sum of the NTT kernels' durations per proof): rocprofv3's per-kernel times on
                         # another box of the pool, a few per cent apart
                         "frac_from_profile": _num(rf and rf.get("frac_from_profile")),
                         "ntt_kernel_ms_from_profile": _num(rf and rf.get("ntt_kernel_ms_per_proof")),
                         # the transform is VALU-issue bound, not HBM bound: SQ_INSTS_VALU of the NTT kernels per algorithmic byte
                         # (31 Goldilocks / 25 BabyBear lane-instructions) at the measured 2.9 cycles per wave64 integer
                         # instruction caps `frac` at valu_ceiling_frac; pass_structure_ceiling_frac is the other cap of this
                         # transform - its 31 n s of physical traffic per from_values column at the chip's plain-copy rate
                         "valu_ceiling_frac": _num(rf and rf.get("valu_ceiling_frac")),
                         "pass_structure_ceiling_frac": _num(rf and rf.get("pass_structure_ceiling_frac")),
                         "physical_frac_from_profile": _num(rf and rf.get("physical_frac")),
                         "profile_source": rj and {"file": rj["profile_file"], "stale": rj["stale"]}},
            "merkle": {"permutations": perms, "Gperm_per_s": _num(perms / (merkle_ms * 1e-3) / 1e9 if merkle_ms else None),
                       "ms": _num(merkle_ms)},
        }
        # ALU roofline of the dominant kernel: leaf sponges.  VALU instructions per permutation come from SQ_INSTS_VALU / grid
        # (PMC pass, tools/pmc_poseidon.py); the peak is the SIMD issue rate (one wave64 VALU instruction per 2 cycles per SIMD).
        pj = _latest_profile("r*_poseidon_valu_%s.json" % fname)
        peak = SIMDS * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9   # G wave-instructions / s
        alu = {"bound": "valu-issue", "kernel": "k_bb_merkle_leaves" if bb else "k_gl_merkle_leaves",
               "unit": "G wave-instr/s", "peak": peak, "leaf_permutations": leaf_perms, "ms": _num(leaves_ms),
               "Gperm_per_s": _num(leaf_perms / (leaves_ms * 1e-3) / 1e9 if leaves_ms else None),
               "achieved": None, "frac": None, "valu_instr_per_permutation": None}
        if pj and leaves_ms:
            ipp = pj["valu_instr_per_permutation"]
            ach = ipp * leaf_perms / 64.0 / (leaves_ms * 1e-3) / 1e9
            # issue_cost_floor_frac: the same instructions priced with the measured per-class issue costs (mad 4.5, plain 32-bit 2.4,
            # carry / select / 64-bit 2.9 cycles; tools/isa_mix.py) over the SIMD cycles the kernel had, live time
            cyc = (pj.get("isa_mix") or {}).get("model_cycles_per_valu_instruction")
            alu.update({"valu_instr_per_permutation": ipp, "achieved": ach, "frac": ach / peak,
                        "issue_cost_floor_frac": _num(ipp * leaf_perms / 64.0 * cyc / (SIMDS * CLOCK_HZ * leaves_ms * 1e-3)) if cyc else None,
                        "source": pj.get("profile_file"), "stale": pj.get("stale")})
            kj = None if bb else _latest_profile("r*_leaf_kernel_probe.json")
            if kj and cyc:   # the same floor at the clock the chip holds under this load (s_memtime probes, tools/probe_leaves.py)
                alu["shader_clock_hz_under_load"] = kj["shader_clock_hz_under_load"]
                alu["issue_cost_floor_frac_at_measured_clock"] = _num(ipp * leaf_perms / 64.0 * cyc / (SIMDS * kj["shader_clock_hz_under_load"] * leaves_ms * 1e-3))
                alu["clock_source"] = {"file": kj["profile_file"], "stale": kj["stale"]}
            mpp = pj.get("mfma_instr_per_permutation")
            if mpp:   # Goldilocks: the MDS layers run on the matrix pipe - v_mfma_i32_32x32x32_i8, 32 cycles each on its SIMD
                alu["mfma_instr_per_permutation"] = mpp
                alu["matrix_pipe_busy_frac"] = _num(mpp * 32.0 * leaf_perms / 64.0 / (SIMDS * CLOCK_HZ * leaves_ms * 1e-3))
        out["roofline_alu"] = alu
        return out

    def free(self):
        for lane in self.lanes:
            lane[1].free()
            lane[3] = None
        for c in self.extra_ctx:
            c.close()
        self.lanes, self.extra_ctx = [], []
        self.ctx.trim()
        self.torch.cuda.empty_cache()


def host_footprint(leg):
    """this rank's host-side footprint for the multi-rank rehearsals: resident set (now / peak), page-locked bytes it asked for
    (the witness block of the "pinned" legs, the library's staging ring of the "vecs" legs), circuit build times"""
    rss = peak = None
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                rss = int(line.split()[1]) / 1024.0
            elif line.startswith("VmHWM:"):
                peak = int(line.split()[1]) / 1024.0
    except OSError:
        pass
    wit = leg.nwires * (1 << leg.log_n) * leg.esz * leg.inflight
    return {"rss_mb": rss, "rss_peak_mb": peak, "pinned_witness_mb": wit / 2.0 ** 20, "library_staging_ring_mb": 4 * max(64.0, (1 << leg.log_n) * leg.esz / 2.0 ** 20),   # four slots of max(64 MiB, one column): include/goldibear_gpu.h
            "circuit_columns_build_s": leg.build_s, "circuit_create_s": leg.circuit_create_s}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """Parent of a multi-GPU run started as `python bench.py --gpus N`: N rank processes under torch.distributed.run (the
    command the driver itself uses), never an exec of this process.  Returns the launcher's exit code (non-zero if any rank failed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_local_cpus(device_index):
    """CPUs of the NUMA node the GPU's PCIe function hangs off (sysfs local_cpulist), or None when it cannot be told"""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        return _parse_cpulist(open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read()) or None
    except Exception:
        return None


def bind_rank_to_local_cpus(local_rank, local_world, device_index):
    """numactl-free placement: this rank's threads (and, by first touch, its page-locked witness) go to the cores nearest its
    GPU; ranks whose GPUs share a NUMA node split that node's cores evenly.  Returns what was set, for the JSON line."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if local_world <= 1 or len(avail) < local_world:
        if local_world > 1:
            print("bench.py rank-local %d: not bound (%d cpus for %d ranks)" % (local_rank, len(avail), local_world), file=sys.stderr)
        return {"cpus": len(avail), "bound": False, "first": avail[0], "last": avail[-1]}
    near = [None] * local_world
    if device_index is not None:
        import torch
        ndev = max(1, torch.cuda.device_count())
        shared = bool(os.environ.get("GB_BENCH_SHARE_DEVICE"))
        near = [_gpu_local_cpus(j % ndev if shared else j) if (shared or j < ndev) else None for j in range(local_world)]
    key = lambda j: tuple(sorted(near[j])) if near[j] else None
    peers = [j for j in range(local_world) if key(j) == key(local_rank)]   # ranks that compete for the same cores
    pool = [c for c in avail if near[local_rank] is None or c in near[local_rank]] or avail
    k, m = peers.index(local_rank), len(peers)
    mine = pool[k * len(pool) // m:(k + 1) * len(pool) // m] or pool
    os.sched_setaffinity(0, mine)
    info = {"cpus": len(mine), "bound": True, "first": mine[0], "last": mine[-1], "numa_local": near[local_rank] is not None}
    print("bench.py rank-local %d: %d cpus [%d..%d]%s" % (local_rank, len(mine), mine[0], mine[-1],
                                                          " (GPU's NUMA node)" if info["numa_local"] else ""), file=sys.stderr)
    return info


def _first_line(e):
    lines = [l for l in str(e).splitlines() if l.strip()]
    return "%s: %s" % (type(e).__name__, lines[0].strip() if lines else "")


def _rccl_precondition(local_rank, stub):
    """what must hold on THIS rank before it may enter an RCCL call; None = fine, else the reason (cheap, local, cannot hang)"""
    import torch
    if os.environ.get("GB_BENCH_FAKE_RCCL"):      # CPU tests of this function: a gloo group stands in for the RCCL one
        return None
    if stub or not torch.cuda.is_available():
        return "no GPU visible to this rank"
    if not 0 <= local_rank < torch.cuda.device_count():
        return "device index %d out of range (%d devices)" % (local_rank, torch.cuda.device_count())
    if os.environ.get("GB_BENCH_SHARE_DEVICE"):
        return "ranks share a device (GB_BENCH_SHARE_DEVICE): RCCL needs one device per rank"
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") != "0":
        return "HSA_ENABLE_IPC_MODE_LEGACY must be 0 on this host driver (dmabuf IPC only)"
    return None


def init_control_plane(world, rank, local_rank, default_backend, stub=False):
    """The data path has no collective; torch.distributed carries one barrier pair and one max per timed region.  The DEFAULT group
    is gloo - rendezvous over TCP, nothing that can fail for GPU reasons - and, unless GB_BENCH_BACKEND says gloo, an RCCL group
    ("nccl") is brought up beside it; it carries the barrier and the reductions only if it came up on EVERY rank.  A job whose proofs
    need no interconnect must not die of its control plane, whichever way RCCL fails:
      1. every rank checks a cheap LOCAL precondition and the ranks agree over gloo - no rank enters an RCCL call unless all pass;
      2. group creation and one probe all-reduce run in a helper thread under a bounded wait (GB_BENCH_RCCL_TIMEOUT seconds, default
         90): a rank that raises reports at once, a rank whose peers never arrive stops waiting; torch's watchdog is told not to
         abort the process over a timed-out collective (TORCH_NCCL_ASYNC_ERROR_HANDLING=0);
      3. the outcome is agreed over gloo (MIN); on failure the half-built group is destroyed where that cannot hang, the run goes on
         over gloo and says so - on stderr and in the line's `control_plane`."""
    import datetime
    import torch
    import torch.distributed as dist
    from plonky2_goldibear_amd import sharding
    want = os.environ.get("GB_BENCH_BACKEND", default_backend)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dist.init_process_group("gloo")
    if want != "nccl":
        return {"backend": "gloo", "requested": want}

    def all_agree(ok, why):
        """MIN over the ranks of `ok` (gloo) and every rank's reason"""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        whys = [None] * world
        dist.all_gather_object(whys, why)
        return int(flag.item()) == 1, whys

    def fall_back(whys, extra=None):
        given = [w for w in whys if w]
        if given and len(given) == len(whys) and len(set(given)) == 1:
            reason = "all %d ranks: %s" % (len(whys), given[0])
        else:
            reason = "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(whys) if w) or "another rank could not bring RCCL up"
        if rank == 0:
            print("bench.py: RCCL control plane unavailable (%s) - barrier and max-over-ranks go over gloo; the proofs use no collective"
                  % reason, file=sys.stderr)
        cp = {"backend": "gloo", "requested": "nccl", "fallback_reason": reason}
        cp.update(extra or {})
        return cp

    pre = _rccl_precondition(local_rank, stub)
    ok, whys = all_agree(pre is None, pre)
    if not ok:
        return fall_back(whys, {"failed_at": "precondition"})

    fake = bool(os.environ.get("GB_BENCH_FAKE_RCCL"))
    timeout = float(os.environ.get("GB_BENCH_RCCL_TIMEOUT", "90"))
    res = {}

    def bring_up():
        try:
            if os.environ.get("GB_BENCH_RCCL_FAIL_RANK") == str(rank):       # test hook: this rank's RCCL raises, its peers wait
                raise RuntimeError("RCCL bring-up failed on this rank (GB_BENCH_RCCL_FAIL_RANK)")
            if not fake:
                torch.cuda.set_device(local_rank)     # the current device is per thread: this helper starts on device 0
            res["group"] = group = dist.new_group(backend="gloo" if fake else "nccl", timeout=datetime.timedelta(seconds=timeout))
            t = torch.ones(1) if fake else torch.ones(1, device=torch.device("cuda", local_rank))
            work = dist.all_reduce(t, group=group, async_op=True)
            work.wait()
            if not fake:
                torch.cuda.synchronize(local_rank)
            if int(t.item()) != world:
                raise RuntimeError("all_reduce over RCCL returned %r, expected %d" % (t.item(), world))
            res["ok"] = True
        except Exception as e:   # noqa: BLE001 - whatever RCCL raises, the job does not need it
            res["why"] = _first_line(e)

    # new_group() is itself a collective over the default group: every rank must call it, also the one that is about to fail
    if os.environ.get("GB_BENCH_RCCL_FAIL_RANK") == str(rank):
        res["group"] = dist.new_group(backend="gloo" if fake else "nccl", timeout=datetime.timedelta(seconds=timeout))
        res["why"] = "RuntimeError: RCCL bring-up failed on this rank (GB_BENCH_RCCL_FAIL_RANK)"
        helper = None
    else:
        helper = threading.Thread(target=bring_up, daemon=True)
        helper.start()
        helper.join(timeout)
    stuck = helper is not None and helper.is_alive()
    if stuck:
        res.setdefault("why", "RCCL bring-up still waiting after %.0f s (a peer never arrived)" % timeout)
    ok, whys = all_agree(bool(res.get("ok")) and not stuck, res.get("why"))
    if ok:
        sharding.set_group(res["group"])
        return {"backend": "gloo (standing in for RCCL: GB_BENCH_FAKE_RCCL)" if fake else "nccl", "requested": "nccl"}
    if not stuck and res.get("group") is not None and not fake:
        try:
            dist.destroy_process_group(res["group"])     # the half-built group: gone before the proofs start
        except Exception as e:   # noqa: BLE001
            print("bench.py: destroying the RCCL group failed: %s" % _first_line(e), file=sys.stderr)
    global _HARD_EXIT
    _HARD_EXIT = stuck or fake    # a helper thread may still sit in the collective: leave without joining it or its group
    return fall_back(whys, {"failed_at": "bring-up", "helper_still_waiting": stuck})


_HARD_EXIT = False


def finish(world, dist):
    """end of a rank: leave the process groups; after an RCCL bring-up that is still blocked in a helper thread there is nothing to
    tear down cleanly - flush and leave"""
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.barrier()
        if _HARD_EXIT:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


def stub_main(args, rank, world, affinity, dist, control_plane=None):
    """GB_BENCH_STUB=1: the multi-rank flow of main() with the GPU work replaced by a sleep (CPU tests of the launch path)"""
    from plonky2_goldibear_amd import sharding
    if os.environ.get("GB_BENCH_STUB_FAIL_RANK") == str(rank):   # test hook: a rank that dies must fail the whole run
        sys.exit(3)
    for _ in range(args.warmup):
        time.sleep(0.001)
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))   # the last rank is the slow one
    sharding.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "proofs/s", "value": world * args.steps / dt, "unit": "proofs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                          "config": {"workload": "STUB (GB_BENCH_STUB=1): no GPU work, launch-path rehearsal only"},
                          "stub": True, "affinity": affinity, "control_plane": control_plane}))
    finish(world, dist)


def workload_name(leg, witness):
    return ("prove(): 2^%d-row %s dummy circuit (2^%d+1 NoopGates), %s with num_challenges=%d, %s, witness %s -> proof bytes "
            "(%d B)" % (leg.log_n, "BabyBear" if leg.bb else "Goldilocks", leg.log_n - 1,
                        "recursion_config_bb_narrow" if leg.bb else "standard_recursion_config_gl", leg.challenges,
                        "Poseidon2-16" if leg.bb else "Poseidon-12", witness, leg.proof_len))


HOST_W = "in page-locked host memory, handed over every step"
VECS_W = ("the reference's own layout (MatrixWitness.wire_values: Vec<Vec<F>>, iop/witness.rs:277-279): %d separately malloc'ed PAGEABLE "
          "numpy columns holding the field type's in-memory words, handed over every step as a table of %d pointers "
          "(gb_prove_cols, GB_INPUT_P3_REPR) and staged through the library's page-locked ring by its copy threads")
N_WITNESSES = 16   # pinned witness seeds a leg cycles through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="prove", choices=["prove", "commit"])
    ap.add_argument("--challenges", type=int, default=None,
                    help="num_challenges; default = the minimum the reference's security assert allows at --log-n "
                         "(Goldilocks 2^20: 3, BabyBear 2^20: 10; circuit_builder.rs:1190-1192)")
    ap.add_argument("--field", default="goldilocks", choices=["goldilocks", "babybear"])
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent proofs in flight per GPU (one host thread + one HIP stream each); a step is then one "
                         "batch of that many proofs")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cols", type=int, default=135)
    ap.add_argument("--cpu-sample-log-n", type=int, default=None,
                    help="rows of the CPU baseline's circuit; default: --log-n itself when a calibration run says it fits --cpu-budget-s")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-babybear", action="store_true", help="skip the BASELINE configs[3] leg of the default run")
    ap.add_argument("--no-resident", action="store_true", help="skip the HBM-resident-witness leg (value_hbm_resident)")
    ap.add_argument("--no-inflight2", action="store_true", help="skip the two-proofs-in-flight legs (value_inflight2)")
    ap.add_argument("--no-vecs", action="store_true", help="skip the Vec<Vec<F>> legs (value_vec_of_vecs)")
    ap.add_argument("--trace-markers", action="store_true",
                    help="one marker dispatch (gb_permute of one state) around every timed region, for tools/roofline_recompute.py --db")
    ap.add_argument("--no-checks", action="store_true",
                    help="skip the proofs made outside the timed region (every witness verified once, the golden SHA-256): kernel traces")
    ap.add_argument("--lib-option", action="append", default=[], metavar="KEY=VALUE",
                    help="gb_ctx_set_option on every context of the run (copy_threads, lde_group, intt_group, ...), repeatable")
    ap.add_argument("--host-witness", action="store_true", help="(default since round 2; kept for old command lines)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a bare shell: this process has touched no GPU (nothing imported so far does) and stays
        # the parent of N fresh rank processes; rank 0's JSON line arrives on the inherited stdout
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` from a bare shell, or under "
                 "torch.distributed.run with --nproc-per-node N)" % (args.gpus, world))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    stub = bool(os.environ.get("GB_BENCH_STUB"))

    import torch
    import torch.distributed as dist

    # rehearsal switches (not used by the driver): GB_BENCH_BACKEND=gloo and GB_BENCH_SHARE_DEVICE=1 let several ranks share
    # the one GPU of a test box, to exercise the multi-rank flow without a multi-GPU node; GB_BENCH_STUB=1 replaces the GPU
    # work of a step by a sleep, so that launch -> rendezvous -> barrier -> max-over-ranks -> JSON line runs on a CPU-only box
    device_index = local_rank
    if os.environ.get("GB_BENCH_SHARE_DEVICE") and not stub:
        device_index = local_rank % max(1, torch.cuda.device_count())
    affinity = bind_rank_to_local_cpus(local_rank, local_world, None if stub else device_index)
    if not stub:
        torch.cuda.set_device(device_index)
    local_rank = device_index
    control_plane = None
    # GB_BENCH_FORCE_CONTROL_PLANE=1 (rehearsal): bring the control plane up even for one rank under the launcher - a one-rank RCCL
    # group is the only way to execute the "RCCL came up" branch (new_group, probe all-reduce, barrier and max over it) on a box
    # with a single GPU, where ranks would otherwise have to share a device
    if world > 1 or (os.environ.get("GB_BENCH_FORCE_CONTROL_PLANE") and "MASTER_PORT" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        control_plane = init_control_plane(world, rank, local_rank, "gloo" if stub else "nccl", stub)
        gathered = [None] * world
        dist.all_gather_object(gathered, affinity)      # every rank's CPU binding goes into the line (default group: gloo)
        affinity = dict(affinity or {}, ranks=gathered)
    if stub:
        return stub_main(args, rank, world, affinity, dist, control_plane)

    from plonky2_goldibear_amd import GpuContext, PolynomialBatch, sharding

    log_n, rate_bits, cap_height = args.log_n, 3, 4
    ctx = GpuContext(local_rank)
    lib_options = [(kv.split("=", 1)[0], int(kv.split("=", 1)[1])) for kv in args.lib_option]
    for k, v in lib_options:
        ctx.set_option(k, v)
    ProveLeg.lib_options = lib_options
    ProveLeg.trace_markers = args.trace_markers
    out = None
    if args.workload == "prove":
        inflight = max(1, args.inflight)
        steps = args.steps
        leg = ProveLeg(args.field, log_n, args.challenges, local_rank, rank, inflight, ctx)
        dt, scopes, retries = leg.timed(steps, args.warmup, "pinned")
        if rank == 0:
            out = {
                "metric": "proofs/s", "value": world * steps * inflight / dt, "unit": "proofs/s", "n_gpus": world, "steps": steps,
                "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32" if leg.bb else "u64", "data": "synthetic",
                "config": {"workload": workload_name(leg, HOST_W), "field": args.field, "log_n": log_n, "rate_bits": rate_bits,
                           "cap_height": cap_height, "proofs_in_flight_per_gpu": inflight, "witness": HOST_W,
                           "sharding": "one independent circuit per GPU, no collective"},
            }
            out.update(leg.report(steps, scopes, inflight, retries))
            out["scopes_ms_per_step"] = {k: v[0] / steps for k, v in scopes.items() if v[1]}
            out["verified"] = leg.verify_last()
            if not args.no_checks:
                out["verified_witnesses"] = "%d of %d" % (leg.verify_all_witnesses(), N_WITNESSES)   # every distinct witness, once, gb_verify
                out["proof_sha256_matches_golden"], out["golden"] = leg.golden_check()   # == the CPU oracle prover's bytes for that witness
            out["perm_arg_retries"] = retries  # InvZeroPermArg re-runs inside the timed steps (BabyBear: ~1 in 5 proofs)
            out["value_no_retry"] = _num(leg.no_retry_rate() and world * leg.no_retry_rate())
            out["witnesses"] = "%d pinned seeds, cycled: every step proves a different witness" % N_WITNESSES
            out["affinity"] = affinity
            out["control_plane"] = control_plane
        if world > 1:   # every rank's host footprint goes into the line (the default group is gloo: host objects)
            foot = [None] * world
            dist.all_gather_object(foot, host_footprint(leg))
            if rank == 0:
                out["ranks_host"] = foot
        elif rank == 0:
            out["ranks_host"] = [host_footprint(leg)]
        if not args.no_resident:
            dt2, scopes2, _ = leg.timed(steps, args.warmup, "hbm")
            if rank == 0:
                out["value_hbm_resident"] = world * steps * inflight / dt2
                out["ms_per_step_hbm_resident"] = dt2 / steps * 1e3
        if not args.no_vecs:
            dtv, scopesv, _ = leg.timed(steps, args.warmup, "vecs")
            if rank == 0:
                out["value_vec_of_vecs"] = world * steps * inflight / dtv
                out["ms_per_step_vec_of_vecs"] = dtv / steps * 1e3
                out["vec_of_vecs"] = VECS_W % (leg.nwires, leg.nwires)
                out["scopes_ms_per_step_vec_of_vecs"] = {k: v[0] / steps for k, v in scopesv.items() if v[1]}
        leg.free()
        del leg
        second = not args.no_inflight2 and inflight == 1   # SURVEY.md 8(d) counts a stream of independent proofs: two of them in
        if second:                                          # flight per GPU (one host thread + one stream each), host witnesses
            leg2 = ProveLeg(args.field, log_n, args.challenges, local_rank, rank, 2, ctx)
            dt3, _, _ = leg2.timed(steps, args.warmup, "pinned")
            if rank == 0:
                out["value_inflight2"] = world * steps * 2 / dt3
                out["ms_per_step_inflight2"] = dt3 / steps * 1e3    # one step = two proofs
            leg2.free()
            del leg2
        if args.field == "goldilocks" and world == 1 and not args.no_babybear and log_n == 20:
            # BASELINE configs[3]: the same measurements for BabyBear + Poseidon2-16 (a 31-bit field needs num_challenges = 10)
            bleg = ProveLeg("babybear", log_n, None, local_rank, rank, inflight, ctx)
            bdt, bscopes, bret = bleg.timed(steps, args.warmup, "pinned")
            bb = {"metric": "proofs/s", "value": steps * inflight / bdt, "ms_per_step": bdt / steps * 1e3, "dtype": "u32",
                  "config": {"workload": workload_name(bleg, HOST_W), "witness": HOST_W}}
            bb.update(bleg.report(steps, bscopes, inflight, bret))
            bb["scopes_ms_per_step"] = {k: v[0] / steps for k, v in bscopes.items() if v[1]}
            bb["verified"] = bleg.verify_last()
            if not args.no_checks:
                bb["verified_witnesses"] = "%d of %d" % (bleg.verify_all_witnesses(), N_WITNESSES)
                bb["proof_sha256_matches_golden"], bb["golden"] = bleg.golden_check()
            bb["perm_arg_retries"] = bret     # at their natural rate: `value` has the re-done proofs inside, value_no_retry has not
            bb["value_no_retry"] = _num(bleg.no_retry_rate())
            if not args.no_resident:
                bdt2, _, _ = bleg.timed(steps, args.warmup, "hbm")
                bb["value_hbm_resident"] = steps * inflight / bdt2
                bb["ms_per_step_hbm_resident"] = bdt2 / steps * 1e3
            if not args.no_vecs:
                bdtv, _, _ = bleg.timed(steps, args.warmup, "vecs")
                bb["value_vec_of_vecs"] = steps * inflight / bdtv
                bb["ms_per_step_vec_of_vecs"] = bdtv / steps * 1e3
                bb["vec_of_vecs"] = VECS_W % (bleg.nwires, bleg.nwires)
            bleg.free()
            if second:
                bleg2 = ProveLeg("babybear", log_n, None, local_rank, rank, 2, ctx)
                bdt3, _, _ = bleg2.timed(steps, args.warmup, "pinned")
                bb["value_inflight2"] = steps * 2 / bdt3
                bb["ms_per_step_inflight2"] = bdt3 / steps * 1e3
                bleg2.free()
            out["babybear"] = bb
        if rank == 0 and not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only (the driver's contract)
            from oracle import oracle as _O
            cores = _O.host_cpu_share()
            bbf = args.field == "babybear"
            ch = args.challenges or max(6 if bbf else 2, -(-100 // ((31 if bbf else 64) - log_n)))
            # default: the benchmark's own circuit when the calibrated estimate fits the budget (~50 s on 16 cores at 2^20 rows)
            out["cpu_baseline"] = cpu_baseline_prove(log_n, ch, args.cpu_sample_log_n, args.field, args.cpu_budget_s)
    else:
        ncols, n = args.cols, 1 << log_n
        host = splitmix64_matrix((0xC0FFEE ^ (ncols << 32) ^ log_n) + rank, ncols, n)
        ftag, esz = 0, 8
        if args.field == "babybear":
            ftag, esz = 1, 4
            host = (host % np.uint64(2013265921)).astype(np.uint32)
            dev = torch.from_numpy(host.view(np.int32)).to("cuda:%d" % local_rank)
        else:
            dev = torch.from_numpy(host.view(np.int64)).to("cuda:%d" % local_rank)
        del host
        torch.cuda.synchronize()

        def step():
            b = PolynomialBatch.from_values(ctx, dev, rate_bits, cap_height, field=ftag)
            b.free()

        def barrier():
            ctx.synchronize()
            torch.cuda.synchronize()
            sharding.barrier()

        for _ in range(args.warmup):
            step()
        ctx.set_profiling(True)
        ctx.scope_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = sharding.max_over_ranks(time.perf_counter() - t0)
        scopes = {s: ctx.scope_ms(s) for s in SCOPES}
        ctx.set_profiling(False)
        if rank == 0:
            steps, N = args.steps, n << rate_bits
            ntt_ms = (scopes["IFFT"][0] + scopes["FFT + blinding"][0]) / steps
            merkle_ms = scopes["build Merkle tree"][0] / steps
            alg_bytes = (2 + (1 << rate_bits)) * n * esz * ncols  # (2 + 2^r) n s per column
            perms = N * (-(-ncols // 8)) + (N - (1 << cap_height))  # leaf sponge + internal nodes
            achieved = alg_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None
            tj = _latest_profile("r*_ntt_traffic_pmc_%s.json" % args.field)
            traffic = (tj["ifft_bytes_per_column"] + tj["lde_bytes_per_column"]) * ncols if (tj and log_n == 20) else None
            metric = "commits/s (PolynomialBatch::from_values, wires oracle of the 2^%d-row circuit)" % log_n
            out = {
                "metric": metric, "value": world * steps / dt, "unit": "commits/s", "n_gpus": world, "steps": steps,
                "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32" if ftag else "u64", "data": "synthetic",
                "config": {"workload": "from_values: %d cols x 2^%d rows %s, rate_bits 3, cap_height 4, %s, input resident in HBM" % (
                               ncols, log_n, args.field, "Poseidon2-16" if ftag else "Poseidon-12"),
                           "field": args.field, "log_n": log_n, "rate_bits": rate_bits, "cap_height": cap_height,
                           "sharding": "one independent batch per GPU, no collective"},
                "roofline": {"bound": "hbm", "kernel": "NTT pass (IFFT + FFT + blinding scopes)", "achieved": _num(achieved),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _num(achieved / HBM_PEAK_GBS if achieved else None),
                             "traffic": traffic, "algorithmic_bytes": alg_bytes, "ms": _num(ntt_ms)},
                "scopes_ms_per_step": {k: v[0] / steps for k, v in scopes.items() if v[1]},
                "merkle": {"permutations": perms, "Gperm_per_s": _num(perms / (merkle_ms * 1e-3) / 1e9 if merkle_ms else None)},
            }
            if not args.no_cpu_baseline and world == 1:
                from oracle import oracle as _O
                cores = _O.host_cpu_share()
                sample = args.cpu_sample_log_n
                if sample is None:
                    sample = max(10, min(log_n, 19, 13 + (cores.bit_length() - 1)))
                out["cpu_baseline"] = cpu_baseline_commit(ncols, log_n, rate_bits, cap_height, sample)
    if rank == 0:
        print(json.dumps(out, allow_nan=False))
    ctx.close()
    finish(world, dist)


if __name__ == "__main__":
    main()
