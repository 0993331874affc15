#!/usr/bin/env python3
"""bench.py - one JSON line for the driver (see DESIGN.md "Measurement").

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM.  Workload `prove` (default, BASELINE.json configs[2]): one full prove() of the 2^20-row
Goldilocks dummy circuit (num_challenges = 3) from a device-resident
MatrixWitness to ProofWithPublicInputs bytes, constants/sigmas commitment pre-resident
(plonk/prover.rs:228-447).  Workload `commit` = PolynomialBatch::from_values on the wires matrix
only (135 columns, rate 3, cap 4 - fri/oracle.rs:68-123 as called at plonk/prover.rs:261-272).
Independent circuits shard one per GPU: every rank runs the same workload on its own device,
no data-path collective ("scaling": "weak").  The roofline object prices the NTT pass (the IFFT
+ LDE kernels) against the 8 TB/s HBM peak with SURVEY.md 8(d)'s algorithmic bytes; the
cpu_baseline object times the CPU oracle (a restatement of the reference algorithm, "port") on
a bounded sample on rank 0's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
GL_P = 0xFFFFFFFF00000001


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


def cpu_baseline_prove(log_n, num_challenges, sample_log_n, field="goldilocks"):
    """The CPU oracle prover (restatement of the reference's prove(), OpenMP where the reference uses
    Rayon) on a smaller dummy circuit of the same shape, scaled linearly in rows."""
    from oracle import oracle as O
    from oracle import plonk_dummy as D
    cores = O.use_host_cpu_share()  # a 1-GPU box grants 16 of the host's CPUs (cgroup quota): more threads only contend
    if field == "babybear":
        from oracle.fields import BB
        circ = D.DummyCircuit(sample_log_n, D.CircuitConfig.babybear(num_challenges), check_security=False, F=BB)
    else:
        circ = D.DummyCircuit(sample_log_n, D.CircuitConfig(num_challenges=num_challenges), check_security=False)
    _ = circ.circuit_digest  # build(): not part of prove()
    w = circ.witness(seed=1)
    t0 = time.perf_counter()
    proof, _dbg = D.prove_cpu(circ, w)
    dt = time.perf_counter() - t0 - D.prove_cpu.last_cs_commit_seconds  # the oracle redoes build()'s constants/sigmas commit
    assert D.verify(circ, proof)
    scale = float(1 << (log_n - sample_log_n))
    return {
        "value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": cores, "kind": "port",
        "sample": "oracle prove() of the 2^%d-row dummy circuit (1/%d of the rows, num_challenges %d) took %.2f s and "
                  "verified; scaled linearly in rows" % (sample_log_n, int(scale), num_challenges, dt),
        "sample_seconds": dt,
    }


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
    ap.add_argument("--cpu-sample-log-n", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-witness", action="store_true",
                    help="prove workload: hand the witness over as a HOST array every step (the drop-in boundary's case: 1.05 GiB "
                         "across PCIe per Goldilocks proof); the default keeps it resident in HBM, which is what `value` is quoted on")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N>1)"
    # rehearsal switches (not used by the driver): GB_BENCH_BACKEND=gloo and GB_BENCH_SHARE_DEVICE=1 let several ranks share
    # the one GPU of a test box, to exercise the multi-rank flow without a multi-GPU node
    if os.environ.get("GB_BENCH_SHARE_DEVICE"):
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GB_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from plonky2_goldibear_amd import CircuitData, GpuContext, PolynomialBatch, sharding
    from plonky2_goldibear_amd import dummy_circuit as DC

    ncols, log_n, rate_bits, cap_height = args.cols, args.log_n, 3, 4
    n = 1 << log_n
    bb = args.field == "babybear"
    if args.challenges is None:
        bits = 31 if bb else 64
        args.challenges = max(6 if bb else 2, -(-100 // (bits - log_n)))
    # per-field circuit shape: standard_recursion_config_gl / recursion_config_bb_narrow (circuit_data.rs:102-139)
    nwires, nrouted, arity_bits, ext_d, esz = (167, 41, 3, 4, 4) if bb else (135, 80, 4, 2, 8)
    ctx = GpuContext(local_rank)
    proof_len = 0
    inflight = max(1, args.inflight) if args.workload == "prove" else 1
    extra_ctx = []
    if args.workload == "prove":
        import threading
        ncols = nwires
        idt = np.int32 if bb else np.int64
        cs, k_is, pi_row, _ = DC.build_dummy_circuit_bb(log_n) if bb else DC.build_dummy_circuit(log_n)
        cs_dev = torch.from_numpy(cs.view(idt)).to("cuda:%d" % local_rank)
        lanes = []  # one (context, circuit, witness) per proof in flight: independent circuits, as across GPUs
        for li in range(inflight):
            lctx = ctx if li == 0 else GpuContext(local_rank)
            if li:
                extra_ctx.append(lctx)
            if bb:  # build(): once per circuit
                circuit = CircuitData.babybear(lctx, log_n, cs_dev, k_is, num_challenges=args.challenges)
                wit = DC.dummy_witness_bb(log_n, pi_row, seed=rank * inflight + li)
            else:
                circuit = CircuitData(lctx, log_n, cs_dev, k_is, num_challenges=args.challenges)
                wit = DC.dummy_witness(log_n, pi_row, seed=rank * inflight + li)
            if args.host_witness:   # page-locked, as a host that wants the PCIe rate would allocate it (hipHostMalloc)
                wit_in = torch.from_numpy(wit.view(idt)).pin_memory().numpy().view(wit.dtype)
            else:
                wit_in = torch.from_numpy(wit.view(idt)).to("cuda:%d" % local_rank)
            lanes.append((lctx, circuit, wit_in))
        # prove_with_partition_witness's retry loop (plonk/prover.rs:183-226): on InvZeroPermArg the random wire - last
        # wire of the PublicInputGate row - is re-drawn and the proof redone; failed attempts stay inside the timed region
        random_wire = (nwires - 1, pi_row)
        rng = np.random.default_rng(1234 + rank)
        retries = [0]
        last_proof = [None]
        del cs, cs_dev, wit
        torch.cuda.synchronize()

        def step():
            nonlocal proof_len
            if inflight == 1:
                last_proof[0] = lanes[0][1].prove(lanes[0][2], random_wire=random_wire, rng=rng)
                proof_len = len(last_proof[0])
                retries[0] += lanes[0][1].perm_arg_retries
                return
            out = [0] * inflight

            def run(i):
                out[i] = len(lanes[i][1].prove(lanes[i][2], random_wire=random_wire, rng=np.random.default_rng(99 + i)))
            ts = [threading.Thread(target=run, args=(i,)) for i in range(inflight)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            proof_len = out[0]
    else:
        host = splitmix64_matrix((0xC0FFEE ^ (ncols << 32) ^ log_n) + rank, ncols, n)
        ftag = 0
        if args.field == "babybear":
            ftag = 1
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
        for c in extra_ctx:
            c.synchronize()
        torch.cuda.synchronize()
        sharding.barrier()

    for _ in range(args.warmup):
        step()
    if args.workload == "prove":
        retries[0] = 0
    if inflight == 1:
        ctx.set_profiling(True)  # per-scope HIP events; with several proofs in flight scopes overlap, so only wall time
    ctx.scope_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0)  # whole job = the slowest rank

    scope_names = ("IFFT", "FFT + blinding", "build Merkle tree", "compute wires commitment", "compute partial products",
                   "compute quotient polys", "construct the opening set", "compute opening proofs",
                   "find proof-of-work witness", "fri query rounds")
    scopes = {s: ctx.scope_ms(s) for s in scope_names}
    ctx.set_profiling(False)
    units_per_step = inflight

    if rank == 0:
        steps = args.steps
        ntt_ms = (scopes["IFFT"][0] + scopes["FFT + blinding"][0]) / steps or float("nan")
        merkle_ms = scopes["build Merkle tree"][0] / steps or float("nan")
        N = n << rate_bits
        if args.workload == "prove":
            c = args.challenges
            nzs, nq = c * (-(-nrouted // 8)), c * 8  # Z + partial products, quotient chunks
            # from_values (2 + 2^r) n s per column (wires, zs/pp), from_coeffs (1 + 2^r) n s (quotient
            # chunks), + the final polynomial's D coordinate columns; the quotient's per-coset inverse NTTs and the
            # small FRI layers are not counted (conservative)
            alg_bytes = ((2 + 8) * (nwires + nzs) + (1 + 8) * (nq + ext_d)) * n * esz
            nlayers, db = 0, log_n
            while db > 5 and db + rate_bits - arity_bits >= cap_height:  # fri/reduction_strategies.rs:44-56
                nlayers, db = nlayers + 1, db - arity_bits
            perms = sum(N * (-(-w // 8)) + (N - 16) for w in (nwires, nzs, nq)) + sum(
                (N >> (arity_bits * (l + 1))) * ((ext_d << arity_bits) // 8) + ((N >> (arity_bits * (l + 1))) - 16)
                for l in range(nlayers))
            metric = "proofs/s"
            workload = ("prove(): 2^%d-row %s dummy circuit (2^%d+1 NoopGates), %s with num_challenges=%d, %s, witness "
                        "resident in HBM -> proof bytes (%d B)" % (
                            log_n, "BabyBear" if bb else "Goldilocks", log_n - 1,
                            "recursion_config_bb_narrow" if bb else "standard_recursion_config_gl", c,
                            "Poseidon2-16" if bb else "Poseidon-12", proof_len))
        else:
            alg_bytes = (2 + (1 << rate_bits)) * n * esz * ncols  # (2 + 2^r) n s per column
            perms = N * (-(-ncols // 8)) + (N - (1 << cap_height))  # leaf sponge + internal nodes
            metric = "commits/s (PolynomialBatch::from_values, wires oracle of the 2^%d-row circuit)" % log_n
            workload = "from_values: %d cols x 2^%d rows %s, rate_bits 3, cap_height 4, %s" % (
                ncols, log_n, args.field, "Poseidon2-16" if args.field == "babybear" else "Poseidon-12")
        achieved = alg_bytes / (ntt_ms * 1e-3) / 1e9
        # physical HBM bytes per column, measured with rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/*.json)
        traffic = None
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ntt_traffic_pmc_%s.json" % args.field)))  # newest round last
        if log_n == 20 and tpaths:
            tj = json.load(open(tpaths[-1]))
            if args.workload == "prove":
                traffic = tj["ifft_bytes_per_column"] * (nwires + nzs) + tj["lde_bytes_per_column"] * (nwires + nzs + nq + ext_d)
            else:
                traffic = (tj["ifft_bytes_per_column"] + tj["lde_bytes_per_column"]) * ncols
        out = {
            "metric": metric, "value": world * steps * units_per_step / dt, "unit": metric.split(" ")[0], "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32" if args.field == "babybear" else "u64", "data": "synthetic",
            "config": {"workload": workload, "field": args.field, "log_n": log_n, "rate_bits": rate_bits,
                       "cap_height": cap_height, "proofs_in_flight_per_gpu": inflight,
                       "witness": "page-locked host memory, copied in every step" if getattr(args, "host_witness", False) else "resident in HBM",
                       "sharding": "one independent circuit per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": "NTT pass = %s (IFFT) + %s (FFT + blinding), all commitments of the step" % (
                             ("k_bb_intt_p1+p2+p3", "k_bb_lde_pa+pb") if bb else ("k_gl_intt16_p1+p2+p3", "k_gl_lde_pa16+pb16")),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes": alg_bytes, "ms": ntt_ms},
            "scopes_ms_per_step": {k: v[0] / steps for k, v in scopes.items() if v[1]},
            "merkle": {"permutations": perms, "Gperm_per_s": perms / (merkle_ms * 1e-3) / 1e9},
        }
        if args.workload == "prove":
            if last_proof[0] is not None:  # outside the timed region: the library's own host-side verifier (gb_verify)
                out["verified"] = bool(lanes[0][1].verify(last_proof[0]))
            out["perm_arg_retries"] = retries[0]  # InvZeroPermArg re-runs inside the timed steps (BabyBear: ~1 in 5 proofs)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only (the driver's contract)
            sample = args.cpu_sample_log_n
            from oracle import oracle as _O
            cores = _O.host_cpu_share()
            if args.workload == "prove":
                if sample is None:
                    sample = max(8, min(log_n, 19 if bb else 18, (15 if bb else 14) + (cores.bit_length() - 1)))  # ~10-30 s of CPU work
                out["cpu_baseline"] = cpu_baseline_prove(log_n, args.challenges, sample, args.field)
            else:
                if sample is None:
                    sample = max(10, min(log_n, 19, 13 + (cores.bit_length() - 1)))
                out["cpu_baseline"] = cpu_baseline_commit(ncols, log_n, rate_bits, cap_height, sample)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
