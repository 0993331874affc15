#!/usr/bin/env python3
"""Soak: N proofs of the 2^20-row dummy circuits (both fields) from fresh random witnesses, alternating the hand-over forms (one
page-locked block, separately allocated pageable columns of p3 words, HBM-resident), EVERY proof checked by gb_verify; the retry path
at its natural rate (BabyBear).  Catches what a 16-witness run can miss: data-dependent paths (the folds' rare carry branch, zero
denominators), allocator luck across many pool cycles, the staging ring's slots coming round hundreds of times.

  gpurun --timeout 900 -- 'python3 tools/soak.py 150 > gpurun_out/soak.txt'
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from plonky2_goldibear_amd import CircuitData, GpuContext
    from plonky2_goldibear_amd import dummy_circuit as DC
    from plonky2_goldibear_amd.native import TooManyPermArgFailuresError
    n_proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = GpuContext(0)
    rng = np.random.default_rng(2026)
    for field in ("goldilocks", "babybear"):
        bb = field == "babybear"
        p = DC.BB_P if bb else DC.P
        dt, idt = (np.uint32, np.int32) if bb else (np.uint64, np.int64)
        cs, k_is, pi_row, _ = (DC.build_dummy_circuit_bb if bb else DC.build_dummy_circuit)(log_n)
        ch = max(6 if bb else 2, -(-100 // ((31 if bb else 64) - log_n)))
        circuit = (CircuitData.babybear(ctx, log_n, cs, k_is, num_challenges=ch) if bb else CircuitData(ctx, log_n, cs, k_is, num_challenges=ch))
        del cs
        nw, first = (167, 8) if bb else (135, 4)
        base = np.zeros((nw, 1 << log_n), dtype=dt)
        pinned = torch.from_numpy(base.view(idt)).pin_memory().numpy().view(dt)
        cols = [np.zeros(1 << log_n, dtype=dt) for _ in range(nw)]
        dev = torch.from_numpy(base.view(idt)).cuda()
        rw = (nw - 1, pi_row)
        t0, retries, forms = time.time(), 0, {"pinned": 0, "vecs": 0, "hbm": 0}
        def prove_form(form, row):
            if form == "pinned":
                pinned[:, pi_row] = row
                return circuit.prove(pinned, random_wire=rw, rng=rng)
            if form == "vecs":
                words = ((row.astype(np.uint64) << np.uint64(32)) % np.uint64(p)).astype(dt) if bb else row
                for c in range(nw):
                    cols[c][pi_row] = words[c]
                return circuit.prove(cols, random_wire=rw, rng=rng, p3_repr=True)
            dev[:, pi_row] = torch.from_numpy(row.view(idt)).cuda()
            torch.cuda.synchronize()
            return circuit.prove(dev, random_wire=rw, rng=rng)

        gave_up = 0
        for i in range(n_proofs):
            row = np.zeros(nw, dtype=dt)
            row[first:] = rng.integers(0, p, nw - first, dtype=np.uint64).astype(dt)
            form = ("pinned", "vecs", "hbm")[i % 3]
            try:
                proof = prove_form(form, row)
            except TooManyPermArgFailuresError:
                # three attempts in a row met a zero denominator: ProverError::TooManyPermArgFailures, as the reference itself returns
                # (plonk/prover.rs:183-226, MAX_PERM_ARG_RETRIES = 3) - 0.2^3 per 2^20-row BabyBear proof
                gave_up += 1
                print("  %s proof %d (%s): TooManyPermArgFailures" % (field, i, form), flush=True)
                retries += 2
                continue
            retries += circuit.perm_arg_retries
            forms[form] += 1
            assert circuit.verify(proof), (field, i, form)
            continue
        dt_s = time.time() - t0
        print("%s 2^%d rows: %d proofs (%s), every one verified by gb_verify, %d InvZeroPermArg retries, %d x TooManyPermArgFailures (three zero denominators in a row: the reference gives up too), %.1f s incl. host-side verification" % (
            field, log_n, n_proofs, ", ".join("%d %s" % (v, k) for k, v in forms.items()), retries, gave_up, dt_s), flush=True)
        circuit.free()
        ctx.trim()
    ctx.close()


if __name__ == "__main__":
    main()
