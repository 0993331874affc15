"""Timing of prove() on a general-gate circuit: the reference's factorial example padded with NoopGates to 2^log_n rows
(the gate-constraint kernel evaluates every gate of the gate set at every LDE point, whatever sits in the rows).
usage: python tools/bench_gates.py [log_n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from plonky2_goldibear_amd import GpuContext  # noqa: E402
from plonky2_goldibear_amd.circuit_builder import NoopGate  # noqa: E402
from circuits import factorial_circuit  # noqa: E402


def main():
    ctx = GpuContext(0)
    ctx.set_profiling(True)
    for log_n in [int(a) for a in sys.argv[1:]] or [12, 16]:
        nch = 2 if log_n <= 14 else 3
        b, pw = factorial_circuit(num_challenges=nch)
        while b.num_gates() < (1 << log_n) - 60:
            b.add_gate(NoopGate())
        t0 = time.time()
        c = b.build(ctx)
        w, pis = c.generate_witness(pw)
        t1 = time.time()
        assert c.degree_bits == log_n
        proof = c.data.prove(w, pis)
        c.data.verify(proof)
        import torch
        wd = torch.from_numpy(w.view(np.int64)).to("cuda:0")
        ctx.scope_reset()
        ts = []
        for _ in range(5):
            ctx.synchronize()
            t = time.time()
            c.data.prove(wd, pis)
            ctx.synchronize()
            ts.append(time.time() - t)
        q = ctx.scope_ms("compute quotient polys")
        print("log_n %d  build+witness %.1fs  prove %.2f ms (min of 5)  quotient scope %.2f ms/proof  proof %d bytes"
              % (log_n, t1 - t0, 1e3 * min(ts), q[0] / max(q[1], 1), len(proof)), flush=True)
        c.data.free()


if __name__ == "__main__":
    main()
