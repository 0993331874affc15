import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from oracle import oracle_bb as B
from plonky2_goldibear_amd import GB_BABYBEAR, GpuContext, PolynomialBatch
from plonky2_goldibear_amd import native as N
O.use_host_cpu_share()
ctx = GpuContext(0)
for field in ("goldilocks", "babybear"):
    for log_n in (21, 22):
        ncols = 2
        if field == "goldilocks":
            vals, tag = O.splitmix64_fill(7, ncols << log_n).reshape(ncols, 1 << log_n), N.GB_GOLDILOCKS
            cpu = O.PolynomialBatch.from_values(vals, 3, 4)
        else:
            vals, tag = B.fill(7, ncols << log_n).reshape(ncols, 1 << log_n), GB_BABYBEAR
            cpu = B.PolynomialBatch.from_values(vals, 3, 4)
        gpu = PolynomialBatch.from_values(ctx, vals, 3, 4, field=tag)
        pc = (gpu.polynomials == cpu.polynomials)
        print(field, log_n, "coeffs ok" if pc.all() else "coeffs BAD: %d wrong, first idx %s" % ((~pc).sum(), np.argwhere(~pc)[:5].tolist()), flush=True)
        co = PolynomialBatch.from_coeffs(ctx, cpu.polynomials, 3, 4, field=tag)
        leaves = co.merkle_tree.leaves
        bad = np.flatnonzero((leaves != cpu.leaves).any(axis=1))
        print(field, log_n, "lde ok" if bad.size == 0 else "lde BAD: %d leaves wrong, first %s ; blocks(2^20) hit: %s" % (bad.size, bad[:8].tolist(), sorted(set((bad >> 20).tolist()))[:40]), flush=True)
        print("   cap ok" if (co.merkle_tree.cap == cpu.cap).all() else "   cap BAD", flush=True)
        gpu.free(); co.free(); ctx.trim()
