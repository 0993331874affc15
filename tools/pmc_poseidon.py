#!/usr/bin/env python3
"""VALU instructions per Poseidon permutation of the leaf-hash kernel, from the SQ counter summary of a
`bench.py --workload commit` PMC pass (tools/pmc_sq_summary.py CSV) -> the JSON bench.py's roofline_alu object reads.

  python tools/pmc_poseidon.py profiles/r02_commit_goldilocks_2p20_sq_counters.csv goldilocks 135 20 profiles/r02_poseidon_valu_goldilocks.json

SQ_INSTS_VALU counts wave-level instructions; one wave hashes 64 leaves, a leaf of `cols` elements takes ceil(cols / 8)
permutations (overwrite-mode sponge, rate 8), so instructions per permutation = SQ_INSTS_VALU / (N * ceil(cols/8) / 64)."""
import csv
import json
import sys

path, field, cols, log_n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
kernel = "gbk::k_gl_merkle_leaves" if field == "goldilocks" else "gbk::k_bb_merkle_leaves"
row = next(r for r in csv.DictReader(open(path)) if r["Kernel"] == kernel)
N = 1 << (log_n + 3)
perms = N * (-(-cols // 8)) * int(row["Dispatches"])
insts = float(row["SQ_INSTS_VALU"])
wave_cycles = float(row["SQ_WAVE_CYCLES"])
res = {
    "source_file": path, "kernel": kernel, "columns": cols, "log_n": log_n, "permutations": perms,
    "SQ_INSTS_VALU": insts, "valu_instr_per_permutation": insts / (perms / 64.0),
    "vgprs": int(row["VGPRs"]), "duration_ns_under_pmc": float(row["TotalDurationNs(under PMC)"]),
    "wait_inst_per_wave_cycle": float(row["wait_inst_per_wave_cycle"]), "wait_any_per_wave_cycle": float(row["wait_any_per_wave_cycle"]),
    "note": "issue cost per wave64 VALU instruction measured by tools/microbench_valu2.hip: v_mov/v_add_u32 ~2.4-2.6 cycles, every "
            "carry, select, shift, mul and v_mad_u64_u32 ~4.2-4.5 cycles per SIMD",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
