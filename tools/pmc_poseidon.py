#!/usr/bin/env python3
"""VALU instructions per Poseidon permutation of the leaf-hash kernel, from the SQ counter summary of a
`bench.py --workload commit` PMC pass (tools/pmc_sq_summary.py CSV) -> the JSON bench.py's roofline_alu object reads.

  python tools/pmc_poseidon.py profiles/r02_commit_goldilocks_2p20_sq_counters.csv goldilocks 135 20 profiles/r02_poseidon_valu_goldilocks.json

SQ_INSTS_VALU counts wave-level instructions; one wave hashes 64 leaves, a leaf of `cols` elements takes ceil(cols / 8)
permutations (overwrite-mode sponge, rate 8), so instructions per permutation = SQ_INSTS_VALU / (N * ceil(cols/8) / 64)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import measured_sha16 as csrc_sha16  # noqa: E402
from isa_mix import kernel_mix    # noqa: E402

path, field, cols, log_n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
kernel = "gbk::k_gl_merkle_leaves" if field == "goldilocks" else "gbk::k_bb_merkle_leaves"
row = next(r for r in csv.DictReader(open(path)) if r["Kernel"] == kernel)
N = 1 << (log_n + 3)
perms = N * (-(-cols // 8)) * int(row["Dispatches"])
insts = float(row["SQ_INSTS_VALU"])
wave_cycles = float(row["SQ_WAVE_CYCLES"])
# issue-cost floor: the kernel's instruction mix priced with the measured per-class issue costs, as a
# fraction of the SIMD cycles the kernel had - 1.0 would mean the VALU port never waited
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sym = "_ZN3gbk18k_gl_merkle_leavesEPKymjyPy" if field == "goldilocks" else "_ZN3gbk18k_bb_merkle_leavesEPKjmjyPj"
src = "kernels_merkle.hip" if field == "goldilocks" else "kernels_bb.hip"
try:   # needs hipcc and the kernel's mangled name: neither is worth losing the whole summary over
    mix = kernel_mix(os.path.join(root, "plonky2_goldibear_amd", "csrc", src), sym)
except (Exception, SystemExit) as e:   # noqa: BLE001
    print("pmc_poseidon: no instruction mix (%s): issue_cost_floor_frac = null" % e, file=sys.stderr)
    mix = None
dur_s = float(row["TotalDurationNs(under PMC)"]) * 1e-9
floor = None
if mix:
    SIMDS, CLOCK = 1024, 2.4e9
    floor = insts * mix["model_cycles_per_valu_instruction"] / (SIMDS * CLOCK * dur_s)
# matrix-pipe share of the same kernel (its MDS layers are i8 MFMAs): from the MFMA counter pass when make_profiles.sh collected one
mfma_per_perm = None
mfma_csv = path.replace("_sq_counters.csv", "_mfma_counters.csv")
if field == "goldilocks" and os.path.exists(mfma_csv):
    mrow = next((r for r in csv.DictReader(open(mfma_csv)) if r["Kernel"] == kernel), None)
    if mrow and float(mrow.get("SQ_INSTS_MFMA", 0) or 0) > 0:
        mperms = N * (-(-cols // 8)) * int(mrow["Dispatches"])
        mfma_per_perm = float(mrow["SQ_INSTS_MFMA"]) / (mperms / 64.0)
res = {
    "csrc_sha16": csrc_sha16(), "issue_cost_floor_frac": floor, "isa_mix": mix, "mfma_instr_per_permutation": mfma_per_perm,
    "source_file": path, "kernel": kernel, "columns": cols, "log_n": log_n, "permutations": perms,
    "SQ_INSTS_VALU": insts, "valu_instr_per_permutation": insts / (perms / 64.0),
    "vgprs": int(row["VGPRs"]), "duration_ns_under_pmc": float(row["TotalDurationNs(under PMC)"]),
    "wait_inst_per_wave_cycle": float(row["wait_inst_per_wave_cycle"]), "wait_any_per_wave_cycle": float(row["wait_any_per_wave_cycle"]),
    "note": "issue cost per wave64 VALU instruction measured by tools/microbench_valu2.hip: v_mov/v_add_u32 ~2.4-2.6 cycles, every "
            "carry, select, shift, mul and v_mad_u64_u32 ~4.2-4.5 cycles per SIMD",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
