#!/usr/bin/env python3
"""Static VALU instruction mix of one kernel (by issue-cost class) from hipcc's assembly of a csrc/*.hip file, and the cycles
per average VALU instruction the issue-cost model of DESIGN.md section 4 gives for it.

  python tools/isa_mix.py plonky2_goldibear_amd/csrc/kernels_merkle.hip _ZN3gbk18k_gl_merkle_leavesEPKymjyPy [-DFLAG ...]

Classes (gfx950, tools/microbench_valu2.hip / microbench_mulmod.hip): v_mad_u64_u32 / v_mad_i64_i32 / v_mul_lo / v_mul_hi 4.5 cycles per wave64
instruction and SIMD; v_mov / v_add_u32 / v_xor / v_perm / v_lshl_add_u32 (plain 32-bit, no carry) 2.4; every carry, select,
64-bit add, shift op 2.9; an MFMA holds the SIMD's vector issue for 8 cycles (MI355X_MICROARCH.md) and runs on the matrix pipe."""
import json
import os
import re
import subprocess
import sys
import tempfile

COST = {"mad": 4.5, "plain32": 2.4, "other": 2.9, "mfma": 8.0}
PLAIN32 = ("v_mov_b32", "v_mov_b64", "v_add_u32", "v_sub_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_perm_b32", "v_lshl_add_u32",
           "v_lshlrev_b32", "v_lshrrev_b32", "v_add3_u32", "v_lshl_or_b32", "v_or3_b32", "v_and_or_b32", "v_bfe_u32", "v_min_u32", "v_max_u32")


def kernel_mix(src, symbol, flags=()):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "plonky2_goldibear_amd", "csrc"),
                               "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", "-o", out, src, *flags],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^%s:.*?s_endpgm" % re.escape(symbol), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found" % symbol)
    mix = {"mad": 0, "plain32": 0, "other": 0, "mfma": 0}
    for line in m.group(0).splitlines():
        t = line.split()
        if not t or not t[0].startswith("v_"):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            mix["mfma"] += 1
        elif op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")):   # 32 x 32 multipliers
            mix["mad"] += 1
        elif any(op.startswith(p) for p in PLAIN32):
            mix["plain32"] += 1
        else:
            mix["other"] += 1
    n = sum(mix.values())
    cyc = sum(mix[k] * COST[k] for k in mix) / n
    return {"kernel": symbol, "static_valu_instructions": n, "mix": mix, "cost_cycles": COST, "model_cycles_per_valu_instruction": cyc}


if __name__ == "__main__":
    print(json.dumps(kernel_mix(sys.argv[1], sys.argv[2], sys.argv[3:]), indent=1))
