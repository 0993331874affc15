#!/usr/bin/env python3
"""profiles/rNN_roofline_recompute.json: the `roofline` object of bench.py recomputed from the committed rocprofv3 summaries alone
(VERDICT r4 item 4), so that the line's live figure (HIP-event scopes) and the profile's figure can be printed side by side.

  python tools/roofline_recompute.py profiles r05 [proofs_in_trace]

Per field it reads  rNN_prove_<field>_2p20_kernel_stats.csv  (kernel trace of `bench.py --field F --steps 5 --warmup 2`: 7 proofs),
rNN_ntt_traffic_pmc_<field>.json (FETCH / WRITE passes of the 135 / 167-column commit) and rNN_commit_<field>_2p20_sq_counters.csv
(SQ_INSTS_VALU per kernel of the same commit) and writes, per NTT kernel: calls, average duration, time per proof, physical bytes per
commit and the fraction of 8 TB/s it moves them at; per field: NTT kernel time per proof, algorithmic bytes per proof (SURVEY.md
8(d): (2 + 2^r) n s per from_values column, (1 + 2^r) n s per from_coeffs column - bench.py ProveLeg.counts), `frac_from_profile`,
the physical fraction, the VALU-issue ceiling of the transform's instruction count and the ceiling of its pass structure."""
import csv
import json
import os
import sys

HBM = 8.0e12
SIMDS, CLOCK = 1024, 2.4e9
COPY_RATE = 5.4e12       # measured plain-copy rate of this chip (profiles/r04_ab_kernel_times.txt: 2.26 GB r+w in 0.42 ms)
MIXED_ISSUE_CYCLES = 2.9  # cycles per wave64 integer VALU instruction in a mixed stream (microbench_valu2)
SHAPE = {"goldilocks": dict(nw=135, nr=80, ch=3, esz=8, d=2), "babybear": dict(nw=167, nr=41, ch=10, esz=4, d=4)}


def algorithmic_bytes(field, log_n=20, rate_bits=3):
    s = SHAPE[field]
    n = 1 << log_n
    nzs, nq = s["ch"] * (-(-s["nr"] // 8)), s["ch"] * 8
    return ((2 + (1 << rate_bits)) * (s["nw"] + nzs) + (1 + (1 << rate_bits)) * (nq + s["d"])) * n * s["esz"], s["nw"] + nzs, nq + s["d"]


def main():
    pdir, rnd = sys.argv[1], sys.argv[2]
    proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    out = {"source": "tools/roofline_recompute.py over %s_prove_*_kernel_stats.csv (%d proofs each), %s_ntt_traffic_pmc_*.json, "
                     "%s_commit_*_sq_counters.csv" % (rnd, proofs, rnd, rnd), "hbm_peak_bytes_per_s": HBM}
    for field in ("goldilocks", "babybear"):
        ks = os.path.join(pdir, "%s_prove_%s_2p20_kernel_stats.csv" % (rnd, field))
        if not os.path.exists(ks):
            continue
        alg, ncols_values, ncols_coeffs = algorithmic_bytes(field)
        kernels, ntt_ns = {}, 0.0
        for r in csv.DictReader(open(ks)):
            name = r["Name"].split("(")[0].replace("void ", "")
            if "intt" in name or "lde_p" in name:
                kernels[name] = {"calls_per_proof": int(r["Calls"]) / proofs, "avg_us": float(r["AverageNs"]) / 1e3,
                                 "ms_per_proof": float(r["TotalDurationNs"]) / proofs / 1e6}
                ntt_ns += float(r["TotalDurationNs"]) / proofs
        f = {"ntt_kernel_ms_per_proof": ntt_ns / 1e6, "algorithmic_bytes_per_proof": alg,
             "frac_from_profile": alg / (ntt_ns * 1e-9) / HBM, "kernels": kernels}
        tj = os.path.join(pdir, "%s_ntt_traffic_pmc_%s.json" % (rnd, field))
        if os.path.exists(tj):
            t = json.load(open(tj))
            phys = t["ifft_bytes_per_column"] * ncols_values + t["lde_bytes_per_column"] * (ncols_values + ncols_coeffs)
            f["physical_bytes_per_proof"] = phys
            f["physical_frac"] = phys / (ntt_ns * 1e-9) / HBM
            ns = (1 << t["log_n"]) * t["elem_bytes"]
            passes = (t["ifft_bytes_per_column"] + t["lde_bytes_per_column"]) / ns   # physical n s units per from_values column
            f["physical_ns_units_per_from_values_column"] = passes
            f["pass_structure_ceiling_frac"] = (2 + 8) / passes * COPY_RATE / HBM     # every pass at the plain-copy rate
            for k, v in t["kernels"].items():
                kk = k.replace("void ", "")
                if kk in kernels:
                    kernels[kk]["physical_bytes_per_commit"] = v["fetch_bytes_corrected"] + v["write_bytes"]
        sq = os.path.join(pdir, "%s_commit_%s_2p20_sq_counters.csv" % (rnd, field))
        if os.path.exists(sq):
            valu = 0.0
            for r in csv.DictReader(open(sq)):
                if "intt" in r["Kernel"] or "lde_p" in r["Kernel"]:
                    valu += float(r["SQ_INSTS_VALU"])
                    kk = r["Kernel"].replace("void ", "")
                    if kk in kernels:
                        kernels[kk]["valu_wave_instr_per_commit"] = float(r["SQ_INSTS_VALU"])
                        kernels[kk]["commit_duration_ms_under_pmc"] = float(r["TotalDurationNs(under PMC)"]) / 1e6
            cols = SHAPE[field]["nw"]
            alg_commit = 10 * cols * (1 << 20) * SHAPE[field]["esz"]
            f["valu_wave_instr_per_from_values_column"] = valu / cols
            f["valu_lane_instr_per_algorithmic_byte"] = valu * 64 / alg_commit
            f["valu_ceiling_frac"] = alg_commit / (valu * MIXED_ISSUE_CYCLES / (SIMDS * CLOCK)) / HBM
            f["valu_ceiling_note"] = ("from_values commit of %d columns: SQ_INSTS_VALU of its NTT kernels at %.1f cycles per wave64 instruction and SIMD "
                                      "(mixed integer stream), memory free, issue port never idle" % (cols, MIXED_ISSUE_CYCLES))
        out[field] = f
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from csrc_hash import measured_sha16
        out["csrc_sha16"] = measured_sha16()
    except Exception:
        pass
    path = os.path.join(pdir, "%s_roofline_recompute.json" % rnd)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} if isinstance(v, dict) else v for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
