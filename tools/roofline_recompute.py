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
import sqlite3
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


# ---------------------------------------------------------------------------------------------------------------------------
# Same process, same box (round 6):  python tools/roofline_recompute.py --db p_results.db --line line.json --out out.json
# `line.json` is the JSON line of `rocprofv3 --kernel-trace -- python3 bench.py --trace-markers ...`, `p_results.db` the rocpd
# database of that very run.  bench.py --trace-markers puts one dispatch of k_gl_poseidon_permute (a kernel no proof launches) in
# front of and behind every timed region, so the dispatches of the timed steps can be cut out of the trace exactly; inside a
# region every transform dispatch is put into the scope whose HIP events cover it in the library (csrc/api.hip commit():
# "IFFT" / "FFT + blinding"; csrc/prover_host.inc: "quotient IFFT", "FRI LDE") by the order in which a proof issues them:
#   ... commitments (inverse transforms -> IFFT, LDE passes -> FFT + blinding) ... k_quotient ... inverse transforms (quotient IFFT)
#   ... the quotient commitment's two LDE passes (FFT + blinding) ... every later LDE dispatch of the proof (FRI LDE).
def classify_region(rows):
    """rows: (name, duration_ns) of one timed region in dispatch order -> {scope: total ns}"""
    tot = {"IFFT": 0, "FFT + blinding": 0, "quotient IFFT": 0, "FRI LDE": 0}
    cnt = dict.fromkeys(tot, 0)
    phase = "commit"
    for name, dur in rows:
        inv, lde = "intt" in name or "ntt_small" in name, "lde_p" in name
        if "k_quotient<" in name or "k_quotient(" in name:
            phase = "q_intt"
            continue
        if not (inv or lde):
            continue
        if phase == "fri" and inv:        # the next proof's wires commitment
            phase = "commit"
        if phase == "q_intt" and lde:     # the quotient commitment (from_coeffs): strided pass, then contiguous pass
            phase = "q_commit"
        if phase == "commit":
            key = "IFFT" if inv else "FFT + blinding"
        elif phase == "q_intt":
            key = "quotient IFFT"
        elif phase == "q_commit":
            key = "FFT + blinding"
            if "lde_pb" in name:          # the contiguous pass ends the commitment: what follows is the FRI layers' coset_fft
                phase = "fri"
        else:
            key = "FRI LDE"
        tot[key] += dur
        cnt[key] += 1
    return tot, cnt


def same_process(db_path, line_path, out_path):
    line = json.loads([l for l in open(line_path) if l.startswith("{")][-1])
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "poseidon_permute" in r[0]]
    assert len(marks) >= 2 and len(marks) % 2 == 0, "run bench.py with --trace-markers under rocprofv3 --kernel-trace (%d marker dispatches)" % len(marks)
    regions = [[(r[0], r[2] - r[1]) for r in rows[marks[2 * i] + 1:marks[2 * i + 1]]] for i in range(len(marks) // 2)]
    # the legs of bench.py's default run in the order it times them: headline (host witness), [hbm], [vecs], [two in flight], then the
    # same for BabyBear
    def legs(obj):
        return 1 + ("value_hbm_resident" in obj) + ("value_vec_of_vecs" in obj) + ("value_inflight2" in obj)
    targets = [(line, 0)]
    if isinstance(line.get("babybear"), dict):
        targets.append((line["babybear"], legs(line)))
    steps = line["steps"]
    for obj, ri in targets:
        roof = obj["roofline"]
        tot, cnt = classify_region(regions[ri])
        trace_ms = {k: v / 1e6 / steps for k, v in tot.items()}
        live = roof["scopes_ms"]
        counted = ("IFFT", "FFT + blinding", "FRI LDE")          # what roofline.ms and the algorithmic bytes cover
        t_ms = sum(trace_ms[k] for k in counted)
        roof["frac_from_profile"] = roof["algorithmic_bytes"] / (t_ms * 1e-3) / HBM
        roof["ntt_kernel_ms_from_profile"] = t_ms
        roof["profile_source"] = {"same_process": True, "trace": os.path.basename(db_path),
                                  "note": "kernel-trace durations of exactly the dispatches the timed steps' scopes cover (markers + dispatch order)"}
        roof["trace_ms"] = trace_ms
        roof["trace_dispatches_per_step"] = {k: v / float(steps) for k, v in cnt.items()}
        # HIP-event span minus the sum of the kernels' own durations: launch gaps inside a scope (and the clocks' disagreement)
        roof["events_minus_trace_ms"] = {k: live[k] - trace_ms[k] for k in trace_ms}
        roof["frac_over_frac_from_profile"] = roof["frac"] / roof["frac_from_profile"]
    line["profile_mode"] = "bench.py --trace-markers under rocprofv3 --kernel-trace: the line and the trace are one process on one box"
    json.dump(line, open(out_path, "w"))
    print(json.dumps({"goldilocks": {k: line["roofline"][k] for k in ("frac", "frac_from_profile", "scopes_ms", "trace_ms", "events_minus_trace_ms")}}, indent=1))


def main():
    if "--db" in sys.argv:
        a = sys.argv
        return same_process(a[a.index("--db") + 1], a[a.index("--line") + 1], a[a.index("--out") + 1])
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
        # round 6: with the run's rocpd database (GB_PROFILE_DB_<FIELD>, bench.py --trace-markers) the sum is taken over exactly the
        # dispatches of the timed steps that the line's scopes cover - IFFT, FFT + blinding, FRI LDE - and the quotient's per-coset
        # inverse transforms, which the algorithmic bytes do not count, are itemised beside it
        dbp = os.environ.get("GB_PROFILE_DB_" + field.upper())
        if dbp and os.path.exists(dbp):
            rows = sqlite3.connect(dbp).execute("select name, start, end from kernels order by start").fetchall()
            marks = [i for i, r in enumerate(rows) if "poseidon_permute" in r[0]]
            if len(marks) >= 2:
                steps = int(os.environ.get("GB_PROFILE_STEPS", "5"))
                tot, _ = classify_region([(r[0], r[2] - r[1]) for r in rows[marks[0] + 1:marks[1]]])
                counted = (tot["IFFT"] + tot["FFT + blinding"] + tot["FRI LDE"]) / steps
                f["all_transform_kernels_ms_per_proof"] = f["ntt_kernel_ms_per_proof"]
                f["ntt_kernel_ms_per_proof"] = counted / 1e6
                f["frac_from_profile"] = alg / (counted * 1e-9) / HBM
                f["scopes_ms_per_proof_from_trace"] = {k: v / steps / 1e6 for k, v in tot.items()}
                f["dispatch_selection"] = "timed steps only (markers), scopes by dispatch order; the quotient's inverse transforms are not in the sum"
                # what the name-level sum holds beyond the timed steps: the circuit's own constants/sigmas commitment and the warm-up proofs
                outside = sum(r[2] - r[1] for r in rows[:marks[0]] + rows[marks[1]:] if "intt" in r[0] or "lde_p" in r[0])
                f["transform_ms_outside_timed_steps"] = outside / 1e6
                f["timed_steps"], f["proofs_in_trace"] = steps, proofs
                ntt_ns = counted
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
