#!/usr/bin/env python3
"""profiles/rNN_recursion_shape.txt from the logs of one gpurun call (tools/recursion_profile.sh): prove() of recursion-shaped
circuits at 2^12 .. 2^14 rows, both fields, several proofs in flight on 4 and 8 hardware queues, and where a 2^12-row proof's wall
time is not kernels (tools/trace_gaps.py).
  python tools/recursion_profile.py gpurun_out/rs profiles/r06_recursion_shape.txt"""
import json
import os
import sys


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")] if os.path.exists(path) else []


def main():
    d, out = sys.argv[1], sys.argv[2]
    o = ["# tools/bench_recursion_shape.py on one MI355X, one gpurun call (tools/recursion_profile.sh): the gate set of the reference's recursion",
         "# circuits at 2^12 .. 2^14 rows, standard_recursion_config (rate_bits 3, 28 query rounds, 16 proof-of-work bits), witness resident in HBM,",
         "# median of 20 proofs, every proof verified.  'with scopes' = the same with the library's timing scopes on (two HIP events each).",
         "# Round 5: 4.28 / 4.55 / 5.71 ms (Goldilocks 2^12 / 2^13 / 2^14), 2.57 ms (BabyBear 2^12), 518 proofs/s six in flight.", ""]
    o.append("%-11s %5s %10s %10s %12s %12s" % ("field", "rows", "median ms", "min ms", "with scopes", "proof bytes"))
    for f in ("gl", "bb", "gl_high_rate"):
        for j in lines(os.path.join(d, f + ".log")):
            o.append("%-11s  2^%2d %10.3f %10.3f %12.3f %12d%s" % (j["field"], j["log_n"], j["prove_ms_median"], j["prove_ms_min"],
                                                                    j.get("prove_ms_median_with_timing_scopes", float("nan")), j["proof_bytes"],
                                                                    "   (high_rate_config: rate_bits 7, 12 query rounds)" if f == "gl_high_rate" else ""))
    o += ["", "several independent circuits proved concurrently (one context = one stream, one host thread each), 2^12 rows, Goldilocks:"]
    for name in sorted(os.listdir(d)):
        if name.startswith("gl_inflight") and name.endswith(".log"):
            for j in lines(os.path.join(d, name)):
                o.append("  %-70s GPU_MAX_HW_QUEUES=%-4s %8.1f proofs/s" % (j["workload"].split(",")[1].strip() + ", " + name[:-4], j.get("GPU_MAX_HW_QUEUES"), j["proofs_per_s"]))
    for name, title in (("scopes", None),):
        js = lines(os.path.join(d, "gl.log"))
        if js:
            o += ["", "timing scopes of the 2^12-row Goldilocks proof (ms per proof; nested scopes overlap):", "  " + json.dumps(js[0]["scopes_ms_per_proof"])]
    g = os.path.join(d, "gaps.txt")
    if os.path.exists(g):
        o += ["", "tools/trace_gaps.py on a kernel trace of the 2^12-row Goldilocks proof (rocprofv3 --kernel-trace, timing scopes off; the tracer adds about 0.2 ms per proof):"]
        o += ["  " + l.rstrip() for l in open(g)]
    open(out, "w").write("\n".join(o) + "\n")
    print("\n".join(o[:40]))


if __name__ == "__main__":
    main()
