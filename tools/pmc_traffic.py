#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, HBM section).  Inputs are the rocpd SQLite databases rocprofv3 writes on this image.

  python tools/pmc_traffic.py fetch.db write.db ncols log_n elem_bytes [out.json]

Units: the counters are in KB.  WRITE_SIZE is exact.  FETCH_SIZE on gfx950 tallies 128-byte requests at 64 bytes
for wide coalesced reads (guide: "double it") and "other access widths are uncalibrated": measured here, 8-byte-per-lane
Goldilocks loads all need x2, 4-byte-per-lane BabyBear loads need x1 when a wave touches 64-byte row segments (p1, p2)
and x2 when it touches >= 128 contiguous bytes.  So each kernel's factor is CALIBRATED to 1 or 2, whichever brings the raw
count closer to the bytes the kernel reads by design (every kernel here reads its input exactly once; design bytes are
listed per kernel), and `fetch_over_design` shows what is left: 1.0 = no re-reads reach the memory side.
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import measured_sha16 as csrc_sha16  # noqa: E402


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0].split("(")[0]: (r[1], r[2] * 1024.0) for r in rows}


# bytes read by design, in units of (ncols * n * elem_bytes); r = rate_bits = 3
DESIGN_READS = {"intt16_p1": 1, "intt16_p2": 1, "intt16_p3": 1, "lde_pa16": 1, "lde_pb16": 8, "merkle_leaves": 8, "to_mont": 1}


def main():
    fdb, wdb, ncols, log_n, es = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    unit = ncols * (1 << log_n) * es
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --workload commit "
                     "--steps 1 --warmup 0` (%d columns x 2^%d, %d-byte elements); KB -> bytes; per-kernel FETCH factor in "
                     "{1, 2} calibrated against the design read bytes (see tools/pmc_traffic.py)" % (ncols, log_n, es),
           "csrc_sha16": csrc_sha16(), "columns": ncols, "log_n": log_n, "elem_bytes": es, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if "copyBuffer" in k:
            continue
        raw = f.get(k, (0, 0.0))[1]
        design = next((m * unit for key, m in DESIGN_READS.items() if key in k), None)
        factor = 2.0
        if design and raw:
            factor = min((1.0, 2.0), key=lambda c: abs(raw * c - design))
        out["kernels"][k] = {"launches": f.get(k, (0, 0))[0], "fetch_bytes_raw": raw, "fetch_factor": factor,
                             "fetch_bytes_corrected": raw * factor, "design_read_bytes": design,
                             "fetch_over_design": (raw * factor / design) if design else None,
                             "write_bytes": w.get(k, (0, 0.0))[1]}
    ks = out["kernels"]
    tot = lambda pred: sum(v["fetch_bytes_corrected"] + v["write_bytes"] for k, v in ks.items() if pred(k))
    out["ifft_bytes_per_column"] = tot(lambda k: "intt" in k) / ncols
    out["lde_bytes_per_column"] = tot(lambda k: "lde_" in k) / ncols
    out["ntt_bytes_per_commit"] = tot(lambda k: "intt" in k or "lde_" in k)
    s = json.dumps(out, indent=1)
    if len(sys.argv) > 6:
        open(sys.argv[6], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
