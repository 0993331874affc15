#!/usr/bin/env python3
"""Per-kernel SQ counter summary (CSV) from a rocprofv3 --pmc rocpd database.
  python tools/pmc_sq_summary.py s_results.db out.csv"""
import collections
import csv
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start), max(vgpr_count), max(lds_block_size) "
                   "from counters_collection group by kernel_name, counter_name").fetchall()
agg, meta = collections.defaultdict(dict), {}
for k, c, v, n, d, vg, lds in rows:
    k = k.split("(")[0]
    agg[k][c] = v
    meta[k] = (n, d, vg, lds)
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["Kernel", "Dispatches", "TotalDurationNs(under PMC)", "VGPRs", "LDS_bytes"] + names +
           ["active_valu_per_wave_cycle", "wait_inst_per_wave_cycle", "wait_any_per_wave_cycle"])
for k in sorted(agg, key=lambda k: -meta[k][1]):
    if "copyBuffer" in k:
        continue
    v = agg[k]
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    w.writerow([k, meta[k][0], meta[k][1], meta[k][2], meta[k][3]] + ["%.6g" % v.get(c, 0) for c in names] +
               ["%.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / wc), "%.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / wc),
                "%.3f" % (v.get("SQ_WAIT_ANY", 0) / wc)])
