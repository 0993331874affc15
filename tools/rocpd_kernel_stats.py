#!/usr/bin/env python3
"""rocprofv3's default output on this image is a rocpd SQLite database; this prints / writes the per-kernel
summary (the same columns as rocprofv3 --stats' kernel_stats.csv) from it.

  python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        "select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by %s "
        "order by 3 desc" % (name, name)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        out.writerow([n, c, t, "%.1f" % a, "%.2f" % (100.0 * t / total), mn, mx])


if __name__ == "__main__":
    main()
