#!/usr/bin/env python3
"""Merged host / device timeline of the LAST proof in a rocprofv3 trace taken with --hip-trace --kernel-trace (rocpd database): every
HIP API call of the proving thread with its duration and every kernel / copy dispatch, in time order, microseconds from the proof's
first call - to see what the host does while the queue is empty (tools/trace_gaps.py says how long it is empty).

  rocprofv3 --hip-trace --kernel-trace -d gpurun_out/rs/ht -o p -- python3 tools/bench_recursion_shape.py 12
  python tools/trace_timeline.py gpurun_out/rs/ht/p_results.db > gpurun_out/rs/timeline.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    kern = db.execute("select name, start, end from kernels order by start").fetchall()
    pows = [i for i, r in enumerate(kern) if "k_pow_grind" in r[0]]
    t_lo, t_hi = kern[pows[-2]][2], kern[pows[-1]][2] + 400000      # one proof's worth: previous grind .. this grind (+ the query phase)
    cols = [r[1] for r in db.execute("pragma table_info(regions)")]
    rows = db.execute("select name, start, end, tid from regions where start >= ? and start < ? order by start", (t_lo, t_hi)).fetchall() if "tid" in cols else []
    ev = [(s, "host", n, e - s) for n, s, e, _ in rows] + [(s, "gpu ", n.split("(")[0].replace("void ", "").replace("gbk::", "")[:44], e - s)
                                                             for n, s, e in kern if t_lo <= s < t_hi]
    ev.sort()
    t0 = ev[0][0]
    agg = {}
    for s, where, n, d in ev:
        print("%9.1f  %s  %-46s %8.1f us" % ((s - t0) / 1e3, where, n, d / 1e3))
        if where == "host":
            a = agg.setdefault(n, [0, 0.0])
            a[0] += 1
            a[1] += d / 1e3
    print("\nhost API totals in this window:")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s x %4d  %9.1f us" % (n, c, us))


if __name__ == "__main__":
    main()
