#!/usr/bin/env python3
"""Where a small proof's wall time is NOT kernels: from the rocpd database of a kernel trace, the idle time between consecutive
dispatches of the steady-state proofs (the last `proofs` ones, cut at the proof-of-work kernel that every proof launches once or
more at its end) - gaps sorted into "back-to-back" (< 6 us: the queue had the next kernel), "launch-bound" (6 .. 25 us: the host
was still launching) and "host round trip" (> 25 us: a read-back of the transcript, host hashing, the next stage's set-up).

  rocprofv3 --kernel-trace -d gpurun_out/rs -o p -- python3 tools/bench_recursion_shape.py 12
  python tools/trace_gaps.py gpurun_out/rs/p_results.db [proofs]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # a proof ends with its query-round gathers; the proof-of-work grind precedes them: cut after the LAST gather that follows a grind
    ends = []
    seen_pow = False
    for i, (n, s, e) in enumerate(rows):
        if "k_pow_grind" in n:
            seen_pow = True
        elif seen_pow and "k_gather" not in n and "copyBuffer" not in n:
            ends.append(i)      # first dispatch of the next proof
            seen_pow = False
    if len(ends) < proofs + 1:
        raise SystemExit("only %d proofs in the trace" % len(ends))
    lo, hi = ends[-proofs - 1], ends[-1]
    seg = rows[lo:hi]
    span = seg[-1][2] - seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    gaps = [(seg[i + 1][1] - seg[i][2], seg[i][0], seg[i + 1][0]) for i in range(len(seg) - 1)]
    cls = {"back-to-back (< 6 us)": [g for g in gaps if g[0] < 6000], "launch-bound (6-25 us)": [g for g in gaps if 6000 <= g[0] < 25000],
           "host round trip (> 25 us)": [g for g in gaps if g[0] >= 25000]}
    print("%d proofs, %.1f dispatches per proof: %.3f ms per proof wall (first start to last end), %.3f ms kernels, %.3f ms idle" % (
        proofs, len(seg) / proofs, span / proofs / 1e6, busy / proofs / 1e6, (span - busy) / proofs / 1e6))
    for k, g in cls.items():
        print("  %-26s %6.1f per proof, %7.3f ms per proof" % (k, len(g) / proofs, sum(x[0] for x in g) / proofs / 1e6))
    big = sorted(cls["host round trip (> 25 us)"], key=lambda x: -x[0])
    where = {}
    for g, a, b in big:
        key = (a.split("(")[0].split("<")[0].replace("void ", "").replace("gbk::", ""), b.split("(")[0].split("<")[0].replace("void ", "").replace("gbk::", ""))
        where.setdefault(key, []).append(g)
    print("  host round trips by (kernel before -> kernel after), us per occurrence:")
    for (a, b), v in sorted(where.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("    %-34s -> %-34s x%5.1f per proof, avg %7.1f us" % (a[:34], b[:34], len(v) / proofs, sum(v) / len(v) / 1e3))
    names = {}
    for n, s, e in seg:
        k = n.split("(")[0].split("<")[0].replace("void ", "").replace("gbk::", "")
        names.setdefault(k, [0, 0])
        names[k][0] += 1
        names[k][1] += e - s
    print("  kernels per proof:")
    for k, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:16]:
        print("    %-40s x%6.1f  %7.3f ms" % (k[:40], c / proofs, t / proofs / 1e6))


if __name__ == "__main__":
    main()
