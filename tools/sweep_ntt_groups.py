#!/usr/bin/env python3
"""Ablation of the column-group knobs of the NTT passes (kernels_ntt.hip; gb_ctx_set_option lde_group / pa_log_split / intt_group) on the
commit workload (from_values of the wires matrix, input resident in HBM): one bench.py process per setting, IFFT / FFT scopes.

  gpurun -- 'python3 tools/sweep_ntt_groups.py goldilocks > gpurun_out/sweep_gl.txt'
"""
import itertools
import json
import os
import subprocess
import sys

field = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
cols = "167" if field == "babybear" else "135"
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"


def run(opts):
    out = subprocess.run([sys.executable, "bench.py", "--workload", "commit", "--field", field, "--cols", cols, "--steps", "4",
                          "--warmup", "1", "--no-cpu-baseline"] + ["--lib-option=%s=%s" % kv for kv in opts.items()],
                         env=dict(os.environ), capture_output=True, text=True, timeout=300)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            s = j["scopes_ms_per_step"]
            return s.get("IFFT"), s.get("FFT + blinding"), j["roofline"]["frac"], j["ms_per_step"]
    return None, None, None, out.stderr[-300:]


print("%-46s %8s %8s %8s %9s" % ("setting", "IFFT ms", "FFT ms", "frac", "step ms"), flush=True)
settings = [{}]
groups = (1, 2, 4) if quick else (1, 2, 3, 4, 8, 16)
splits = (0, 2) if quick else (0, 1, 2, 3)
for g, sp in itertools.product(groups, splits):
    settings.append({"lde_group": g, "pa_log_split": sp})
settings.append({"pa_log_split": 1})
settings.append({"pa_log_split": 3})
for g in ((4, 8) if quick else (1, 2, 4, 8, 16)):
    settings.append({"intt_group": g})
for st in settings:
    r = run(st)
    name = " ".join("%s=%s" % kv for kv in sorted(st.items())) or "(all columns per launch)"
    if r[0] is None:
        print("%-46s FAILED %s" % (name, r[3]), flush=True)
    else:
        print("%-46s %8.3f %8.3f %8.4f %9.2f" % (name, r[0], r[1], r[2], r[3]), flush=True)
