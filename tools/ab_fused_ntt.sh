#!/bin/bash
# from_values at 2^20 rows with and without the fused last-inverse / first-LDE pass (gb_ctx_set_option "fuse_intt_lde"): per-kernel
# times of the commit workload (input resident in HBM) under rocprofv3 --kernel-trace, and the live scopes.
#   gpurun -- 'bash tools/ab_fused_ntt.sh goldilocks'
F=${1:-goldilocks}
COLS=$([ $F = babybear ] && echo 167 || echo 135)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out/ab_fused
for fuse in 1 0 1 0; do
    d=gpurun_out/ab_fused/${F}_$fuse
    rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --workload commit --field $F --cols $COLS --steps 3 --warmup 1 \
        --no-cpu-baseline --lib-option fuse_intt_lde=$fuse > $d.log 2>&1
    python3 tools/rocpd_kernel_stats.py $d/p_results.db $d.csv
    echo "== $F fuse_intt_lde=$fuse"; python3 - <<PY
import csv, json
tot = 0.0
for r in csv.DictReader(open("$d.csv")):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "lde_p" in n or "intt" in n:
        per = float(r["TotalDurationNs"]) / 4 / 1e3
        tot += per
        print("  %-34s calls %4s avg %9.1f us   per commit %9.1f us" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, per))
j = [json.loads(l) for l in open("$d.log") if l.startswith("{")][-1]
print("  NTT kernels per commit %.3f ms | live scopes IFFT %.3f + FFT %.3f ms, roofline.frac %.4f" % (
    tot / 1e3, j["scopes_ms_per_step"].get("IFFT", 0), j["scopes_ms_per_step"]["FFT + blinding"], j["roofline"]["frac"]))
PY
done
