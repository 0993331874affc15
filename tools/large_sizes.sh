#!/bin/bash
# from_values of a wires-sized batch above 2^20 rows: NTT scopes and per-kernel times, product library against library variants
# (tools/bin/libs/NAME.so, e.g. the round-5 build with its outer radix step around the 2^20-row passes).
#   gpurun -- 'bash tools/large_sizes.sh product r05'        -> gpurun_out/large_sizes.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out/ls
LIB=plonky2_goldibear_amd/lib/libgoldibear_gpu.so
cp $LIB gpurun_out/ls/.product_lib.so && trap 'cp gpurun_out/ls/.product_lib.so $LIB; rm -f gpurun_out/ls/.product_lib.so' EXIT
OUT=gpurun_out/large_sizes.txt
: > $OUT
for v in "$@"; do
    [ "$v" = product ] && cp gpurun_out/ls/.product_lib.so $LIB || cp tools/bin/libs/$v.so $LIB
    for F in goldilocks babybear; do
        C=135; [ $F = babybear ] && C=167
        for L in ${GB_LS_SIZES:-20 21 22}; do
            d=gpurun_out/ls/$v-$F-$L
            rm -rf $d
            timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --workload commit --field $F --log-n $L --cols $C \
                --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || { echo "FAILED $v $F $L" | tee -a $OUT; tail -5 $d.log | tee -a $OUT; continue; }
            python tools/rocpd_kernel_stats.py $d/p_results.db $d.csv
            python - <<PY | tee -a $OUT
import csv, json
line = [l for l in open("$d.log") if l.startswith("{")][-1]
j = json.loads(line)
sc = j["scopes_ms_per_step"]
print("== %s %s 2^%s x %s cols: IFFT %.2f ms  FFT+blinding %.2f ms  Merkle %.2f ms  step %.2f ms  (NTT per 2^20-row column equivalent: %.1f us)" % (
    "$v", "$F", "$L", "$C", sc["IFFT"], sc["FFT + blinding"], sc["build Merkle tree"], j["ms_per_step"],
    (sc["IFFT"] + sc["FFT + blinding"]) * 1e3 / $C / (1 << ($L - 20))))
for r in csv.DictReader(open("$d.csv")):
    n = r["Name"].split("(")[0].replace("void ", "").replace("gbk::", "")
    if any(k in n for k in ("lde_p", "intt", "deinterleave", "combine")):
        print("   %-44s calls %4s avg %10.1f us  total %9.2f ms" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
        done
    done
done
