#!/bin/bash
# A/B of library variants under tools/bin/libs/*.so on the GPU box: per-kernel average times of the commit workload.
#   gpurun -- 'bash tools/ab_kernel_times.sh base lds3'
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out/ab
LIB=plonky2_goldibear_amd/lib/libgoldibear_gpu.so
cp $LIB gpurun_out/ab/.product_lib.so && trap 'cp gpurun_out/ab/.product_lib.so $LIB; rm -f gpurun_out/ab/.product_lib.so' EXIT   # the variants are copied over the product library: put it back
WL=${GB_AB_WORKLOAD:---workload commit}
for v in "$@"; do
    cp tools/bin/libs/$v.so plonky2_goldibear_amd/lib/libgoldibear_gpu.so
    rm -rf gpurun_out/ab/$v
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ab/$v -o p -- python3 bench.py $WL --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$v.log 2>&1
    python tools/rocpd_kernel_stats.py gpurun_out/ab/$v/p_results.db gpurun_out/ab/$v.csv
    echo "== $v"; python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ab/$v.csv")):
    n = r["Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("lde_p", "intt", "merkle_leaves", "merkle_level", "quotient", "eval_partial")):
        print("%-40s calls %4s avg %10.1f us" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
