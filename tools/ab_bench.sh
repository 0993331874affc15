#!/bin/bash
# A/B of library variants under tools/bin/libs/*.so on the GPU box with the bench itself (both fields, short run).
#   gpurun -- 'bash tools/ab_bench.sh nopf pf'
mkdir -p gpurun_out/ab
LIB=plonky2_goldibear_amd/lib/libgoldibear_gpu.so
cp $LIB gpurun_out/ab/.product_lib.so && trap 'cp gpurun_out/ab/.product_lib.so $LIB; rm -f gpurun_out/ab/.product_lib.so' EXIT   # the variants are copied over the product library: put it back
for v in "$@"; do
    cp tools/bin/libs/$v.so plonky2_goldibear_amd/lib/libgoldibear_gpu.so
    timeout -k 10 200 python3 bench.py --steps ${GB_AB_STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/ab/bench_$v.json 2> gpurun_out/ab/bench_$v.err || exit 1
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab/bench_$v.json").read().strip().splitlines()[-1])
b = d.get("babybear", {})
print("== $v GL value %.3f resident %.3f hash %.2f ms merkle %.2f ms | BB value %.3f noretry %.3f hash %.2f ms merkle %.2f ms" % (
    d["value"], d.get("value_hbm_resident", 0), d["scopes_ms_per_step"]["hash leaves"], d["merkle"]["ms"],
    b.get("value", 0), b.get("value_no_retry", 0), b.get("scopes_ms_per_step", {}).get("hash leaves", 0), b.get("merkle", {}).get("ms", 0)))
PY
done
