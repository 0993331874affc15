#!/bin/bash
# value / value_vec_of_vecs with different numbers of copy threads of the staging ring (gb_ctx_set_option "copy_threads").
#   gpurun -- 'bash tools/ab_vecs.sh -1 0 2 4 8'
mkdir -p gpurun_out/ab_vecs
for t in "$@"; do
    timeout -k 10 240 python3 bench.py --steps ${GB_AB_STEPS:-12} --warmup 3 --no-cpu-baseline --no-inflight2 --no-resident \
        --lib-option copy_threads=$t > gpurun_out/ab_vecs/bench_$t.json 2> gpurun_out/ab_vecs/bench_$t.err || exit 1
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_vecs/bench_$t.json").read().strip().splitlines()[-1])
b = d.get("babybear", {})
sv, sp = d.get("scopes_ms_per_step_vec_of_vecs", {}), d["scopes_ms_per_step"]
print("copy_threads=%3s  GL value %.3f vecs %.3f (%.1f%%)  wires-commit scope %.2f vs %.2f ms | BB value %.3f vecs %.3f (%.1f%%)" % (
    "$t", d["value"], d["value_vec_of_vecs"], 100 * (d["value_vec_of_vecs"] / d["value"] - 1), sv.get("compute wires commitment", 0),
    sp.get("compute wires commitment", 0), b.get("value", 0), b.get("value_vec_of_vecs", 0),
    100 * (b.get("value_vec_of_vecs", 0) / b.get("value", 1) - 1)))
PY
done
