#!/bin/bash
# A/B on one box: the leaf-sponge segments of the wires commitment on a second stream beside the transforms of the next upload
# chunks (gb_ctx_set_option "overlap_hash") against everything on one stream.  usage (GPU box): bash tools/ab_overlap_hash.sh
set -e
for F in goldilocks babybear; do
    for rep in 1 2; do
        for OPT in 0 1; do
            L=$(python3 bench.py --field $F --steps 10 --warmup 3 --no-babybear --no-resident --no-inflight2 --no-vecs --no-cpu-baseline --no-checks --lib-option overlap_hash=$OPT 2>/dev/null | grep '"metric"')
            python3 - "$F" "$OPT" "$L" <<'PY'
import json, sys
l = json.loads(sys.argv[3])
print("%s overlap_hash=%s value %.3f proofs/s  ms/step %.2f  roofline.frac %.4f" % (sys.argv[1], sys.argv[2], l["value"], l["ms_per_step"], l["roofline"]["frac"]), flush=True)
PY
        done
    done
done
