#!/bin/bash
# SQ / SQC counter passes over the leaf-hash kernel of the commit workload (k_gl_merkle_leaves / k_bb_merkle_leaves): instruction
# classes, issue activity, instruction-cache and scalar-cache behaviour, LDS / SMEM waits - the counter side of the attribution of
# the kernel's idle issue slots (profiles/r05_leaf_kernel_attribution_*.txt).  Counters only, never combined with tracing.
#   gpurun -- 'bash tools/pmc_leaf_kernel.sh goldilocks'
F=${1:-goldilocks}
COLS=$([ $F = babybear ] && echo 167 || echo 135)
OUT=gpurun_out/pmc_leaf_$F
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
C="python3 bench.py --workload commit --field $F --cols $COLS --steps 1 --warmup 0 --no-cpu-baseline"
i=0
while read -r line; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $line --kernel-include-regex "merkle_leaves" -d $OUT/p$i -o c -- $C > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $line"; tail -3 $OUT/p$i.log; continue; }
    python3 tools/pmc_sq_summary.py $OUT/p$i/c_results.db $OUT/pass$i.csv && echo "pass $i ok: $line"
done <<PASSES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS SQ_WAVE_CYCLES
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_CYCLES
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQ_WAVE_CYCLES
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU
PASSES
python3 - <<PY
import csv, glob
rows = {}
for f in sorted(glob.glob("$OUT/pass*.csv")):
    for r in csv.DictReader(open(f)):
        if "merkle_leaves" not in r["Kernel"]:
            continue
        d = rows.setdefault(r["Kernel"], {})
        for k, v in r.items():
            if k.startswith(("SQ", "Total", "Disp")):
                d.setdefault(k, v)
with open("$OUT/leaf_kernel_counters.txt", "w") as o:
    for k, d in rows.items():
        o.write(k + "\n")
        for c in sorted(d):
            o.write("  %-34s %s\n" % (c, d[c]))
print(open("$OUT/leaf_kernel_counters.txt").read())
PY
