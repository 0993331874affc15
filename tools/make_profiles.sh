#!/bin/bash
# Regenerates profiles/${R}_* on an MI355X box (the command sequence that produced the committed files):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/make_profiles.sh counters'   then, locally:  bash tools/make_profiles.sh --collect
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/make_profiles.sh bench'      then, locally:  bash tools/make_profiles.sh --collect
# (two calls: a box lives for one call of at most 20 minutes; the second call ships the summaries the first one brought back)
# rocprofv3 writes rocpd SQLite databases on this image; tools/rocpd_kernel_stats.py, tools/pmc_traffic.py,
# tools/pmc_sq_summary.py and tools/pmc_poseidon.py turn them into the CSV / JSON summaries.  PMC passes are separate runs
# (FETCH_SIZE and WRITE_SIZE do not fit one pass) and never combined with tracing.
# Order on the box: the PMC passes, the s_memtime attribution of the leaf-hash kernel and the per-field kernel traces first, their
# summaries (incl. rNN_roofline_recompute.json) written into profiles/ of the box's copy of the tree, THEN the default bench line
# that quotes them - so that no committed bench line says "stale": true about a sibling of the same run.  Everything that has to
# come back is also written under gpurun_out/profiles/ (only gpurun_out/ travels back).
set -e
R=${GB_PROFILE_ROUND:-r06}
OUT=gpurun_out
P=$OUT/profiles
if [ "$1" = "--collect" ]; then
    cp $P/${R}_* profiles/
    ls -la profiles | grep " ${R}_"
    exit 0
fi
STAGE=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p $P
python3 tools/csrc_hash.py > $OUT/profile_csrc_sha16.txt   # the sources these figures are measured on
if [ "$STAGE" != bench ]; then
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
for F in goldilocks babybear; do
    COLS=$([ $F = babybear ] && echo 167 || echo 135)
    ES=$([ $F = babybear ] && echo 4 || echo 8)
    C="python3 bench.py --workload commit --field $F --cols $COLS --steps 1 --warmup 0 --no-cpu-baseline"
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_${F}_f -o f -- $C > $OUT/pmc_${F}_f.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_${F}_w -o w -- $C > $OUT/pmc_${F}_w.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc $SQ -d $OUT/pmc_${F}_sq -o s -- $C > $OUT/pmc_${F}_sq.log 2>&1
    echo "pmc $F done"
done
# matrix-pipe counters of the Goldilocks hash kernels (the MDS layers run as i8 MFMAs); a pass of its own, allowed to fail
# (counter names differ between ROCm releases)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_goldilocks_mfma -o m -- python3 bench.py --workload commit --field goldilocks --cols 135 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_goldilocks_mfma.log 2>&1 || true
[ -f $OUT/pmc_goldilocks_mfma/m_results.db ] && python3 tools/pmc_sq_summary.py $OUT/pmc_goldilocks_mfma/m_results.db $P/${R}_commit_goldilocks_2p20_mfma_counters.csv || true
for F in goldilocks babybear; do
    COLS=$([ $F = babybear ] && echo 167 || echo 135)
    ES=$([ $F = babybear ] && echo 4 || echo 8)
    python3 tools/pmc_traffic.py $OUT/pmc_${F}_f/f_results.db $OUT/pmc_${F}_w/w_results.db $COLS 20 $ES $P/${R}_ntt_traffic_pmc_$F.json > /dev/null
    python3 tools/pmc_sq_summary.py $OUT/pmc_${F}_sq/s_results.db $P/${R}_commit_${F}_2p20_sq_counters.csv
    python3 tools/pmc_poseidon.py $P/${R}_commit_${F}_2p20_sq_counters.csv $F $COLS 20 $P/${R}_poseidon_valu_$F.json > /dev/null
done
# where the wave cycles of the leaf-hash kernel go (attribution build tools/bin/libs/probe.so, built in the build container) and the
# shader clock the chip holds under that load
if [ -f tools/bin/libs/probe.so ]; then
    GB_PROBE_JSON=$P/${R}_leaf_kernel_probe.json timeout -k 10 300 python3 tools/probe_leaves.py > $P/${R}_leaf_kernel_probe_attribution.txt 2> $OUT/probe_leaves.err || true
fi
# one field per run, host-witness leg only, under the kernel trace: 7 proofs each (5 timed + 2 warm-up), no verification proofs
for F in goldilocks babybear; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$F -o p -- python3 bench.py --field $F --steps 5 --warmup 2 --no-babybear --no-resident --no-inflight2 --no-vecs --no-cpu-baseline --no-checks --trace-markers > $OUT/prof_$F.log 2>&1
    python3 tools/rocpd_kernel_stats.py $OUT/prof_$F/p_results.db $P/${R}_prove_${F}_2p20_kernel_stats.csv
    grep '"metric"' $OUT/prof_$F.log > $P/${R}_bench_prove_${F}_2p20.json
done
GB_PROFILE_DB_GOLDILOCKS=$OUT/prof_goldilocks/p_results.db GB_PROFILE_DB_BABYBEAR=$OUT/prof_babybear/p_results.db GB_PROFILE_STEPS=5 python3 tools/roofline_recompute.py $P $R 7 > /dev/null
cp $P/${R}_* profiles/          # the box's own tree: what the bench line below quotes
fi
[ "$STAGE" = counters ] && { ls -la $P; exit 0; }
# the driver's own command under the kernel trace: the line and the trace are ONE process on ONE box (round 6; the program itself
# after `--`), with a marker dispatch around every timed region (--trace-markers) so that tools/roofline_recompute.py can sum exactly
# the dispatches the line's scopes cover -> rNN_bench_default.json carries `frac` (HIP events) and `frac_from_profile` (kernel trace)
# of the same dispatches.  Then once more bare: the line the driver will reproduce (rNN_bench_untraced.json).
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o p -- python3 bench.py --steps 10 --warmup 3 --trace-markers > $OUT/prof_bench.log 2>&1
python3 tools/rocpd_kernel_stats.py $OUT/prof_bench/p_results.db $P/${R}_bench_default_kernel_stats.csv
grep '"metric"' $OUT/prof_bench.log > $OUT/prof_bench_line.json
python3 tools/roofline_recompute.py --db $OUT/prof_bench/p_results.db --line $OUT/prof_bench_line.json --out $P/${R}_bench_default.json > $OUT/prof_bench_recompute.log 2>&1
timeout -k 10 700 python3 bench.py --steps 20 --warmup 5 > $P/${R}_bench_untraced.json 2> $OUT/bench_default.err
ls -la $P
