#!/bin/bash
# Regenerates profiles/${R}_* on an MI355X box (the command sequence that produced the committed files):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/make_profiles.sh'      then, locally:  bash tools/make_profiles.sh --summarise
# rocprofv3 writes rocpd SQLite databases on this image; tools/rocpd_kernel_stats.py, tools/pmc_traffic.py,
# tools/pmc_sq_summary.py and tools/pmc_poseidon.py turn them into the CSV / JSON summaries.  PMC passes are separate runs
# (FETCH_SIZE and WRITE_SIZE do not fit one pass) and never combined with tracing.
set -e
R=${GB_PROFILE_ROUND:-r03}
OUT=gpurun_out
if [ "$1" != "--summarise" ]; then
    cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
    python3 tools/csrc_hash.py > $OUT/profile_csrc_sha16.txt   # the code these figures are measured on (bench.py: "stale")
    SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
    # the driver's own command: kernel trace + stats of the default bench line (both fields, host and HBM-resident witness)
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 > $OUT/prof_bench.log 2>&1
    for F in goldilocks babybear; do
        COLS=$([ $F = babybear ] && echo 167 || echo 135)
        timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$F -o p -- python3 bench.py --field $F --steps 5 --warmup 2 --no-babybear --no-resident --no-cpu-baseline > $OUT/prof_$F.log 2>&1
        timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_${F}_f -o f -- python3 bench.py --workload commit --field $F --cols $COLS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_${F}_f.log 2>&1
        timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_${F}_w -o w -- python3 bench.py --workload commit --field $F --cols $COLS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_${F}_w.log 2>&1
        timeout -k 10 300 rocprofv3 --pmc $SQ -d $OUT/pmc_${F}_sq -o s -- python3 bench.py --workload commit --field $F --cols $COLS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_${F}_sq.log 2>&1
    done
    # matrix-pipe counters of the Goldilocks hash kernels (the MDS layers run as i8 MFMAs); a pass of its own, allowed to fail
    # (counter names differ between ROCm releases)
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_goldilocks_mfma -o m -- python3 bench.py --workload commit --field goldilocks --cols 135 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_goldilocks_mfma.log 2>&1 || true
    exit 0
fi
python tools/rocpd_kernel_stats.py $OUT/prof_bench/p_results.db profiles/${R}_bench_default_kernel_stats.csv
grep '"metric"' $OUT/prof_bench.log > profiles/${R}_bench_default.json
[ -f $OUT/pmc_goldilocks_mfma/m_results.db ] && python tools/pmc_sq_summary.py $OUT/pmc_goldilocks_mfma/m_results.db profiles/${R}_commit_goldilocks_2p20_mfma_counters.csv || true
for F in goldilocks babybear; do
    COLS=$([ $F = babybear ] && echo 167 || echo 135)
    ES=$([ $F = babybear ] && echo 4 || echo 8)
    python tools/rocpd_kernel_stats.py $OUT/prof_$F/p_results.db profiles/${R}_prove_${F}_2p20_kernel_stats.csv
    python tools/pmc_traffic.py $OUT/pmc_${F}_f/f_results.db $OUT/pmc_${F}_w/w_results.db $COLS 20 $ES profiles/${R}_ntt_traffic_pmc_$F.json > /dev/null
    python tools/pmc_sq_summary.py $OUT/pmc_${F}_sq/s_results.db profiles/${R}_commit_${F}_2p20_sq_counters.csv
    python tools/pmc_poseidon.py profiles/${R}_commit_${F}_2p20_sq_counters.csv $F $COLS 20 profiles/${R}_poseidon_valu_$F.json > /dev/null
    grep '"metric"' $OUT/prof_$F.log > profiles/${R}_bench_prove_${F}_2p20.json
done
ls -la profiles
